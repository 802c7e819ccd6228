"""Training solver for example/clip_fdt and example/clip on the MI355X engine.

Restates the reference's ClsSolver (example/clip_fdt/train_solver.py:92-717, example/clip/train_solver.py:157-624):
same CLI (--config --output_path --batch_size --debug --exp_name --ckpt_path), same YAML keys, same run-directory and
checkpoint layout ({'model': DDP state_dict with the 'module.' prefix, 'optimizer', 'last_iter'} in
<out>/checkpoints/ckpt_{step}.pth.tar), same step body order (train_solver.py:348-439) and the iterated-learning
reset schedule (:545-557).  Differences, all deliberate:
  * no per-step host synchronisation: the reference's three barrier()+.cpu() and three .item() per step
    (linklink/__init__.py:30-34, misc.py:38-40) are replaced by device-side meters read every print_freq steps;
  * the reset call passes seed=curr_step (the published call passes no seed and would raise TypeError at the
    first reset, train_solver.py:546 vs clip_fdt.py:256) and the codebook copy is taken at the reset step (the
    published code only stores it on resume, :548-549) -- SURVEY.md section 5;
  * wandb, the webdataset pipeline and the in-loop SugarCrepe evaluation are outside the hot path: data comes from
    `--synthetic` (the benchmark's generator) or from any iterable of (image[B,3,H,W], text) batches passed to
    ClsSolver(train_data=...).
"""
import argparse
import json
import logging
import os
import pprint
import time

import torch

from .prototype import linklink as link
from .prototype.loss_functions import ClipInfoCELoss
from .prototype.lr_scheduler import scheduler_entry
from .prototype.model import model_entry
from .prototype.optimizer import optim_entry
from .prototype.utils import torch_ddp_dist as D
from .prototype.utils.misc import (EasyDict, accuracy, count_params, create_logger, get_logger, load_state_model,
                                   load_state_optimizer, makedir, param_group_all, parse_config)


class DeviceMeter:
    """Sliding-window mean kept on the device; read (one host sync) only when a log line is printed."""

    def __init__(self, length, device):
        self.buf = torch.zeros(max(1, length), device=device)
        self.n = 0

    def update(self, t):
        self.buf[self.n % self.buf.numel()] = t.detach().reshape(())
        self.n += 1

    def read(self):
        k = min(self.n, self.buf.numel())
        if k == 0:
            return 0.0, 0.0
        window = self.buf[:k]
        stats = torch.stack([self.buf[(self.n - 1) % self.buf.numel()], window.mean()])
        if link.get_world_size() > 1 and torch.distributed.is_initialized():
            link.allreduce(stats)            # values were pre-divided by world_size, as in the reference
        val, avg = stats.tolist()
        return val, avg


class SyntheticPairs:
    """Synthetic 224x224 + 77-token pairs, resident on the device."""

    def __init__(self, batch_size, num_batches, res, ctx, device, seed=1234):
        g = torch.Generator().manual_seed(seed + link.get_rank())
        self.num_batches = num_batches
        self.images = torch.randn(batch_size, 3, res, res, generator=g).to(device)
        tok = torch.zeros(batch_size, ctx, dtype=torch.int64)
        pad = torch.full((batch_size, ctx), float("-inf"))
        lens = torch.randint(min(8, ctx), ctx + 1, (batch_size,), generator=g)
        for b in range(batch_size):
            n = int(lens[b])
            tok[b, 0] = 49407
            tok[b, 1:n - 1] = torch.randint(0, 49406, (n - 2,), generator=g)
            tok[b, n - 1] = 49408
            pad[b, :n] = 0
        # caption lengths travel with the batch as host metadata (what a tokenising loader knows), so the text tower runs on
        # the valid tokens only
        self.text = (tok.to(device), pad.to(device), [int(n) for n in lens])

    def set_epoch(self, epoch):
        pass

    def __iter__(self):
        for _ in range(self.num_batches):
            yield self.images, self.text

    @property
    def dataloader(self):
        return self


class DevicePrefetcher:
    """Feeds the step from any iterable of (image [B,3,H,W] CPU tensor, captions list[str]) batches -- what the
    reference's loader yields (prototype/data/clip_dataset_wsd.py:158-240) -- one batch ahead of the GPU
    (SURVEY.md 8f-1/2): a worker thread tokenises the captions with the C++ BPE (the reference does it in Python inside
    forward(), on the training thread), pins the image batch and copies it to the device on its own stream; the
    consumer receives (image_cuda, (tokens, pad_mask, lengths)), i.e. the text tower can run on the valid tokens and
    the H2D copy / tokenisation of batch i+1 overlap the step on batch i.  Already-tokenised text passes through.
    Images may arrive as uint8 ([B,H,W,3] as decoders produce them, or [B,3,H,W]), alone or as (image_u8, flags) with
    per-sample uint8 flags (bit 0 horizontal flip, bit 1 grayscale): the batch then crosses PCIe at a quarter of the fp32
    size and ToTensor + Normalize (+ the flip / grayscale of MOCOV2_single) run on the device
    (ops.image_u8_normalize; reference prototype/data/imagenet_dataloader.py:13-14,59-68).
    A batch whose image part is a LIST of decoded uint8 [H,W,3] images of any sizes (what a loader yields when it stops after
    decode) takes the whole MOCOV2_single augmentation on the device: the random draws are made here on the host
    (ops.mocov2_params, seeded by `augment_seed` and the batch index), the images cross PCIe back to back in one pinned buffer,
    and RandomResizedCrop / ColorJitter / RandomGrayscale / GaussianBlur / flip / ToTensor / Normalize run in
    ops.image_augment on this prefetcher's stream, beside the step on the previous batch."""

    def __init__(self, loader, tokenize, device, depth=2, augment_seed=0, out_size=224):
        self.loader, self.tokenize, self.device, self.depth = loader, tokenize, torch.device(device), depth
        self.dataloader = self
        self.num_batches = getattr(loader, "num_batches", None)
        self.augment_seed, self.out_size, self._batches = augment_seed, out_size, 0

    def _augment(self, images):
        """list of uint8 [H,W,3] CPU tensors -> fp32 [B,3,out,out] on the device (current stream)"""
        import random
        from . import ops
        sizes = [(int(im.shape[0]), int(im.shape[1])) for im in images]
        for im in images:
            if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3:
                raise RuntimeError("DevicePrefetcher: decoded images must be uint8 [H,W,3], got %s %s" % (im.dtype, tuple(im.shape)))
        params = ops.mocov2_params(sizes, random.Random((self.augment_seed << 32) ^ self._batches), out_size=self.out_size)
        self._batches += 1
        total = sum(h * w * 3 for h, w in sizes)
        flat = torch.empty(total, dtype=torch.uint8).pin_memory()
        offs, o = [], 0
        for im, (h, w) in zip(images, sizes):
            offs.append(o)
            flat[o:o + h * w * 3].copy_(im.reshape(-1))
            o += h * w * 3
        dev = self.device
        return ops.image_augment(flat.to(dev, non_blocking=True), torch.tensor(offs, dtype=torch.int64).pin_memory().to(dev, non_blocking=True),
                                 torch.tensor(sizes, dtype=torch.int32).pin_memory().to(dev, non_blocking=True), params, self.out_size)

    def __len__(self):
        return len(self.loader)

    def set_epoch(self, epoch):
        if hasattr(self.loader, "set_epoch"):
            self.loader.set_epoch(epoch)

    def _stage(self, image, text, stream):
        cuda = self.device.type == "cuda"
        if isinstance(text, (list, tuple)) and text and isinstance(text[0], str):
            tokens, lengths, pad = self.tokenize(list(text), return_length=True)
            text = (tokens, pad, lengths.tolist())
        flags = None
        decoded = isinstance(image, (tuple, list)) and len(image) > 0 and all(torch.is_tensor(t) and t.dim() == 3 for t in image) and \
            not (len(image) == 2 and image[1].dim() == 1)
        if isinstance(image, (tuple, list)) and not decoded:
            image, flags = image
        if cuda and decoded:
            with torch.cuda.stream(stream):
                image = self._augment(list(image))
                if isinstance(text, tuple) and torch.is_tensor(text[0]):
                    text = tuple(t.pin_memory().to(self.device, non_blocking=True) if torch.is_tensor(t) else t for t in text)
                done = torch.cuda.Event()
                done.record(stream)
            return image, text, done
        if cuda:
            with torch.cuda.stream(stream):
                image = (image if image.is_pinned() else image.pin_memory()).to(self.device, non_blocking=True)
                if image.dtype == torch.uint8:
                    from . import ops
                    if flags is not None:
                        flags = flags.to(torch.uint8).pin_memory().to(self.device, non_blocking=True)
                    image = ops.image_u8_normalize(image.contiguous(), flags=flags)     # on this thread's current stream
                if isinstance(text, tuple) and torch.is_tensor(text[0]):
                    text = tuple(t.pin_memory().to(self.device, non_blocking=True) if torch.is_tensor(t) else t for t in text)
                done = torch.cuda.Event()
                done.record(stream)
            return image, text, done
        return image, text, None

    def __iter__(self):
        import queue
        import threading
        q = queue.Queue(maxsize=self.depth)
        stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None
        stop = object()

        def work():
            try:
                for image, text in self.loader:
                    q.put(self._stage(image, text, stream))
                q.put(stop)
            except Exception as e:          # re-raised in the consumer
                q.put(e)

        worker = threading.Thread(target=work, name="ilvlm-prefetch", daemon=True)
        worker.start()
        while True:
            item = q.get()
            if item is stop:
                break
            if isinstance(item, Exception):
                raise item
            image, text, done = item
            if done is not None:
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(done)                                        # order the step after the copies
                image.record_stream(cur)
                if isinstance(text, tuple):
                    for t in text:
                        if torch.is_tensor(t) and t.is_cuda:
                            t.record_stream(cur)
            yield image, text
        worker.join()


class AsyncCheckpointWriter:
    """Checkpoint I/O off the training thread (the reference blocks every rank in torch.save + barrier,
    train_solver.py:521-543).  save() snapshots the state on the device (a device-to-device copy, ordered on the training
    stream, ~1 ms for 1.8 GB) and returns; a writer thread copies the snapshot to the host on its own stream and writes
    the files.  Same {'model', 'optimizer', 'last_iter'} layout, byte-compatible with torch.load."""

    def __init__(self):
        self._thread = None
        self._stream = None
        self.error = None

    @staticmethod
    def _map(obj, fn):
        if torch.is_tensor(obj):
            return fn(obj)
        if isinstance(obj, dict):
            return type(obj)((k, AsyncCheckpointWriter._map(v, fn)) for k, v in obj.items())
        if isinstance(obj, (list, tuple)):
            return type(obj)(AsyncCheckpointWriter._map(v, fn) for v in obj)
        return obj

    def save(self, state, paths):
        import threading
        self.wait()
        snap = self._map(state, lambda t: t.detach().clone() if t.is_cuda else t.detach().clone())
        dev = next((t.device for t in self._tensors(snap) if t.is_cuda), None)
        if dev is not None:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=dev)
            self._stream.wait_stream(torch.cuda.current_stream(dev))       # after the snapshot copies

        def work():
            try:
                if dev is not None:
                    with torch.cuda.stream(self._stream):
                        host = self._map(snap, lambda t: t.to("cpu", non_blocking=True) if t.is_cuda else t)
                    self._stream.synchronize()
                else:
                    host = snap
                for path in paths:
                    tmp = path + ".tmp"
                    torch.save(host, tmp)
                    os.replace(tmp, path)
            except Exception as e:       # surfaced by wait()
                self.error = e

        self._thread = threading.Thread(target=work, name="ilvlm-ckpt-writer", daemon=False)
        self._thread.start()

    @staticmethod
    def _tensors(obj):
        if torch.is_tensor(obj):
            yield obj
        elif isinstance(obj, dict):
            for v in obj.values():
                yield from AsyncCheckpointWriter._tensors(v)
        elif isinstance(obj, (list, tuple)):
            for v in obj:
                yield from AsyncCheckpointWriter._tensors(v)

    def wait(self):
        if self._thread is not None:
            self._thread.join()
            self._thread = None
        if self.error is not None:
            err, self.error = self.error, None
            raise RuntimeError("checkpoint writer failed: %r" % (err,))


class ClsSolver:
    def __init__(self, args, train_data=None):
        self.args = args
        self.config = parse_config(args.config)
        self.config.output_path = args.output_path
        self.config.data.train.batch_size = args.batch_size
        self.fdt = "fdt" in self.config.model.kwargs
        if "reset" not in self.config:
            self.config.reset = EasyDict(enable=False, reset_steps=0, reset_nums=0, smooth_steps=0)
        self.train_data = train_data
        self.setup_env()
        self.build_model()
        self.build_optimizer()
        self.build_data()
        self.build_lr_scheduler()

    # ------------------------------------------------------------------------------------------
    def setup_env(self):
        D.set_random_seed()
        self.rank, self.world_size, self.local_rank = D.get_rank(), D.get_world_size(), D.get_local_rank()
        rs = self.config.reset
        exp = (self.args.exp_name or "") + "_Reset_%s_steps_%s_smooth_%s" % (rs.enable, rs.reset_steps, rs.smooth_steps)
        if self.args.debug:
            exp += "_debug"
        self.output_path = os.path.join(self.config.output_path, exp)
        self.save_path = os.path.join(self.output_path, "checkpoints")
        self.result_path = os.path.join(self.output_path, "results")
        for p in (self.output_path, self.save_path, self.result_path):
            makedir(p)
        create_logger(os.path.join(self.output_path, "log.txt"))
        self.logger = get_logger(__name__)
        self.logger.critical("config: %s" % pprint.pformat(self.config))
        if self.rank == 0:
            with open(os.path.join(self.output_path, "config.json"), "w") as f:
                json.dump(self.config, f)
        if self.args.ckpt_path:
            self.state = torch.load(self.args.ckpt_path, map_location="cpu", weights_only=False)
            self.logger.info("load ckpt from %s" % self.args.ckpt_path)
        else:
            self.state = {"last_iter": 0}

    def build_model(self):
        self.model = model_entry(self.config.model)
        self.model.cuda()
        count_params(self.model)
        self.model = D.convert_to_ddp_model(self.model, self.local_rank)
        if "model" in self.state:
            load_state_model(self.model, self.state["model"])

    def build_optimizer(self):
        oc = self.config.optimizer
        oc.kwargs.lr = self.config.lr_scheduler.kwargs.base_lr
        pconfig = {}
        if oc.get("no_wd", False):
            for k in ("conv_b", "linear_b", "bn_w", "bn_b", "ln_w", "ln_b"):
                pconfig[k] = {"weight_decay": 0.0}
        if "pconfig" in oc:
            pconfig.update(oc["pconfig"])
        kwargs = dict(oc.kwargs)
        kwargs["params"] = param_group_all(self.model, pconfig)[0]
        self.optimizer = optim_entry(dict(type=oc.type, kwargs=kwargs))
        # the loop below calls zero_grad() right before the one backward of a step and nothing reads gradients after step():
        # the optimizer may zero the arena itself, beside the next forward (ILVLM_PREZERO=0 restores the memset in zero_grad)
        self.optimizer.prezero_grads = os.environ.get("ILVLM_PREZERO", "1") == "1"
        # train_step() below keeps the reference's order (zero_grad, one backward, step, nothing touching gradients in
        # between), which is what the in-backward update needs
        # (opt-in: on one GPU it measured +-0, the chip is already full during backward)
        if os.environ.get("ILVLM_ADAMW_IN_BACKWARD", "0") == "1":
            if self.config.grad_clip.type in self.GRAD_CLIPS:
                raise ValueError("ILVLM_ADAMW_IN_BACKWARD=1 updates blocks before backward ends; grad_clip.type=%r needs the "
                                 "complete gradient first" % self.config.grad_clip.type)
            self.optimizer.overlap_backward(True)
        if "optimizer" in self.state and not self.fdt:
            # the baseline solver restores optimizer state (example/clip/train_solver.py:279-280); the FDT solver does not
            load_state_optimizer(self.optimizer, self.state["optimizer"])

    def build_lr_scheduler(self):
        kw = dict(self.config.lr_scheduler.kwargs)
        kw.setdefault("max_iter", self.config.data.get("max_iter", 0))
        kw["last_iter"] = self.state["last_iter"]
        kw["reset_steps"] = self.config.reset.reset_steps if self.fdt else kw.get("reset_steps", 0)
        kw["optimizer"] = self.optimizer
        self.lr_scheduler = scheduler_entry(dict(type=self.config.lr_scheduler.type, kwargs=kw))

    def build_data(self):
        if self.train_data is not None:
            if os.environ.get("ILVLM_PREFETCH", "1") == "1" and not isinstance(self.train_data, (SyntheticPairs, DevicePrefetcher)):
                self.train_data = DevicePrefetcher(getattr(self.train_data, "dataloader", self.train_data),
                                                   self.model.module.encode_text.tokenize,
                                                   torch.device("cuda", self.local_rank))
            return
        tc = self.config.data.train
        if getattr(self.args, "synthetic", False) or tc.get("synthetic", False):
            m = self.model.module
            nb = self.args.max_steps or max(1, tc.num_samples // (tc.batch_size * self.world_size))
            self.train_data = SyntheticPairs(tc.batch_size, nb, m.visual.input_resolution, m.encode_text.context_length,
                                             torch.device("cuda", self.local_rank))
            return
        raise RuntimeError("the webdataset input pipeline of the reference (prototype/data) is outside the accelerated hot "
                           "path: pass --synthetic, or construct ClsSolver(args, train_data=<iterable of (image, text)>)")

    # ------------------------------------------------------------------------------------------
    def store_codebook_value(self):
        self.stored_codebook = self.model.module.space_dict.data.clone()

    def keep_codebook_value(self):
        """Pin the codebook to its stored value (train_solver.py:214,553).  The write goes through `.data`, which neither the
        version counters nor the storage check of Engine.prepare() see, and AdamW has just written the bf16 shadow of the
        post-step codebook: without mark_dirty() the FDT score GEMM would keep reading the un-pinned values."""
        m = self.model.module
        m.space_dict.data.copy_(self.stored_codebook)
        eng = getattr(m, "engine", None)
        if eng is not None:
            eng.mark_dirty()

    # grad_clip.type, all of the reference's (train_solver.py:374-415, 467-470), every one without a host read:
    #   parameter clips   logit_scale_param_value | logit_scale_param_abs_min | constant | logit_scale_param (the step may move
    #                     logit_scale by at most `value`) | logit_scale_param_ema (EMA_logit_scale, train_solver.py:61-83)
    #   gradient clips    norm | value | logit_scale_grad  (prototype/utils/grad_clip.py), applied between backward and step
    PARAM_CLIPS = ("logit_scale_param_value", "logit_scale_param_abs_min", "constant", "logit_scale_param", "logit_scale_param_ema")
    GRAD_CLIPS = ("norm", "value", "logit_scale_grad")

    def _param_clip_before(self):
        gc = self.config.grad_clip
        from . import ops
        ls = self.model.module.logit_scale
        if gc.type == "logit_scale_param_value":
            ops.clamp_(ls.data, gc.value, gc.max_value)
        elif gc.type == "logit_scale_param_abs_min":
            ops.clamp_(ls.data, gc.value, float("inf"))
        elif gc.type == "constant":
            # The reference sets requires_grad = False HERE, between forward and backward (train_solver.py:374-375).  In its
            # first step the autograd graph of the forward already holds logit_scale as a leaf that requires a gradient, so
            # that step still updates it once; from the second step on it is frozen.  The engine decides what to differentiate
            # at backward time, so the flag is flipped after the first step's update to give the same trajectory.
            if getattr(self, "_ls_const_frozen", False):
                ls.requires_grad = False
            else:
                self._ls_freeze_after_step = True
        elif gc.type == "logit_scale_param":
            self._ls_before = ls.data.clone()                       # the reference reads .item() here: a host sync per step
        elif gc.type not in self.GRAD_CLIPS and gc.type != "logit_scale_param_ema":
            raise NotImplementedError("grad_clip.type=%r" % gc.type)

    def _param_clip_after(self):
        gc = self.config.grad_clip
        from . import ops
        ls = self.model.module.logit_scale
        if gc.type == "logit_scale_param_value":
            ops.clamp_(ls.data, gc.value, gc.max_value)
        elif gc.type == "logit_scale_param_abs_min":
            ops.clamp_(ls.data, gc.value, float("inf"))
        elif gc.type == "logit_scale_param":
            # after - before > value -> before + value; before - after > value -> before - value: a clamp around `before`
            ls.data.copy_(torch.minimum(torch.maximum(ls.data, self._ls_before - gc.value), self._ls_before + gc.value))
        elif gc.type == "constant" and getattr(self, "_ls_freeze_after_step", False):
            ls.requires_grad = False                  # frozen from the second step on (see _param_clip_before)
            self._ls_freeze_after_step, self._ls_const_frozen = False, True

    def _grad_clip_before(self):
        """between backward and optimizer.step() (train_solver.py:402-411, 431), on the flat gradient arena"""
        gc = self.config.grad_clip
        if gc.type not in self.GRAD_CLIPS:
            return
        from . import ops
        eng = self.model.module.engine
        arena = eng.arena
        arena.wait_grads()                  # the data-parallel mean of every range has to be in before gradients are judged
        if gc.type == "norm":
            # parameters without a gradient hold zeros in the arena and add nothing to the norm (the reference skips them)
            self.grad_norm_sq = ops.clip_grad_norm_(arena.G, gc.value, getattr(self, "grad_norm_sq", None))
        elif gc.type == "value":
            ops.clamp_(arena.G, -gc.value, gc.value)
        else:
            ops.clamp_(eng.Gr["logit_scale"].view(-1), -gc.value, gc.value)

    def _ema_clip(self):
        """EMA_logit_scale.clamp() then .update() at the end of the iteration (train_solver.py:61-83, 467-470): logit_scale may
        stray at most `value` from its running mean (momentum 0.9, start 3.125); clip_number counts the clamps on the device"""
        gc = self.config.grad_clip
        if gc.type != "logit_scale_param_ema":
            return
        ls = self.model.module.logit_scale
        if getattr(self, "_ema_buf", None) is None:
            extra = self.state.get("solver_extra") or {}             # resumed run: continue the running mean and the count
            self._ema_buf = (extra["ema_logit_scale"].to(ls.device).to(ls.dtype).reshape(ls.shape).clone()
                             if "ema_logit_scale" in extra else torch.full_like(ls.data, 3.125))
            self.clip_number = torch.full((), int(extra.get("clip_number", 0)), dtype=torch.int64, device=ls.device)
        lo, hi = self._ema_buf - gc.value, self._ema_buf + gc.value
        self.clip_number += ((ls.data > hi) | (ls.data < lo)).sum()
        ls.data.copy_(torch.minimum(torch.maximum(ls.data, lo), hi))
        self._ema_buf.mul_(0.9).add_(ls.data, alpha=0.1)

    def _temperature(self, curr_step):
        td = self.config.get("t_decay")
        if not (self.fdt and td) or curr_step % td.sd_T_decay_iter:
            return
        t = max(td.org_t * (td.sd_T_decay_w ** (curr_step / td.sd_T_decay_iter)), td.sd_T_min)
        self.model.module.img_query_model.temperature = t
        self.model.module.txt_query_model.temperature = t

    def train_step(self, image, text, curr_step):
        """One optimisation step in the reference's order (train_solver.py:348-439)."""
        self.lr_scheduler.step(curr_step)
        self._temperature(curr_step)
        image = image.cuda(non_blocking=True)
        out = self.model(image, text)
        logits = out[0] if self.fdt else out
        loss, target = self.criterion(logits[0], logits[1])
        if self.world_size > 1:         # reference: unconditional (train_solver.py:420); at one rank it is x / 1
            loss = loss / self.world_size
        prec1, prec5 = accuracy(logits[0], target, topk=(1, self.topk))
        self.optimizer.zero_grad()
        self._param_clip_before()
        loss.backward()
        self._grad_clip_before()
        self.optimizer.step()
        self._param_clip_after()
        self._ema_clip()
        return loss, prec1 / self.world_size, prec5 / self.world_size

    def save_checkpoint(self, curr_step):
        if self.rank == 0:
            name = "ckpt_%d.pth.tar" % curr_step if self.config.saver.save_many else "ckpt.pth.tar"
            self.state["model"] = self.model.state_dict()
            self.state["optimizer"] = self.optimizer.state_dict()
            self.state["last_iter"] = curr_step
            # solver-side state of grad_clip.type = logit_scale_param_ema / constant (an extra key: the reference keeps neither, so
            # its EMA restarts at 3.125 on resume; files stay loadable by the reference, which reads the three keys above)
            extra = {}
            if getattr(self, "_ema_buf", None) is not None:
                extra.update(ema_logit_scale=self._ema_buf.detach().cpu().clone(), clip_number=int(self.clip_number))
            if getattr(self, "_ls_const_frozen", False):
                extra["logit_scale_frozen"] = True
            if extra:
                self.state["solver_extra"] = extra
            paths = [os.path.join(self.save_path, name)]
            if curr_step % (self.config.saver.save_freq * 10) == 0:
                k_path = self.save_path + "_k_times"
                os.makedirs(k_path, exist_ok=True)
                paths.append(os.path.join(k_path, "ckpt_%d.pth.tar" % curr_step))
            if os.environ.get("ILVLM_ASYNC_CKPT", "1") == "1":
                if getattr(self, "ckpt_writer", None) is None:
                    self.ckpt_writer = AsyncCheckpointWriter()
                self.ckpt_writer.save(self.state, paths)      # returns after a device-side snapshot
            else:
                for path in paths:
                    torch.save(self.state, path)
        link.barrier()

    def finish_checkpoints(self):
        """block until the last checkpoint is on disk (called at the end of train())"""
        w = getattr(self, "ckpt_writer", None)
        if w is not None:
            w.wait()

    def iterated_learning(self, curr_step, start_step):
        """train_solver.py:545-557 with the two documented fixes (module docstring)."""
        rs = self.config.reset
        if not (self.fdt and rs.enable and rs.reset_steps < curr_step < rs.reset_steps * rs.reset_nums):
            return
        m = self.model.module
        phase = curr_step % rs.reset_steps
        if phase == 0 or (curr_step == start_step + 1 and phase < rs.smooth_steps):
            self.store_codebook_value()
            m.reset_text_encoder(curr_step)
            self.logger.info("step %d: reset text encoder" % curr_step)
        elif phase < rs.smooth_steps:
            self.keep_codebook_value()
        if phase == rs.smooth_steps:
            m.freeze_unfreeze_vision_weights(unfreeze=True, freeze_codebook=False)
            m.train()          # re-freezes conv1 exactly as the next train()/eval() call does in the reference
            self.logger.info("step %d: unfreeze vision encoder" % curr_step)

    def train(self):
        cfg = self.config
        dev = torch.device("cuda", self.local_rank)
        self.model.train()
        self.criterion = ClipInfoCELoss()
        self.model.module.find_always_freeze_weight()
        self.topk = 5
        pf = cfg.saver.print_freq
        meters = {k: DeviceMeter(pf, dev) for k in ("loss", "top1", "top5")}
        loader = getattr(self.train_data, "dataloader", self.train_data)     # the reference's wrapper or a plain iterable
        each_epoch = getattr(loader, "num_batches", None) or len(loader)
        total_step = cfg.data.train.epoch * each_epoch
        start_step = curr_step = self.state["last_iter"]
        end = time.time()
        losses = []
        for epoch_id in range(cfg.data.train.epoch):
            if hasattr(self.train_data, "set_epoch"):
                self.train_data.set_epoch(epoch_id)
            for image, text in loader:
                curr_step += 1
                loss, p1, p5 = self.train_step(image, text, curr_step)
                meters["loss"].update(loss); meters["top1"].update(p1); meters["top5"].update(p5)
                if curr_step % pf == 0:
                    lv, la = meters["loss"].read()
                    _, t1 = meters["top1"].read()
                    _, t5 = meters["top5"].read()
                    bt = (time.time() - end) / pf
                    end = time.time()
                    ls = float(self.model.module.logit_scale.detach())
                    losses.append(la)
                    if self.rank == 0:
                        self.logger.critical(
                            "Iter: [%d/%d]\tTime %.3f\tLoss_all %.4f (%.4f)\tPrec@1 (%.3f)\tPrec@5 (%.3f)\tLR %.6f\t"
                            "logit_scale_exp %.4f\tlogit_scale %.4f\tpairs/s %.1f" % (
                                curr_step, total_step, bt, lv, la, t1, t5, self.lr_scheduler.get_lr()[0],
                                float(torch.tensor(ls).exp()), ls, cfg.data.train.batch_size * self.world_size / max(bt, 1e-9)))
                if curr_step % cfg.saver.save_freq == 0 or curr_step == total_step:
                    self.save_checkpoint(curr_step)
                self.iterated_learning(curr_step, start_step)
                if self.args.max_steps and curr_step - start_step >= self.args.max_steps:
                    self.finish_checkpoints()
                    return losses
        self.finish_checkpoints()
        return losses


def main(argv=None):
    ap = argparse.ArgumentParser(description="CLIP / CLIP+FDT training on MI355X")
    ap.add_argument("--config", required=True, type=str)
    ap.add_argument("--output_path", default="./output", type=str)
    ap.add_argument("--batch_size", default=256, type=int)
    ap.add_argument("--debug", action="store_true")
    ap.add_argument("--exp_name", default="", type=str)
    ap.add_argument("--ckpt_path", default="", type=str)
    ap.add_argument("--lipreg", default=0, type=float, help="accepted for CLI compatibility with example/clip; must be 0")
    ap.add_argument("--synthetic", action="store_true", help="train on synthetic pairs resident on the device")
    ap.add_argument("--max_steps", default=0, type=int, help="stop after this many steps (0 = run the configured epochs)")
    args = ap.parse_args(argv)
    if args.lipreg:
        raise NotImplementedError("--lipreg (Lipschitz regulariser, off by default in the reference) is not on the hot path")
    if D.get_world_size() > 1:
        D.init_ddp()
    else:
        torch.cuda.set_device(D.get_local_rank())
        logging.getLogger().setLevel(logging.INFO)
    solver = ClsSolver(args)
    solver.train()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

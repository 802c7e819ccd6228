"""Shared host logic of the two contrastive models: engine ownership, text input handling, the single
autograd node that wraps a whole forward/backward of the hot path."""
import os

import torch
from torch import nn

from ... import ops
from ...engine import Engine


def default_precision():
    return os.environ.get("ILVLM_PRECISION", "bf16")


class _StepFn(torch.autograd.Function):
    """One autograd node for the whole model forward.  Parameters are inputs only so that autograd schedules the
    backward; their gradients are accumulated straight into the gradient arena by the kernels (returned as None)."""

    @staticmethod
    def forward(ctx, model, images, tokens, pad_mask, seq, *params):
        li, lt, saved = model._forward_impl(images, tokens, pad_mask, True, seq)
        ctx.model, ctx.saved, ctx.n = model, saved, len(params)
        f8 = model._eng.fp8
        ctx.fp8_active = None if f8 is None else f8.active     # the saved buffers have THIS mode's layout
        ctx.fp8_steps = None if f8 is None else f8.steps       # ... and were quantised with THIS step's scales
        return li, lt

    @staticmethod
    def backward(ctx, dli, dlt):
        saved, ctx.saved = ctx.saved, None
        if saved is None:
            raise RuntimeError("ilvlm: backward through the same forward twice is not supported")
        ctx.model._eng.arena.finish_zero_grad()          # a gradient memset still queued behind a deferred optimizer update
        li = saved["head"][9]
        dli = torch.zeros_like(li) if dli is None else dli
        dlt = torch.zeros_like(li) if dlt is None else dlt
        f8 = ctx.model._eng.fp8
        if f8 is None:
            ctx.model._backward_impl(saved, dli, dlt)
        else:
            # The saved e4m3 activation copies and the fp8 weight copies are de-quantised with the scales the state holds NOW.
            # A later TRAINING forward has rewritten them (and re-quantised the weights in place): this backward would produce
            # silently wrong gradients.  The reference's loop (one forward, one backward) never does that; refuse rather than
            # guess.  (Inference forwards in between are fine: they advance nothing.)
            if ctx.fp8_active and f8.steps != ctx.fp8_steps:
                raise RuntimeError("ilvlm fp8: backward of training forward #%d after training forward #%d has replaced its delayed "
                                   "scales and fp8 weight copies; run each backward before the next training forward (or use "
                                   "precision bf16 for multi-forward schemes)" % (ctx.fp8_steps, f8.steps))
            # run the backward in the fp8 mode of ITS forward (another forward may have switched the state since)
            now, f8.active = f8.active, ctx.fp8_active
            try:
                ctx.model._backward_impl(saved, dli, dlt)
            finally:
                f8.active = now
            f8.bwd_seen += 1
        return (None,) * (5 + ctx.n)


def _prefix_lengths(pad_mask):
    """lengths of a host-side pad mask whose valid positions (0) form a prefix of every row; None otherwise"""
    valid = (pad_mask == 0)
    lens = valid.sum(1)
    if bool((valid.int().cumprod(1).sum(1) == lens).all()) and int(lens.min()) >= 1:
        return lens.tolist()
    return None


class ContrastiveBase(nn.Module):
    def _init_engine(self, cfg):
        import weakref
        object.__setattr__(self, "_eng", Engine(self, cfg))
        for enc in (self.visual, self.encode_text):          # encoders reach the engine for their inference-only calls
            object.__setattr__(enc, "_owner", weakref.ref(self))
        object.__setattr__(self, "_grad_sync", None)     # set by the data-parallel wrapper

    @property
    def engine(self):
        return self._eng

    # Training steps run the text tower on the valid tokens only (include/ilvlm_hip.h, "packed text rows") whenever the
    # caption lengths are known on the host without a device synchronisation: captions given as strings (tokenised here),
    # (tokens, pad_mask) given as CPU tensors, or a third element -- the lengths (list / CPU tensor) or a ready
    # ops.PackedSeq -- next to device tensors.  ILVLM_TEXT_PACK=0 or model.pack_text = False keeps every position.
    pack_text = os.environ.get("ILVLM_TEXT_PACK", "1") != "0"

    def _text_inputs(self, texts, device, want_seq=False):
        """list[str] (tokenised here, as the reference does inside forward) or a (tokens, pad_mask[, lengths]) tuple.
        Returns tokens, pad_mask on the device (+ the packed-row descriptor or None when want_seq)."""
        lengths = None
        if isinstance(texts, (tuple, list)) and len(texts) in (2, 3) and torch.is_tensor(texts[0]):
            tokens, pad_mask = texts[0], texts[1]
            if len(texts) == 3:
                lengths = texts[2]
            elif want_seq and not pad_mask.is_cuda:
                lengths = _prefix_lengths(pad_mask)
        elif hasattr(texts, "out1"):
            tokens, pad_mask = texts.out1, texts.out2
            if want_seq and not pad_mask.is_cuda:
                lengths = _prefix_lengths(pad_mask)
        else:
            tokens, lengths, pad_mask = self.encode_text.tokenize(texts, context_length=self.encode_text.context_length,
                                                                  return_length=True)
        tokens = tokens.to(device=device, dtype=torch.int64).contiguous()
        pad_mask = pad_mask.to(device=device, dtype=torch.float32).contiguous()
        if not want_seq:
            return tokens, pad_mask
        seq = None
        if self.pack_text and lengths is not None and tokens.shape[1] <= 128:
            if isinstance(lengths, ops.PackedSeq):
                seq = lengths
            else:
                seq = ops.PackedSeq(lengths.tolist() if torch.is_tensor(lengths) else lengths, tokens.shape[1], device)
        return tokens, pad_mask, seq

    def _run(self, images, texts):
        eng = self._eng
        training = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        eng.prepare(training=training)
        dev = eng.arena.P.device
        if not images.is_cuda:
            raise RuntimeError("images must be on the GPU (the solver calls image.cuda())")
        params = [p for _, p in eng.arena.named]
        if training:
            tokens, pad_mask, seq = self._text_inputs(texts, dev, want_seq=True)
            return _StepFn.apply(self, images, tokens, pad_mask, seq, *params)
        tokens, pad_mask = self._text_inputs(texts, dev)
        li, lt, _ = self._forward_impl(images, tokens, pad_mask, False)
        return li, lt

    def all_gather(self, input):
        from ... import comm
        return comm.gather_pair(input, input)[0]

    def zero_grad(self, set_to_none=False):
        if self._eng.arena is not None:
            self._eng.arena.zero_grad()
        else:
            super().zero_grad(set_to_none=set_to_none)

    def _sync(self, what):
        if self._grad_sync is not None:
            self._grad_sync(what)
        # optimizer.overlap_backward(): the block's update starts now, on the optimizer's stream
        opt = getattr(self._eng.arena, "eager_opt", None)
        if opt is not None and what.endswith("."):
            opt.block_grads_final(what, self._eng._wg.get(torch.cuda.current_stream().cuda_stream))

"""Host-side execution engine of the CLIP / CLIP+FDT step on MI355X.

The engine owns
  * a flat fp32 PARAMETER ARENA (the nn.Module parameters are views into it, reference names and
    shapes untouched), a flat GRADIENT ARENA (every .grad is a view into it; weight gradients are
    accumulated there directly by the kernels), and a bf16 SHADOW of the parameters that the MFMA
    GEMMs read (refreshed by one cast kernel per forward);
  * the forward / backward composition of the hot path as explicit kernel sequences (no autograd graph
    inside: activations needed by the backward pass are kept in plain lists), one C-ABI call per kernel.
PyTorch supplies device memory, streams and torch.distributed only.

Reference semantics restated here (file:line relative to the reference root):
  VisualTransformer.forward        prototype/model/image_encoder/visual_transformer.py:55-91
  ResidualAttentionBlock.forward   prototype/model/image_encoder/base_transformer.py:50-62 (text twin :50-59)
  TextTransformer.forward          prototype/model/text_encoder/text_transformer.py:211-338
  Query_model.forward              prototype/model/clip_fdt.py:96-161
  Clip_FDT.forward                 prototype/model/clip_fdt.py:390-428
  CLIP.forward                     prototype/model/clip.py:125-149
"""
import contextlib
import math
import os

import torch

from . import ops
from . import lib as _lib
from .lib import (F32, BF16, ACT_QUICKGELU, ACT_GELU_ERF, ACT_QUICKGELU_BWD, ACT_GELU_ERF_BWD, POOL_MAX, POOL_MEAN,
                  POOL_SUM)

ALIGN = 64   # elements; keeps every parameter view 256-byte aligned
POOLS = {"max": POOL_MAX, "mean": POOL_MEAN, "sum": POOL_SUM}


class ParamArena:
    """Flat fp32 parameter / gradient storage with an optional bf16 shadow."""

    def __init__(self, module, precision):
        self.named = [(n, p) for n, p in module.named_parameters()]
        dev = self.named[0][1].device
        if dev.type != "cuda":
            raise RuntimeError("ilvlm: the model must be on the GPU (model.cuda()) before its first forward; "
                               "there is no CPU execution path")
        self.offsets, off = {}, 0
        for n, p in self.named:
            self.offsets[n] = off
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.total = off
        self.P = torch.zeros(off, device=dev, dtype=torch.float32)
        self.G = torch.zeros(off, device=dev, dtype=torch.float32)
        self.S = torch.zeros(off, device=dev, dtype=torch.bfloat16) if precision in ("bf16", "fp8") else None
        self.views, self.gviews, self.sviews = {}, {}, {}
        self.inactive = set()     # parameters the loss never reaches (no gradient, untouched by the optimizer)
        self.shadow_fresh = False # set by the fused optimizer: its kernel wrote the bf16 shadow of what it updated
        self.packed_fresh = False # ... and the fragment-order images of the packed GEMM weights (ilvlm_adamw_step_packed)
        self.packed = None        # engine.PackedWeights of this arena (bf16 mode), for the optimizer
        self._prezero = None      # (event, G version) of a gradient memset issued ahead of zero_grad() (prezero_grads)
        self._zstream = None
        self.late_event = None    # FusedAdamW.defer_late_blocks: the update of the blocks from late_from up, on the side stream
        self.late_from = 0
        self._versions = None
        self.reducer = None       # comm.GradReducer installed by the data-parallel wrapper
        self.eager_opt = None     # FusedAdamW.overlap_backward(): the optimizer that updates blocks from inside backward
        for n, p in self.named:
            o, k = self.offsets[n], p.numel()
            v = self.P[o:o + k].view(p.shape)
            v.copy_(p.data)
            p.data = v
            self.views[n] = v
            self.gviews[n] = self.G[o:o + k].view(p.shape)
            p.grad = self.gviews[n]
            p._ilvlm_arena = (self, n)
            if self.S is not None:
                self.sviews[n] = self.S[o:o + k].view(p.shape)

    def sync_in(self):
        """Re-adopt parameters whose storage was replaced behind our back (p.data = ..., load_state_dict on a
        moved module) and re-attach gradient views dropped by zero_grad(set_to_none=True).  Returns True when a
        parameter's storage had been replaced."""
        swapped = False
        for n, p in self.named:
            v = self.views[n]
            if p.data_ptr() != v.data_ptr():
                v.copy_(p.data.to(v.device, torch.float32))
                p.data = v
                swapped = True
            if p.grad is None or p.grad.data_ptr() != self.gviews[n].data_ptr():
                if p.grad is not None:
                    self.gviews[n].copy_(p.grad)
                p.grad = self.gviews[n]
        return swapped

    def versions(self):
        """autograd version counters of the parameters: every in-place write through the Parameter (load_state_dict,
        reset_parameters, p.copy_, p.clamp_ under no_grad) bumps one"""
        return tuple(p._version for _, p in self.named)

    def refresh_shadow(self):
        if self.S is not None:
            ops.cast_f32(self.P, self.S)

    def zero_grad(self):
        ev = self._prezero
        self._zero_wait = None
        if ev is not None and self.G._version == ev[1]:
            # the optimizer zeroed the arena on a side stream right after its update (FusedAdamW.prezero_grads) and nothing
            # has written gradients through torch since: only order this stream behind that memset -- at once, or, while a
            # deferred update (FusedAdamW.defer_late_blocks) still runs in front of that memset, when backward begins
            # (finish_zero_grad; gradients are only written there): waiting here would hold the forward back behind the update
            if self.late_event is not None:
                self._zero_wait = ev[0]
            else:
                torch.cuda.current_stream(self.G.device).wait_event(ev[0])
            self._prezero = None
            return
        self._prezero = None
        self.wait_late_update()                       # a deferred update may still be reading the gradients
        self.G.zero_()

    def finish_zero_grad(self):
        """called at the start of backward, on the stream backward starts from"""
        ev = getattr(self, "_zero_wait", None)
        if ev is not None:
            torch.cuda.current_stream(self.G.device).wait_event(ev)
            self._zero_wait = None

    def side_stream(self):
        if self._zstream is None:
            self._zstream = torch.cuda.Stream(device=self.G.device)
        return self._zstream

    def wait_late_update(self):
        """order the current stream behind a deferred optimizer update (FusedAdamW.defer_late_blocks); cheap when it is done"""
        if self.late_event is not None:
            torch.cuda.current_stream(self.G.device).wait_event(self.late_event)

    def prezero_grads(self):
        """zero the gradient arena on a side stream, ordered after everything enqueued on the current stream (the optimizer's
        update has read the gradients); the next zero_grad() waits for it instead of issuing its own memset.  Kernel writes
        into the arena only happen in backward, which follows that zero_grad(); writes through torch bump G's version
        counter and make zero_grad() fall back to its own memset."""
        dev = self.G.device
        if self._zstream is None:
            self._zstream = torch.cuda.Stream(device=dev)
        self._zstream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self._zstream):
            self.G.zero_()
            ev = torch.cuda.Event()
            ev.record(self._zstream)
        self._prezero = (ev, self.G._version)

    def reducing(self):
        """True when a data-parallel wrapper exchanges gradients over ranks in this process (a world of one rank does not,
        unless comm.force_collectives is set): only then must each block's gradients be final as soon as its backward ends"""
        from . import comm
        return self.reducer is not None and comm._active(self.reducer.W)

    def wait_grads(self):
        """Order the current stream after any in-flight gradient all-reduce."""
        if self.reducer is not None:
            self.reducer.wait()

    def range_of(self, prefix):
        """[begin, end) element range of the parameters whose name starts with prefix (contiguous by construction)."""
        names = [n for n, _ in self.named if n.startswith(prefix)]
        if not names:
            return (0, 0)
        last = names[-1]
        p = dict(self.named)[last]
        return self.offsets[names[0]], self.offsets[last] + (p.numel() + ALIGN - 1) // ALIGN * ALIGN


class Fp8State:
    """Per-tensor delayed scaling for the fp8 mode (BASELINE.json configs[4]): the QKV / out / MLP GEMMs of both towers run
    on fp8 operands in the forward pass (activations and weights e4m3) and in the input-gradient pass (gradients e5m2,
    transposed e4m3 weights) and, unless ILVLM_FP8_WGRAD=0, in the weight-gradient pass (e5m2 gradients x the e4m3
    activation copies of the forward pass, both K-strided).  Every quantised tensor kind has a SLOT: its scale is derived
    from the amax history of the last HIST steps (ilvlm_fp8_scale_update); the first step runs the bf16 kernels and only
    observes the amaxes (there is no history to scale by yet)."""
    HIST = 16
    ACT, GRAD, WEIGHT = ("h1", "att", "h2", "g"), ("dout", "du", "dmid", "dqkv"), ("in_w", "out_w", "fc_w", "proj_w")
    WNAME = dict(in_w="attn.in_proj_weight", out_w="attn.out_proj.weight", fc_w="mlp.c_fc.weight", proj_w="mlp.c_proj.weight")

    def __init__(self, arena, block_prefixes):
        dev = arena.P.device
        self.arena = arena
        self.slots = {}
        fmt_max = []
        table = []
        for pre in block_prefixes:
            for k in self.ACT + self.WEIGHT:
                self.slots[pre + k] = len(fmt_max); fmt_max.append(448.0)
            for k in self.GRAD:
                self.slots[pre + k] = len(fmt_max); fmt_max.append(57344.0)
            for k in self.WEIGHT:
                name = pre + self.WNAME[k]
                r, c = arena.views[name].shape
                if r % 64 or c % 64 or arena.offsets[name] % 64:
                    raise RuntimeError("fp8 mode: weight %s [%d,%d] is not a multiple of 64 x 64" % (name, r, c))
                for r0 in range(0, r, 64):
                    for c0 in range(0, c, 64):
                        table.append((arena.offsets[name] // 64, r, c, self.slots[pre + k], r0, c0))
        n = len(fmt_max)
        self.n = n
        self.fmt_max = torch.tensor(fmt_max, dtype=torch.float32).to(dev)
        self.amax = torch.zeros(n, dtype=torch.float32, device=dev)
        self.hist = torch.zeros((n, self.HIST), dtype=torch.float32, device=dev)
        self.scale = torch.ones(n, dtype=torch.float32, device=dev)
        self.inv = torch.ones(n, dtype=torch.float32, device=dev)
        self.table = torch.tensor(table, dtype=torch.int32).to(dev)
        self.W8 = torch.zeros(arena.total, dtype=torch.uint8, device=dev)
        self.W8T = torch.zeros(arena.total, dtype=torch.uint8, device=dev)
        # fragment-order copies for the streaming kernel (ILVLM_FP8_PACK=0: the direct-to-LDS kernel on W8 / W8T)
        self.pack = os.environ.get("ILVLM_FP8_PACK", "1") != "0" and all(r % 128 == 0 and c % 128 == 0 for _, r, c, _, _, _ in table)
        self.W8P = torch.zeros(arena.total, dtype=torch.uint8, device=dev) if self.pack else None
        self.W8TP = torch.zeros(arena.total, dtype=torch.uint8, device=dev) if self.pack else None
        self.steps = 0            # training forwards begun = scale updates done; 0 = nothing observed yet
        self.bwd_seen = 0         # backward passes observed (their gradient amaxes are what the e5m2 scales come from)
        self.w_observed = False   # weight amaxes recorded once
        self.active = False       # False: observe only (bf16 GEMMs); True: fp8 GEMMs

    def _quantize_weights(self, st):
        if self.pack:
            ops.L.check(ops.L.load().ilvlm_fp8_quantize_weights_packed(
                self.arena.P.data_ptr(), self.W8.data_ptr(), self.W8T.data_ptr(), self.W8P.data_ptr(), self.W8TP.data_ptr(),
                self.table.data_ptr(), self.table.shape[0], self.scale.data_ptr(), self.amax.data_ptr(), st), "fp8_quantize_weights_packed")
            return
        ops.L.check(ops.L.load().ilvlm_fp8_quantize_weights(self.arena.P.data_ptr(), self.W8.data_ptr(), self.W8T.data_ptr(),
                                                            self.table.data_ptr(), self.table.shape[0], self.scale.data_ptr(),
                                                            self.amax.data_ptr(), st), "fp8_quantize_weights")

    def _scale_update(self, st, pos):
        ops.L.check(ops.L.load().ilvlm_fp8_scale_update(self.amax.data_ptr(), self.hist.data_ptr(), self.scale.data_ptr(),
                                                        self.inv.data_ptr(), self.fmt_max.data_ptr(), self.n, self.HIST,
                                                        pos, st), "fp8_scale_update")

    def begin_step(self, training):
        """Once per forward.  Only a TRAINING forward (gradients enabled) advances the delayed-scaling state: the amaxes the
        previous step recorded enter the history, the scales follow, the weights are re-quantised.  A no-grad forward
        (`encode_image`, `extract_*`, the data-parallel wrapper's constructor) advances nothing -- it would otherwise consume
        the observe-only step with every activation / gradient amax still 0 and the first real step would quantise e5m2
        gradients at scale 1, i.e. flush most of them to zero.  It only re-quantises the weights with the scales in force
        (they may have been loaded or reset since); on the very first call the weight scales come from an observe pass whose
        history slot the first training step rewrites.
        fp8 GEMMs are switched on (`active`) once a training forward AND a backward have been observed, so that every
        activation and every gradient slot has a non-empty history."""
        st = ops._stream()
        if not self.w_observed:   # weights: observe their amax first so that the very first quantisation is scaled
            self._quantize_weights(st)
            if not training:
                self._scale_update(st, self.steps % self.HIST)
            self.w_observed = True
        if training:
            self._scale_update(st, self.steps % self.HIST)
        self._quantize_weights(st)
        if training:
            self.active = self.steps >= 1 and self.bwd_seen >= 1
            self.steps += 1

    def s(self, key):
        i = self.slots[key]
        return self.scale[i:i + 1], self.inv[i:i + 1], self.amax[i:i + 1]

    def w8(self, pre, k, transposed=False):
        name = pre + self.WNAME[k]
        o = self.arena.offsets[name]
        r, c = self.arena.views[name].shape
        return (self.W8T[o:o + r * c].view(c, r) if transposed else self.W8[o:o + r * c].view(r, c))

    def w8p(self, pre, k, transposed=False):
        """fragment-order copy of w8(...) (flat), None without the packed copies"""
        if not self.pack:
            return None
        name = pre + self.WNAME[k]
        o = self.arena.offsets[name]
        r, c = self.arena.views[name].shape
        return (self.W8TP if transposed else self.W8P)[o:o + r * c]

    def quant(self, x, key, e5m2=False):
        """fp8 copy of x for slot `key` (None while only observing); always records the amax"""
        sc, _, am = self.s(key)
        if not self.active:
            ops.fp8_quantize(x, None, None, am, e5m2)
            return None
        return ops.fp8_quantize(x, torch.empty(x.shape, dtype=torch.uint8, device=x.device), sc, am, e5m2)


class PackedWeights:
    """Fragment-order copies of the GEMM weights of the residual attention blocks (bf16 mode): the forward image (B operand of
    x W^T) and the input-gradient image (B operand of dY W) of every [out, in] weight, at the weight's own arena offset in
    two arena-shaped bf16 buffers.  One launch (ilvlm_pack_weights) re-packs all of them from the bf16 shadow; the eight
    store-type GEMMs of a block then run the streaming kernel (include/ilvlm_hip.h, ilvlm_gemm_epilogue.b_packed), which
    reads them in whole 1 KiB wave loads straight into MFMA operand registers."""
    WNAME = Fp8State.WNAME

    def __init__(self, arena, block_prefixes):
        dev = arena.P.device
        self.arena = arena
        table = []
        for pre in block_prefixes:
            for k, wn in self.WNAME.items():
                name = pre + wn
                r, c = arena.views[name].shape
                if r % 64 or c % 64 or arena.offsets[name] % 64:
                    raise RuntimeError("packed weights: %s [%d,%d] is not a multiple of 64 x 64" % (name, r, c))
                for r0 in range(0, r, 64):
                    for c0 in range(0, c, 64):
                        table.append((arena.offsets[name] // 64, r, c, r0, c0))
        self.table = torch.tensor(table, dtype=torch.int32).to(dev)
        self.names = frozenset(pre + wn for pre in block_prefixes for wn in self.WNAME.values())
        self.fwd = torch.zeros(arena.total, dtype=torch.bfloat16, device=dev)
        self.bwd = torch.zeros(arena.total, dtype=torch.bfloat16, device=dev)

    def refresh(self):
        ops.pack_weights(self.arena.S, self.fwd, self.bwd, self.table)

    def view(self, name, backward=False):
        o = self.arena.offsets[name]
        n = self.arena.views[name].numel()
        return (self.bwd if backward else self.fwd)[o:o + n]


class TowerSaved:
    """what ilvlm_tower_fwd leaves for the backward: the tower input, the block outputs [n, M, E] and the per-block saved
    activations (n x stride bytes)"""
    __slots__ = ("x0", "xs", "ws", "stride")

    def __init__(self, x0, xs, ws, stride):
        self.x0, self.xs, self.ws, self.stride = x0, xs, ws, stride


def _empty(shape, dtype, like):
    return torch.empty(shape, dtype=dtype, device=like.device)


class Engine:
    """Executes the towers of one model instance.  `cfg` keys: precision ('bf16'|'fp32'), v_heads, t_heads,
    patch, res, ctx, fdt (bool) and for FDT: att_func, pool."""

    def __init__(self, module, cfg):
        self.m = module
        self.cfg = cfg
        self.precision = cfg["precision"]
        if self.precision not in ("bf16", "fp32", "fp8"):
            raise ValueError("precision must be 'bf16', 'fp32' or 'fp8', got %r" % (self.precision,))
        # fp8: bf16 storage and kernels everywhere except the QKV / out / MLP GEMMs (forward + input gradient), which take
        # fp8 operands with per-tensor delayed scaling (Fp8State)
        self.T = torch.float32 if self.precision == "fp32" else torch.bfloat16
        self.fp8 = None
        self.packed = None        # PackedWeights (bf16 mode): fragment-order weight copies for the streaming GEMM kernel
        self.use_packed = os.environ.get("ILVLM_PACKED_WEIGHTS", "1") == "1"
        self.arena = None
        self._side = None
        self.concurrent_towers = True     # False: everything on the current stream (per-kernel timing, debugging)
        # weight-gradient GEMMs leave the dgrad -> LayerNorm -> attention chain for a companion stream per tower (they are
        # only needed by the gradient reduction / optimizer): -1.3 % step time; stream priorities added nothing
        self.wgrad_streams = os.environ.get("ILVLM_WGRAD_STREAMS", "1") == "1"
        self._wg = {}
        self._wg_keep = {}      # tower stream -> tensors its companion stream still reads
        self.composite = os.environ.get("ILVLM_COMPOSITE", "1") == "1"    # one C call per transformer block
        self.tower_calls = os.environ.get("ILVLM_TOWER", "1") == "1"      # ... and one per TOWER in training steps (host time)
        self.tower_count = [0, 0]         # tower forward / backward calls issued (tests)
        self.fused_fdt = os.environ.get("ILVLM_FUSED_FDT", "1") == "1"    # codebook scores + token max-pool in one GEMM
        self.trust_shadow = os.environ.get("ILVLM_TRUST_SHADOW", "1") == "1"
        self.defer_ln = os.environ.get("ILVLM_DEFER_LN", "1") == "1"      # one LayerNorm-gradient reduction per tower
        self._f8_carry = None
        self.join_each_block = os.environ.get("ILVLM_JOIN_EACH_BLOCK", "0") == "1"
        # opt-in: split-K weight gradients through slab workspaces (one per executing stream) instead of fp32 atomics --
        # bit-reproducible weight gradients (the sum order is fixed), at -3.3 % step throughput (measured)
        self.slab_splitk = os.environ.get("ILVLM_SLAB_SPLITK", "0") == "1"
        self._slabs = {}
        # fp8 mode: weight gradients on fp8 operands as well (e5m2 gradient copies x the e4m3 activation copies of the forward)
        self.fp8_wgrad = os.environ.get("ILVLM_FP8_WGRAD", "1") == "1"
        self._ln_defer = {}
        self._blk = {}          # block prefix -> ilvlm_block descriptor (rebuilt when requires_grad flags change)

    @property
    def side_stream(self):
        """Second HIP stream: the text tower runs on it concurrently with the vision tower (independent until the
        contrastive head), which fills CUs that one tower's launches leave idle.  With concurrent_towers off this is
        the current stream itself, so the same code path runs serialised."""
        if not self.concurrent_towers:
            return torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream(priority=int(os.environ.get("ILVLM_SIDE_PRIO", "0")))
        return self._side

    # ------------------------------------------------------------------ parameters
    def prepare(self, training=False):
        """Adopt / re-check the parameters before a forward.  `training`: a forward whose backward will run (gradients
        enabled); only such a forward advances the fp8 delayed-scaling state (Fp8State.begin_step)."""
        if training:      # regime hint for the C library's kernel choices (weight-gradient tile): towers + companions in flight?
            ops.gemm_set_concurrent(self.concurrent_towers and self.wgrad_streams)
        if self.arena is None:
            self.arena = ParamArena(self.m, self.precision)
            self.arena.inactive = set(self.m.unused_parameter_names())
            swapped = True
        else:
            swapped = self.arena.sync_in()
        a = self.arena
        # The bf16 shadow is re-cast from the fp32 masters (0.2 ms for 155 M parameters) unless it is provably current:
        # the fused AdamW wrote the shadow of everything it updated, no parameter storage was replaced and no Parameter
        # was written in place since (version counters).  A write through `p.data` is invisible to both checks -- the
        # reference solver only does that to logit_scale, which the kernels read in fp32 -- so code that edits GEMM weights
        # that way calls mark_dirty() (or sets ILVLM_TRUST_SHADOW=0).
        vers = a.versions()
        current = self.trust_shadow and a.shadow_fresh and not swapped and vers == a._versions
        # a deferred optimizer update (FusedAdamW.defer_late_blocks) is waited for by the towers in front of the first block it
        # touched -- when this forward takes the tower calls with current copies; every other forward waits here
        if a.late_event is not None and not (training and current and a.packed_fresh and self.tower_calls and self.precision == "bf16"):
            a.wait_late_update()
            a.late_event = None
        if not current:
            a.refresh_shadow()
        # the fragment-order images are current under the same conditions when the optimizer wrote them with its update
        packed_current = current and a.packed_fresh
        a.shadow_fresh = a.packed_fresh = False
        a._versions = vers
        self.Wf = a.views                                     # fp32 masters
        self.Wc = a.views if self.precision == "fp32" else a.sviews   # GEMM operands
        self.Gr = a.gviews
        pres = ["visual.transformer.resblocks.%d." % i for i in range(self.cfg["v_layers"])] + \
               ["encode_text.transformer.resblocks.%d." % i for i in range(self.cfg["t_layers"])]
        if self.precision == "fp8":
            if self.fp8 is None:
                self.fp8 = Fp8State(a, pres)
            self.fp8.begin_step(bool(training))
        if self.precision == "bf16" and self.use_packed:
            # the streaming GEMM kernel reads the block weights in fragment order: re-pack from the (now current) shadow
            if self.packed is None:
                self.packed = a.packed = PackedWeights(a, pres)
                self._blk = {}
                packed_current = False
            if not packed_current:
                self.packed.refresh()
        req = {n: p.requires_grad for n, p in a.named}
        if req != getattr(self, "req", None):
            self._blk = {}                                    # frozen / unfrozen parameters: new gradient slots
        self.req = req

    def mark_dirty(self):
        """the fp32 masters were edited behind the engine's back (through p.data): re-cast the bf16 shadow next forward"""
        if self.arena is not None:
            self.arena.shadow_fresh = False
            self.arena._versions = None

    def _wgrad_stream(self):
        """companion stream of the current stream for weight-gradient GEMMs (ILVLM_WGRAD_STREAMS=0 switches it off)"""
        if not self.wgrad_streams or not self.concurrent_towers:
            return None
        cur = torch.cuda.current_stream()
        key = cur.cuda_stream
        if key not in self._wg:
            self._wg[key] = torch.cuda.Stream(priority=int(os.environ.get("ILVLM_WGRAD_PRIO", "0")))
        return self._wg[key]

    SLAB_BYTES, SLAB_TILES = 48 << 20, 4096

    def _slab_ws(self):
        """(workspace, zeroed ticket counters) of the stream this tower's weight-gradient GEMMs run on, or None"""
        if not self.slab_splitk:
            return None
        wg = self._wgrad_stream()
        key = (wg if wg is not None else torch.cuda.current_stream()).cuda_stream
        st = self._slabs.get(key)
        if st is None:
            dev = self.arena.P.device
            st = self._slabs[key] = (torch.empty(self.SLAB_BYTES, dtype=torch.uint8, device=dev),
                                     torch.zeros(self.SLAB_TILES, dtype=torch.int32, device=dev))
        return st

    def join_wgrad(self):
        """order the current stream after its companion's weight-gradient GEMMs"""
        wg = self._wg.get(torch.cuda.current_stream().cuda_stream)
        if wg is not None:
            torch.cuda.current_stream().wait_stream(wg)
            # tensors freed from here on are reused by work enqueued after this join, i.e. after the weight gradients
            self._wg_keep.pop(torch.cuda.current_stream().cuda_stream, None)

    def _block_desc(self, pre, E, H, causal):
        """ilvlm_block descriptor of the transformer block with parameter prefix `pre`, or None when the composite path
        does not apply: switched off (ILVLM_COMPOSITE=0), a GEMM profiler is attached (it times the individual launches),
        or a LayerNorm parameter of the block is frozen."""
        if not self.composite or ops._gemm_profiler is not None:
            return None
        d = self._blk.get(pre)
        if d is None:
            names = dict(ln1_w="ln_1.weight", ln1_b="ln_1.bias", ln2_w="ln_2.weight", ln2_b="ln_2.bias",
                         in_w="attn.in_proj_weight", in_b="attn.in_proj_bias", out_w="attn.out_proj.weight",
                         out_b="attn.out_proj.bias", fc_w="mlp.c_fc.weight", fc_b="mlp.c_fc.bias",
                         proj_w="mlp.c_proj.weight", proj_b="mlp.c_proj.bias")
            if not all(self.req[pre + names[k]] for k in ("ln1_w", "ln1_b", "ln2_w", "ln2_b")):
                d = False
            else:
                params = {k: (self.Wc if k in ("in_w", "out_w", "fc_w", "proj_w") else self.Wf)[pre + n] for k, n in names.items()}
                grads = {"g_" + k: (self.Gr[pre + n] if self.req[pre + n] else None) for k, n in names.items()}
                d = ops.block_desc(E, H, causal, self.T, params, grads)
                if self.packed is not None:
                    for k in PackedWeights.WNAME:
                        setattr(d, k + "p", self.packed.view(pre + names[k]).data_ptr())
                        setattr(d, k + "pt", self.packed.view(pre + names[k], backward=True).data_ptr())
                if self.fp8 is not None:
                    f8 = self.fp8
                    base = f8.slots[pre + "h1"]
                    for k in f8.WEIGHT:
                        setattr(d, k + "8", f8.w8(pre, k).data_ptr())
                        setattr(d, k + "8t", f8.w8(pre, k, transposed=True).data_ptr())
                        if f8.pack:
                            setattr(d, k + "8p", f8.w8p(pre, k).data_ptr())
                            setattr(d, k + "8tp", f8.w8p(pre, k, transposed=True).data_ptr())
                    d.f8_scale = f8.scale[base:].data_ptr()
                    d.f8_inv = f8.inv[base:].data_ptr()
                    d.f8_amax = f8.amax[base:].data_ptr()
            self._blk[pre] = d
        if d and self.fp8 is not None:
            d.fp8 = (3 if self.fp8_wgrad else 2) if self.fp8.active else 1
        return d or None

    def _packed(self, name, backward=False):
        """fragment-order copy of a block weight (None: not a packed weight / packing off)"""
        if self.packed is None or not name.endswith(tuple(PackedWeights.WNAME.values())) or ".resblocks." not in name:
            return None
        return self.packed.view(name, backward)

    def _mat(self, name):
        w = self.Wc[name]
        return w if w.dim() == 2 else w.reshape(w.shape[0], -1)

    # ------------------------------------------------------------------ helpers
    def _linear_bwd(self, dy, x, wname, bname, need_dx=True, dx_act=0, dx_aux=None, fp8_keys=None, x8=None):
        """dy: [M,N] T; x: [M,K] T.  Accumulates dW (and db) into the gradient arena, returns dx (T) or None.
        fp8_keys = (block prefix, gradient slot, weight slot): the input gradient runs on e5m2 x transposed-e4m3 operands
        in fp8 mode; x8 = (e4m3 copy of x kept by the forward pass, its slot): the weight gradient does too."""
        M, N = dy.shape
        K = x.shape[1]
        dy8 = None
        if fp8_keys is not None and self.fp8 is not None:
            pre, kg, kw_ = fp8_keys
            dy8 = self.fp8.quant(dy, pre + kg, e5m2=True)
        need_b = bname is not None and self.req[bname]
        fuse_b = need_b and self.req[wname] and self.T == torch.bfloat16 and ops.rowsum_fusable(N, M)
        wg = self._wgrad_stream()
        slab = self._slab_ws()
        if wg is not None:                   # weight gradients are off the dgrad chain: a companion stream takes them
            wg.wait_stream(torch.cuda.current_stream())
            # the operands must outlive the launch: the saved activation x does (the autograd node holds it until backward
            # returns, after join_wgrad); dy is kept referenced until the join instead of paying record_stream's
            # allocator events
            keep = self._wg_keep.setdefault(torch.cuda.current_stream().cuda_stream, [])
            keep.append(dy)
            keep.append(x)       # x may be a temporary of the caller (the un-padded patch rows of a trainable conv1)
            keep.append(dy8)
        # thread-local launch-stream override: cheaper than entering a torch stream context per GEMM
        with ops.stream_override(wg.cuda_stream if wg is not None else None):
            if self.req[wname] and dy8 is not None and x8 is not None and x8[0] is not None:
                ops.gemm_fp8_wgrad(dy8, x8[0], self.Gr[wname].reshape(N, -1), self.fp8.s(pre + kg)[1], self.fp8.s(pre + x8[1])[1],
                                   split_k=ops.wgrad_split(N, K, M, 128), rowsum=self.Gr[bname] if fuse_b else None, slab=slab)
            elif self.req[wname]:
                # dW[N,K] += dy^T x ; the bias gradient sum_m dy[m,:] rides along as the row sums of the A operand
                ops.gemm(dy, x, self.Gr[wname].reshape(N, -1), trans_a=True, trans_b=True, accumulate=True,
                         split_k=ops.wgrad_split(N, K, M, 128 if self.T == torch.bfloat16 else 64),
                         a_rowsum=self.Gr[bname] if fuse_b else None, slab=slab if self.T == torch.bfloat16 else None)
            if need_b and not fuse_b:
                ops.colsum(dy, self.Gr[bname])
        if not need_dx:
            return None
        dx = _empty((M, K), self.T, dy)
        if dy8 is not None:
            ops.gemm_fp8(dy8, self.fp8.w8(pre, kw_, transposed=True), dx, self.fp8.s(pre + kg)[1], self.fp8.s(pre + kw_)[1],
                         a_e5m2=True, aux=dx_aux, act=dx_act, b_packed=self.fp8.w8p(pre, kw_, transposed=True))
        else:
            ops.gemm(dy, self._mat(wname), dx, trans_b=True, aux=dx_aux, act=dx_act, b_packed=self._packed(wname, backward=True))
        return dx

    # ------------------------------------------------------------------ transformer block
    def block_fwd(self, x_in, pre, B, L, H, causal, save, seq=None):
        """seq: ops.PackedSeq when the rows are the valid tokens only (text tower of a training step)."""
        M, E = x_in.shape
        desc = self._block_desc(pre, E, H, causal)
        if desc is not None:      # one C call for the 8 kernels of the block (ilvlm_block_fwd)
            ws = torch.empty(ops.block_saved_bytes(desc, M, B, L), dtype=torch.uint8, device=x_in.device)
            x_out = _empty((M, E), torch.float32, x_in)
            ops.block_fwd(desc, x_in, x_out, ws, B, L, seq)
            return x_out, ((x_in, ws) if save else None)
        T = self.T
        Wf = self.Wf
        f8 = self.fp8

        def lin(x, key_a, key_w, wname, out, **kw):
            """x [M,K] (T) . W^T: fp8 operands when the fp8 mode is active, else the compute-dtype GEMM"""
            x8 = f8.quant(x, pre + key_a) if f8 is not None else None
            if x8 is not None:
                ops.gemm_fp8(x8, f8.w8(pre, key_w), out, f8.s(pre + key_a)[1], f8.s(pre + key_w)[1], b_packed=f8.w8p(pre, key_w), **kw)
            else:
                ops.gemm(x, self._mat(pre + wname), out, b_packed=self._packed(pre + wname), **kw)
            return x8

        h1 = _empty((M, E), T, x_in); mean1 = _empty((M,), torch.float32, x_in); rstd1 = torch.empty_like(mean1)
        ops.layernorm_fwd(x_in, Wf[pre + "ln_1.weight"], Wf[pre + "ln_1.bias"], h1, mean1, rstd1, M, E)
        qkv = _empty((M, 3 * E), T, x_in)
        h1_8 = lin(h1, "h1", "in_w", "attn.in_proj_weight", qkv, bias=Wf[pre + "attn.in_proj_bias"])
        att = _empty((M, E), T, x_in); lse = _empty((B, H, L), torch.float32, x_in)
        ops.attention_fwd(qkv, att, lse, B, L, H, causal, seq)
        x_mid = _empty((M, E), torch.float32, x_in)
        att8 = lin(att, "att", "out_w", "attn.out_proj.weight", x_mid, bias=Wf[pre + "attn.out_proj.bias"], residual=x_in)
        h2 = _empty((M, E), T, x_in); mean2 = torch.empty_like(mean1); rstd2 = torch.empty_like(mean1)
        ops.layernorm_fwd(x_mid, Wf[pre + "ln_2.weight"], Wf[pre + "ln_2.bias"], h2, mean2, rstd2, M, E)
        u = _empty((M, 4 * E), T, x_in); g = _empty((M, 4 * E), T, x_in)
        h2_8 = lin(h2, "h2", "fc_w", "mlp.c_fc.weight", g, bias=Wf[pre + "mlp.c_fc.bias"], aux=u, act=ACT_QUICKGELU)
        x_out = _empty((M, E), torch.float32, x_in)
        g8 = lin(g, "g", "proj_w", "mlp.c_proj.weight", x_out, bias=Wf[pre + "mlp.c_proj.bias"], residual=x_mid)
        # the e4m3 copies are the X operands of the fp8 weight gradients
        x8s = (h1_8, att8, h2_8, g8) if (self.fp8_wgrad and g8 is not None) else (None,) * 4
        saved = (x_in, h1, mean1, rstd1, qkv, att, lse, x_mid, h2, mean2, rstd2, u, g, x8s) if save else None
        return x_out, saved

    def _ln_defer_begin(self, tower, n_blocks, E):
        """Deferred LayerNorm-gradient reduction for one tower's backward (composite path, no data-parallel reducer that
        wants each block's gradients final right away): every block's two LayerNorm backward launches leave their partial
        rows in their own slots, and ONE kernel adds all of them up at the end of the tower -- 1 launch instead of 24
        8 us launches on the dgrad chain.  Returns the slot tensor [n_blocks, 2, 2 * LN_WS_BLOCKS * E] or None."""
        if not (self.defer_ln and self.composite and ops._gemm_profiler is None and not self.arena.reducing()
                and self.arena.eager_opt is None):
            return None
        key = (tower, n_blocks, E)
        st = self._ln_defer.get(key)
        if st is None:
            pre = "visual." if tower == "v" else "encode_text."
            ptrs = []
            for i in range(n_blocks):
                b = "%stransformer.resblocks.%d." % (pre, i)
                for ln in ("ln_2", "ln_1"):                 # slot order of ilvlm_block_bwd
                    ptrs += [self.Gr[b + ln + ".weight"].data_ptr(), self.Gr[b + ln + ".bias"].data_ptr()]
            dev = self.arena.P.device
            st = (torch.empty((n_blocks, 2, 2 * ops.LN_WS_BLOCKS * E), dtype=torch.float32, device=dev),
                  torch.tensor(ptrs, dtype=torch.int64).to(dev))
            self._ln_defer[key] = st
        return st

    def _ln_defer_end(self, st, rows, E):
        slots, ptrs = st
        ops.ln_reduce_batched(slots, slots.shape[2], slots.shape[0] * 2, rows, E, ptrs)

    def block_bwd(self, saved, pre, dx_f32, dx_lp, B, L, H, causal, seq=None, ln_slots=None, f8_in=None, f8_next=None):
        """dx_f32: fp32 gradient of the block output; dx_lp: the same in T (None in fp32 mode).  Returns the pair
        for the block input.  ln_slots: this block's two deferred LayerNorm slots (composite path only)."""
        if len(saved) == 2:       # saved by the composite forward: composite backward (ilvlm_block_bwd)
            x_in, ws = saved
            M, E = x_in.shape
            desc = self._block_desc(pre, E, H, causal)
            lp = self.T != torch.float32
            din = _empty((M, E), torch.float32, x_in)
            f8 = self.fp8 if (self.fp8 is not None and self.fp8.active) else None
            # fp8 weight gradients on: a consumer block whose four weight matrices are trainable reads only the e5m2 copy of
            # this call's input gradient, so the bf16 copy is not produced
            only8 = (f8 is not None and self.fp8_wgrad and f8_next is not None
                     and all(self.req[f8_next + n] for n in Fp8State.WNAME.values())
                     and all(self.req[pre + n] for n in Fp8State.WNAME.values())
                     and self._block_desc(f8_next, E, H, causal) is not None)
            din_lp = _empty((M, E), self.T, x_in) if (lp and not only8) else None
            scratch = torch.empty(ops.block_scratch_bytes(desc, M), dtype=torch.uint8, device=x_in.device)
            wg = self._wgrad_stream()
            slab = self._slab_ws()
            if slab is not None:
                desc.splitk_ws, desc.splitk_ws_bytes = slab[0].data_ptr(), slab[0].numel()
                desc.splitk_cnt, desc.splitk_cnt_len = slab[1].data_ptr(), slab[1].numel()
            else:
                desc.splitk_ws, desc.splitk_cnt = None, None
            if wg is not None:    # the scratch holds the dY operands of the weight-gradient GEMMs: alive until the join
                self._wg_keep.setdefault(torch.cuda.current_stream().cuda_stream, []).append(scratch)
                self._wg_keep[torch.cuda.current_stream().cuda_stream].append(dx_lp if lp else dx_f32)
                if f8_in is not None:     # the e5m2 copy of dx is the dY operand of this block's proj weight gradient
                    self._wg_keep[torch.cuda.current_stream().cuda_stream].append(f8_in)
            # fp8 mode: f8_in = e5m2 copy of dx_lp made by the block processed before this one; f8_next = prefix of the block
            # that consumes this call's input gradient (its d(x_out) slot scales the copy this call emits)
            din8 = sc8 = am8 = None
            if self.fp8 is not None and self.fp8.active and f8_next is not None:
                din8 = torch.empty((M, E), dtype=torch.uint8, device=x_in.device)
                sc8, _, am8 = self.fp8.s(f8_next + "dout")
            ops.block_bwd(desc, x_in, ws, dx_f32, dx_lp, din, din_lp, scratch, B, L, seq, wg, ln_slots,
                          dx8=f8_in if (self.fp8 is not None and self.fp8.active) else None, din8=din8, din8_scale=sc8,
                          din8_amax=am8)
            self._f8_carry = din8
            return din, din_lp
        x_in, h1, mean1, rstd1, qkv, att, lse, x_mid, h2, mean2, rstd2, u, g, (h1_8, att8, h2_8, g8) = saved
        M, E = x_in.shape
        T, Wf, Gr = self.T, self.Wf, self.Gr
        lp = T != torch.float32
        dy = dx_lp if lp else dx_f32
        # MLP: x_out = x_mid + c_proj(quickgelu(c_fc(h2)))
        du = self._linear_bwd(dy, g, pre + "mlp.c_proj.weight", pre + "mlp.c_proj.bias", dx_act=ACT_QUICKGELU_BWD, dx_aux=u,
                              fp8_keys=(pre, "dout", "proj_w"), x8=(g8, "g"))
        dh2 = self._linear_bwd(du, h2, pre + "mlp.c_fc.weight", pre + "mlp.c_fc.bias", fp8_keys=(pre, "du", "fc_w"), x8=(h2_8, "h2"))
        dmid = _empty((M, E), torch.float32, x_in)
        dmid_lp = _empty((M, E), T, x_in) if lp else None
        ops.layernorm_bwd(dh2, x_mid, mean2, rstd2, Wf[pre + "ln_2.weight"], Gr[pre + "ln_2.weight"], Gr[pre + "ln_2.bias"],
                          M, E, dres=dx_f32, dx_f32=dmid, dx_lp=dmid_lp)
        dy = dmid_lp if lp else dmid
        # attention: x_mid = x_in + out_proj(attn(in_proj(h1)))
        da = self._linear_bwd(dy, att, pre + "attn.out_proj.weight", pre + "attn.out_proj.bias", fp8_keys=(pre, "dmid", "out_w"),
                              x8=(att8, "att"))
        dqkv = _empty((M, 3 * E), T, x_in)
        ops.attention_bwd(da, qkv, att, lse, dqkv, B, L, H, causal, seq)
        dh1 = self._linear_bwd(dqkv, h1, pre + "attn.in_proj_weight", pre + "attn.in_proj_bias", fp8_keys=(pre, "dqkv", "in_w"),
                               x8=(h1_8, "h1"))
        din = _empty((M, E), torch.float32, x_in)
        din_lp = _empty((M, E), T, x_in) if lp else None
        ops.layernorm_bwd(dh1, x_in, mean1, rstd1, Wf[pre + "ln_1.weight"], Gr[pre + "ln_1.weight"], Gr[pre + "ln_1.bias"],
                          M, E, dres=dmid, dx_f32=din, dx_lp=din_lp)
        return din, din_lp

    # ------------------------------------------------------------------ whole towers in one C call (training steps)
    def _tower_descs(self, fmt, n, E, H, causal):
        """block descriptors of a tower, or None when any block is outside the composite path"""
        if not (self.tower_calls and self.composite and not self.join_each_block):
            return None
        descs = [self._block_desc(fmt % i, E, H, causal) for i in range(n)]
        return None if any(d is None for d in descs) else descs

    def tower_fwd(self, x0, fmt, n, B, L, H, causal, seq=None):
        """all n blocks of a tower through ilvlm_tower_fwd.  Returns (x_out, saved) or None when the tower path does not apply
        (then the caller walks the blocks itself).  The activations of the whole tower are two allocations: the block outputs
        xs [n, M, E] fp32 and the per-block workspaces n x stride bytes."""
        M, E = x0.shape
        descs = self._tower_descs(fmt, n, E, H, causal)
        a = self.arena
        if descs is None:
            a.wait_late_update()
            return None
        stride = max(ops.block_saved_bytes(d, M, B, L) for d in descs)
        xs = torch.empty((n, M, E), dtype=torch.float32, device=x0.device)
        ws = torch.empty(n * stride, dtype=torch.uint8, device=x0.device)
        k = a.late_from if a.late_event is not None else 0
        if 0 < k < n:
            # the optimizer is still updating the blocks from k up on its side stream: run the blocks below k, then wait
            ops.tower_fwd(descs[:k], x0, xs[:k], ws[:k * stride], stride, B, L, seq)
            a.wait_late_update()
            ops.tower_fwd(descs[k:], xs[k - 1], xs[k:], ws[k * stride:], stride, B, L, seq)
        else:
            a.wait_late_update()
            ops.tower_fwd(descs, x0, xs, ws, stride, B, L, seq)
        self.tower_count[0] += 1
        return xs[n - 1], TowerSaved(x0, xs, ws, stride)

    def tower_bwd(self, ts, fmt, n, dx_f32, dx_lp, B, L, H, causal, seq=None, st=None):
        """backward of tower_fwd's blocks through ilvlm_tower_bwd; returns the gradient pair of the tower input.  Per block:
        the same arguments block_bwd would pass (which copies of the input gradient exist, the fp8 scale slots of the block
        that consumes them, the LayerNorm slots), the gradients of all blocks in three allocations, one scratch allocation;
        the per-block announcement (data-parallel reducer, in-backward optimizer) comes back through the `done` callback."""
        x0, xs, ws, stride = ts.x0, ts.xs, ts.ws, ts.stride
        M, E = x0.shape
        descs = self._tower_descs(fmt, n, E, H, causal)
        if descs is None:
            raise RuntimeError("tower_bwd: the forward used the tower call, the backward cannot (composite path switched off in between?)")
        dev = x0.device
        lp = self.T != torch.float32
        f8 = self.fp8 if (self.fp8 is not None and self.fp8.active) else None
        sstride = max(ops.block_scratch_bytes(d, M) for d in descs)
        d_f32 = torch.empty((n, M, E), dtype=torch.float32, device=dev)
        d_lp = torch.empty((n, M, E), dtype=self.T, device=dev) if lp else None
        d8 = torch.empty((n, M, E), dtype=torch.uint8, device=dev) if f8 is not None else None
        scratch = torch.empty(n * sstride, dtype=torch.uint8, device=dev)
        per = (_lib.TowerGrad * n)()
        lnws = None if st is not None else ops._ln_workspace(dev, E)
        trainable = [all(self.req[(fmt % i) + w] for w in Fp8State.WNAME.values()) for i in range(n)]
        for i in range(n):
            g = per[i]
            # fp8 weight gradients on: a consumer block whose four weight matrices are trainable reads only the e5m2 copy of
            # this block's input gradient, so the bf16 copy is not produced (block_bwd's `only8`)
            only8 = f8 is not None and self.fp8_wgrad and i > 0 and trainable[i] and trainable[i - 1]
            g.din_lp = d_lp[i].data_ptr() if (lp and not only8) else None
            if f8 is not None and i > 0:
                sc8, _, am8 = f8.s((fmt % (i - 1)) + "dout")
                g.din8, g.din8_scale, g.din8_amax = d8[i].data_ptr(), sc8.data_ptr(), am8.data_ptr()
            if st is not None:
                g.ln_ws, g.ln_ws_blocks = st[0][i].data_ptr(), -ops.LN_WS_BLOCKS
            else:
                g.ln_ws, g.ln_ws_blocks = lnws.data_ptr(), ops.LN_WS_BLOCKS
        slab = self._slab_ws()
        for d in descs:
            if slab is not None:
                d.splitk_ws, d.splitk_ws_bytes = slab[0].data_ptr(), slab[0].numel()
                d.splitk_cnt, d.splitk_cnt_len = slab[1].data_ptr(), slab[1].numel()
            else:
                d.splitk_ws, d.splitk_cnt = None, None
        wg = self._wgrad_stream()
        if wg is not None:      # operands of the weight-gradient GEMMs: alive until the join
            self._wg_keep.setdefault(torch.cuda.current_stream().cuda_stream, []).extend(
                [scratch, d_f32, d_lp, d8, dx_lp if lp else dx_f32, xs, ws])
        need_cb = self.m._grad_sync is not None or getattr(self.arena, "eager_opt", None) is not None
        ops.tower_bwd(descs, per, x0, xs, ws, stride, dx_f32, dx_lp if lp else None, d_f32, scratch, sstride, B, L, seq, wg,
                      done=(lambda i: self.m._sync(fmt % i)) if need_cb else None)
        self.tower_count[1] += 1
        return d_f32[0], (d_lp[0] if lp else None)

    # ------------------------------------------------------------------ vision tower
    def vision_fwd(self, images, save):
        cfg, Wf, T = self.cfg, self.Wf, self.T
        if images.dim() != 4 or images.shape[1] != 3 or images.shape[2] != cfg["res"] or images.shape[3] != cfg["res"]:
            raise RuntimeError("images must be [B,3,%d,%d], got %s" % (cfg["res"], cfg["res"], tuple(images.shape)))
        images = images.contiguous().float()
        B, ps = images.shape[0], cfg["patch"]
        g = cfg["res"] // ps
        P, Lv = g * g, g * g + 1
        W = Wf["visual.class_embedding"].shape[0]
        K = 3 * ps * ps
        Kp = (K + 63) // 64 * 64            # patch length padded to the GEMM K-tile (ViT-L/14: 588 -> 640)
        patches = _empty((B * P, Kp), T, images)
        ops.patchify(images, patches, ps)
        wconv = self._mat("visual.conv1.weight")
        if Kp != K:                          # zero-padded copy of the (tiny) patch-embedding matrix; memory op only
            wpad = torch.zeros((W, Kp), dtype=T, device=images.device)
            wpad[:, :K].copy_(wconv)
            wconv = wpad
        tokens = _empty((B * Lv, W), torch.float32, images)
        pos = Wf["visual.positional_embedding"]
        ops.cls_rows(Wf["visual.class_embedding"], pos, tokens, B, Lv, W)
        ops.gemm(patches, wconv, tokens, rowbias=pos, out_group=P, out_skip=1)
        x = _empty((B * Lv, W), torch.float32, images)
        mean0 = _empty((B * Lv,), torch.float32, images); rstd0 = torch.empty_like(mean0)
        ops.layernorm_fwd(tokens, Wf["visual.ln_pre.weight"], Wf["visual.ln_pre.bias"], x, mean0, rstd0, B * Lv, W)
        tw = self.tower_fwd(x, "visual.transformer.resblocks.%d.", cfg["v_layers"], B, Lv, cfg["v_heads"], 0) if save else None
        if tw is not None:
            x, blocks = tw
        else:
            blocks = []
            for i in range(cfg["v_layers"]):
                x, s = self.block_fwd(x, "visual.transformer.resblocks.%d." % i, B, Lv, cfg["v_heads"], 0, save)
                blocks.append(s)
        saved = dict(B=B, P=P, Lv=Lv, W=W, patches=patches if self.req["visual.conv1.weight"] else None, tokens=tokens,
                     mean0=mean0, rstd0=rstd0, blocks=blocks) if save else None
        return x, saved     # x: final residual stream [B*Lv, W] fp32 (dense patch tokens are rows 1.. of each image)

    def vision_bwd(self, saved, dx_f32, dx_lp):
        cfg, Wf, Gr = self.cfg, self.Wf, self.Gr
        B, Lv, W = saved["B"], saved["Lv"], saved["W"]
        tower = isinstance(saved["blocks"], TowerSaved)
        composite = tower or all(s is not None and len(s) == 2 for s in saved["blocks"])
        st = self._ln_defer_begin("v", cfg["v_layers"], W) if composite and all(
            self.req["visual.transformer.resblocks.%d.ln_1.weight" % i] for i in range(cfg["v_layers"])) else None
        carry = None
        if tower:
            dx_f32, dx_lp = self.tower_bwd(saved["blocks"], "visual.transformer.resblocks.%d.", cfg["v_layers"], dx_f32, dx_lp, B, Lv,
                                           cfg["v_heads"], 0, None, st)
        for i in (() if tower else reversed(range(cfg["v_layers"]))):
            self._f8_carry = None
            dx_f32, dx_lp = self.block_bwd(saved["blocks"][i], "visual.transformer.resblocks.%d." % i, dx_f32, dx_lp, B, Lv,
                                           cfg["v_heads"], 0, ln_slots=st[0][i] if st is not None else None, f8_in=carry,
                                           f8_next=("visual.transformer.resblocks.%d." % (i - 1)) if i > 0 else None)
            carry = self._f8_carry
            if self.join_each_block:      # (a gradient reducer waits for the companion stream itself: comm.reduce_range)
                self.join_wgrad()
            self.m._sync("visual.transformer.resblocks.%d." % i)       # this block's gradients are complete
        if st is not None:
            self._ln_defer_end(st, B * Lv, W)
        dtok = _empty((B * Lv, W), torch.float32, dx_f32)
        ops.layernorm_bwd(dx_f32, saved["tokens"], saved["mean0"], saved["rstd0"], Wf["visual.ln_pre.weight"],
                          Gr["visual.ln_pre.weight"], Gr["visual.ln_pre.bias"], B * Lv, W, dx_f32=dtok)
        need_pos, need_cls = self.req["visual.positional_embedding"], self.req["visual.class_embedding"]
        if need_pos or need_cls:
            scratch = Gr["visual.positional_embedding"] if need_pos else torch.zeros_like(Gr["visual.positional_embedding"])
            ops.batch_sum(dtok, scratch, Gr["visual.class_embedding"] if need_cls else None, B, Lv, W)
        if self.req["visual.conv1.weight"]:     # frozen by train() in the reference; honoured if someone unfreezes it
            dpatch = dtok.view(B, Lv, W)[:, 1:, :].reshape(B * saved["P"], W).to(self.T).contiguous()
            K = self.Gr["visual.conv1.weight"][0].numel()
            pt = saved["patches"]
            if pt.shape[1] != K:             # drop the K padding of the saved patches
                pt = pt[:, :K].contiguous()
            self._linear_bwd(dpatch, pt, "visual.conv1.weight", None, need_dx=False)

    def vision_pooled(self, x_final, B, Lv, save):
        """ln_post(cls) @ proj -> (projected [B,D] fp32, saved).  Used by the baseline CLIP loss and encode_image."""
        Wf = self.Wf
        W = x_final.shape[1]
        idx = torch.zeros(B, dtype=torch.int64, device=x_final.device)
        cls = _empty((B, W), torch.float32, x_final)
        ops.gather_rows(x_final, idx, cls, B, Lv, W)
        feat = torch.empty_like(cls); mean = _empty((B,), torch.float32, cls); rstd = torch.empty_like(mean)
        ops.layernorm_fwd(cls, Wf["visual.ln_post.weight"], Wf["visual.ln_post.bias"], feat, mean, rstd, B, W)
        proj = Wf["visual.proj"]                     # [W, D] stored K-major -> trans_b
        out = _empty((B, proj.shape[1]), torch.float32, cls)
        ops.gemm(feat, proj, out, trans_b=True)
        return out, feat, ((idx, cls, feat, mean, rstd) if save else None)

    def vision_pooled_bwd(self, saved, dout, dx_stream, B, Lv):
        """Adds the pooled-path gradient into the fp32 stream gradient dx_stream [B*Lv, W]."""
        idx, cls, feat, mean, rstd = saved
        Wf, Gr = self.Wf, self.Gr
        W = cls.shape[1]
        if self.req["visual.proj"]:
            ops.gemm(feat, dout, Gr["visual.proj"], trans_a=True, trans_b=True, accumulate=True)   # [W,D] += feat^T dout
        dfeat = torch.empty_like(feat)
        ops.gemm(dout, Wf["visual.proj"], dfeat)      # dout [B,D] . proj[W,D]^T
        dcls = torch.empty_like(cls)
        ops.layernorm_bwd(dfeat, cls, mean, rstd, Wf["visual.ln_post.weight"], Gr["visual.ln_post.weight"],
                          Gr["visual.ln_post.bias"], B, W, dx_f32=dcls)
        ops.scatter_rows(dcls, idx, dx_stream, B, Lv, W)

    # ------------------------------------------------------------------ text tower
    def text_fwd(self, tokens, save, seq=None):
        """seq (ops.PackedSeq): run the tower on the valid tokens only -- rows [seq.offs[b], seq.offs[b+1]) of every text
        tensor belong to caption b.  Without it every one of the B*ctx positions is computed, as the reference does."""
        cfg, Wf = self.cfg, self.Wf
        if tokens.dim() != 2 or tokens.shape[1] != cfg["ctx"] or tokens.dtype != torch.int64:
            raise RuntimeError("tokens must be int64 [B,%d], got %s %s" % (cfg["ctx"], tuple(tokens.shape), tokens.dtype))
        tokens = tokens.contiguous()
        B, Lt = tokens.shape
        table = Wf["encode_text.token_embedding.weight"]
        Wt = table.shape[1]
        if seq is not None and (seq.B != B or seq.ctx != Lt):
            raise RuntimeError("packed text rows: descriptor is for [%d,%d], tokens are [%d,%d]" % (seq.B, seq.ctx, B, Lt))
        x = _empty((seq.rows if seq is not None else B * Lt, Wt), torch.float32, table)
        ops.embed_fwd(tokens, table, Wf["encode_text.positional_embedding"], x, seq)
        tw = self.tower_fwd(x, "encode_text.transformer.resblocks.%d.", cfg["t_layers"], B, Lt, cfg["t_heads"], 1, seq) if save else None
        if tw is not None:
            x, blocks = tw
        else:
            blocks = []
            for i in range(cfg["t_layers"]):
                x, s = self.block_fwd(x, "encode_text.transformer.resblocks.%d." % i, B, Lt, cfg["t_heads"], 1, save, seq)
                blocks.append(s)
        saved = dict(B=B, Lt=Lt, Wt=Wt, tokens=tokens, blocks=blocks, seq=seq) if save else None
        return x, saved       # final residual stream BEFORE ln_final

    def text_bwd(self, saved, dx_f32, dx_lp):
        cfg, Gr = self.cfg, self.Gr
        B, Lt = saved["B"], saved["Lt"]
        tower = isinstance(saved["blocks"], TowerSaved)
        composite = tower or all(s is not None and len(s) == 2 for s in saved["blocks"])
        st = self._ln_defer_begin("t", cfg["t_layers"], saved["Wt"]) if composite and all(
            self.req["encode_text.transformer.resblocks.%d.ln_1.weight" % i] for i in range(cfg["t_layers"])) else None
        carry = None
        if tower:
            dx_f32, dx_lp = self.tower_bwd(saved["blocks"], "encode_text.transformer.resblocks.%d.", cfg["t_layers"], dx_f32, dx_lp, B,
                                           Lt, cfg["t_heads"], 1, saved["seq"], st)
        for i in (() if tower else reversed(range(cfg["t_layers"]))):
            self._f8_carry = None
            dx_f32, dx_lp = self.block_bwd(saved["blocks"][i], "encode_text.transformer.resblocks.%d." % i, dx_f32, dx_lp, B,
                                           Lt, cfg["t_heads"], 1, saved["seq"], ln_slots=st[0][i] if st is not None else None,
                                           f8_in=carry,
                                           f8_next=("encode_text.transformer.resblocks.%d." % (i - 1)) if i > 0 else None)
            carry = self._f8_carry
            if self.join_each_block:      # (a gradient reducer waits for the companion stream itself: comm.reduce_range)
                self.join_wgrad()
            self.m._sync("encode_text.transformer.resblocks.%d." % i)
        if st is not None:
            self._ln_defer_end(st, dx_f32.shape[0], saved["Wt"])
        need_tab, need_pos = self.req["encode_text.token_embedding.weight"], self.req["encode_text.positional_embedding"]
        if need_tab:
            ops.embed_bwd(saved["tokens"], dx_f32, Gr["encode_text.token_embedding.weight"],
                          Gr["encode_text.positional_embedding"] if need_pos else None, saved["seq"])
        elif need_pos:
            if saved["seq"] is not None:
                raise NotImplementedError("packed text rows with a frozen token embedding and a trainable positional embedding")
            ops.batch_sum(dx_f32, Gr["encode_text.positional_embedding"], None, B, Lt, saved["Wt"])

    def text_words(self, x_final, save):
        """ln_final over every token -> word features [B*Lt, Wt] in T."""
        Wf = self.Wf
        M, Wt = x_final.shape
        words = _empty((M, Wt), self.T, x_final); mean = _empty((M,), torch.float32, x_final); rstd = torch.empty_like(mean)
        ops.layernorm_fwd(x_final, Wf["encode_text.ln_final.weight"], Wf["encode_text.ln_final.bias"], words, mean, rstd, M, Wt)
        return words, ((x_final, mean, rstd) if save else None)

    def text_words_bwd(self, saved, dwords_f32):
        x_final, mean, rstd = saved
        M, Wt = x_final.shape
        lp = self.T != torch.float32
        dx = torch.empty_like(x_final)
        dx_lp = _empty((M, Wt), self.T, x_final) if lp else None
        ops.layernorm_bwd(dwords_f32, x_final, mean, rstd, self.Wf["encode_text.ln_final.weight"],
                          self.Gr["encode_text.ln_final.weight"], self.Gr["encode_text.ln_final.bias"], M, Wt, dx_f32=dx,
                          dx_lp=dx_lp)
        return dx, dx_lp

    def text_pooled(self, x_final, tokens, B, Lt, save, seq=None):
        """ln_final at the EOT position (argmax of the ids, text_transformer.py:248) then text_projection."""
        Wf = self.Wf
        Wt = x_final.shape[1]
        idx = tokens.argmax(dim=-1)
        row = _empty((B, Wt), torch.float32, x_final)
        ops.gather_rows(x_final, idx, row, B, Lt, Wt, seq)
        feat = torch.empty_like(row); mean = _empty((B,), torch.float32, row); rstd = torch.empty_like(mean)
        ops.layernorm_fwd(row, Wf["encode_text.ln_final.weight"], Wf["encode_text.ln_final.bias"], feat, mean, rstd, B, Wt)
        w = Wf["encode_text.text_projection.weight"]
        out = _empty((B, w.shape[0]), torch.float32, row)
        ops.gemm(feat, w, out, bias=Wf["encode_text.text_projection.bias"])
        return out, feat, ((idx, row, feat, mean, rstd, seq) if save else None)

    def text_pooled_bwd(self, saved, dout, dx_stream, B, Lt):
        idx, row, feat, mean, rstd, seq = saved
        Wf, Gr = self.Wf, self.Gr
        Wt = row.shape[1]
        w = Wf["encode_text.text_projection.weight"]
        if self.req["encode_text.text_projection.weight"]:
            ops.gemm(dout, feat, Gr["encode_text.text_projection.weight"], trans_a=True, trans_b=True, accumulate=True)
        if self.req["encode_text.text_projection.bias"]:
            ops.colsum(dout, Gr["encode_text.text_projection.bias"])
        dfeat = torch.empty_like(feat)
        ops.gemm(dout, w, dfeat, trans_b=True)
        drow = torch.empty_like(row)
        ops.layernorm_bwd(dfeat, row, mean, rstd, Wf["encode_text.ln_final.weight"], Gr["encode_text.ln_final.weight"],
                          Gr["encode_text.ln_final.bias"], B, Wt, dx_f32=drow)
        ops.scatter_rows(drow, idx, dx_stream, B, Lt, Wt, seq)

    # ------------------------------------------------------------------ FDT query model
    def qmap_fwd(self, ft, side, rows, ftdim, group, skip, save):
        """q_map: LN -> Linear -> erf-GELU -> LN -> Linear.  ft is the fp32 token stream (image, remapped rows) or the
        T word features (text).  Returns q [rows, d] in T."""
        Wf, T = self.Wf, self.T
        pre = side + "q_map."
        d = Wf[pre + "1.weight"].shape[0]
        a0 = _empty((rows, ftdim), T, ft); m0 = _empty((rows,), torch.float32, ft); r0 = torch.empty_like(m0)
        ops.layernorm_fwd(ft, Wf[pre + "0.weight"], Wf[pre + "0.bias"], a0, m0, r0, rows, ftdim, group=group, skip=skip)
        pre1 = _empty((rows, d), T, ft); h1 = _empty((rows, d), T, ft)
        ops.gemm(a0, self._mat(pre + "1.weight"), h1, bias=Wf[pre + "1.bias"], aux=pre1, act=ACT_GELU_ERF)
        a3 = _empty((rows, d), T, ft); m3 = torch.empty_like(m0); r3 = torch.empty_like(m0)
        ops.layernorm_fwd(h1, Wf[pre + "3.weight"], Wf[pre + "3.bias"], a3, m3, r3, rows, d)
        q = _empty((rows, d), T, ft)
        ops.gemm(a3, self._mat(pre + "4.weight"), q, bias=Wf[pre + "4.bias"])
        return q, ((ft, a0, m0, r0, pre1, h1, a3, m3, r3, q, group, skip) if save else None)

    def qmap_bwd(self, saved, side, dq, dstream_f32=None, dstream_lp=None):
        """dq [rows,d] T.  Image side: writes the token-stream gradient into dstream_* (remapped rows, pre-zeroed).
        Text side: returns d(word features) fp32."""
        ft, a0, m0, r0, pre1, h1, a3, m3, r3, q, group, skip = saved
        Wf, Gr, T = self.Wf, self.Gr, self.T
        pre = side + "q_map."
        rows, d = dq.shape
        ftdim = a0.shape[1]
        da3 = self._linear_bwd(dq, a3, pre + "4.weight", pre + "4.bias")
        dpre1 = _empty((rows, d), T, dq)
        ops.layernorm_bwd(da3, h1, m3, r3, Wf[pre + "3.weight"], Gr[pre + "3.weight"], Gr[pre + "3.bias"], rows, d,
                          dx_lp=dpre1, act=ACT_GELU_ERF_BWD, act_aux=pre1)
        da0 = self._linear_bwd(dpre1, a0, pre + "1.weight", pre + "1.bias")
        if group > 0:
            ops.layernorm_bwd(da0, ft, m0, r0, Wf[pre + "0.weight"], Gr[pre + "0.weight"], Gr[pre + "0.bias"], rows, ftdim,
                              dx_f32=dstream_f32, dx_lp=dstream_lp, group=group, skip=skip)
            return None
        dwords = _empty((rows, ftdim), torch.float32, dq)
        ops.layernorm_bwd(da0, ft, m0, r0, Wf[pre + "0.weight"], Gr[pre + "0.weight"], Gr[pre + "0.bias"], rows, ftdim,
                          dx_f32=dwords)
        return dwords

    def fdt_fwd(self, q, B, Tn, mask, temperature, save, seq=None):
        """codebook scores -> token pooling -> sparsemax/softmax -> weighted codebook sum (clip_fdt.py:113-154).
        With seq (packed text rows) q holds the valid tokens only and the pad mask is implied by the row layout."""
        cfg, Wf = self.cfg, self.Wf
        sd = Wf["space_dict"]
        Cn, d = sd.shape
        pooled = _empty((B, Cn), torch.float32, q)
        pool = POOLS[cfg["pool"]]
        argmax = _empty((B, Cn), torch.int32, q) if pool == POOL_MAX else None
        fused = (self.fused_fdt and pool == POOL_MAX and self.T == torch.bfloat16 and d % 64 == 0 and float(temperature) > 0
                 and (mask is None or seq is not None))
        if fused:
            # scores, scale and the token max + argmax in the GEMM epilogue: the [rows, 4096] fp32 scores never reach HBM
            ops.fdt_score_pool_fwd(q, self._mat("space_dict"), pooled, argmax, B, Tn, math.sqrt(d), float(temperature), seq)
        else:
            scores = _empty((q.shape[0], Cn), torch.float32, q)
            ops.gemm(q, self._mat("space_dict"), scores)
            ops.fdt_pool_fwd(scores, mask, pooled, argmax, B, Tn, Cn, math.sqrt(d), float(temperature), pool, seq)
            del scores
        att_w = torch.empty_like(pooled)
        att_n, rsum = att_w, None       # att_n: the operand of the weighted codebook sum
        if cfg["att_func"] == "sparsemax":
            ops.sparsemax_fwd(pooled, att_w)
        elif cfg["att_func"] == "softmax":
            ops.softmax_fwd(pooled, att_w)
        elif cfg["att_func"] == "sigmoid":
            # clip_fdt.py:76,156-157: weights = sigmoid, the weighted sum is divided by the weights' row sum
            att_n, rsum = torch.empty_like(pooled), _empty((B,), torch.float32, q)
            ops.sigmoid_norm_fwd(pooled, att_w, att_n, rsum)
        else:
            raise NotImplementedError("att_func_type=%r (reference: softmax, sigmoid, sparsemax)" % cfg["att_func"])
        # [B,C] x [C,d]: only (B/64)*(d/64) output tiles -> split the 4096-deep reduction over 8 workgroups each
        att_ft = torch.zeros((B, d), dtype=torch.float32, device=q.device)
        ops.gemm(att_n, sd, att_ft, trans_b=True, accumulate=True, split_k=8 if Cn >= 1024 else 1)
        return att_w, att_ft, ((q, argmax, mask, att_w, float(temperature), B, Tn, seq, att_n, rsum) if save else None)

    def fdt_bwd(self, saved, datt_ft):
        """Returns dq [B*Tn, d] in T; accumulates d space_dict."""
        q, argmax, mask, att_w, temperature, B, Tn, seq, att_n, rsum = saved
        cfg, Wf, Gr, T = self.cfg, self.Wf, self.Gr, self.T
        sd = Wf["space_dict"]
        Cn, d = sd.shape
        need_sd = self.req["space_dict"]
        datt_w = torch.empty_like(att_w)
        ops.gemm(datt_ft, sd, datt_w)                                      # [B,d] . sd[C,d]^T
        if need_sd:
            ops.gemm(att_n, datt_ft, Gr["space_dict"], trans_a=True, trans_b=True, accumulate=True)   # att_n^T datt_ft
        dpooled = torch.empty_like(att_w)
        if cfg["att_func"] == "sigmoid":
            ops.sigmoid_norm_bwd(att_w, att_n, rsum, datt_w, dpooled)
        else:
            (ops.sparsemax_bwd if cfg["att_func"] == "sparsemax" else ops.softmax_bwd)(att_w, datt_w, dpooled)
        rows = q.shape[0]
        dscores = _empty((rows, Cn), T, q)
        ops.fdt_pool_bwd(dpooled, argmax, mask, dscores, B, Tn, Cn, math.sqrt(d), temperature, POOLS[cfg["pool"]], seq)
        if need_sd:
            ops.gemm(dscores, q, Gr["space_dict"], trans_a=True, trans_b=True, accumulate=True,
                     split_k=ops.wgrad_split(Cn, d, rows, 128 if T == torch.bfloat16 else 64))
        dq = _empty((rows, d), T, q)
        ops.gemm(dscores, self._mat("space_dict"), dq, trans_b=True)
        return dq

    # ------------------------------------------------------------------ contrastive head
    def head_fwd(self, img_ft, txt_ft, eps_i, eps_t, save):
        """L2-normalise, temperature, global-batch gather, two logit matrices (clip_fdt.py:410-422 / clip.py:133-147).
        img_ft, txt_ft: fp32 [B,D].  Returns logits_per_image, logits_per_text fp32 [B, W*B]."""
        from . import comm
        Wf = self.Wf
        B, D = img_ft.shape
        img_n = torch.empty_like(img_ft); ni = _empty((B,), torch.float32, img_ft)
        txt_n = torch.empty_like(txt_ft); nt = torch.empty_like(ni)
        ops.l2norm_fwd(img_ft, img_n, ni, eps_i)
        ops.l2norm_fwd(txt_ft, txt_n, nt, eps_t)
        scale = _empty((1,), torch.float32, img_ft)
        ops.logit_scale_fwd(Wf["logit_scale"], scale, 100.0)
        g_img, g_txt = comm.gather_pair(img_n, txt_n)
        Bg = g_img.shape[0]
        li = _empty((B, Bg), torch.float32, img_ft); lt = torch.empty_like(li)
        ops.gemm(img_n, g_txt, li, alpha_ptr=scale)
        ops.gemm(txt_n, g_img, lt, alpha_ptr=scale)
        saved = (img_ft, txt_ft, img_n, txt_n, ni, nt, scale, g_img, g_txt, li, lt, eps_i, eps_t) if save else None
        return li, lt, saved

    def head_bwd(self, saved, dli, dlt):
        """Returns (d img_ft, d txt_ft) fp32 [B,D]; accumulates d logit_scale."""
        from . import comm
        img_ft, txt_ft, img_n, txt_n, ni, nt, scale, g_img, g_txt, li, lt, eps_i, eps_t = saved
        B, D = img_ft.shape
        Bg = g_img.shape[0]
        dli = dli.contiguous(); dlt = dlt.contiguous()
        d_img_n = torch.empty_like(img_n); d_txt_n = torch.empty_like(txt_n)
        ops.gemm(dli, g_txt, d_img_n, trans_b=True, alpha_ptr=scale)          # dli [B,Bg] . g_txt [Bg,D]
        ops.gemm(dlt, g_img, d_txt_n, trans_b=True, alpha_ptr=scale)
        dg_txt = _empty((Bg, D), torch.float32, img_ft); dg_img = torch.empty_like(dg_txt)
        ops.gemm(dli, img_n, dg_txt, trans_a=True, trans_b=True, alpha_ptr=scale)   # dli^T [Bg,B] . img_n [B,D]
        ops.gemm(dlt, txt_n, dg_img, trans_a=True, trans_b=True, alpha_ptr=scale)
        s_img, s_txt = comm.reduce_gathered(dg_img, dg_txt, B)
        ops.add_inplace(d_img_n, s_img.contiguous())      # own slice of the gathered-matrix gradient
        ops.add_inplace(d_txt_n, s_txt.contiguous())
        if self.req["logit_scale"]:
            ops.logit_scale_bwd(dli, li, dlt, lt, self.Wf["logit_scale"], scale, self.Gr["logit_scale"])
        d_img = torch.empty_like(img_ft); d_txt = torch.empty_like(txt_ft)
        ops.l2norm_bwd(img_ft, ni, d_img_n, d_img, eps_i)
        ops.l2norm_bwd(txt_ft, nt, d_txt_n, d_txt, eps_t)
        return d_img, d_txt

"""Optimizer registry (reference prototype/optimizer/__init__.py:18-26).  'AdamW' resolves to the fused
multi-tensor HIP AdamW over the parameter arena; its param_groups / state / state_dict() keep
torch.optim.AdamW's layout ('step', 'exp_avg', 'exp_avg_sq') so reference checkpoints round-trip."""
import bisect
import ctypes as C
import os
import re

import torch

from ... import lib as L

CHUNK = 4096
INACTIVE_GROUP = 15


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False):
        if amsgrad:
            raise NotImplementedError("amsgrad=True is not used by the reference configs")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False))
        if len(self.param_groups) >= INACTIVE_GROUP:
            raise ValueError("at most %d parameter groups" % (INACTIVE_GROUP - 1))
        self._arena = None
        self._sig = None
        self._step = 0
        self._pending_state = False
        self._overlap = False       # overlap_backward(): per-block updates issued from inside backward
        self._ostreams = set()      # streams that carry in-backward updates of the current step
        self._eager = []            # chunk-index ranges already updated in the current backward
        self._ranges = {}           # block prefix -> [i0, i1) of the (offset-sorted) chunk table
        self._tiles = None          # tile table of the GEMM weights kept in fragment order (ilvlm_adamw_step_packed)
        # opt-in for loops in the reference's order (zero_grad right before the ONE backward of a step, nothing reading
        # gradients between step() and zero_grad()): step() zeroes the gradient arena itself, on a side stream beside the
        # next forward, and zero_grad() only waits for that -- the 0.11 ms memset leaves the critical path
        self.prezero_grads = False
        # defer_late_blocks(K): step() updates what the next forward reads FIRST (embeddings, heads, the transformer blocks
        # below K of each tower) on the calling stream and everything else -- the blocks from K up, most of the parameters -- on
        # a side stream, beside the first blocks of that forward; the towers wait for it in front of their block K
        self._defer_from = 0
        self._late = None           # (chunk offsets, counts, groups, tile table) of the deferred part

    # -- arena binding -----------------------------------------------------------------------
    def _bind(self):
        arena = None
        for g in self.param_groups:
            for p in g["params"]:
                a = getattr(p, "_ilvlm_arena", None)
                if a is None:
                    raise RuntimeError("FusedAdamW: parameter is not owned by an ilvlm engine arena (run one forward of "
                                       "the model on the GPU before optimizer.step())")
                if arena is None:
                    arena = a[0]
                elif arena is not a[0]:
                    raise RuntimeError("FusedAdamW: parameters of several models in one optimizer are not supported")
        self._arena = arena
        arena.eager_opt = self if self._overlap else None
        self.M = torch.zeros_like(arena.P)
        self.V = torch.zeros_like(arena.P)

    def _views(self, p):
        arena, name = p._ilvlm_arena
        o = arena.offsets[name]
        return self.M[o:o + p.numel()].view(p.shape), self.V[o:o + p.numel()].view(p.shape)

    def _build_table(self):
        arena = self._arena
        sig = (tuple(tuple(bool(p.requires_grad) and (p._ilvlm_arena[1] not in arena.inactive) for p in g["params"])
                     for g in self.param_groups), self._overlap, getattr(arena, "packed", None) is not None, self._defer_from)
        if sig == self._sig:
            return
        offs, cnts, grps = [], [], []
        # GEMM weights the engine keeps in MFMA-fragment order (bf16 mode): updated tile by tile by the kernel that also
        # writes both packed images, so the per-step re-pack launch disappears.  Not with in-backward updates (their launches
        # are slices of the chunk table) -- those keep the separate re-pack.
        packed = getattr(arena, "packed", None)
        tiled = packed.names if (packed is not None and not self._overlap and arena.S is not None and
                                 os.environ.get("ILVLM_ADAMW_PACK", "1") == "1") else ()
        tiles = []
        K = self._defer_from if not self._overlap else 0
        blk = re.compile(r"^(?:visual|encode_text)\.transformer\.resblocks\.(\d+)\.")

        def late(name):
            m = blk.match(name)
            return K > 0 and m is not None and int(m.group(1)) >= K
        l_offs, l_cnts, l_grps, l_tiles = [], [], [], []
        for gi, g in enumerate(self.param_groups):
            for p, act in zip(g["params"], sig[0][gi]):
                name = p._ilvlm_arena[1]
                o, n = arena.offsets[name], p.numel()
                if name in tiled:
                    r, c = p.shape
                    grp = gi if act else INACTIVE_GROUP
                    (l_tiles if late(name) else tiles).extend((o // 64, r, c, r0, c0, grp) for r0 in range(0, r, 64) for c0 in range(0, c, 64))
                    if act and p not in self.state:
                        m, v = self._views(p)
                        self.state[p] = dict(step=torch.tensor(float(self._step)), exp_avg=m, exp_avg_sq=v)
                    continue
                to = (l_offs, l_cnts, l_grps) if late(name) else (offs, cnts, grps)
                for c in range(0, n, CHUNK):
                    to[0].append(o + c)
                    to[1].append(min(CHUNK, n - c))
                    to[2].append(gi if act else INACTIVE_GROUP)
                if act and p not in self.state:
                    m, v = self._views(p)
                    self.state[p] = dict(step=torch.tensor(float(self._step)), exp_avg=m, exp_avg_sq=v)
        dev = arena.P.device
        # sorted by arena offset: the chunks of one transformer block (a contiguous arena range) are one slice of the table
        order = sorted(range(len(offs)), key=offs.__getitem__)
        offs, cnts, grps = [offs[i] for i in order], [cnts[i] for i in order], [grps[i] for i in order]
        self._offs_host = offs
        self._coff = torch.tensor(offs, dtype=torch.int64, device=dev)
        self._ccnt = torch.tensor(cnts, dtype=torch.int32, device=dev)
        self._cgrp = torch.tensor(grps, dtype=torch.int32, device=dev)
        self._tiles = torch.tensor(tiles, dtype=torch.int32, device=dev) if tiles else None
        self._late = None
        if l_offs or l_tiles:
            self._late = (torch.tensor(l_offs, dtype=torch.int64, device=dev), torch.tensor(l_cnts, dtype=torch.int32, device=dev),
                          torch.tensor(l_grps, dtype=torch.int32, device=dev),
                          torch.tensor(l_tiles, dtype=torch.int32, device=dev) if l_tiles else None)
        self._sig = sig
        self._ranges = {}

    def _hyper(self):
        h = L.AdamWHyper()
        b1 = b2 = eps = None
        for gi, g in enumerate(self.param_groups):
            h.lr[gi], h.weight_decay[gi], h.active[gi] = float(g["lr"]), float(g["weight_decay"]), 1
            if b1 is None:
                (b1, b2), eps = g["betas"], g["eps"]
            elif (b1, b2) != tuple(g["betas"]) or eps != g["eps"]:
                raise NotImplementedError("per-group betas/eps are not supported by the fused kernel")
        h.active[INACTIVE_GROUP] = 0
        h.beta1, h.beta2, h.eps = float(b1), float(b2), float(eps)
        return h

    def _launch(self, i0, i1, h, stream, tab=None):
        if i1 <= i0:
            return
        arena = self._arena
        coff, ccnt, cgrp = tab if tab is not None else (self._coff, self._ccnt, self._cgrp)
        L.check(L.load().ilvlm_adamw_step(arena.P.data_ptr(), arena.G.data_ptr(), self.M.data_ptr(), self.V.data_ptr(),
                                          arena.S.data_ptr() if arena.S is not None else None, coff.data_ptr() + 8 * i0,
                                          ccnt.data_ptr() + 4 * i0, cgrp.data_ptr() + 4 * i0, int(i1 - i0),
                                          C.byref(h), stream), "adamw_step")

    def _launch_tiles(self, tiles, h, stream):
        arena, pk = self._arena, self._arena.packed
        L.check(L.load().ilvlm_adamw_step_packed(arena.P.data_ptr(), arena.G.data_ptr(), self.M.data_ptr(), self.V.data_ptr(),
                                                 arena.S.data_ptr(), pk.fwd.data_ptr(), pk.bwd.data_ptr(), tiles.data_ptr(),
                                                 int(tiles.shape[0]), C.byref(h), stream), "adamw_step_packed")

    # -- update beside the next forward ---------------------------------------------------------
    def defer_late_blocks(self, first_block=2):
        """Opt-in for training loops in which nothing reads parameters between step() and the next forward except through
        flush() (the solver and bench.py): step() updates on the calling stream only what the next forward reads first -- the
        embeddings, the heads, the transformer blocks below `first_block` of each tower -- and the rest (blocks first_block.. of
        both towers: two thirds of the parameters) on the gradient-memset side stream, where the 5 TB/s HBM stream of the update
        runs beside the first blocks' GEMMs instead of in front of them.  The engine's tower calls wait for that part in front
        of block `first_block` (engine.tower_fwd); any other reader goes through flush() / engine.prepare().  AdamW is
        element-wise: bit-identical results.  0 switches it off.  bf16 mode (fp8 re-quantises every weight at the start of a
        forward) and not with overlap_backward()."""
        self._defer_from = max(0, int(first_block))

    def flush(self):
        """order the current stream behind a deferred part of the last step() (no-op without one)"""
        a = self._arena
        if a is not None and a.late_event is not None:
            torch.cuda.current_stream(a.P.device).wait_event(a.late_event)

    # -- update inside backward ---------------------------------------------------------------
    def overlap_backward(self, enabled=True):
        """Opt-in for training loops in the reference's order (train_solver.py:348-439: zero_grad, ONE backward, step, and
        nothing that reads or rescales gradients in between): the update of a transformer block is issued on a companion
        stream as soon as the block's gradients are final, while backward continues on the earlier blocks.  AdamW is
        element-wise, hence the result is bit-identical to the update done in step(), which then only takes the ranges that
        are left and joins the streams.  Not valid with gradient accumulation over several backward calls.  Measured on one
        MI355X (ViT-B/32 + FDT, batch 256): 17.9 ms per step either way -- two towers plus their weight-gradient streams
        already fill the chip during backward, so the 0.84 ms update only moves; off by default."""
        self._overlap = bool(enabled)
        if self._arena is not None:
            self._arena.eager_opt = self if self._overlap else None

    def block_grads_final(self, prefix, wg=None):
        """called by the model (base._sync) from inside backward, on the stream that produced the block's gradients; wg =
        that stream's weight-gradient companion"""
        if not self._overlap or self._pending_state:
            return
        arena = self._arena
        if self._sig is None:
            self._build_table()
        r = self._ranges.get(prefix)
        if r is None:
            b, e = arena.range_of(prefix)
            r = self._ranges[prefix] = (bisect.bisect_left(self._offs_host, b), bisect.bisect_left(self._offs_host, e))
        if r[1] <= r[0]:
            return
        # No stream of its own: HIP multiplexes streams onto a few hardware queues, and a fifth stream shares one with a tower
        # stream, whose later kernels then queue behind this update's wait for the weight gradients (measured: 19.4 instead
        # of 17.9 ms per step).  The update goes where its last producer already runs: the gradient-mean stream when
        # ranks exchange gradients, else the weight-gradient companion of the calling stream, else the calling stream.
        cur = torch.cuda.current_stream()
        red = arena.reducer
        if red is not None and red.stream is not None and red.pending:
            ost = red.stream                          # in order behind this block's mean over ranks
        elif wg is not None:
            ost = wg                                  # in order behind this block's weight gradients
            ost.wait_stream(cur)                      # LayerNorm / bias gradients come from the calling stream
        else:
            ost = cur
        self._ostreams.add(ost)
        h = self._hyper()
        h.step = self._step + 1
        self._launch(r[0], r[1], h, ost.cuda_stream)
        self._eager.append(r)

    def _ingest_loaded_state(self):
        """After load_state_dict: copy the loaded moments into the arenas and re-point the state at the views."""
        steps = [0]
        for p, st in list(self.state.items()):
            m, v = self._views(p)
            if st["exp_avg"].data_ptr() != m.data_ptr():
                m.copy_(st["exp_avg"]); v.copy_(st["exp_avg_sq"])
                st["exp_avg"], st["exp_avg_sq"] = m, v
            steps.append(int(float(st["step"])))
        self._step = max(steps)
        self._pending_state = False

    # -- torch.optim.Optimizer surface ---------------------------------------------------------
    def zero_grad(self, set_to_none=False):
        """Zeroes the flat gradient arena with one memset (gradient views stay attached)."""
        if self._arena is None:
            try:
                self._bind()
            except RuntimeError:
                return super().zero_grad(set_to_none=False)
        self._arena.zero_grad()
        if self._overlap:
            self._build_table()       # requires_grad flags are read once per step, before backward starts using the table

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closures are not supported")
        if self._arena is None:
            self._bind()
        if self._pending_state:
            self._ingest_loaded_state()
        arena = self._arena
        arena.wait_grads()
        self.flush()                                   # the deferred part of the previous step owns the same moments
        if not self._eager:
            self._build_table()
        self._step += 1
        h = self._hyper()
        h.step = self._step
        st = torch.cuda.current_stream()
        cur = 0
        for i0, i1 in sorted(self._eager):             # what the backward pass has not updated already
            self._launch(cur, i0, h, st.cuda_stream)
            cur = max(cur, i1)
        self._launch(cur, len(self._offs_host), h, st.cuda_stream)
        packed_done = False
        if self._tiles is not None:
            self._launch_tiles(self._tiles, h, st.cuda_stream)
            packed_done = True
        arena.late_event, arena.late_from = None, 0
        if self._late is not None:
            # the blocks from defer_from up: on the side stream, behind everything enqueued here (the gradients are final,
            # the clip has scaled them); the gradient memset of prezero_grads follows on the same stream
            zs = arena.side_stream()
            zs.wait_stream(st)
            coff, ccnt, cgrp, ltiles = self._late
            self._launch(0, int(coff.shape[0]), h, zs.cuda_stream, (coff, ccnt, cgrp))
            if ltiles is not None:
                self._launch_tiles(ltiles, h, zs.cuda_stream)
                packed_done = True
            ev = torch.cuda.Event()
            ev.record(zs)
            arena.late_event, arena.late_from = ev, self._defer_from
        if self._eager:
            for o in self._ostreams:
                if o is not st:
                    st.wait_stream(o)
            self._ostreams.clear()
            self._eager = []
        arena.shadow_fresh = arena.S is not None       # the kernel wrote the bf16 shadow of every element it updated
        arena.packed_fresh = packed_done               # ... and both fragment-order images of every packed weight it updated
        if self.prezero_grads:
            arena.prezero_grads()

    def state_dict(self):
        self.flush()
        for st in self.state.values():
            st["step"] = torch.tensor(float(self._step))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._pending_state = True
        if self._arena is not None:
            self._ingest_loaded_state()


AdamW = FusedAdamW


def optim_entry(config):
    kwargs = dict(config["kwargs"])
    if config["type"] != "AdamW":
        raise NotImplementedError("optimizer type %r: only AdamW (the shipped configs' choice) runs on the HIP path" % config["type"])
    return FusedAdamW(**kwargs)

"""Data-parallel exchange steps of the contrastive path over torch.distributed (backend 'nccl' = RCCL over
xGMI on MI355X; 'gloo' in the CPU tests).  Device-agnostic plumbing: no arithmetic besides the collectives and the
dtype change of the gradient buckets.

Reference behaviour restated (file:line relative to the reference root):
  AllGather.forward   prototype/model/clip_fdt.py:166-178 -- dist.all_gather of [B,D] twice per step (image, text)
  AllGather.backward  prototype/model/clip_fdt.py:180-188 -- all_reduce(SUM) of the full [W,B,D] gradient, then [rank]
  DDP gradient mean   prototype/utils/torch_ddp_dist.py:52-67
Here: ONE fused all-gather of [2,B,D] forward, ONE reduce-scatter of [W,2,B,D] backward (mathematically the
all-reduce + slice of the reference at 1/W of the traffic), and the gradient mean of ranges of the flat gradient arena
on a side stream, per transformer block as soon as the block's backward is complete (prototype/utils/torch_ddp_dist.py).

Gradient buckets.  Default: fp32, the reference's arithmetic.  `ILVLM_GRAD_BUCKET=bf16` (opt-in, see NativeDDP) sends the
mean in bf16: the fp32 range is cast into a bf16 bucket on the communication stream, reduced, and widened back into the
arena -- 309 MB instead of 618 MB per step on the wire, at one bf16 rounding of every averaged gradient.  `ILVLM_GRAD_ALGO=rs_ag` expresses the mean as reduce-scatter + all-gather (each rank owns 1/W of the
bucket), the formulation whose two halves RCCL can run as direct exchanges over the 7 xGMI links of the full mesh;
`allreduce` (default) leaves the choice to RCCL.

One code path: the collectives below are issued at every world size when `force_collectives(True)` (or
ILVLM_COMM_FORCE=1) is set -- a world of one rank then runs the same RCCL calls a world of eight does, which is how the
one-GPU box exercises the production branch (tests/test_comm_nccl_gpu.py).  Without the switch a single rank skips them.
"""
import contextlib
import os

import torch
import torch.distributed as dist

_FORCE = os.environ.get("ILVLM_COMM_FORCE", "0") == "1"
_TRACE = None        # list of (op, begin, end, dtype, elements) per collective issued, in issue order (trace(True))


def trace(on=True):
    """Record every collective this module issues -- (op, range begin, range end, dtype, elements) in issue order -- and return
    the list (None when switched off).  RCCL matches collectives by their ORDER on each rank: a step whose sequence differed
    between ranks (say, because two tower streams announced their ranges in a timing-dependent order) would deadlock or mix
    buffers at world size 8.  The sequence here is a function of the host program alone; tests compare it across ranks."""
    global _TRACE
    _TRACE = [] if on else None
    return _TRACE


def _note(op, begin, end, t):
    if _TRACE is not None:
        _TRACE.append((op, int(begin), int(end), str(t.dtype).replace("torch.", ""), int(t.numel())))


def force_collectives(on):
    """issue the collectives even at world size 1 (test hook; results are unchanged by construction)"""
    global _FORCE
    _FORCE = bool(on)


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _active(W):
    return W > 1 or (_FORCE and dist.is_available() and dist.is_initialized())


def _has_reduce_scatter():
    return dist.get_backend() == "nccl"       # gloo implements neither reduce_scatter nor ReduceOp.AVG


def gather_pair(img, txt):
    """[B,D] x2 -> rank-major [W*B,D] x2 (own rows included)."""
    rank, W = world()
    if not _active(W):
        return img, txt
    B, D = img.shape
    send = torch.stack([img, txt], 0).contiguous()                  # [2,B,D]
    recv = torch.empty((W, 2, B, D), dtype=img.dtype, device=img.device)
    _note("all_gather", 0, 0, send)
    dist.all_gather_into_tensor(recv.view(W * 2 * B, D), send.view(2 * B, D))
    return recv[:, 0].reshape(W * B, D).contiguous(), recv[:, 1].reshape(W * B, D).contiguous()


def reduce_gathered(dg_img, dg_txt, B):
    """Gradients of the gathered matrices [W*B,D] x2 -> this rank's [B,D] slices summed over ranks."""
    rank, W = world()
    if not _active(W):
        return dg_img, dg_txt
    D = dg_img.shape[1]
    send = torch.stack([dg_img.view(W, B, D), dg_txt.view(W, B, D)], 1).contiguous()   # [W,2,B,D]
    _note("reduce_scatter", 0, 0, send)
    if _has_reduce_scatter():
        out = torch.empty((2, B, D), dtype=send.dtype, device=send.device)
        dist.reduce_scatter_tensor(out.view(2 * B, D), send.view(W * 2 * B, D), op=dist.ReduceOp.SUM)
        return out[0], out[1]
    dist.all_reduce(send, op=dist.ReduceOp.SUM)       # gloo has no reduce-scatter: reference formulation
    return send[rank, 0].contiguous(), send[rank, 1].contiguous()


def _narrow(src, dst):
    """dst (bf16) = src (fp32)"""
    if src.is_cuda:
        from . import ops
        ops.cast_f32(src, dst)
    else:
        dst.copy_(src)


def _widen(src, dst):
    """dst (fp32) = src (bf16)"""
    if src.is_cuda:
        from . import ops
        ops.cast_to_f32(src, dst)
    else:
        dst.copy_(src)


class GradReducer:
    """Mean over ranks of ranges of a flat fp32 gradient buffer, on a dedicated communication stream.

    bucket: 'fp32' reduces the arena in place; 'bf16' reduces a bf16 copy of each chunk (half the bytes on the wire).
    algo:   'allreduce' | 'rs_ag' (reduce-scatter of the chunk + all-gather of the owned shards)."""

    def __init__(self, flat, bucket=None, algo=None):
        self.flat = flat
        self.rank, self.W = world()
        self.cuda = flat.is_cuda
        self.stream = torch.cuda.Stream(device=flat.device) if self.cuda else None
        self.pending = False
        self.bucket = bucket or os.environ.get("ILVLM_GRAD_BUCKET", "fp32")
        self.algo = algo or os.environ.get("ILVLM_GRAD_ALGO", "allreduce")
        if self.bucket not in ("fp32", "bf16") or self.algo not in ("allreduce", "rs_ag"):
            raise ValueError("GradReducer: bucket must be fp32|bf16 and algo allreduce|rs_ag, got %r %r" % (self.bucket, self.algo))
        self._staging = {}          # chunk elements -> reusable bf16 / shard buffers (allocated on the comm stream)
        self.bytes_sent = 0         # bytes handed to the collectives since construction (tests)
        self.calls = 0              # collective calls since construction
        self.timing = None          # set to [] (bench.py): wait() records (consumer-stream event, comm-stream event) pairs

    def _buf(self, key, n, dtype):
        b = self._staging.get((key, dtype))
        if b is None or b.numel() < n:
            b = torch.empty(n, dtype=dtype, device=self.flat.device)
            self._staging[(key, dtype)] = b
        return b[:n]

    def _mean(self, t, begin=0, end=0):
        """in-place mean over ranks of the 1-D tensor t (fp32 or bf16), the arena range [begin, end)"""
        W = self.W
        self.bytes_sent += t.numel() * t.element_size()
        self.calls += 1
        _note("grad_mean:" + self.algo, begin, end, t)
        if self.algo == "rs_ag":
            n = t.numel()
            per = (n + W - 1) // W
            if per * W != n:                       # pad the tail so that every rank owns an equal shard
                padded = self._buf("pad", per * W, t.dtype)
                padded[:n].copy_(t)
                padded[n:].zero_()
            else:
                padded = t
            shard = self._buf("shard", per, t.dtype)
            if _has_reduce_scatter():
                dist.reduce_scatter_tensor(shard, padded, op=dist.ReduceOp.AVG)
                dist.all_gather_into_tensor(padded, shard)
            else:
                # gloo (CPU tests) has neither reduce_scatter nor AVG: the same shard bookkeeping over an all-reduce of the
                # padded buffer, the owned shard divided, then gathered back -- the ragged-range logic is what is under test
                dist.all_reduce(padded, op=dist.ReduceOp.SUM)
                shard.copy_(padded[self.rank * per:(self.rank + 1) * per])
                shard.div_(W)
                parts = [torch.empty_like(shard) for _ in range(W)]
                dist.all_gather(parts, shard)
                padded.copy_(torch.cat(parts))
            if padded is not t:
                t.copy_(padded[:n])
        elif _has_reduce_scatter():
            dist.all_reduce(t, op=dist.ReduceOp.AVG)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            t.div_(W)

    def reduce_range(self, begin, end, chunk_elems=32 * 1024 * 1024, also_wait=None):
        """Enqueue the mean of flat[begin:end]; the producer stream's work so far is waited for (and that of `also_wait`, the
        weight-gradient companion of the producer stream: the communication stream waits for it, the producer does not)."""
        if not _active(self.W) or end <= begin:
            return
        if self.cuda:
            self.stream.wait_stream(torch.cuda.current_stream(self.flat.device))
            if also_wait is not None:
                self.stream.wait_stream(also_wait)
            ctx = torch.cuda.stream(self.stream)
        else:
            ctx = contextlib.nullcontext()
        with ctx:
            for a in range(begin, end, chunk_elems):
                piece = self.flat[a:min(end, a + chunk_elems)]
                if self.bucket == "bf16":
                    # arena offsets are multiples of 64 elements, so both views keep the 16-byte alignment of the casts
                    lp = self._buf("lp", piece.numel(), torch.bfloat16)
                    _narrow(piece, lp)
                    self._mean(lp, a, a + piece.numel())
                    _widen(lp, piece)
                else:
                    self._mean(piece, a, a + piece.numel())
        self.pending = True

    def wait(self):
        """Make the current stream wait for every enqueued reduction (call before the optimizer reads gradients)."""
        if self.pending and self.cuda:
            cur = torch.cuda.current_stream(self.flat.device)
            if self.timing is not None:
                # exposed communication of this step = how much later the communication stream finishes than the consumer
                # stream reaches this wait (both events on their own streams; read after a synchronise)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
                e1.record(self.stream)
                self.timing.append((e0, e1))
            cur.wait_stream(self.stream)
        self.pending = False

    def exposed_ms(self):
        """mean over the recorded waits of max(0, comm-stream end - consumer arrival) in ms (after a device synchronise)"""
        if not self.timing:
            return 0.0
        return sum(max(0.0, a.elapsed_time(b)) for a, b in self.timing) / len(self.timing)

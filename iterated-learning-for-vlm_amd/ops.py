"""Thin tensor-level wrappers over the C ABI (include/ilvlm_hip.h).  PyTorch is used only for device
memory and the current HIP stream; every computation below is a kernel of libilvlm_hip.so.  The
wrappers check shapes/dtypes on the host (a kernel that indexes out of bounds can take the GPU
down) and raise RuntimeError on any failure; there is no fallback path."""
import ctypes as C
import math
import os
import threading

import torch

from . import lib as L
from .lib import F32, BF16, GemmEpilogue  # noqa: F401

_TD = {torch.float32: F32, torch.bfloat16: BF16}


def dt(t):
    return _TD[t.dtype]


def torch_dtype(code):
    return torch.float32 if code == F32 else torch.bfloat16


# Launch-stream override (raw stream handle): inside `with stream_override(h):` every launch of THIS thread goes to h
# instead of torch's current stream -- the engine sends weight-gradient GEMMs to a companion stream that way, which is
# cheaper than entering a torch stream context per GEMM.  Thread-local on purpose: ctypes releases the GIL during a launch,
# and the prefetcher / checkpoint threads (solver.DevicePrefetcher) launch kernels of their own on their own streams; a
# process-global switch would route those onto the weight-gradient stream, un-ordered against the copies they depend on.
_tls = threading.local()


class stream_override:
    __slots__ = ("handle", "prev")

    def __init__(self, handle):
        self.handle = handle

    def __enter__(self):
        self.prev = getattr(_tls, "override", None)
        _tls.override = self.handle
        return self

    def __exit__(self, *exc):
        _tls.override = self.prev
        return False


def _override():
    return getattr(_tls, "override", None)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)      # the handle without building a Stream object (~1 us vs ~10)
_cur_device = getattr(torch._C, "_cuda_getDevice", torch.cuda.current_device)


def _stream():
    s = getattr(_tls, "override", None)
    if s is not None:
        return s
    if _raw_stream is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _event_stream():
    """torch stream object matching _stream() (for event records of the GEMM profiler)"""
    s = getattr(_tls, "override", None)
    return torch.cuda.ExternalStream(s) if s is not None else torch.cuda.current_stream()


def _p(t):
    return None if t is None else t.data_ptr()


def _chk(t, name, dtype=None, shape=None):
    if not t.is_cuda:
        raise RuntimeError("%s must be a device tensor (no CPU fallback)" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError("%s: expected %s got %s" % (name, dtype, t.dtype))
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise RuntimeError("%s: expected shape %s got %s" % (name, tuple(shape), tuple(t.shape)))


def selftest_fragments():
    out = torch.zeros(5, 64, 8, device="cuda", dtype=torch.float32)
    L.check(L.load().ilvlm_selftest_fragments(out.data_ptr(), _stream()), "selftest")
    return out


# ---------------------------------------------------------------------------------------------
class GemmProfiler:
    """Brackets every bf16 GEMM launch with HIP events on the launch stream (bench.py's roofline leg).  `ms` is the sum of
    the launch durations; `union_ms` the time during which at least one of them was running (they overlap when the towers
    run on their own streams), measured against a base event recorded when the profiler was created."""

    def __init__(self):
        self.records = []     # (start_event, end_event, flops, algorithmic bytes)
        self.f32_flops = 0.0  # fp32 GEMM launches (not timed): att_w @ sd, logits and their gradients
        self.base = torch.cuda.Event(enable_timing=True)
        self.base.record()

    def summary(self):
        torch.cuda.synchronize()
        ms = sum(r[0].elapsed_time(r[1]) for r in self.records)
        f8 = [r for r in self.records if len(r) > 4]
        fp8 = dict(launches=len(f8), ms=sum(r[0].elapsed_time(r[1]) for r in f8), flops=float(sum(r[2] for r in f8)),
                   bytes=float(sum(r[3] for r in f8)))
        spans = sorted((self.base.elapsed_time(r[0]), self.base.elapsed_time(r[1])) for r in self.records)
        union, cur_a, cur_b = 0.0, None, None
        for a, b in spans:
            if cur_b is None or a > cur_b:
                if cur_b is not None:
                    union += cur_b - cur_a
                cur_a, cur_b = a, b
            else:
                cur_b = max(cur_b, b)
        if cur_b is not None:
            union += cur_b - cur_a
        return dict(launches=len(self.records), ms=ms, union_ms=union, flops=float(sum(r[2] for r in self.records)),
                    bytes=float(sum(r[3] for r in self.records)), f32_flops=self.f32_flops, fp8=fp8)


_gemm_profiler = None


def set_gemm_profiler(p):
    global _gemm_profiler
    _gemm_profiler = p


def gemm(a, b, out, *, trans_a=False, trans_b=False, bias=None, rowbias=None, residual=None, aux=None, act=0,
         alpha=1.0, alpha_ptr=None, accumulate=False, out_group=0, out_skip=0, split_k=1, M=None, N=None, K=None,
         a_rowsum=None, slab=None, b_packed=None):
    """out[m,n] = epilogue(sum_k A(m,k) B(n,k)); see ilvlm_gemm.  a, b: 2-D bf16 or fp32 (same dtype);
    out: 2-D.  With out_group > 0, `out` is the token-stream tensor the rows are mapped into.
    b_packed: the B operand in MFMA-fragment order (gemm_pack_b / pack_weights), n * k bf16 elements: the streaming
    kernel takes store-type bf16 GEMMs with it (bit-identical results)."""
    if a.dtype != b.dtype:
        raise RuntimeError("gemm: operand dtypes differ: %s %s" % (a.dtype, b.dtype))
    _chk(a, "gemm.a"); _chk(b, "gemm.b"); _chk(out, "gemm.out")
    m = (a.shape[1] if trans_a else a.shape[0]) if M is None else M
    k = (a.shape[0] if trans_a else a.shape[1]) if K is None else K
    n = (b.shape[1] if trans_b else b.shape[0]) if N is None else N
    kb = b.shape[0] if trans_b else b.shape[1]
    if kb != k and K is None:
        raise RuntimeError("gemm: K mismatch %d vs %d" % (k, kb))
    lda, ldb, ldc = a.stride(0), b.stride(0), out.stride(0)
    rows_out = m if out_group <= 0 else (m // out_group) * (out_group + out_skip)
    if out_group > 0 and m % out_group:
        raise RuntimeError("gemm: M=%d not a multiple of out_group=%d" % (m, out_group))
    if out.shape[0] < rows_out or out.shape[1] != n:
        raise RuntimeError("gemm: out shape %s too small for [%d,%d]" % (tuple(out.shape), rows_out, n))
    if bias is not None:
        _chk(bias, "gemm.bias", torch.float32, (n,))
    if rowbias is not None:
        _chk(rowbias, "gemm.rowbias", torch.float32, (out_group + out_skip, n))
    if residual is not None:
        _chk(residual, "gemm.residual", torch.float32, out.shape)
    if aux is not None:
        _chk(aux, "gemm.aux", a.dtype, (m, n))
        if aux.stride(0) != ldc:
            raise RuntimeError("gemm: aux stride must equal out stride")
    if accumulate and out.dtype != torch.float32:
        raise RuntimeError("gemm: accumulate needs an fp32 output")
    if a_rowsum is not None:
        _chk(a_rowsum, "gemm.a_rowsum", torch.float32, (m,))
    epi = GemmEpilogue(_p(bias), _p(rowbias), _p(residual), _p(aux), _p(alpha_ptr), float(alpha), int(act), dt(out),
                       int(bool(accumulate)), int(out_group), int(out_skip), _p(a_rowsum))
    if slab is not None:          # (workspace uint8, counters int32): slab split-K instead of atomics (ilvlm_gemm_epilogue)
        epi.splitk_ws, epi.splitk_ws_bytes = slab[0].data_ptr(), slab[0].numel()
        epi.splitk_cnt, epi.splitk_cnt_len = slab[1].data_ptr(), slab[1].numel()
    if b_packed is not None:
        _chk(b_packed, "gemm.b_packed", torch.bfloat16)
        if b_packed.numel() != n * k or a.dtype != torch.bfloat16:
            raise RuntimeError("gemm: b_packed must hold n * k = %d bf16 elements (bf16 GEMMs only), got %d" % (n * k, b_packed.numel()))
        epi.b_packed = b_packed.data_ptr()
    prof = _gemm_profiler if (_gemm_profiler is not None and a.dtype == torch.bfloat16) else None
    if _gemm_profiler is not None and prof is None:
        _gemm_profiler.f32_flops += 2.0 * m * n * k
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # the events go on the stream the kernel is launched on (a weight-gradient GEMM runs on the companion stream)
        est = _event_stream()
        ev0.record(est)
    L.check(L.load().ilvlm_gemm(dt(a), int(trans_a), int(trans_b), m, n, k, a.data_ptr(), lda, b.data_ptr(), ldb,
                                out.data_ptr(), ldc, C.byref(epi), int(split_k), _stream()), "gemm")
    if prof is not None:
        ev1.record(est)
        # algorithmic HBM bytes: both operands once, the output once (read-modify-write when accumulating), every
        # epilogue operand once
        nbytes = 2.0 * (m * k + n * k) + m * n * out.element_size() * (2 if accumulate else 1)
        nbytes += m * n * (4 if residual is not None else 0) + (m * n * aux.element_size() if aux is not None else 0)
        prof.records.append((ev0, ev1, 2.0 * m * n * k, nbytes))
    return out


def gemm_fp8(a8, b8, out, scale_a, scale_b, *, a_e5m2=False, bias=None, residual=None, aux=None, act=0, b_packed=None):
    """out[m,n] = epilogue(inv_a * inv_b * sum_k a8[m,k] b8[n,k]); a8 [M,K], b8 [N,K] uint8 holding OCP fp8 (e4m3; a8
    e5m2 with a_e5m2), scale_a / scale_b: 1-element fp32 device tensors with the de-quantisation factors.
    b_packed: b8 in fragment order (gemm_pack_b8 / the packed weight quantiser): the streaming kernel, same results."""
    _chk(a8, "gemm_fp8.a", torch.uint8); _chk(b8, "gemm_fp8.b", torch.uint8); _chk(out, "gemm_fp8.out")
    m, k = a8.shape
    n = b8.shape[0]
    if b8.shape[1] != k or tuple(out.shape) != (m, n):
        raise RuntimeError("gemm_fp8: shapes %s %s -> %s" % (tuple(a8.shape), tuple(b8.shape), tuple(out.shape)))
    if bias is not None:
        _chk(bias, "gemm_fp8.bias", torch.float32, (n,))
    if residual is not None:
        _chk(residual, "gemm_fp8.residual", torch.float32, out.shape)
    if aux is not None:
        _chk(aux, "gemm_fp8.aux", torch.bfloat16, (m, n))
    epi = GemmEpilogue(_p(bias), None, _p(residual), _p(aux), scale_a.data_ptr(), 1.0, int(act), dt(out), 0, 0, 0, None,
                       None, None, None, 0, scale_b.data_ptr())
    if b_packed is not None:
        _chk(b_packed, "gemm_fp8.b_packed", torch.uint8)
        if b_packed.numel() != n * k:
            raise RuntimeError("gemm_fp8: b_packed must hold n * k = %d bytes, got %d" % (n * k, b_packed.numel()))
        epi.b_packed = b_packed.data_ptr()
    prof = _gemm_profiler
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        est = _event_stream()
        ev0.record(est)
    L.check(L.load().ilvlm_gemm(L.FP8_BF8A if a_e5m2 else L.FP8, 0, 0, m, n, k, a8.data_ptr(), a8.stride(0), b8.data_ptr(),
                                b8.stride(0), out.data_ptr(), out.stride(0), C.byref(epi), 1, _stream()), "gemm(fp8)")
    if prof is not None:
        ev1.record(est)
        nbytes = 1.0 * (m * k + n * k) + m * n * out.element_size() + m * n * (4 if residual is not None else 0) + (
            m * n * 2 if aux is not None else 0)
        prof.records.append((ev0, ev1, 2.0 * m * n * k, nbytes, "fp8"))
    return out


def gemm_fp8_wgrad(dy8, x8, out, scale_dy, scale_x, *, split_k=1, rowsum=None, K=None, slab=None):
    """out[m,n] += inv_dy * inv_x * sum_t dy8[t,m] x8[t,n] (weight gradient in fp8 mode): dy8 [T,M] uint8 holding e5m2,
    x8 [T,N] uint8 holding e4m3, out fp32 [M,N]; rowsum[m] += inv_dy * sum_t dy8[t,m] (bias gradient)."""
    _chk(dy8, "gemm_fp8_wgrad.dy", torch.uint8); _chk(x8, "gemm_fp8_wgrad.x", torch.uint8)
    _chk(out, "gemm_fp8_wgrad.out", torch.float32)
    k = dy8.shape[0] if K is None else K
    m, n = dy8.shape[1], x8.shape[1]
    if x8.shape[0] < k or dy8.shape[0] < k or tuple(out.shape) != (m, n):
        raise RuntimeError("gemm_fp8_wgrad: shapes %s %s -> %s" % (tuple(dy8.shape), tuple(x8.shape), tuple(out.shape)))
    if rowsum is not None:
        _chk(rowsum, "gemm_fp8_wgrad.rowsum", torch.float32, (m,))
    epi = GemmEpilogue(None, None, None, None, scale_dy.data_ptr(), 1.0, 0, L.F32, 1, 0, 0, _p(rowsum),
                       None, None, None, 0, scale_x.data_ptr())
    if slab is not None:
        epi.splitk_ws, epi.splitk_ws_bytes = slab[0].data_ptr(), slab[0].numel()
        epi.splitk_cnt, epi.splitk_cnt_len = slab[1].data_ptr(), slab[1].numel()
    prof = _gemm_profiler
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        est = _event_stream()
        ev0.record(est)
    L.check(L.load().ilvlm_gemm(L.FP8_BF8A, 1, 1, m, n, k, dy8.data_ptr(), dy8.stride(0), x8.data_ptr(), x8.stride(0),
                                out.data_ptr(), out.stride(0), C.byref(epi), int(split_k), _stream()), "gemm(fp8 wgrad)")
    if prof is not None:
        ev1.record(est)
        prof.records.append((ev0, ev1, 2.0 * m * n * k, 1.0 * (m * k + n * k) + 8.0 * m * n, "fp8"))
    return out


def fp8_quantize(src, dst, scale, amax, e5m2=False):
    """dst (uint8, same shape; None = observe only) = fp8(src * scale[0]); amax[0] = max(amax[0], max|src|)"""
    _chk(src, "fp8_quantize.src")
    if dst is not None:
        _chk(dst, "fp8_quantize.dst", torch.uint8, src.shape)
    L.check(L.load().ilvlm_fp8_quantize(src.data_ptr(), dt(src), _p(dst), src.numel(), _p(scale), _p(amax), int(bool(e5m2)),
                                        _stream()), "fp8_quantize")
    return dst


def gemm_pack_b(b, trans_b=False, out=None):
    """B operand (2-D bf16; [N,K], or [K,N] with trans_b) in MFMA-fragment order for gemm(b_packed=...)"""
    _chk(b, "gemm_pack_b.b", torch.bfloat16)
    n, k = (b.shape[1], b.shape[0]) if trans_b else (b.shape[0], b.shape[1])
    if out is None:
        out = torch.empty(n * k, dtype=torch.bfloat16, device=b.device)
    _chk(out, "gemm_pack_b.out", torch.bfloat16, (n * k,))
    L.check(L.load().ilvlm_gemm_pack_b(int(trans_b), n, k, b.data_ptr(), b.stride(0), out.data_ptr(), _stream()), "gemm_pack_b")
    return out


def gemm_pack_b8(b8, out=None):
    """fp8 B operand (2-D uint8 [N,K]) in the fragment order gemm_fp8(b_packed=...) reads"""
    _chk(b8, "gemm_pack_b8.b", torch.uint8)
    n, k = b8.shape
    if out is None:
        out = torch.empty(n * k, dtype=torch.uint8, device=b8.device)
    _chk(out, "gemm_pack_b8.out", torch.uint8, (n * k,))
    L.check(L.load().ilvlm_gemm_pack_b8(n, k, b8.data_ptr(), b8.stride(0), out.data_ptr(), _stream()), "gemm_pack_b8")
    return out


def pack_weights(arena_bf16, fwd, bwd, table):
    """fragment-order copies (forward / input-gradient images) of every GEMM weight listed in `table` (int32 [n, 5])"""
    for t, nm in ((arena_bf16, "arena"), (fwd, "fwd"), (bwd, "bwd")):
        _chk(t, "pack_weights." + nm, torch.bfloat16)
    _chk(table, "pack_weights.table", torch.int32)
    if fwd.numel() != arena_bf16.numel() or bwd.numel() != arena_bf16.numel() or table.dim() != 2 or table.shape[1] != 5:
        raise RuntimeError("pack_weights: arena-shaped outputs and an [n, 5] table are required")
    L.check(L.load().ilvlm_pack_weights(arena_bf16.data_ptr(), fwd.data_ptr(), bwd.data_ptr(), table.data_ptr(), table.shape[0],
                                        _stream()), "pack_weights")


_WGRAD_GROUP_SLOTS = int(os.environ.get("ILVLM_WGRAD_GROUP_SLOTS", "512"))


def wgrad_group_split(tiles, rows, slots=None, fp8=False):
    """K-slices ilvlm_wgrad_group picks for `tiles` 128 x 128 output tiles over `rows` token rows (mirror of the C rule)"""
    slots = slots or _WGRAD_GROUP_SLOTS
    nt = -(-rows // (128 if fp8 else 64))
    cap = min(16, nt, rows // 256 if rows >= 256 else 1)
    best, split = None, 1
    for sp in range(1, cap + 1):
        c = -(-tiles * sp // slots) * (-(-nt // sp) + (8.0 if sp == 1 else 25.0))
        if best is None or c < best:
            best, split = c, sp
    return split


def wgrad_group(problems, rows, target=None, fp8=False):
    """ONE launch for the weight (and bias) gradients of up to four linears that share their token rows.  problems: list of
    (dy [rows, n], x [rows, k], gw fp32 [n, k], gb fp32 [n] or None[, inv_g, inv_x]); bf16 operands, or fp8 (dy e5m2 /
    x e4m3 as uint8, with their de-quantisation scale tensors).  Accumulates; see ilvlm_wgrad_group for the single-writer rule."""
    if not 1 <= len(problems) <= L.WGRAD_GROUP_MAX:
        raise RuntimeError("wgrad_group: 1..%d problems" % L.WGRAD_GROUP_MAX)
    arr = (L.WgradProblem * len(problems))()
    od = torch.uint8 if fp8 else torch.bfloat16
    for i, pr in enumerate(problems):
        dy, x, gw, gb = pr[:4]
        _chk(dy, "wgrad_group.dy", od)
        _chk(x, "wgrad_group.x", od)
        _chk(gw, "wgrad_group.gw", torch.float32)
        if dy.dim() != 2 or x.dim() != 2 or dy.shape[0] != rows or x.shape[0] != rows or tuple(gw.shape) != (dy.shape[1], x.shape[1]):
            raise RuntimeError("wgrad_group: problem %d: dy [rows, n], x [rows, k], gw [n, k] expected, got %s %s %s" % (
                i, tuple(dy.shape), tuple(x.shape), tuple(gw.shape)))
        if gb is not None:
            _chk(gb, "wgrad_group.gb", torch.float32)
            if gb.numel() != dy.shape[1]:
                raise RuntimeError("wgrad_group: problem %d: gb must have n elements" % i)
        arr[i].dy, arr[i].x, arr[i].gw = dy.data_ptr(), x.data_ptr(), gw.data_ptr()
        arr[i].gb = gb.data_ptr() if gb is not None else None
        arr[i].n, arr[i].k = dy.shape[1], x.shape[1]
        if fp8:
            arr[i].inv_g, arr[i].inv_x = pr[4].data_ptr(), pr[5].data_ptr()
    L.check(L.load().ilvlm_wgrad_group(L.FP8_BF8A if fp8 else L.BF16, arr, len(problems), int(rows),
                                       int(target if target is not None else _WGRAD_GROUP_SLOTS), _stream()), "wgrad_group")


def gemm_set_variant(v):
    L.check(L.load().ilvlm_gemm_set_variant(int(v)), "gemm_set_variant")


def gemm_set_tile_rows(rows=-1):
    """streaming kernel: tile height 128 / 96 / 64 rows, 0 = per-launch cost model, -1 = default (128)"""
    L.check(L.load().ilvlm_gemm_set_tile_rows(int(rows)), "gemm_set_tile_rows")


def gemm_set_wgrad_tile(rows=-1):
    """bf16 weight-gradient kernel: workgroup tile 128 (128 x 128), 256 (256 x 128, two stages), 257 (256 x 128, one stage), -1 = default"""
    L.check(L.load().ilvlm_gemm_set_wgrad_tile(int(rows)), "gemm_set_wgrad_tile")


def gemm_set_concurrent(concurrent):
    """regime hint: several GEMM streams in flight (towers + weight-gradient companions) -> kernels measured best inside the step"""
    L.check(L.load().ilvlm_gemm_set_concurrent(int(bool(concurrent))), "gemm_set_concurrent")


def gemm_set_persistent(slots=0, epi_sep=-1, stagger=-1):
    """persistent streaming kernel: workgroups per launch (0 = two per CU), epilogue LDS placement (-1 = default) and the start
    delay of each CU's second workgroup in cycles per K-tile (-1 = default)"""
    L.check(L.load().ilvlm_gemm_set_persistent(int(slots), int(epi_sep), int(stagger)), "gemm_set_persistent")


def rowsum_fusable(m, k):
    """True when a weight-gradient GEMM with output rows m and reduction k can also produce the bias gradient (the
    direct-to-LDS kernel takes any reduction length when both operands are K-strided, as they are in a weight gradient)."""
    return m % 8 == 0 and m >= 8


_WGRAD_TARGET = int(os.environ.get("ILVLM_WGRAD_TARGET", "384"))


def wgrad_split(out_rows, out_cols, k, tile=128):
    """split-K factor for a weight-gradient GEMM: enough workgroups to fill 256 CUs."""
    tiles = math.ceil(out_rows / tile) * math.ceil(out_cols / tile)
    s = max(1, min(16, round(_WGRAD_TARGET / tiles)))
    return max(1, min(s, k // 256 if k >= 256 else 1))


# ---------------------------------------------------------------------------------------------
def layernorm_fwd(x, gamma, beta, y, mean, rstd, rows, cols, eps=1e-5, group=0, skip=0):
    L.check(L.load().ilvlm_layernorm_fwd(x.data_ptr(), dt(x), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), dt(y),
                                         mean.data_ptr(), rstd.data_ptr(), rows, cols, eps, group, skip, _stream()),
            "layernorm_fwd")


# workgroups of a LayerNorm backward launch = partial dgamma / dbeta rows for its second stage.  512 (x 4 waves, grid-stride
# over the rows): the backward kernel holds 144 VGPRs at width 768, so only 768 workgroups are resident at once and a grid
# of 1024 ran a second, quarter-full round; measured 45.9 -> 41.1 us for backward + second stage at 12800 x 768.
LN_WS_BLOCKS = int(os.environ.get("ILVLM_LN_WS_BLOCKS", "512"))
_ln_ws = {}


def _ln_workspace(device, cols):
    """per-device scratch for the two-stage dgamma/dbeta reduction (2 x LN_WS_BLOCKS x cols floats)"""
    key = (device, torch.cuda.current_stream().cuda_stream)
    ws = _ln_ws.get(key)
    if ws is None or ws.numel() < 2 * LN_WS_BLOCKS * cols:
        ws = torch.empty(2 * LN_WS_BLOCKS * max(cols, 1024), device=device, dtype=torch.float32)
        _ln_ws[key] = ws
    return ws


def layernorm_bwd(dy, x, mean, rstd, gamma, dgamma, dbeta, rows, cols, dres=None, dx_f32=None, dx_lp=None, act=0,
                  act_aux=None, group=0, skip=0, two_stage=True):
    ws = _ln_workspace(dy.device, cols) if two_stage else None
    L.check(L.load().ilvlm_layernorm_bwd(dy.data_ptr(), dt(dy), x.data_ptr(), dt(x), mean.data_ptr(), rstd.data_ptr(),
                                         gamma.data_ptr(), _p(dres), _p(dx_f32), _p(dx_lp),
                                         dt(dx_lp) if dx_lp is not None else 0, act, _p(act_aux), dgamma.data_ptr(),
                                         dbeta.data_ptr(), rows, cols, group, skip, _p(ws), LN_WS_BLOCKS, _stream()),
            "layernorm_bwd")


def block_desc(E, H, causal, dtype, params, grads):
    """ilvlm_block from tensors: params / grads are dicts keyed by the struct field names (grads: missing or None =
    frozen)."""
    b = L.Block()
    for k, t in params.items():
        setattr(b, k, t.data_ptr())
    for k in ("g_ln1_w", "g_ln1_b", "g_ln2_w", "g_ln2_b", "g_in_w", "g_in_b", "g_out_w", "g_out_b", "g_fc_w", "g_fc_b",
              "g_proj_w", "g_proj_b"):
        t = grads.get(k)
        setattr(b, k, None if t is None else t.data_ptr())
    b.E, b.H, b.causal, b.dtype = int(E), int(H), int(causal), _TD[dtype]
    return b


def block_saved_bytes(desc, rows, B, Lq):
    n = L.load().ilvlm_block_saved_bytes(C.byref(desc), rows, B, Lq)
    if n <= 0:
        raise RuntimeError("ilvlm block_saved_bytes: bad arguments")
    return n


def block_scratch_bytes(desc, rows):
    n = L.load().ilvlm_block_scratch_bytes(C.byref(desc), rows)
    if n <= 0:
        raise RuntimeError("ilvlm block_scratch_bytes: bad arguments")
    return n


def block_fwd(desc, x_in, x_out, saved, B, Lq, seq=None):
    rows = x_in.shape[0]
    _chk(x_in, "block.x_in", torch.float32, (rows, desc.E)); _chk(x_out, "block.x_out", torch.float32, (rows, desc.E))
    _chk(saved, "block.saved", torch.uint8)
    L.check(L.load().ilvlm_block_fwd(C.byref(desc), x_in.data_ptr(), x_out.data_ptr(), saved.data_ptr(), rows, B, Lq,
                                     seq.cap if seq is not None else Lq, seq.offs.data_ptr() if seq is not None else None,
                                     _stream()), "block_fwd")


def ln_reduce_batched(ws, slot_stride, n_slots, rows, cols, grad_ptrs):
    """second stage of n_slots deferred LayerNorm backward launches in one kernel (ilvlm_layernorm_bwd_reduce_batched)"""
    _chk(ws, "ln_reduce.ws", torch.float32); _chk(grad_ptrs, "ln_reduce.ptrs", torch.int64, (2 * n_slots,))
    if ws.numel() < n_slots * slot_stride:
        raise RuntimeError("ln_reduce_batched: workspace too small")
    L.check(L.load().ilvlm_layernorm_bwd_reduce_batched(ws.data_ptr(), slot_stride, n_slots, rows, LN_WS_BLOCKS, cols,
                                                        grad_ptrs.data_ptr(), _stream()), "layernorm_bwd_reduce_batched")


def block_bwd(desc, x_in, saved, dx_f32, dx_lp, din_f32, din_lp, scratch, B, Lq, seq=None, wgrad_stream=None, ln_slots=None,
              dx8=None, din8=None, din8_scale=None, din8_amax=None):
    """ln_slots: fp32 tensor of 2 slots x 2 * LN_WS_BLOCKS * E floats -> the LayerNorm dgamma / dbeta second stages are
    deferred (ln_2's partials in slot 0, ln_1's in slot 1; ln_reduce_batched adds them up); None: reduced immediately."""
    rows = x_in.shape[0]
    _chk(dx_f32, "block.dx", torch.float32, (rows, desc.E)); _chk(din_f32, "block.din", torch.float32, (rows, desc.E))
    _chk(scratch, "block.scratch", torch.uint8); _chk(saved, "block.saved", torch.uint8)
    if ln_slots is not None:
        _chk(ln_slots, "block.ln_slots", torch.float32)
        if ln_slots.numel() < 4 * LN_WS_BLOCKS * desc.E:
            raise RuntimeError("block_bwd: ln_slots too small")
    ws = ln_slots if ln_slots is not None else _ln_workspace(x_in.device, desc.E)
    L.check(L.load().ilvlm_block_bwd(C.byref(desc), x_in.data_ptr(), saved.data_ptr(), dx_f32.data_ptr(), _p(dx_lp),
                                     din_f32.data_ptr(), _p(din_lp), scratch.data_ptr(), ws.data_ptr(),
                                     -LN_WS_BLOCKS if ln_slots is not None else LN_WS_BLOCKS, rows,
                                     B, Lq, seq.cap if seq is not None else Lq,
                                     seq.offs.data_ptr() if seq is not None else None, _WGRAD_TARGET, _stream(),
                                     None if wgrad_stream is None else wgrad_stream.cuda_stream, _p(dx8), _p(din8),
                                     _p(din8_scale), _p(din8_amax)), "block_bwd")


def tower_fwd(descs, x0, xs, saved, saved_stride, B, Lq, seq=None):
    """ilvlm_tower_fwd: all blocks of a tower in one call; xs [n, rows, E] receives the block outputs, saved n x saved_stride bytes"""
    n, rows, E = xs.shape
    _chk(x0, "tower.x0", torch.float32, (rows, E)); _chk(xs, "tower.xs", torch.float32); _chk(saved, "tower.saved", torch.uint8)
    if len(descs) != n or saved.numel() < n * saved_stride:
        raise RuntimeError("tower_fwd: %d descriptors for %d blocks / saved buffer too small" % (len(descs), n))
    arr = (L.Block * n)(*descs)
    L.check(L.load().ilvlm_tower_fwd(arr, n, x0.data_ptr(), xs.data_ptr(), saved.data_ptr(), saved_stride, rows, B, Lq,
                                     seq.cap if seq is not None else Lq, seq.offs.data_ptr() if seq is not None else None,
                                     _stream()), "tower_fwd")


def tower_bwd(descs, per_block, x0, xs, saved, saved_stride, dtop_f32, dtop_lp, d_f32, scratch, scratch_stride, B, Lq, seq=None,
              wgrad_stream=None, done=None):
    """ilvlm_tower_bwd.  per_block: (lib.TowerGrad * n) filled by the caller; done(i): host callback after block i's backward
    has been enqueued (an exception it raises is re-raised here, after the call has returned)."""
    n, rows, E = xs.shape
    _chk(dtop_f32, "tower.dtop", torch.float32, (rows, E)); _chk(d_f32, "tower.d", torch.float32, (n, rows, E))
    _chk(scratch, "tower.scratch", torch.uint8); _chk(saved, "tower.saved", torch.uint8)
    if len(descs) != n or scratch.numel() < n * scratch_stride or saved.numel() < n * saved_stride:
        raise RuntimeError("tower_bwd: descriptor count / buffer sizes do not match %d blocks" % n)
    arr = (L.Block * n)(*descs)
    failure = []

    def _cb(i, _user):
        if failure:
            return
        try:
            done(i)
        except BaseException as exc:      # a ctypes callback cannot propagate: keep it and re-raise below
            failure.append(exc)
    cb = L.BLOCK_DONE_FN(_cb) if done is not None else L.BLOCK_DONE_FN()
    rc = L.load().ilvlm_tower_bwd(arr, n, per_block, x0.data_ptr(), xs.data_ptr(), saved.data_ptr(), saved_stride,
                                  dtop_f32.data_ptr(), _p(dtop_lp), d_f32.data_ptr(), scratch.data_ptr(), scratch_stride, rows, B, Lq,
                                  seq.cap if seq is not None else Lq, seq.offs.data_ptr() if seq is not None else None, _WGRAD_TARGET,
                                  _stream(), None if wgrad_stream is None else wgrad_stream.cuda_stream, cb, None)
    if failure:
        raise failure[0]
    L.check(rc, "tower_bwd")


class PackedSeq(object):
    """Row layout of a packed text batch (include/ilvlm_hip.h, "packed text rows"): sequence b owns rows
    [offs[b], offs[b+1]).  `lengths` is a host-side int sequence; the offsets live on the device as int32."""

    def __init__(self, lengths, ctx, device):
        lens = [int(v) for v in lengths]
        if not lens or min(lens) < 1 or max(lens) > ctx:
            raise RuntimeError("packed text rows: lengths must be in [1, %d]" % ctx)
        offs = [0]
        for v in lens:
            offs.append(offs[-1] + v)
        self.B, self.ctx, self.rows, self.cap = len(lens), int(ctx), offs[-1], max(lens)
        self.lengths = lens
        self.offs = torch.tensor(offs, dtype=torch.int32).to(device, non_blocking=True)
        self._row_seq = None
        self._device = device

    @property
    def row_seq(self):
        """int32 [rows]: the sequence every packed row belongs to (host metadata like `offs`; built on first use)"""
        if self._row_seq is None:
            lens = torch.tensor(self.lengths, dtype=torch.int64)
            self._row_seq = torch.repeat_interleave(torch.arange(self.B, dtype=torch.int32), lens).to(self._device, non_blocking=True)
        return self._row_seq


def attention_fwd(qkv, out, lse, B, Lq, H, causal, seq=None):
    rows = seq.rows if seq is not None else B * Lq
    _chk(qkv, "attn.qkv", None, (rows, 3 * 64 * H)); _chk(out, "attn.out", qkv.dtype, (rows, 64 * H))
    _chk(lse, "attn.lse", torch.float32, (B, H, Lq))
    if seq is not None:
        L.check(L.load().ilvlm_attention_packed_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), dt(qkv), B, Lq, seq.cap, H,
                                                    int(causal), seq.offs.data_ptr(), _stream()), "attention_packed_fwd")
        return
    L.check(L.load().ilvlm_attention_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), dt(qkv), B, Lq, H, int(causal),
                                         _stream()), "attention_fwd")


def attention_bwd(dout, qkv, out, lse, dqkv, B, Lq, H, causal, seq=None):
    rows = seq.rows if seq is not None else B * Lq
    _chk(dout, "attn.dout", qkv.dtype, (rows, 64 * H)); _chk(dqkv, "attn.dqkv", qkv.dtype, (rows, 3 * 64 * H))
    if seq is not None:
        L.check(L.load().ilvlm_attention_packed_bwd(dout.data_ptr(), qkv.data_ptr(), out.data_ptr(), lse.data_ptr(),
                                                    dqkv.data_ptr(), dt(qkv), B, Lq, seq.cap, H, int(causal),
                                                    seq.offs.data_ptr(), _stream()), "attention_packed_bwd")
        return
    L.check(L.load().ilvlm_attention_bwd(dout.data_ptr(), qkv.data_ptr(), out.data_ptr(), lse.data_ptr(),
                                         dqkv.data_ptr(), dt(qkv), B, Lq, H, int(causal), _stream()), "attention_bwd")


def embed_fwd(tokens, table, pos, x, seq=None):
    B, Lq = tokens.shape
    rows = seq.rows if seq is not None else B * Lq
    _chk(tokens, "embed.tokens", torch.int64); _chk(x, "embed.x", torch.float32, (rows, table.shape[1]))
    _chk(pos, "embed.pos", torch.float32, (Lq, table.shape[1]))
    if seq is not None:
        L.check(L.load().ilvlm_embed_packed_fwd(tokens.data_ptr(), seq.offs.data_ptr(), table.data_ptr(), pos.data_ptr(),
                                                x.data_ptr(), B, Lq, table.shape[1], table.shape[0], _stream()),
                "embed_packed_fwd")
        return
    L.check(L.load().ilvlm_embed_fwd(tokens.data_ptr(), table.data_ptr(), pos.data_ptr(), x.data_ptr(), B, Lq,
                                     table.shape[1], table.shape[0], _stream()), "embed_fwd")


def embed_bwd(tokens, dx, dtable, dpos, seq=None):
    B, Lq = tokens.shape
    rows = seq.rows if seq is not None else B * Lq
    _chk(dx, "embed.dx", torch.float32, (rows, dtable.shape[1]))
    if seq is not None:
        L.check(L.load().ilvlm_embed_packed_bwd(tokens.data_ptr(), seq.offs.data_ptr(), dx.data_ptr(), dtable.data_ptr(),
                                                _p(dpos), B, Lq, dtable.shape[1], dtable.shape[0], _stream()),
                "embed_packed_bwd")
        return
    L.check(L.load().ilvlm_embed_bwd(tokens.data_ptr(), dx.data_ptr(), dtable.data_ptr(), _p(dpos), B, Lq,
                                     dtable.shape[1], dtable.shape[0], _stream()), "embed_bwd")


def patchify(images, patches, ps):
    B, Cc, res, res2 = images.shape
    g = res // ps
    _chk(images, "patchify.images", torch.float32); _chk(patches, "patchify.out")
    if res != res2 or res % ps:
        raise RuntimeError("patchify: bad image size")
    if patches.shape[0] != B * g * g or patches.shape[1] < Cc * ps * ps:
        raise RuntimeError("patchify: output shape %s too small" % (tuple(patches.shape),))
    L.check(L.load().ilvlm_patchify(images.data_ptr(), patches.data_ptr(), dt(patches), B, Cc, res, ps, patches.shape[1],
                                    _stream()), "patchify")


def cls_rows(cls, pos, tokens, B, Lq, W):
    _chk(tokens, "cls_rows.tokens", torch.float32, (B * Lq, W))
    L.check(L.load().ilvlm_cls_rows(cls.data_ptr(), pos.data_ptr(), tokens.data_ptr(), B, Lq, W, _stream()), "cls_rows")


def batch_sum(x, out, out0, B, Lq, W):
    _chk(x, "batch_sum.x", torch.float32, (B * Lq, W)); _chk(out, "batch_sum.out", torch.float32, (Lq, W))
    L.check(L.load().ilvlm_batch_sum(x.data_ptr(), out.data_ptr(), _p(out0), B, Lq, W, _stream()), "batch_sum")


def gather_rows(x, idx, y, B, Lq, W, seq=None):
    rows = seq.rows if seq is not None else B * Lq
    _chk(x, "gather.x", torch.float32, (rows, W)); _chk(y, "gather.y", torch.float32, (B, W))
    _chk(idx, "gather.idx", torch.int64, (B,))
    if seq is not None:
        L.check(L.load().ilvlm_gather_packed_rows(x.data_ptr(), idx.data_ptr(), seq.offs.data_ptr(), y.data_ptr(), B, W,
                                                  _stream()), "gather_packed_rows")
        return
    L.check(L.load().ilvlm_gather_rows(x.data_ptr(), idx.data_ptr(), y.data_ptr(), B, Lq, W, _stream()), "gather_rows")


def scatter_rows(dy, idx, dx, B, Lq, W, seq=None):
    rows = seq.rows if seq is not None else B * Lq
    _chk(dx, "scatter.dx", torch.float32, (rows, W)); _chk(dy, "scatter.dy", torch.float32, (B, W))
    if seq is not None:
        L.check(L.load().ilvlm_scatter_packed_rows(dy.data_ptr(), idx.data_ptr(), seq.offs.data_ptr(), dx.data_ptr(), B, W,
                                                   _stream()), "scatter_packed_rows")
        return
    L.check(L.load().ilvlm_scatter_rows(dy.data_ptr(), idx.data_ptr(), dx.data_ptr(), B, Lq, W, _stream()),
            "scatter_rows")


def fdt_pool_fwd(scores, mask, pooled, argmax, B, T, Cn, sqrt_d, temp, pool, seq=None):
    rows = seq.rows if seq is not None else B * T
    _chk(scores, "fdt.scores", torch.float32, (rows, Cn)); _chk(pooled, "fdt.pooled", torch.float32, (B, Cn))
    if mask is not None:
        _chk(mask, "fdt.mask", torch.float32, (B, T))
    if argmax is not None:
        _chk(argmax, "fdt.argmax", torch.int32, (B, Cn))
    if seq is not None:
        L.check(L.load().ilvlm_fdt_pool_packed_fwd(scores.data_ptr(), seq.offs.data_ptr(), pooled.data_ptr(), _p(argmax), B, T,
                                                   Cn, sqrt_d, temp, pool, _stream()), "fdt_pool_packed_fwd")
        return
    L.check(L.load().ilvlm_fdt_pool_fwd(scores.data_ptr(), _p(mask), pooled.data_ptr(), _p(argmax), B, T, Cn, sqrt_d,
                                        temp, pool, _stream()), "fdt_pool_fwd")


def fdt_score_pool_fwd(q, sd, pooled, argmax, B, T, sqrt_d, temp, seq=None):
    """fused codebook scores + scale + token max-pool (ilvlm_fdt_score_pool_fwd): q [rows,d] bf16, sd [C,d] bf16"""
    rows, d = q.shape
    Cn = sd.shape[0]
    _chk(q, "fdt.q", torch.bfloat16); _chk(sd, "fdt.sd", torch.bfloat16, (Cn, d))
    _chk(pooled, "fdt.pooled", torch.float32, (B, Cn)); _chk(argmax, "fdt.argmax", torch.int32, (B, Cn))
    if rows != (seq.rows if seq is not None else B * T):
        raise RuntimeError("fdt_score_pool_fwd: q has %d rows, layout says %d" % (rows, seq.rows if seq is not None else B * T))
    ws = torch.empty((B, Cn), dtype=torch.int64, device=q.device)
    L.check(L.load().ilvlm_fdt_score_pool_fwd(q.data_ptr(), sd.data_ptr(), ws.data_ptr(), pooled.data_ptr(), argmax.data_ptr(),
                                              rows, B, T, Cn, d, sqrt_d, temp, seq.offs.data_ptr() if seq is not None else None,
                                              seq.row_seq.data_ptr() if seq is not None else None, _stream()),
            "fdt_score_pool_fwd")


def fdt_pool_bwd(dpooled, argmax, mask, dscores, B, T, Cn, sqrt_d, temp, pool, seq=None):
    rows = seq.rows if seq is not None else B * T
    _chk(dpooled, "fdt.dpooled", torch.float32, (B, Cn)); _chk(dscores, "fdt.dscores", None, (rows, Cn))
    if seq is not None:
        L.check(L.load().ilvlm_fdt_pool_packed_bwd(dpooled.data_ptr(), _p(argmax), seq.offs.data_ptr(), dscores.data_ptr(),
                                                   dt(dscores), B, T, Cn, sqrt_d, temp, pool, _stream()), "fdt_pool_packed_bwd")
        return
    L.check(L.load().ilvlm_fdt_pool_bwd(dpooled.data_ptr(), _p(argmax), _p(mask), dscores.data_ptr(), dt(dscores), B, T,
                                        Cn, sqrt_d, temp, pool, _stream()), "fdt_pool_bwd")


def _rows2(name, fn, *ts):
    rows, cols = ts[0].shape
    for t in ts:
        _chk(t, name, torch.float32, (rows, cols))
    L.check(fn(*[t.data_ptr() for t in ts], rows, cols, _stream()), name)


def sparsemax_fwd(z, out):
    _rows2("sparsemax_fwd", L.load().ilvlm_sparsemax_fwd, z, out)


def sparsemax_bwd(out, g, dz):
    _rows2("sparsemax_bwd", L.load().ilvlm_sparsemax_bwd, out, g, dz)


def softmax_fwd(z, out):
    _rows2("softmax_fwd", L.load().ilvlm_softmax_fwd, z, out)


def softmax_bwd(out, g, dz):
    _rows2("softmax_bwd", L.load().ilvlm_softmax_bwd, out, g, dz)


def sigmoid_norm_fwd(z, w, wn, rowsum):
    """w = sigmoid(z), wn = w / rowsum(w) (att_func_type 'sigmoid', clip_fdt.py:76,156-157)"""
    rows, cols = z.shape
    for t, nm in ((z, "z"), (w, "w"), (wn, "wn")):
        _chk(t, "sigmoid_norm_fwd." + nm, torch.float32, (rows, cols))
    _chk(rowsum, "sigmoid_norm_fwd.rowsum", torch.float32, (rows,))
    L.check(L.load().ilvlm_sigmoid_norm_fwd(z.data_ptr(), w.data_ptr(), wn.data_ptr(), rowsum.data_ptr(), rows, cols, _stream()),
            "sigmoid_norm_fwd")


def sigmoid_norm_bwd(w, wn, rowsum, g, dz):
    rows, cols = w.shape
    for t, nm in ((w, "w"), (wn, "wn"), (g, "g"), (dz, "dz")):
        _chk(t, "sigmoid_norm_bwd." + nm, torch.float32, (rows, cols))
    _chk(rowsum, "sigmoid_norm_bwd.rowsum", torch.float32, (rows,))
    L.check(L.load().ilvlm_sigmoid_norm_bwd(w.data_ptr(), wn.data_ptr(), rowsum.data_ptr(), g.data_ptr(), dz.data_ptr(), rows, cols,
                                            _stream()), "sigmoid_norm_bwd")


def l2norm_fwd(x, y, norm, eps):
    rows, cols = x.shape
    _chk(x, "l2norm.x", torch.float32); _chk(y, "l2norm.y", torch.float32, x.shape); _chk(norm, "l2norm.norm", torch.float32, (rows,))
    L.check(L.load().ilvlm_l2norm_fwd(x.data_ptr(), y.data_ptr(), norm.data_ptr(), rows, cols, eps, _stream()), "l2norm_fwd")


def l2norm_bwd(x, norm, dy, dx, eps):
    rows, cols = x.shape
    _chk(dy, "l2norm.dy", torch.float32, x.shape); _chk(dx, "l2norm.dx", torch.float32, x.shape)
    L.check(L.load().ilvlm_l2norm_bwd(x.data_ptr(), norm.data_ptr(), dy.data_ptr(), dx.data_ptr(), rows, cols, eps,
                                      _stream()), "l2norm_bwd")


def logit_scale_fwd(param, out, max_scale=100.0):
    L.check(L.load().ilvlm_logit_scale_fwd(param.data_ptr(), out.data_ptr(), max_scale, _stream()), "logit_scale_fwd")


def logit_scale_bwd(dli, li, dlt, lt, param, scale_used, dparam):
    for t in (dli, li, dlt, lt):
        _chk(t, "logit_scale_bwd", torch.float32, li.shape)
    L.check(L.load().ilvlm_logit_scale_bwd(dli.data_ptr(), li.data_ptr(), dlt.data_ptr(), lt.data_ptr(), li.numel(),
                                           param.data_ptr(), scale_used.data_ptr(), dparam.data_ptr(), _stream()),
            "logit_scale_bwd")


def infonce_fwd(li, lt, label_offset, loss, dli, dlt):
    B, Bg = li.shape
    for t in (li, lt, dli, dlt):
        _chk(t, "infonce", torch.float32, (B, Bg))
    L.check(L.load().ilvlm_infonce_fwd(li.data_ptr(), lt.data_ptr(), B, Bg, label_offset, loss.data_ptr(),
                                       dli.data_ptr(), dlt.data_ptr(), _stream()), "infonce_fwd")


def topk_accuracy(logits, label_offset, k, out):
    B, Bg = logits.shape
    _chk(logits, "topk.logits", torch.float32); _chk(out, "topk.out", torch.float32, (2,))
    L.check(L.load().ilvlm_topk_accuracy(logits.data_ptr(), B, Bg, label_offset, k, out.data_ptr(), _stream()), "topk")


def colsum(x, out):
    rows, cols = x.shape
    _chk(out, "colsum.out", torch.float32, (cols,))
    if not x.is_cuda or x.stride(1) != 1:
        raise RuntimeError("colsum: bad input")
    L.check(L.load().ilvlm_colsum(x.data_ptr(), dt(x), out.data_ptr(), rows, cols, x.stride(0), _stream()), "colsum")


def cast_f32(src, dst):
    _chk(src, "cast.src", torch.float32); _chk(dst, "cast.dst")
    if src.numel() != dst.numel():
        raise RuntimeError("cast: size mismatch")
    L.check(L.load().ilvlm_cast_f32(src.data_ptr(), dst.data_ptr(), dt(dst), src.numel(), _stream()), "cast_f32")


IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)     # imagenet_dataloader.py:13-14


def image_u8_normalize(src, dst=None, flags=None, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """uint8 image batch [B,H,W,3] or [B,3,H,W] (device) -> fp32 [B,3,H,W], ToTensor + Normalize, optional per-sample
    flags (uint8 [B]: bit 0 horizontal flip, bit 1 grayscale)."""
    _chk(src, "image.src", torch.uint8)
    if src.dim() != 4 or (src.shape[3] != 3 and src.shape[1] != 3):
        raise RuntimeError("image_u8_normalize: expected [B,H,W,3] or [B,3,H,W] uint8, got %s" % (tuple(src.shape),))
    if src.shape[3] == 3 and src.shape[1] == 3:
        raise RuntimeError("image_u8_normalize: ambiguous layout (3 rows or 3 columns); pass a real image size")
    nhwc = src.shape[3] == 3
    B = src.shape[0]
    H, W = (src.shape[1], src.shape[2]) if nhwc else (src.shape[2], src.shape[3])
    if dst is None:
        dst = torch.empty((B, 3, H, W), dtype=torch.float32, device=src.device)
    _chk(dst, "image.dst", torch.float32, (B, 3, H, W))
    if flags is not None:
        _chk(flags, "image.flags", torch.uint8, (B,))
    m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    L.check(L.load().ilvlm_image_u8_normalize(src.data_ptr(), int(nhwc), _p(flags), dst.data_ptr(), B, H, W, m3, s3, _stream()),
            "image_u8_normalize")
    return dst


def mocov2_params(sizes, rng, out_size=224, scale=(0.2, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), jitter=(0.4, 0.4, 0.4, 0.1),
                  p_jitter=0.8, p_gray=0.2, p_blur=0.5, sigma=(0.1, 2.0), p_flip=0.5):
    """The random draws of MOCOV2_single (reference prototype/data/imagenet_dataloader.py:59-68) for a batch of decoded images
    of the given (height, width) sizes, made on the host with `rng` (a random.Random): RandomResizedCrop.get_params (ten
    attempts at an area / aspect draw, then the centre-crop fallback), RandomApply(ColorJitter) with ColorJitter.get_params
    (a permutation of the four operations and U(1 - x, 1 + x) / U(-hue, hue) factors), RandomGrayscale, RandomApply(
    GaussianBlur) with sigma ~ U(0.1, 2), RandomHorizontalFlip.  Returns a ctypes array of lib.AugmentParams."""
    import math
    arr = (L.AugmentParams * len(sizes))()
    for a, (H, W) in zip(arr, sizes):
        area = H * W
        for _ in range(10):
            target = area * rng.uniform(scale[0], scale[1])
            ar = math.exp(rng.uniform(math.log(ratio[0]), math.log(ratio[1])))
            w, h = int(round(math.sqrt(target * ar))), int(round(math.sqrt(target / ar)))
            if 0 < w <= W and 0 < h <= H:
                a.crop_top, a.crop_left, a.crop_h, a.crop_w = rng.randint(0, H - h), rng.randint(0, W - w), h, w
                break
        else:                                   # fallback: central crop at the nearest admissible aspect ratio
            in_ratio = W / H
            if in_ratio < ratio[0]:
                w, h = W, int(round(W / ratio[0]))
            elif in_ratio > ratio[1]:
                h, w = H, int(round(H * ratio[1]))
            else:
                w, h = W, H
            a.crop_top, a.crop_left, a.crop_h, a.crop_w = (H - h) // 2, (W - w) // 2, h, w
        a.jitter = int(rng.random() < p_jitter)
        order = [0, 1, 2, 3]
        rng.shuffle(order)
        a.jitter_order = order[0] | (order[1] << 2) | (order[2] << 4) | (order[3] << 6)
        a.brightness = rng.uniform(max(0.0, 1 - jitter[0]), 1 + jitter[0])
        a.contrast = rng.uniform(max(0.0, 1 - jitter[1]), 1 + jitter[1])
        a.saturation = rng.uniform(max(0.0, 1 - jitter[2]), 1 + jitter[2])
        a.hue = rng.uniform(-jitter[3], jitter[3])
        a.grayscale = int(rng.random() < p_gray)
        a.blur_sigma = rng.uniform(sigma[0], sigma[1]) if rng.random() < p_blur else 0.0
        a.flip = int(rng.random() < p_flip)
    return arr


def image_augment(src, offsets, hw, params, out_size=224, dst=None, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """MOCOV2_single on the device (ilvlm_image_augment): src = the decoded uint8 images back to back ([H][W][3] each) on the
    device, offsets int64 [B] (byte offset of each), hw int32 [B,2]; params: lib.AugmentParams array (mocov2_params) or a uint8
    device tensor holding it.  Returns fp32 [B,3,out_size,out_size]."""
    _chk(src, "augment.src", torch.uint8); _chk(offsets, "augment.offsets", torch.int64); _chk(hw, "augment.hw", torch.int32)
    B = offsets.numel()
    if tuple(hw.shape) != (B, 2):
        raise RuntimeError("image_augment: hw must be int32 [B,2]")
    if not torch.is_tensor(params):
        if len(params) != B:
            raise RuntimeError("image_augment: %d parameter records for %d images" % (len(params), B))
        max_rows = max(p.crop_h for p in params)
        params = torch.frombuffer(bytearray(bytes(params)), dtype=torch.uint8).to(src.device)
    else:
        _chk(params, "augment.params", torch.uint8, (B * C.sizeof(L.AugmentParams),))
        max_rows = int(hw[:, 0].max())
    if dst is None:
        dst = torch.empty((B, 3, out_size, out_size), dtype=torch.float32, device=src.device)
    _chk(dst, "augment.dst", torch.float32, (B, 3, out_size, out_size))
    n = L.load().ilvlm_image_augment_scratch_floats(B, out_size, max_rows)
    scratch = torch.empty(n, dtype=torch.float32, device=src.device)
    m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    L.check(L.load().ilvlm_image_augment(src.data_ptr(), offsets.data_ptr(), hw.data_ptr(), params.data_ptr(), dst.data_ptr(),
                                         scratch.data_ptr(), B, out_size, max_rows, m3, s3, _stream()), "image_augment")
    return dst


def cast_to_f32(src, dst):
    """dst (fp32) = src (bf16)"""
    _chk(src, "cast_to_f32.src", torch.bfloat16); _chk(dst, "cast_to_f32.dst", torch.float32)
    if src.numel() != dst.numel():
        raise RuntimeError("cast_to_f32: size mismatch")
    L.check(L.load().ilvlm_cast_to_f32(src.data_ptr(), dt(src), dst.data_ptr(), src.numel(), _stream()), "cast_to_f32")


def scale(x, y, a):
    L.check(L.load().ilvlm_scale(x.data_ptr(), y.data_ptr(), float(a), x.numel(), _stream()), "scale")


def scale_dev(x, a_dev, y=None):
    """y = a_dev[0] * x with a device scalar (fp32)."""
    _chk(x, "scale_dev.x", torch.float32)
    y = torch.empty_like(x) if y is None else y
    a = a_dev.reshape(-1)
    if a.dtype != torch.float32 or not a.is_cuda or a.numel() != 1:
        raise RuntimeError("scale_dev: scalar must be a 1-element fp32 device tensor")
    L.check(L.load().ilvlm_scale_dev(x.data_ptr(), y.data_ptr(), a.data_ptr(), x.numel(), _stream()), "scale_dev")
    return y


def add_inplace(y, x):
    _chk(y, "add_inplace.y", torch.float32); _chk(x, "add_inplace.x", torch.float32, y.shape)
    L.check(L.load().ilvlm_add_inplace(y.data_ptr(), x.data_ptr(), y.numel(), _stream()), "add_inplace")
    return y


_SUMSQ_WS = {}


def clip_grad_norm_(flat, max_norm, scratch=None):
    """clip_grad_norm_ (2-norm) over a flat fp32 gradient buffer, entirely on the device; returns the 1-element tensor holding
    the SQUARED total norm (read it only when logging)"""
    _chk(flat, "clip_grad_norm.flat", torch.float32)
    ss = scratch if scratch is not None else torch.empty(1, dtype=torch.float32, device=flat.device)
    ss.zero_()
    # per-workgroup partials of the (bitwise reproducible) two-stage sum; one small buffer per device, reused on the stream
    key = flat.device.index
    part = _SUMSQ_WS.get(key)
    if part is None:
        part = _SUMSQ_WS[key] = torch.empty(L.load().ilvlm_sumsq_partials(), dtype=torch.float32, device=flat.device)
    L.check(L.load().ilvlm_sumsq(flat.data_ptr(), flat.numel(), ss.data_ptr(), part.data_ptr(), _stream()), "sumsq")
    L.check(L.load().ilvlm_clip_by_norm(flat.data_ptr(), flat.numel(), ss.data_ptr(), float(max_norm), _stream()), "clip_by_norm")
    return ss


def clamp_(x, lo, hi):
    _chk(x, "clamp.x", torch.float32)
    L.check(L.load().ilvlm_clamp(x.data_ptr(), float(lo), float(hi), x.numel(), _stream()), "clamp")
    return x

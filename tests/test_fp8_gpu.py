"""fp8 path (BASELINE configs[4]): quantisation kernels against torch's OCP float8 conversions, the fp8 MFMA GEMM
against the de-quantised fp32 product (exact products, fp32 sums), delayed scaling bookkeeping."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import clip_oracle as O  # noqa: E402

F8 = {False: (torch.float8_e4m3fn, 448.0), True: (torch.float8_e5m2, 57344.0)}


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def to_f8(x, scale, e5m2):
    dt, mx = F8[e5m2]
    return (x.float() * scale).clamp(-mx, mx).to(dt)


@pytest.mark.parametrize("e5m2", [False, True])
@pytest.mark.parametrize("src_dtype", [torch.bfloat16, torch.float32])
def test_quantize_matches_torch_float8(e5m2, src_dtype):
    from ilvlm_amd import ops
    x = (rnd(1000, 64, seed=1) * 3).to(src_dtype)
    x[0, :8] = torch.tensor([0.0, -0.0, 1e-9, 500.0, -1e6, 448.0, 0.0625, -3.5]).to(src_dtype)
    for s in (1.0, 37.5, 0.01):
        scale = torch.tensor([s], device="cuda")
        amax = torch.zeros(1, device="cuda")
        dst = torch.empty(x.shape, dtype=torch.uint8, device="cuda")
        ops.fp8_quantize(x.cuda(), dst, scale, amax, e5m2)
        want = to_f8(x, s, e5m2).view(torch.uint8)
        got = dst.cpu()
        # -0.0 vs +0.0 after saturation / flush may differ in the sign bit only
        diff = got != want
        assert not bool((diff & ((got & 0x7f) != 0)).any()), "fp8 codes differ in %d places" % int(diff.sum())
        assert float(amax) == float(x.float().abs().max())
    amax = torch.zeros(1, device="cuda")
    ops.fp8_quantize(x.cuda(), None, None, amax)            # observe only
    assert float(amax) == float(x.float().abs().max())


@pytest.mark.parametrize("a_e5m2", [False, True])
@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (300, 200, 256), (1000, 768, 768), (64, 512, 3072)])
def test_fp8_gemm_matches_dequantised_product(M, N, K, a_e5m2):
    from ilvlm_amd import ops
    a, w = rnd(M, K, seed=1), rnd(N, K, seed=2) * 0.05
    if a_e5m2:
        a = a * 1e-4                                   # gradient-like magnitudes
    sa = F8[a_e5m2][1] / float(a.abs().max())
    sw = 448.0 / float(w.abs().max())
    a8, w8 = to_f8(a, sa, a_e5m2), to_f8(w, sw, False)
    ref = (a8.float() @ w8.float().t()) / (sa * sw)
    inv_a, inv_w = torch.tensor([1.0 / sa], device="cuda"), torch.tensor([1.0 / sw], device="cuda")
    A, W = a8.view(torch.uint8).cuda(), w8.view(torch.uint8).cuda()
    out = torch.full((M, N), float("nan"), device="cuda")
    ops.gemm_fp8(A, W, out, inv_a, inv_w, a_e5m2=a_e5m2)
    assert float((out.cpu() - ref).abs().max()) < 1e-4 * float(ref.abs().max())     # exact products, fp32 sums, two scale roundings
    # and it is a faithful GEMM of the unquantised operands to fp8 accuracy
    full = a @ w.t()
    assert float((out.cpu() - full).abs().max()) < (0.08 if not a_e5m2 else 0.15) * float(full.abs().max())
    # epilogues: bias + QuickGELU with saved pre-activation (bf16), bias + residual (fp32 out)
    bias, res = rnd(N, seed=3).cuda(), rnd(M, N, seed=4).cuda()
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); aux = torch.empty_like(y)
    ops.gemm_fp8(A, W, y, inv_a, inv_w, a_e5m2=a_e5m2, bias=bias, aux=aux, act=1)
    pre = ref + bias.cpu()
    assert float((aux.float().cpu() - pre).abs().max()) < 1e-2 * float(pre.abs().max())
    assert float((y.float().cpu() - O.quick_gelu(pre)).abs().max()) < 1e-2 * float(pre.abs().max())
    z = torch.empty(M, N, device="cuda")
    ops.gemm_fp8(A, W, z, inv_a, inv_w, a_e5m2=a_e5m2, bias=bias, residual=res)
    assert float((z.cpu() - (pre + res.cpu())).abs().max()) < 1e-4 * float((pre + res.cpu()).abs().max())


def test_weight_quantisation_and_delayed_scaling():
    """the batched weight kernel: e4m3 copy and its transpose at the tensors' arena offsets, per-tensor amax; the scale
    update: history ring, scale = fmt_max / max(history), amax reset"""
    from ilvlm_amd import lib as L
    shapes = [(128, 192), (64, 64), (256, 128)]
    offs, o = [], 0
    for r, c in shapes:
        offs.append(o)
        o += (r * c + 63) // 64 * 64 + 64
    P = (rnd(o, seed=5) * 0.3).cuda()
    table = []
    for slot, ((r, c), off) in enumerate(zip(shapes, offs)):
        for r0 in range(0, r, 64):
            for c0 in range(0, c, 64):
                table.append([off // 64, r, c, slot, r0, c0])
    tab = torch.tensor(table, dtype=torch.int32).cuda()
    scale = torch.tensor([100.0, 50.0, 200.0], device="cuda")
    amax = torch.zeros(3, device="cuda")
    W8 = torch.zeros(o, dtype=torch.uint8, device="cuda"); W8T = torch.zeros_like(W8)
    h = L.load()
    L.check(h.ilvlm_fp8_quantize_weights(P.data_ptr(), W8.data_ptr(), W8T.data_ptr(), tab.data_ptr(), len(table), scale.data_ptr(),
                                         amax.data_ptr(), torch.cuda.current_stream().cuda_stream), "quantize_weights")
    for slot, ((r, c), off) in enumerate(zip(shapes, offs)):
        w = P[off:off + r * c].view(r, c).cpu()
        want = to_f8(w, float(scale[slot]), False).view(torch.uint8)
        got = W8[off:off + r * c].view(r, c).cpu()
        gotT = W8T[off:off + r * c].view(c, r).cpu()
        same = (got == want) | ((got & 0x7f) == 0) & ((want & 0x7f) == 0)
        assert bool(same.all()) and torch.equal(gotT, got.t().contiguous())
        assert float(amax[slot]) == float(w.abs().max())
    hist = torch.zeros(3, 4, device="cuda"); inv = torch.zeros(3, device="cuda")
    fmt_max = torch.tensor([448.0, 448.0, 57344.0], device="cuda")
    first = amax.clone()
    for pos, bump in enumerate((1.0, 0.5, 3.0, 0.25, 0.25, 0.25, 0.25)):
        amax.copy_(first * bump)
        L.check(h.ilvlm_fp8_scale_update(amax.data_ptr(), hist.data_ptr(), scale.data_ptr(), inv.data_ptr(), fmt_max.data_ptr(), 3, 4,
                                         pos % 4, torch.cuda.current_stream().cuda_stream), "scale_update")
        assert float(amax.abs().max()) == 0.0
        window = [1.0, 0.5, 3.0, 0.25, 0.25, 0.25, 0.25][max(0, pos - 3):pos + 1]
        want = fmt_max.cpu() / (first.cpu() * max(window))
        assert torch.allclose(scale.cpu(), want, rtol=1e-6) and torch.allclose(inv.cpu(), 1 / want, rtol=1e-6)

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


def _pack_b8_host(w8):
    """the layout include/ilvlm_hip.h documents for ilvlm_gemm_pack_b8, restated with numpy indexing"""
    n, k = w8.shape
    v = w8.numpy().reshape(n // 16, 16, k // 128, 2, 4, 16)       # [n16, row, k128, half, g, byte]
    return torch.from_numpy(np.ascontiguousarray(v.transpose(0, 2, 3, 4, 1, 5))).reshape(-1)   # [n16, k128, half, g, row, byte]


@pytest.mark.parametrize("a_e5m2", [False, True])
@pytest.mark.parametrize("M,N,K", [(128, 256, 128), (300, 208, 256), (1000, 768, 768), (77, 512, 3072), (12800, 2304, 768),
                                   (11319, 512, 2048), (1, 16, 128)])
def test_fp8_streaming_kernel_equals_the_direct_to_lds_kernel(M, N, K, a_e5m2):
    """b_packed with fp8 operands: the streaming kernel (A through the LDS ring, fragment-order B straight into the operand
    registers of the scaled MFMA) performs the same MFMA sequence per output element as the direct-to-LDS kernel on the
    row-major operand -- bit-identical outputs, every epilogue, ragged M, N not a multiple of the tile width; the packed
    layout equals the one the header documents"""
    from ilvlm_amd import ops
    a, w = rnd(M, K, seed=1), rnd(N, K, seed=2) * 0.05
    if a_e5m2:
        a = a * 1e-4
    sa, sw = F8[a_e5m2][1] / float(a.abs().max()), 448.0 / float(w.abs().max())
    a8, w8 = to_f8(a, sa, a_e5m2).view(torch.uint8), to_f8(w, sw, False).view(torch.uint8)
    inv_a, inv_w = torch.tensor([1.0 / sa], device="cuda"), torch.tensor([1.0 / sw], device="cuda")
    # A sits in front of poison rows: the streaming kernel must not read past row M
    Abuf = torch.full((M + 130, K), 0x7b, dtype=torch.uint8, device="cuda"); Abuf[:M] = a8.cuda()
    A, W = Abuf[:M], w8.cuda()
    Wp = ops.gemm_pack_b8(W)
    assert torch.equal(Wp.cpu(), _pack_b8_host(w8))
    bias, res = rnd(N, seed=3).cuda(), rnd(M, N, seed=4).cuda()
    for kw, dt_ in ((dict(), torch.float32), (dict(), torch.bfloat16), (dict(bias=bias, act=1), torch.bfloat16),
                    (dict(bias=bias, residual=res), torch.float32)):
        outs = []
        for bp in (None, Wp):
            y = torch.full((M, N), float("nan"), device="cuda").to(dt_)
            k2 = dict(kw)
            if "act" in k2:
                k2["aux"] = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
            ops.gemm_fp8(A, W, y, inv_a, inv_w, a_e5m2=a_e5m2, b_packed=bp, **k2)
            outs.append((y, k2.get("aux")))
        assert torch.equal(outs[0][0], outs[1][0]) and not bool(torch.isnan(outs[1][0].float()).any())
        if outs[0][1] is not None:
            assert torch.equal(outs[0][1], outs[1][1])
    ref = (a8.view(F8[a_e5m2][0]).float() @ w8.view(torch.float8_e4m3fn).float().t()) / (sa * sw)
    assert float((outs[0][0].cpu() - bias.cpu() - res.cpu() - ref).abs().max()) < 1e-4 * float(ref.abs().max()) + 1e-5


def test_packed_weight_quantiser_writes_the_fragment_order_copies():
    """ilvlm_fp8_quantize_weights_packed: W8 / W8T as the plain call, W8P / W8TP = ilvlm_gemm_pack_b8 of them"""
    from ilvlm_amd import lib as L, ops
    shapes = [(128, 384), (256, 128), (384, 256)]
    offs, o = [], 0
    for r, c in shapes:
        offs.append(o)
        o += r * c + 64
    P = (rnd(o, seed=5) * 0.3).cuda()
    table = [[off // 64, r, c, slot, r0, c0] for slot, ((r, c), off) in enumerate(zip(shapes, offs))
             for r0 in range(0, r, 64) for c0 in range(0, c, 64)]
    tab = torch.tensor(table, dtype=torch.int32).cuda()
    scale = torch.tensor([100.0, 50.0, 200.0], device="cuda")
    bufs = [torch.zeros(o, dtype=torch.uint8, device="cuda") for _ in range(6)]
    am1, am2 = torch.zeros(3, device="cuda"), torch.zeros(3, device="cuda")
    h, st = L.load(), torch.cuda.current_stream().cuda_stream
    L.check(h.ilvlm_fp8_quantize_weights(P.data_ptr(), bufs[0].data_ptr(), bufs[1].data_ptr(), tab.data_ptr(), len(table),
                                         scale.data_ptr(), am1.data_ptr(), st), "quantize_weights")
    L.check(h.ilvlm_fp8_quantize_weights_packed(P.data_ptr(), bufs[2].data_ptr(), bufs[3].data_ptr(), bufs[4].data_ptr(),
                                                bufs[5].data_ptr(), tab.data_ptr(), len(table), scale.data_ptr(), am2.data_ptr(), st),
            "quantize_weights_packed")
    assert torch.equal(bufs[0], bufs[2]) and torch.equal(bufs[1], bufs[3]) and torch.equal(am1, am2)
    for (r, c), off in zip(shapes, offs):
        assert torch.equal(bufs[4][off:off + r * c], ops.gemm_pack_b8(bufs[2][off:off + r * c].view(r, c)))
        assert torch.equal(bufs[5][off:off + r * c], ops.gemm_pack_b8(bufs[3][off:off + r * c].view(c, r)))


@pytest.mark.parametrize("T,M,N,split", [(128, 128, 128, 1), (256, 64, 192, 2), (1000, 768, 768, 3), (11319, 512, 2048, 8),
                                         (77, 16, 48, 1)])
def test_fp8_weight_gradient_gemm(T, M, N, split):
    """the (1,1) accumulate form: both operands K-strided fp8 (dY e5m2, X e4m3), any reduction length (rows past T are
    never read), split-K atomics into an fp32 gradient that already holds a value, bias gradient as a by-product"""
    from ilvlm_amd import ops
    dy, x = rnd(T, M, seed=1) * 1e-4, rnd(T, N, seed=2)
    sd, sx = 57344.0 / float(dy.abs().max()), 448.0 / float(x.abs().max())
    dy8, x8 = to_f8(dy, sd, True), to_f8(x, sx, False)
    ref = (dy8.float().t().double() @ x8.float().double()).float() / (sd * sx)
    inv_d, inv_x = torch.tensor([1.0 / sd], device="cuda"), torch.tensor([1.0 / sx], device="cuda")
    # operands sit inside larger buffers whose following rows are poison: the kernel must not read past row T
    D = torch.full((T + 130, M), 0x7b, dtype=torch.uint8, device="cuda"); D[:T] = dy8.view(torch.uint8).cuda()
    X = torch.full((T + 130, N), 0x7e, dtype=torch.uint8, device="cuda"); X[:T] = x8.view(torch.uint8).cuda()
    base = rnd(M, N, seed=3) * float(ref.abs().max())
    out = base.cuda()
    rs = torch.full((M,), 2.0, device="cuda")
    ops.gemm_fp8_wgrad(D, X, out, inv_d, inv_x, split_k=split, rowsum=rs, K=T)
    tol = 2e-4 * float(ref.abs().max()) + 1e-6 * float(base.abs().max())
    assert float((out.cpu() - base - ref).abs().max()) < tol
    want_rs = dy8.float().double().sum(0).float() / sd
    assert float((rs.cpu() - 2.0 - want_rs).abs().max()) < 1e-4 * float(want_rs.abs().max()) + 1e-6
    # faithful to the unquantised product at fp8 accuracy
    full = dy.t() @ x
    assert float((out.cpu() - base - full).abs().max()) < 0.15 * float(full.abs().max())


@pytest.mark.parametrize("rows,E", [(12800, 768), (11319, 512), (1000, 256), (640, 384)])
def test_fp8_grouped_weight_gradients_on_both_tiles(rows, E):
    """ilvlm_wgrad_group with fp8 operands (dY e5m2, X e4m3, both K-strided): the four products and bias gradients of a block,
    accumulated into what the slots hold, against the de-quantised fp32 reference -- on the 128 x 128 tile (non-scaled fp8 MFMA)
    and on the round-4 form: 256 x 128 tiles on the block-scaled MFMA for the output columns from 128 up, the first tile column
    (which carries the bias gradient's row sums) on the 128 x 128 kernel.  Ragged reduction lengths; E = 384: 3 E = 1152 is not a
    multiple of 256 rows, so that block falls back to the narrow tile as a whole."""
    from ilvlm_amd import ops
    dims = ((3 * E, E), (E, E), (4 * E, E), (E, 4 * E))
    data = []
    for i, (n, k) in enumerate(dims):
        dy, x = rnd(rows, n, seed=10 + i) * 1e-4, rnd(rows, k, seed=20 + i)
        sd, sx = 57344.0 / float(dy.abs().max()), 448.0 / float(x.abs().max())
        dy8, x8 = to_f8(dy, sd, True), to_f8(x, sx, False)
        ref = (dy8.float().t().double() @ x8.float().double()).float() / (sd * sx)
        rs = dy8.float().double().sum(0).float() / sd
        base = rnd(n, k, seed=30 + i) * float(ref.abs().max())
        bb = rnd(n, seed=40 + i) * float(rs.abs().max())
        data.append((dy8.view(torch.uint8).cuda(), x8.view(torch.uint8).cuda(), base, bb, ref, rs,
                     torch.tensor([1.0 / sd], device="cuda"), torch.tensor([1.0 / sx], device="cuda")))
    outs = {}
    try:
        for tile in (128, 256):
            ops.gemm_set_wgrad_tile(tile)
            prob = [(d[0], d[1], d[2].clone().cuda(), d[3].clone().cuda(), d[6], d[7]) for d in data]
            ops.wgrad_group(prob, rows, fp8=True)
            for (dy8, x8, gw, gb, _, _), d in zip(prob, data):
                base, bb, ref, rs = d[2], d[3], d[4], d[5]
                assert float((gw.cpu() - base - ref).abs().max()) < 2e-4 * float(ref.abs().max()) + 1e-6 * float(base.abs().max()), tile
                assert float((gb.cpu() - bb - rs).abs().max()) < 1e-4 * float(rs.abs().max()) + 1e-6 * float(bb.abs().max()), tile
            outs[tile] = [p[2] for p in prob]
    finally:
        ops.gemm_set_wgrad_tile(-1)
    if E % 256 != 0 and (3 * E) % 256 != 0:             # the whole block stayed on the narrow tile: same kernel, same K-slices
        pass
    for a, b in zip(outs[128], outs[256]):              # two MFMA forms of the same exact products: fp32 summation order only
        assert float((a - b).abs().max()) < 1e-4 * float(a.abs().max())


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
        window = [1.0, 0.5, 3.0, 0.25, 0.25, 0.25, 0.25][max(0, pos - 3):pos + 1]
        # the running amax restarts at 0.9 x the value just recorded (keeps the per-wave atomic maxima rare), not at 0 -- and
        # not at 0.9 x the window maximum, which made an outlier decay by 10 % per WINDOW instead of per step
        assert torch.allclose(amax.cpu(), 0.9 * first.cpu() * bump, rtol=1e-6)
        want = fmt_max.cpu() / (first.cpu() * max(window))
        assert torch.allclose(scale.cpu(), want, rtol=1e-6) and torch.allclose(inv.cpu(), 1 / want, rtol=1e-6)


def _tiny(precision, seed=11):
    import sys, os
    from configs import CFG, FDT_VARIANTS, model_kwargs, state_shapes
    from detfill import det_state
    from ilvlm_amd.prototype.model import model_entry
    c, v = CFG["a"], FDT_VARIANTS[0]
    kw = model_kwargs(c, v)
    kw["precision"] = precision
    model = model_entry(dict(type="clip_fdt_vitb32", kwargs=kw))
    model.load_state_dict({k: torch.from_numpy(a) for k, a in det_state(state_shapes(c, True), seed).items()})
    return model.cuda().train(), c


def _cos(a, b):
    a, b = a.detach().double().flatten().cpu(), b.detach().double().flatten().cpu()
    return float((a * b).sum() / (a.norm() * b.norm()).clamp_min(1e-300))


def _oracle_step(p0, img, tok, mask, cfg, names, fp8):
    """fp32 oracle step (fp8 = False) or the fp8-mode control (fp8 = True): the SAME oracle with e4m3 / e5m2 operands in the
    four GEMMs of every residual attention block and bf16 operands elsewhere (oracle/clip_oracle.py: fp8_blocks, rounding)"""
    import contextlib
    p = {k: v.detach().clone().requires_grad_(k in names) for k, v in p0.items()}
    with contextlib.ExitStack() as st:
        if fp8:
            st.enter_context(O.rounding(O.bf16_ste))
            st.enter_context(O.fp8_blocks())
        o = O.clip_fdt_forward(p, img, tok, mask, cfg)
        loss, _ = O.info_nce(o["logits_i"], o["logits_t"])
        loss.backward()
    return o["logits_i"].detach(), {n: p[n].grad.detach() for n in names}


# The fp8 tolerances are CONTROLLED, not measured-and-frozen (round 2 asserted min cos > 0.8 / 0.85 because that is what the
# kernels gave): the HIP fp8 gradients must point at the fp32 oracle's at least as well as the fp8-operand oracle's do,
# minus MARGIN; the logits error may be CTRL x the control's + FLOOR (the pattern of tests/test_parity_bf16_gpu.py).
MARGIN, CTRL, FLOOR = 0.03, 3.0, 2e-3


def _check_against_control(tag, li_hip, g_hip, ref, ctl):
    (li_ref, g_ref), (li_ctl, g_ctl) = ref, ctl
    sc = float(li_ref.abs().max())
    e_hip = float((li_hip.float().cpu() - li_ref).abs().max()) / sc
    e_ctl = float((li_ctl - li_ref).abs().max()) / sc
    ch = {n: _cos(g_hip[n], g_ref[n]) for n in g_ref}
    cc = {n: _cos(g_ctl[n], g_ref[n]) for n in g_ref}
    worst = min(ch, key=lambda n: ch[n] - cc[n])
    med = lambda d: sorted(d.values())[len(d) // 2]
    print("%s: logits error HIP fp8 %.3e, fp8-operand oracle %.3e; gradient cosine vs fp32 oracle: HIP min %.4f median %.4f, "
          "control min %.4f median %.4f; largest shortfall %.4f (%s)" % (tag, e_hip, e_ctl, min(ch.values()), med(ch),
                                                                        min(cc.values()), med(cc), cc[worst] - ch[worst], worst))
    assert e_hip <= CTRL * e_ctl + FLOOR
    for n in g_ref:
        assert ch[n] >= cc[n] - MARGIN, "fp8 gradient of %s: cos %.4f, the dtype's own (control) %.4f" % (n, ch[n], cc[n])
    assert med(ch) >= med(cc) - MARGIN / 2


def test_fp8_mode_tiny_model_tracks_bf16():
    """fp8 mode on the tiny golden geometry: the first step only observes (bf16 kernels, identical to bf16 mode), from the
    second step on the block GEMMs run in fp8; scales are derived from the observed amaxes; logits and the gradients of every
    block weight are as close to the fp32 oracle as the fp8-operand oracle (the dtype's own error) is"""
    from configs import CFG, FDT_VARIANTS, oracle_cfg, state_shapes
    from detfill import det_images, det_tokens, det_state
    from ilvlm_amd.prototype.loss_functions import ClipInfoCELoss
    m8, c = _tiny("fp8")
    mb, _ = _tiny("bf16")
    img = torch.from_numpy(det_images(c["batch"], c["res"], 5)).cuda()
    tok, mask = det_tokens(c["batch"], c["ctx"], 5)
    texts = (torch.from_numpy(tok), torch.from_numpy(mask))
    outs = {}
    for name, m in (("fp8", m8), ("bf16", mb)):
        res = []
        for it in range(3):
            (li, lt), _ = m(img, texts)
            loss, _ = ClipInfoCELoss()(li, lt)
            m.zero_grad()
            loss.backward()
            torch.cuda.synchronize()
            res.append((li.detach().float().cpu(), {n: p.grad.detach().float().cpu().clone() for n, p in m.named_parameters()
                                                    if p.grad is not None}))
        outs[name] = res
    f8 = m8.engine.fp8
    assert f8 is not None and f8.active and f8.steps == 3 and f8.bwd_seen == 3
    assert torch.equal(outs["fp8"][0][0], outs["bf16"][0][0])          # observing step == bf16 step
    assert float(f8.scale.min()) > 0 and bool((f8.scale != 1).any())
    names = [n for n, g in outs["bf16"][2][1].items() if "resblocks" in n and n.endswith("weight") and g.dim() == 2]
    p0 = {k: torch.from_numpy(a) for k, a in det_state(state_shapes(c, True), 11).items()}
    cfg = oracle_cfg(c, FDT_VARIANTS[0])
    args = (img.cpu(), torch.from_numpy(tok), torch.from_numpy(mask), cfg, names)
    _check_against_control("tiny model", outs["fp8"][2][0], outs["fp8"][2][1], _oracle_step(p0, *args, fp8=False),
                           _oracle_step(p0, *args, fp8=True))


def test_fp8_mode_vitl14_tracks_bf16():
    """fp8 mode at the ViT-L/14 + FDT geometry (width 1024, 257 tokens: the forward attention emits the e4m3 copy itself,
    the key-block backward its e5m2 copy), batch 8, 24 layers of e5m2 gradients: logits and sampled weight gradients of
    blocks 0 / 11 / 23 against the fp32 oracle, bounded by what the fp8-operand oracle itself loses"""
    import bench as B
    from ilvlm_amd.prototype.model import model_entry
    from ilvlm_amd.prototype.loss_functions import ClipInfoCELoss
    images, tokens, pad, lens = B.synthetic_batch(8, 0, "cuda")
    torch.manual_seed(0)
    m = model_entry(dict(type="clip_fdt_vitL14", kwargs=B.fdt_kwargs("fp8", "vitl14")))
    p0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.cuda().train()
    for it in range(2):
        (li, lt), _ = m(images, (tokens, pad, lens))
        loss, _ = ClipInfoCELoss()(li, lt)
        m.zero_grad()
        loss.backward()
    torch.cuda.synchronize()
    assert m.engine.fp8.active
    g_hip = {n: p.grad.detach().float().cpu().clone() for n, p in m.named_parameters()
             if p.grad is not None and ("resblocks.0." in n or "resblocks.11." in n or "resblocks.23." in n) and p.dim() == 2}
    assert len(g_hip) >= 16
    names = list(g_hip)
    torch.set_num_threads(min(32, __import__("os").cpu_count() or 8))
    cfg = dict(v_heads=16, t_heads=12, temperature=1000.0, att_func="sparsemax", pool="max")
    args = (images.cpu(), tokens.cpu(), pad.cpu(), cfg, names)
    _check_against_control("ViT-L/14 B=8", li.detach(), g_hip, _oracle_step(p0, *args, fp8=False), _oracle_step(p0, *args, fp8=True))


def test_fp8_scales_follow_an_amax_jump_without_overflow():
    """delayed scaling under a sudden change: ln_1's gain of one block is multiplied by 100 for one step (the in-projection's
    input jumps 100 x against a scale derived from the history).  That step saturates instead of overflowing (finite
    logits, loss and gradients), its amax is recorded, and the very next step's scale has followed the jump."""
    from detfill import det_images, det_tokens
    from ilvlm_amd.prototype.loss_functions import ClipInfoCELoss
    m, c = _tiny("fp8")
    img = torch.from_numpy(det_images(c["batch"], c["res"], 5)).cuda()
    tok, mask = det_tokens(c["batch"], c["ctx"], 5)
    texts = (torch.from_numpy(tok), torch.from_numpy(mask))

    def step():
        (li, lt), _ = m(img, texts)
        loss, _ = ClipInfoCELoss()(li, lt)
        m.zero_grad()
        loss.backward()
        torch.cuda.synchronize()
        grads = torch.cat([p.grad.flatten() for p in m.parameters() if p.grad is not None])
        return li, loss, grads
    for _ in range(3):
        step()
    f8 = m.engine.fp8
    assert f8.active
    pre = "visual.transformer.resblocks.0."
    slot = f8.slots[pre + "h1"]
    s_before = float(f8.scale[slot])
    g = m.visual.transformer.resblocks[0].ln_1.weight
    with torch.no_grad():
        g.mul_(100.0)
    li, loss, grads = step()                     # quantised with the OLD scale: saturates at +-448 / scale
    assert bool(torch.isfinite(li).all()) and bool(torch.isfinite(loss)) and bool(torch.isfinite(grads).all())
    with torch.no_grad():
        g.div_(100.0)
    li, loss, grads = step()                     # begin_step folded the jump's amax into the history
    s_after = float(f8.scale[slot])
    assert s_after < s_before / 50, (s_before, s_after)
    assert bool(torch.isfinite(li).all()) and bool(torch.isfinite(grads).all())
    # recovery: the running amax restarts each step at 0.9 x the window maximum (csrc/fp8.hip: keeps thousands of waves from
    # hammering one word), so an outlier decays by 10 % per step once it is the window maximum -- conservative (the scale
    # is never too large), back within 2 x of the old scale after HIST + ~50 steps
    for _ in range(f8.HIST + 1):
        step()
    assert float(f8.scale[slot]) > s_after
    for _ in range(55):
        step()
    assert float(f8.scale[slot]) > s_before / 2, (s_before, float(f8.scale[slot]))


def test_fp8_weight_copies_follow_a_text_encoder_reset():
    """iterated learning under fp8 (train_solver.py:545-557): reset_text_encoder() re-initialises the text tower's fp32 masters
    behind Fp8State; the next training forward must run on e4m3 copies of the NEW weights (and their transposes), quantised
    with the scale then in force"""
    from detfill import det_images, det_tokens
    from ilvlm_amd.prototype.loss_functions import ClipInfoCELoss
    m, c = _tiny("fp8")
    img = torch.from_numpy(det_images(c["batch"], c["res"], 5)).cuda()
    tok, mask = det_tokens(c["batch"], c["ctx"], 5)
    texts = (torch.from_numpy(tok), torch.from_numpy(mask))

    def step():
        (li, lt), _ = m(img, texts)
        loss, _ = ClipInfoCELoss()(li, lt)
        m.zero_grad()
        loss.backward()
        torch.cuda.synchronize()
        return li
    for _ in range(3):
        step()
    f8 = m.engine.fp8
    pre = "encode_text.transformer.resblocks.0."
    name = pre + "mlp.c_fc.weight"
    old8 = f8.w8(pre, "fc_w").clone()
    old_master = m.engine.arena.views[name].clone()
    torch.manual_seed(123)
    m.reset_text_encoder(1)
    assert not torch.equal(m.engine.arena.views[name], old_master)
    li = step()
    assert bool(torch.isfinite(li).all())
    scale = float(f8.scale[f8.slots[pre + "fc_w"]])
    w = m.engine.arena.views[name].detach().cpu()
    # the optimizer has not stepped: the masters are the reset values; their e4m3 image at the current scale
    want = to_f8(w, scale, False).view(torch.uint8)
    got = f8.w8(pre, "fc_w").cpu()
    assert not torch.equal(got, old8.cpu())
    assert not bool(((got != want) & ((got & 0x7f) != 0)).any()), "fp8 weight copy is not the quantised NEW weight"
    got_t = f8.w8(pre, "fc_w", transposed=True).cpu()
    assert torch.equal(got_t, got.t().contiguous())


def test_fp8_state_is_not_advanced_by_inference_forwards():
    """advisor finding (round 2): a no-grad forward (eval entry points; the data-parallel wrapper's prepare()) consumed the
    observe-only step, so the first training step quantised e5m2 gradients at scale 1.  Only training forwards advance the
    delayed-scaling state, fp8 GEMMs switch on after one observed forward AND backward, and a backward runs in the mode of
    its own forward."""
    from detfill import det_images, det_tokens
    from ilvlm_amd.prototype.loss_functions import ClipInfoCELoss
    m, c = _tiny("fp8")
    img = torch.from_numpy(det_images(c["batch"], c["res"], 5)).cuda()
    tok, mask = det_tokens(c["batch"], c["ctx"], 5)
    texts = (torch.from_numpy(tok), torch.from_numpy(mask))
    m.engine.prepare()                                   # what NativeDDP.__init__ does
    with torch.no_grad():
        m.encode_image(img)
        m(img, texts)
    f8 = m.engine.fp8
    assert f8.steps == 0 and not f8.active and f8.bwd_seen == 0
    wslot = f8.slots["visual.transformer.resblocks.0.fc_w"]
    assert float(f8.scale[wslot]) != 1.0                 # ... but the weights are quantised with a real scale already
    (li, lt), _ = m(img, texts)                          # training forward 1: observe only
    assert f8.steps == 1 and not f8.active
    (li2, lt2), _ = m(img, texts)                        # a second forward BEFORE any backward: still observe only
    assert f8.steps == 2 and not f8.active
    loss, _ = ClipInfoCELoss()(li2, lt2)
    m.zero_grad()
    loss.backward()
    assert f8.bwd_seen == 1
    (li3, lt3), _ = m(img, texts)                        # forward + backward observed: fp8 GEMMs from here on
    assert f8.active
    gslot = f8.slots["visual.transformer.resblocks.0.dout"]
    assert float(f8.scale[gslot]) != 1.0, "gradient slots must have a history before fp8 GEMMs switch on"
    # backward of the observe-only forward (li, lt) AFTER the state switched to active: runs in its own forward's mode
    loss_old, _ = ClipInfoCELoss()(li, lt)
    m.zero_grad()
    loss_old.backward()
    torch.cuda.synchronize()
    g_old = m.visual.transformer.resblocks[0].mlp.c_fc.weight.grad.clone()
    assert f8.active and bool(torch.isfinite(g_old).all())
    mb, _ = _tiny("bf16")
    (lb, ltb), _ = mb(img, texts)
    lossb, _ = ClipInfoCELoss()(lb, ltb)
    mb.zero_grad()
    lossb.backward()
    torch.cuda.synchronize()
    gb = mb.visual.transformer.resblocks[0].mlp.c_fc.weight.grad
    # (split-K fp32 atomics: equal up to the summation order)
    assert float((g_old - gb).abs().max()) <= 1e-3 * float(gb.abs().max()), "observe-mode backward must equal the bf16 one"
    # advisor finding (round 3): two ACTIVE training forwards before the first backward -- the second one rewrites the delayed
    # scales and re-quantises the weights in place, so the first forward's saved fp8 copies can no longer be de-quantised:
    # refused loudly instead of producing silently wrong gradients; the latest forward's own backward is fine
    loss3, _ = ClipInfoCELoss()(li3, lt3)
    (li4, lt4), _ = m(img, texts)
    with pytest.raises(RuntimeError, match="replaced its delayed scales"):
        loss3.backward()
    loss4, _ = ClipInfoCELoss()(li4, lt4)
    m.zero_grad()
    loss4.backward()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(m.visual.transformer.resblocks[0].mlp.c_fc.weight.grad).all())


def test_fp8_loss_curve_tracks_bf16_at_real_size():
    """BASELINE configs[4] validation at reduced length (benchmarks/fp8_loss_curve.py runs the >= 200-step curve): the
    shipped geometry, per-GPU batch 64, AdamW + the cosine schedule, 24 steps on one resident synthetic batch (what
    bench.py trains on), fp8 against bf16 from the same initial weights -- the two loss curves stay together"""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "benchmarks"))
    import fp8_loss_curve as FC
    l8 = FC.run("fp8", steps=24, batch=64, n_batches=1)
    lb = FC.run("bf16", steps=24, batch=64, n_batches=1)
    rel = [abs(a - b) / max(abs(b), 1e-3) for a, b in zip(l8, lb)]
    print("fp8 vs bf16 loss: first %.4f / %.4f, last %.4f / %.4f, max relative gap %.3f" % (l8[0], lb[0], l8[-1], lb[-1], max(rel)))
    assert lb[-1] < 0.7 * lb[0], "the bf16 run itself must learn"
    assert l8[-1] < 0.7 * l8[0]
    gap = max(abs(a - b) for a, b in zip(l8, lb))
    assert gap < 0.1 * lb[0], "largest absolute gap %.4f of an initial loss of %.4f" % (gap, lb[0])


def test_fp8_loss_curve_on_distinct_learnable_batches():
    """the same comparison where every batch is different and the task can be learnt (512 latent concepts, a fresh sample of
    128 plus fresh pixel noise per step; benchmarks/fp8_loss_curve.py --structured runs the 400-step, batch-256 curves of
    profiles/round3/): delayed scaling has to follow activations and gradients that change as training moves.  Both
    precisions learn, and fp8 reaches a loss of 1.0 within 1.5 x the steps (+ 16) bf16 needs.  The peak is
    1e-4 because at the shipped warm-up target of 5e-4 this task is unstable in EVERY precision, fp32 included: on the
    shipped schedule (500 warm-up steps) all three spike at step ~20 (loss 6.3 .. 6.9) and what follows is luck -- fp8 ends
    at 0.025, fp32 learns more slowly, bf16 stays at ln(batch); with 50 warm-up steps fp8 and bf16 both stay at ln(batch)
    (profiles/round3/f*_loss_curve_structured_*.json).  A pointwise comparison there measures chaos, not precision."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "benchmarks"))
    import fp8_loss_curve as FC
    l8 = FC.run("fp8", steps=200, batch=128, peak_lr=1e-4, structured=True)
    lb = FC.run("bf16", steps=200, batch=128, peak_lr=1e-4, structured=True)
    tail = lambda x: sum(x[-16:]) / 16

    def reach(x, thr=1.0):       # first step from which the 8-step running mean stays below thr
        for i in range(8, len(x) + 1):
            if sum(x[i - 8:i]) / 8 < thr:
                return i
        return None
    r8, rb = reach(l8), reach(lb)
    print("fp8 vs bf16 on distinct learnable batches: first %.3f / %.3f, loss < 1.0 from step %s / %s, mean of the last 16 steps "
          "%.4f / %.4f" % (l8[0], lb[0], r8, rb, tail(l8), tail(lb)))
    assert all(math.isfinite(v) for v in l8)
    # Weight-gradient K-slices meet in fp32 atomics, so two runs of ONE precision differ by tens of steps in when the loss
    # breaks away from ln(batch) (measured: the last-16 mean at step 160 ranged 0.08..0.30 for bf16 and 0.10..0.81 for fp8 over
    # five runs).  What is asserted is therefore when each precision gets there, not a pointwise gap.
    assert rb is not None and r8 is not None, "both precisions must learn (loss below 1.0 of an initial %.2f)" % lb[0]
    assert r8 <= 1.5 * rb + 16, "fp8 needs %d steps to reach a loss of 1.0, bf16 %d" % (r8, rb)
    assert tail(l8) < 1.0 and tail(lb) < 1.0


def test_fused_fp8_copies_equal_the_separate_quantise_pass():
    """LayerNorm forward / backward and the GEMM epilogue can emit the fp8 copy of their output themselves (fp8 mode's fused
    quantisation): same bytes and same amax as quantising the stored output afterwards"""
    import ctypes as C
    from ilvlm_amd import ops, lib as L
    h = L.load()
    st = torch.cuda.current_stream().cuda_stream
    rows, cols = 777, 768
    x = rnd(rows, cols, seed=1).cuda()
    gamma, beta = (1 + 0.1 * rnd(cols, seed=2)).cuda(), (0.1 * rnd(cols, seed=3)).cuda()
    y = torch.empty(rows, cols, device="cuda", dtype=torch.bfloat16); mean = torch.empty(rows, device="cuda"); rstd = torch.empty_like(mean)
    y8 = torch.zeros(rows, cols, dtype=torch.uint8, device="cuda")
    scale = torch.tensor([37.0], device="cuda"); amax = torch.zeros(1, device="cuda")
    L.check(h.ilvlm_layernorm_fwd_q8(x.data_ptr(), L.F32, gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), L.BF16, mean.data_ptr(),
                                     rstd.data_ptr(), rows, cols, 1e-5, 0, 0, y8.data_ptr(), scale.data_ptr(), amax.data_ptr(), st), "ln_fwd_q8")
    # the fused copy is taken from the fp32 values before the bf16 rounding of y: compare against quantising those
    yf = torch.empty(rows, cols, device="cuda")
    ops.layernorm_fwd(x, gamma, beta, yf, mean, rstd, rows, cols)
    want = to_f8(yf.cpu(), 37.0, False).view(torch.uint8)
    got = y8.cpu()
    assert not bool(((got != want) & ((got & 0x7f) != 0)).any())
    assert abs(float(amax) - float(yf.abs().max())) <= 1e-6 * float(yf.abs().max())
    # LayerNorm backward: e5m2 copy of dx_lp
    dy = (rnd(rows, cols, seed=4) * 1e-3).to(torch.bfloat16).cuda()
    dg, db = torch.zeros(cols, device="cuda"), torch.zeros(cols, device="cuda")
    dx = torch.empty(rows, cols, device="cuda"); dx_lp = torch.empty(rows, cols, device="cuda", dtype=torch.bfloat16)
    dx8 = torch.zeros(rows, cols, dtype=torch.uint8, device="cuda")
    s2 = torch.tensor([2.0e4], device="cuda"); a2 = torch.zeros(1, device="cuda")
    L.check(h.ilvlm_layernorm_bwd_q8(dy.data_ptr(), L.BF16, x.data_ptr(), L.F32, mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                     None, dx.data_ptr(), dx_lp.data_ptr(), L.BF16, 0, None, dg.data_ptr(), db.data_ptr(), rows, cols,
                                     0, 0, None, 1, dx8.data_ptr(), s2.data_ptr(), a2.data_ptr(), st), "ln_bwd_q8")
    want = to_f8(dx.cpu(), 2.0e4, True).view(torch.uint8)
    got = dx8.cpu()
    assert not bool(((got != want) & ((got & 0x7f) != 0)).any())
    assert abs(float(a2) - float(dx.abs().max())) <= 1e-6 * float(dx.abs().max())
    # GEMM epilogue: QuickGELU forward with the activation's e4m3 copy, ragged rows (generic path on the last row tile)
    M, N, K = 300, 256, 128
    a = rnd(M, K, seed=5).to(torch.bfloat16).cuda(); w = (rnd(N, K, seed=6) * 0.1).to(torch.bfloat16).cuda()
    bias = rnd(N, seed=7).cuda()
    g = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); u = torch.empty_like(g)
    g8 = torch.zeros(M, N, dtype=torch.uint8, device="cuda")
    s3 = torch.tensor([55.0], device="cuda"); a3 = torch.zeros(1, device="cuda")
    epi = L.GemmEpilogue(bias.data_ptr(), None, None, u.data_ptr(), None, 1.0, 1, L.BF16, 0, 0, 0, None, None, None, None, 0, None,
                         g8.data_ptr(), s3.data_ptr(), a3.data_ptr(), 0)
    L.check(h.ilvlm_gemm(L.BF16, 0, 0, M, N, K, a.data_ptr(), K, w.data_ptr(), K, g.data_ptr(), N, C.byref(epi), 1, st), "gemm out8")
    pre = a.float() @ w.float().t() + bias
    gf = O.quick_gelu(pre).cpu()
    want = to_f8(gf, 55.0, False)
    got = g8.cpu().view(torch.float8_e4m3fn)
    # fp32 sums differ in the last bit between the MFMA order and torch: allow one fp8 code of difference at rounding ties
    d = (got.float() - want.float()).abs()
    assert float(d.max()) <= 0.13 * float(want.float().abs().max()) and float((d > 0).float().mean()) < 0.02
    assert abs(float(a3) - float(gf.abs().max())) < 1e-3 * float(gf.abs().max())
    # ... and with no main output at all: the same copy, u still written, g untouched
    g2 = torch.full((M, N), 7.0, device="cuda", dtype=torch.bfloat16); u2 = torch.empty_like(g2)
    g8b = torch.zeros(M, N, dtype=torch.uint8, device="cuda"); a3b = torch.zeros(1, device="cuda")
    epi = L.GemmEpilogue(bias.data_ptr(), None, None, u2.data_ptr(), None, 1.0, 1, L.BF16, 0, 0, 0, None, None, None, None, 0, None,
                         g8b.data_ptr(), s3.data_ptr(), a3b.data_ptr(), 0)
    L.check(h.ilvlm_gemm(L.BF16, 0, 0, M, N, K, a.data_ptr(), K, w.data_ptr(), K, None, N, C.byref(epi), 1, st), "gemm out8 only")
    assert torch.equal(g8b, g8) and torch.equal(u2, u) and float(a3b) == float(a3)
    epi = L.GemmEpilogue(bias.data_ptr(), None, None, None, None, 1.0, 0, L.BF16, 0, 0, 0, None)
    assert h.ilvlm_gemm(L.BF16, 0, 0, M, N, K, a.data_ptr(), K, w.data_ptr(), K, None, N, C.byref(epi), 1, st) == -1   # no copy either
    # LayerNorm with only the fp8 copy kept
    y8b = torch.zeros_like(y8); amaxb = torch.zeros(1, device="cuda")
    L.check(h.ilvlm_layernorm_fwd_q8(x.data_ptr(), L.F32, gamma.data_ptr(), beta.data_ptr(), None, L.BF16, mean.data_ptr(),
                                     rstd.data_ptr(), rows, cols, 1e-5, 0, 0, y8b.data_ptr(), scale.data_ptr(), amaxb.data_ptr(), st), "ln_fwd_q8")
    assert torch.equal(y8b, y8) and float(amaxb) == float(amax)
    dx8b = torch.zeros_like(dx8); a2b = torch.zeros(1, device="cuda"); dxb = torch.empty_like(dx)
    dg2, db2 = torch.zeros(cols, device="cuda"), torch.zeros(cols, device="cuda")
    L.check(h.ilvlm_layernorm_bwd_q8(dy.data_ptr(), L.BF16, x.data_ptr(), L.F32, mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                     None, dxb.data_ptr(), None, L.BF16, 0, None, dg2.data_ptr(), db2.data_ptr(), rows, cols,
                                     0, 0, None, 1, dx8b.data_ptr(), s2.data_ptr(), a2b.data_ptr(), st), "ln_bwd_q8")
    assert torch.equal(dx8b, dx8) and torch.equal(dxb, dx) and float(a2b) == float(a2)


@pytest.mark.parametrize("packed,Lx", [(False, 50), (True, 50), (False, 257)])
def test_attention_kernels_emit_the_fp8_copy_of_their_output(packed, Lx):
    """fp8 mode: attention forward / backward write the e4m3 copy of `out` / the e5m2 copy of `dqkv` themselves: the same
    bytes and amax as a quantise pass over the bf16 tensor; the backward may drop the bf16 tensor altogether.  L = 257 is
    ViT-L/14's token count: the key-block backward kernel (round 3: no separate quantise pass for long sequences either)"""
    from ilvlm_amd import ops, lib as L
    h = L.load()
    st = torch.cuda.current_stream().cuda_stream
    B, H = 5, 4
    E = 64 * H
    if packed:
        lens = [50, 7, 33, 1, 16]
        seq = ops.PackedSeq(lens, Lx, "cuda")
        rows, causal = sum(lens), 1
        offs, cap = seq.offs.data_ptr(), seq.cap
    else:
        seq, rows, causal, offs, cap = None, B * Lx, 0, None, Lx
    qkv = rnd(rows, 3 * E, seed=1).to(torch.bfloat16).cuda()
    out = torch.empty(rows, E, device="cuda", dtype=torch.bfloat16); lse = torch.zeros(B, H, Lx, device="cuda")
    ops.attention_fwd(qkv, out, lse, B, Lx, H, causal, seq)
    out2 = torch.empty_like(out); lse2 = torch.zeros_like(lse)
    o8 = torch.zeros(rows, E, dtype=torch.uint8, device="cuda")
    sc = torch.tensor([300.0], device="cuda"); am = torch.zeros(1, device="cuda")
    L.check(h.ilvlm_attention_fwd_q8(qkv.data_ptr(), out2.data_ptr(), lse2.data_ptr(), L.BF16, B, Lx, cap, H, causal, offs,
                                     o8.data_ptr(), sc.data_ptr(), am.data_ptr(), st), "attention_fwd_q8")
    assert torch.equal(out2, out) and torch.equal(lse2, lse)
    ref8 = torch.zeros_like(o8); ram = torch.zeros(1, device="cuda")
    ops.fp8_quantize(out, ref8, sc, ram)
    assert torch.equal(o8, ref8) and float(am) == float(ram)
    # backward
    dout = (rnd(rows, E, seed=2) * 1e-3).to(torch.bfloat16).cuda()
    dqkv = torch.empty(rows, 3 * E, device="cuda", dtype=torch.bfloat16)
    ops.attention_bwd(dout, qkv, out, lse, dqkv, B, Lx, H, causal, seq)
    d8 = torch.zeros(rows, 3 * E, dtype=torch.uint8, device="cuda"); dq2 = torch.empty_like(dqkv)
    s2 = torch.tensor([1.0e5], device="cuda"); a2 = torch.zeros(1, device="cuda")
    L.check(h.ilvlm_attention_bwd_q8(dout.data_ptr(), qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), dq2.data_ptr(), L.BF16, B, Lx,
                                     cap, H, causal, offs, d8.data_ptr(), s2.data_ptr(), a2.data_ptr(), st), "attention_bwd_q8")
    assert torch.equal(dq2, dqkv)
    ref8 = torch.zeros_like(d8); ram = torch.zeros(1, device="cuda")
    ops.fp8_quantize(dqkv, ref8, s2, ram, e5m2=True)
    assert torch.equal(d8, ref8) and float(a2) == float(ram)
    d8b = torch.zeros_like(d8); a2b = torch.zeros(1, device="cuda")
    L.check(h.ilvlm_attention_bwd_q8(dout.data_ptr(), qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), None, L.BF16, B, Lx,
                                     cap, H, causal, offs, d8b.data_ptr(), s2.data_ptr(), a2b.data_ptr(), st), "attention_bwd_q8 (copy only)")
    assert torch.equal(d8b, d8) and float(a2b) == float(a2)

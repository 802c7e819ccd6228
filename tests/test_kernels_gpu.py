"""GPU parity tests of every C-ABI kernel against a plain PyTorch fp32 reference of the same op
(computed on the CPU) and, where one exists, the oracle / golden fixture.  Run with `-m gpu`.
Tolerances: fp32 kernels 1e-5..1e-4 relative (summation order), bf16 kernels 1e-2 (north_star)."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import clip_oracle as O  # noqa: E402


def _ops():
    from ilvlm_amd import ops
    return ops


def dev(t):
    return t.cuda().contiguous()


def rel(a, b):
    a = a.detach().float().cpu().double()
    b = b.detach().float().cpu().double()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def test_fragment_maps():
    """MFMA C/D map, operand K coverage and the ds_read_b64_tr_b16 lane map the kernels rely on."""
    out = _ops().selftest_fragments().cpu()
    lane = torch.arange(64)
    g, c = lane // 16, lane % 16
    assert torch.all(out[0, :, :4] == 496.0)
    for j in range(8):
        assert torch.equal(out[1, :, j], (8 * g + j).float()), "tr read k map, j=%d" % j
        assert torch.equal(out[4, :, j], c.float()), "tr read col map, j=%d" % j
    for r in range(4):
        want = ((4 * g + r) * (c + 1)).float()
        assert torch.equal(out[2, :, r], want), "bf16 mfma C/D map"
        assert torch.equal(out[3, :, r], want), "f32 mfma C/D map"


GEMM_SHAPES = [(128, 128, 64), (200, 136, 72), (24, 384, 128), (328, 64, 40), (1000, 768, 512), (256, 256, 1032)]
GEMM_SHAPES_F32_ODD = [(3, 3, 64), (5, 15, 33), (67, 3, 130), (3, 130, 5)]


@pytest.mark.parametrize("variant", [5, 15])
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 1), (1, 0)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (1000, 768, 512), (640, 392, 256), (136, 2304, 768), (8, 8, 128), (512, 256, 64),
                                   (776, 520, 192)])
def test_gemm_direct_to_lds_variants(variant, ta, tb, M, N, K):
    """the direct-to-LDS 128x128 kernel under both selector settings (5 = always; 15 = default, which only differs where a
    packed B operand is offered), incl. ragged M/N tiles, split-K and the fused bias-gradient row sum"""
    ops = _ops()
    a, b = rnd(M, K, seed=1).to(torch.bfloat16), rnd(N, K, seed=2).to(torch.bfloat16)
    want = a.float() @ b.float().t()
    A, B = dev(a.t() if ta else a), dev(b.t() if tb else b)
    try:
        ops.gemm_set_variant(variant)
        out = torch.full((M, N), float("nan"), device="cuda")
        ops.gemm(A, B, out, trans_a=bool(ta), trans_b=bool(tb))
        assert rel(out, want) < 2e-5
        outb = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        ops.gemm(A, B, outb, trans_a=bool(ta), trans_b=bool(tb), bias=dev(rnd(N, seed=3)))
        assert rel(outb, want + rnd(N, seed=3)) < 1e-2
        for split in (1, 3):
            acc = torch.ones(M, N, device="cuda")
            rs = torch.ones(M, device="cuda")
            ops.gemm(A, B, acc, trans_a=bool(ta), trans_b=bool(tb), accumulate=True, split_k=split,
                     a_rowsum=rs if M % 8 == 0 else None)
            assert rel(acc, want + 1) < 2e-5
            if M % 8 == 0:
                assert rel(rs, 1 + a.float().sum(1)) < 2e-5
    finally:
        ops.gemm_set_variant(15)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 1), (1, 0)])
@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_plain(dtype, ta, tb, M, N, K):
    ops = _ops()
    a = rnd(M, K, seed=1).to(dtype)
    b = rnd(N, K, seed=2).to(dtype)
    want = a.float() @ b.float().t()
    A = dev(a.t() if ta else a)
    B = dev(b.t() if tb else b)
    out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.float32)
    ops.gemm(A, B, out, trans_a=bool(ta), trans_b=bool(tb))
    assert rel(out, want) < 2e-5
    if dtype == torch.bfloat16:
        outb = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        ops.gemm(A, B, outb, trans_a=bool(ta), trans_b=bool(tb))
        assert rel(outb, want) < 1e-2


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 1), (1, 0)])
@pytest.mark.parametrize("M,N,K", GEMM_SHAPES_F32_ODD)
def test_gemm_f32_odd_leading_dimensions(ta, tb, M, N, K):
    """logit matrices of odd local batch sizes: leading dimensions that are not multiples of 4"""
    ops = _ops()
    a, b = rnd(M, K, seed=1), rnd(N, K, seed=2)
    want = a @ b.t()
    out = torch.full((M, N), float("nan"), device="cuda")
    ops.gemm(dev(a.t() if ta else a), dev(b.t() if tb else b), out, trans_a=bool(ta), trans_b=bool(tb))
    assert rel(out, want) < 2e-5
    acc = torch.ones(M, N, device="cuda")
    ops.gemm(dev(a.t() if ta else a), dev(b.t() if tb else b), acc, trans_a=bool(ta), trans_b=bool(tb), accumulate=True)
    assert rel(acc, want + 1) < 2e-5


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gemm_epilogues(dtype):
    ops = _ops()
    M, N, K = 392, 264, 136
    a, b = rnd(M, K, seed=3).to(dtype), rnd(N, K, seed=4).to(dtype)
    bias, res = rnd(N, seed=5), rnd(M, N, seed=6)
    base = a.float() @ b.float().t() / math.sqrt(K)
    A, B = dev(a), dev(b)
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    # bias + residual, fp32 out, alpha through both host and device scalars
    out = torch.empty(M, N, device="cuda")
    alpha_dev = torch.tensor([0.5], device="cuda")
    ops.gemm(A, B, out, bias=dev(bias), residual=dev(res), alpha=2.0 / math.sqrt(K), alpha_ptr=alpha_dev)
    assert rel(out, base + bias + res) < 2e-5
    # activations forward: pre-activation saved to aux, output in compute dtype
    for act, fn in ((1, O.quick_gelu), (2, O.gelu_erf)):
        aux = torch.empty(M, N, device="cuda", dtype=dtype)
        y = torch.empty(M, N, device="cuda", dtype=dtype)
        ops.gemm(A, B, y, bias=dev(bias), aux=aux, act=act, alpha=1 / math.sqrt(K))
        assert rel(aux, base + bias) < tol
        assert rel(y, fn(base + bias)) < tol
        # backward multiply by act'(aux)
        u = (base + bias).to(dtype).float().requires_grad_(True)
        fn(u).sum().backward()
        dy = torch.empty(M, N, device="cuda", dtype=dtype)
        ops.gemm(A, B, dy, aux=aux, act=act + 2, alpha=1 / math.sqrt(K))
        assert rel(dy, base * u.grad) < (5e-5 if dtype == torch.float32 else 2e-2)
    # row remap + rowbias (patch embedding epilogue): 8 images x 49 patches -> [8*50, N]
    rb = rnd(50, N, seed=7)
    tokens = torch.zeros(8 * 50, N, device="cuda")
    ops.gemm(A, B, tokens, rowbias=dev(rb), out_group=49, out_skip=1)
    want = torch.zeros(8, 50, N)
    want[:, 1:, :] = (base * math.sqrt(K)).reshape(8, 49, N) + rb[1:]
    assert rel(tokens, want.reshape(400, N)) < 2e-5
    assert float(tokens.reshape(8, 50, N)[:, 0].abs().max()) == 0.0
    # accumulate with split-K on top of existing contents (wgrad form: both operands K-strided)
    for split in (1, 3):
        acc = dev(res.clone())
        ops.gemm(dev(a.t()), dev(b.t()), acc, trans_a=True, trans_b=True, accumulate=True, split_k=split)
        assert rel(acc, res + base * math.sqrt(K)) < 2e-5


def test_gemm_rejects_bad_arguments():
    ops = _ops()
    a = torch.zeros(16, 12, device="cuda", dtype=torch.bfloat16)   # lda = 12 not a multiple of 8
    out = torch.zeros(16, 16, device="cuda")
    with pytest.raises(RuntimeError):
        ops.gemm(a, a, out)
    with pytest.raises(RuntimeError):
        ops.gemm(torch.zeros(16, 16), torch.zeros(16, 16), torch.zeros(16, 16))   # CPU tensors: no fallback


@pytest.mark.parametrize("cols", [768, 512, 128, 1024, 32])
@pytest.mark.parametrize("xdt,ydt", [(torch.float32, torch.float32), (torch.float32, torch.bfloat16),
                                     (torch.bfloat16, torch.bfloat16)])
def test_layernorm(cols, xdt, ydt):
    ops = _ops()
    rows = 101 if cols != 768 else 5003      # 5003 rows: more rows than workgroups x 4 (grid-stride + workspace cap)
    x = (rnd(rows, cols, seed=1) * 3 + 0.5).to(xdt)
    w, b = 1 + 0.1 * rnd(cols, seed=2), 0.1 * rnd(cols, seed=3)
    xr = x.float().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y_ref = O.layer_norm(xr, wr, br)
    dy = rnd(rows, cols, seed=4).to(ydt)
    dres = rnd(rows, cols, seed=5)
    y_ref.backward(dy.float())
    X, W_, B_ = dev(x), dev(w), dev(b)
    y = torch.empty(rows, cols, device="cuda", dtype=ydt)
    mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    ops.layernorm_fwd(X, W_, B_, y, mean, rstd, rows, cols)
    tol = 1e-5 if ydt == torch.float32 else 1e-2
    assert rel(y, y_ref) < tol
    dg, db = torch.zeros(cols, device="cuda"), torch.zeros(cols, device="cuda")
    dx32 = torch.empty(rows, cols, device="cuda")
    dxlp = torch.empty(rows, cols, device="cuda", dtype=ydt)
    for two_stage in (True, False):     # workspace reduction and the atomic fallback; both accumulate ("+=")
        dg.fill_(1.0); db.fill_(1.0)
        ops.layernorm_bwd(dev(dy), X, mean, rstd, W_, dg, db, rows, cols, dres=dev(dres), dx_f32=dx32, dx_lp=dxlp,
                          two_stage=two_stage)
        assert rel(dx32, xr.grad + dres) < 2e-5
        assert rel(dxlp, xr.grad + dres) < tol
        assert rel(dg, 1 + wr.grad) < 2e-5 and rel(db, 1 + br.grad) < 2e-5


def test_layernorm_row_remap_and_act():
    """Patch-token remap (group 49, skip 1) and the fused gelu' multiply of the q_map backward."""
    ops = _ops()
    Bn, P, Wd = 3, 49, 128
    stream = rnd(Bn * (P + 1), Wd, seed=1)
    w, b = 1 + 0.1 * rnd(Wd, seed=2), 0.1 * rnd(Wd, seed=3)
    dense = stream.reshape(Bn, P + 1, Wd)[:, 1:].reshape(Bn * P, Wd).clone().requires_grad_(True)
    y_ref = O.layer_norm(dense, w, b)
    dy = rnd(Bn * P, Wd, seed=4)
    y_ref.backward(dy)
    y = torch.empty(Bn * P, Wd, device="cuda")
    mean, rstd = torch.empty(Bn * P, device="cuda"), torch.empty(Bn * P, device="cuda")
    ops.layernorm_fwd(dev(stream), dev(w), dev(b), y, mean, rstd, Bn * P, Wd, group=P, skip=1)
    assert rel(y, y_ref) < 1e-5
    dg, db = torch.zeros(Wd, device="cuda"), torch.zeros(Wd, device="cuda")
    dstream = torch.zeros(Bn * (P + 1), Wd, device="cuda")
    ops.layernorm_bwd(dev(dy), dev(stream), mean, rstd, dev(w), dg, db, Bn * P, Wd, dx_f32=dstream, group=P, skip=1)
    got = dstream.reshape(Bn, P + 1, Wd)
    assert float(got[:, 0].abs().max()) == 0.0
    assert rel(got[:, 1:].reshape(Bn * P, Wd), dense.grad) < 2e-5
    # act multiply on the low-precision copy
    pre = rnd(Bn * P, Wd, seed=6).requires_grad_(True)
    O.gelu_erf(pre).sum().backward()
    dxlp = torch.empty(Bn * P, Wd, device="cuda")
    dg.zero_(); db.zero_()
    ops.layernorm_bwd(dev(dy), dev(dense.detach()), mean, rstd, dev(w), dg, db, Bn * P, Wd, dx_lp=dxlp, act=4,
                      act_aux=dev(pre.detach()))
    assert rel(dxlp, dense.grad * pre.grad) < 2e-5


def attn_ref(qkv, B, L, H, causal):
    E = 64 * H
    q, k, v = qkv.reshape(B, L, 3 * E).split(E, dim=-1)
    q = q.reshape(B, L, H, 64).transpose(1, 2) * 0.125
    k = k.reshape(B, L, H, 64).transpose(1, 2)
    v = v.reshape(B, L, H, 64).transpose(1, 2)
    s = q @ k.transpose(-1, -2)
    if causal:
        s = s + O.causal_mask(L)
    p = torch.softmax(s, -1)
    return (p @ v).transpose(1, 2).reshape(B * L, E), torch.logsumexp(s, -1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,L,H,causal", [(3, 50, 2, 0), (2, 77, 2, 1), (2, 5, 1, 0), (3, 24, 3, 1), (2, 64, 1, 1),
                                          (2, 10, 1, 0), (1, 100, 2, 0), (1, 128, 1, 1), (2, 257, 2, 0), (1, 197, 1, 0),
                                          (1, 160, 2, 1), (1, 288, 1, 1), (1, 16, 1, 1), (2, 17, 2, 0), (1, 33, 1, 1),
                                          (1, 96, 2, 0), (1, 112, 1, 1), (2, 80, 2, 1), (1, 1, 1, 0), (1, 129, 1, 1)])
def test_attention(dtype, B, L, H, causal):
    ops = _ops()          # fp32: LDS-resident kernels up to L = 96 / 80, global-memory kernels beyond (ViT-L/14 parity mode)
    E = 64 * H
    qkv = rnd(B * L, 3 * E, seed=1).to(dtype)
    dout = rnd(B * L, E, seed=2).to(dtype)
    qr = qkv.float().requires_grad_(True)
    o_ref, lse_ref = attn_ref(qr, B, L, H, causal)
    o_ref.backward(dout.float())
    Q = dev(qkv)
    out = torch.full((B * L, E), float("nan"), device="cuda", dtype=dtype)
    lse = torch.empty(B, H, L, device="cuda")
    ops.attention_fwd(Q, out, lse, B, L, H, causal)
    tol = 2e-5 if dtype == torch.float32 else 1.5e-2
    assert rel(out, o_ref) < tol
    assert rel(lse, lse_ref) < (1e-5 if dtype == torch.float32 else 5e-3)
    dqkv = torch.full((B * L, 3 * E), float("nan"), device="cuda", dtype=dtype)
    ops.attention_bwd(dev(dout), Q, out, lse, dqkv, B, L, H, causal)
    assert rel(dqkv, qr.grad) < (5e-5 if dtype == torch.float32 else 2.5e-2)


def test_embedding_and_tokens():
    ops = _ops()
    B, L, W, V = 5, 24, 64, 1000
    g = torch.Generator().manual_seed(0)
    tok = torch.randint(0, V, (B, L), generator=g)
    tok[0, :4] = 7          # repeated ids: scatter-add collisions
    tok[1, :4] = 7
    table, pos = rnd(V, W, seed=1), rnd(L, W, seed=2)
    x = torch.empty(B * L, W, device="cuda")
    ops.embed_fwd(dev(tok), dev(table), dev(pos), x)
    assert rel(x, (table[tok] + pos).reshape(B * L, W)) == 0.0
    dx = rnd(B * L, W, seed=3)
    dtab, dpos = torch.zeros(V, W, device="cuda"), torch.zeros(L, W, device="cuda")
    ops.embed_bwd(dev(tok), dev(dx), dtab, dpos)
    want = torch.zeros(V, W).index_add_(0, tok.reshape(-1), dx)
    assert rel(dtab, want) < 1e-6
    assert rel(dpos, dx.reshape(B, L, W).sum(0)) < 1e-6
    # patchify == unfold of conv2d(k=s=ps)
    for ps, res in ((32, 64), (14, 42), (16, 48)):
        img = rnd(3, 3, res, res, seed=4)
        gdim = res // ps
        for dtp in (torch.float32, torch.bfloat16):
            out = torch.empty(3 * gdim * gdim, 3 * ps * ps, device="cuda", dtype=dtp)
            ops.patchify(dev(img), out, ps)
            want = torch.nn.functional.unfold(img, ps, stride=ps).transpose(1, 2).reshape(-1, 3 * ps * ps)
            assert rel(out, want.to(dtp)) == 0.0
            kp = (3 * ps * ps + 63) // 64 * 64          # zero-padded rows (K-tile alignment)
            outp = torch.full((3 * gdim * gdim, kp), 7.0, device="cuda", dtype=dtp)
            ops.patchify(dev(img), outp, ps)
            assert rel(outp[:, :3 * ps * ps], want.to(dtp)) == 0.0
            if kp > 3 * ps * ps:
                assert float(outp[:, 3 * ps * ps:].abs().max()) == 0.0
    cls, p2 = rnd(W, seed=5), rnd(L, W, seed=6)
    toks = torch.zeros(B * L, W, device="cuda")
    ops.cls_rows(dev(cls), dev(p2), toks, B, L, W)
    assert rel(toks.reshape(B, L, W)[:, 0], (cls + p2[0]).expand(B, W)) == 0.0
    s_out, s0 = torch.zeros(L, W, device="cuda"), torch.zeros(W, device="cuda")
    ops.batch_sum(dev(dx), s_out, s0, B, L, W)
    assert rel(s_out, dx.reshape(B, L, W).sum(0)) < 1e-6 and rel(s0, dx.reshape(B, L, W)[:, 0].sum(0)) < 1e-6
    idx = torch.tensor([3, 0, 23, 5, 9])
    y = torch.empty(B, W, device="cuda")
    ops.gather_rows(dev(dx), dev(idx), y, B, L, W)
    assert rel(y, dx.reshape(B, L, W)[torch.arange(B), idx]) == 0.0
    acc = torch.zeros(B * L, W, device="cuda")
    ops.scatter_rows(y, dev(idx), acc, B, L, W)
    want = torch.zeros(B, L, W)
    want[torch.arange(B), idx] = dx.reshape(B, L, W)[torch.arange(B), idx]
    assert rel(acc, want.reshape(B * L, W)) == 0.0


@pytest.mark.parametrize("pool", ["max", "mean", "sum"])
@pytest.mark.parametrize("temp", [1.0, 1000.0])
def test_fdt_pool(pool, temp):
    ops = _ops()
    B, T, C, d = 4, 13, 320, 64
    s = rnd(B * T, C, seed=1) * 5
    mask = torch.zeros(B, T)
    mask[1, 9:] = float("-inf")
    mask[3, 3:] = float("-inf")
    sr = s.clone().requires_grad_(True)
    dot = sr.reshape(B, T, C) / math.sqrt(d)
    dot = dot * ((mask == 0) * 1).unsqueeze(-1)
    dot = dot / temp
    pooled_ref = dot.max(1)[0] if pool == "max" else (dot.mean(1) if pool == "mean" else dot.sum(1))
    gp = rnd(B, C, seed=2)
    pooled_ref.backward(gp)
    code = {"max": 0, "mean": 1, "sum": 2}[pool]
    pooled = torch.empty(B, C, device="cuda")
    am = torch.zeros(B, C, device="cuda", dtype=torch.int32)
    ops.fdt_pool_fwd(dev(s), dev(mask), pooled, am, B, T, C, math.sqrt(d), temp, code)
    assert rel(pooled, pooled_ref) < 1e-6
    for dtp, tol in ((torch.float32, 1e-6), (torch.bfloat16, 1e-2)):
        ds = torch.empty(B * T, C, device="cuda", dtype=dtp)
        ops.fdt_pool_bwd(dev(gp), am, dev(mask), ds, B, T, C, math.sqrt(d), temp, code)
        assert rel(ds, sr.grad) < tol


def test_sparsemax_softmax_golden(golden_dir):
    ops = _ops()
    g = np.load(os.path.join(golden_dir, "g3_ops.npz"))
    z, go = torch.from_numpy(g["sparsemax.z"]), torch.from_numpy(g["sparsemax.g"])
    out = torch.empty_like(z, device="cuda")
    ops.sparsemax_fwd(dev(z), out)
    assert rel(out, torch.from_numpy(g["sparsemax.out"])) < 1e-5      # reference Sparsemax output
    assert abs(float(out.sum(-1).sub(1).abs().max())) < 1e-4
    dz = torch.empty_like(out)
    ops.sparsemax_bwd(out, dev(go), dz)
    assert rel(dz, torch.from_numpy(g["sparsemax.dz"])) < 5e-5       # reference autograd through sort/cumsum
    # random rows at odd widths + softmax
    for cols in (128, 320, 4096, 5000):
        zz = rnd(6, cols, seed=cols) * 3
        zr = zz.clone().requires_grad_(True)
        gg = rnd(6, cols, seed=cols + 1)
        O.sparsemax(zr).backward(gg)
        o2 = torch.empty(6, cols, device="cuda")
        ops.sparsemax_fwd(dev(zz), o2)
        assert rel(o2, O.sparsemax(zz)) < 1e-5
        d2 = torch.empty_like(o2)
        ops.sparsemax_bwd(o2, dev(gg), d2)
        assert rel(d2, zr.grad) < 5e-5
        zr2 = zz.clone().requires_grad_(True)
        torch.softmax(zr2, -1).backward(gg)
        ops.softmax_fwd(dev(zz), o2)
        assert rel(o2, torch.softmax(zz, -1)) < 1e-5
        ops.softmax_bwd(o2, dev(gg), d2)
        assert rel(d2, zr2.grad) < 5e-5


def test_l2norm_scale_infonce_topk(golden_dir):
    ops = _ops()
    x = rnd(9, 96, seed=1)
    x[4] = 0          # zero row: x / (0 + eps) = 0, gradient dy / eps
    for eps in (1e-10, 0.0):
        xx = x.clone()
        if eps == 0.0:
            xx[4] = 1.0
        xr = xx.clone().requires_grad_(True)
        y_ref = xr / (xr.norm(dim=-1, keepdim=True) + eps)
        dy = rnd(9, 96, seed=2)
        y_ref.backward(dy)
        y, n = torch.empty(9, 96, device="cuda"), torch.empty(9, device="cuda")
        ops.l2norm_fwd(dev(xx), y, n, eps)
        assert rel(y, y_ref) < 1e-6
        dx = torch.empty_like(y)
        ops.l2norm_bwd(dev(xx), n, dev(dy), dx, eps)
        keep = torch.ones(9, dtype=torch.bool)
        if eps > 0:
            keep[4] = False     # reference gives dy/eps (1e10 scale) there; checked separately
            assert rel(dx[4], dy[4] / eps) < 1e-5
        assert rel(dx[keep.cuda()], xr.grad[keep]) < 1e-5
    # temperature and its gradient, including the clamp regime exp(5) > 100
    for ls in (math.log(1 / 0.07), 5.0):
        p = torch.tensor([ls], requires_grad=True)
        cos_i, cos_t = rnd(6, 12, seed=3), rnd(6, 12, seed=4)
        scale = torch.clamp(p.exp().detach(), max=100) + (p.exp() - p.exp().detach())
        li, lt = cos_i * scale, cos_t * scale
        gi, gt = rnd(6, 12, seed=5), rnd(6, 12, seed=6)
        (li * gi + lt * gt).sum().backward()
        sc = torch.empty(1, device="cuda")
        ops.logit_scale_fwd(dev(p.detach()), sc)
        assert rel(sc, scale.detach()) < 1e-6
        dp = torch.zeros(1, device="cuda")
        ops.logit_scale_bwd(dev(gi), dev(li.detach()), dev(gt), dev(lt.detach()), dev(p.detach()), sc, dp)
        assert rel(dp, p.grad) < 1e-5
    g = np.load(os.path.join(golden_dir, "g3_ops.npz"))
    li, lt = torch.from_numpy(g["ce.li"]), torch.from_numpy(g["ce.lt"])
    loss = torch.empty(1, device="cuda")
    dli, dlt = torch.empty(4, 16, device="cuda"), torch.empty(4, 16, device="cuda")
    ops.infonce_fwd(dev(li), dev(lt), 8, loss, dli, dlt)          # rank 2 of 4, local batch 4
    assert abs(float(loss) - float(g["ce.loss"])) < 1e-5 * abs(float(g["ce.loss"]))
    assert rel(dli, torch.from_numpy(g["ce.dli"])) < 1e-5 and rel(dlt, torch.from_numpy(g["ce.dlt"])) < 1e-5
    lg = rnd(32, 96, seed=7)
    labels = 32 + torch.arange(32)
    want = O.accuracy(lg, labels, topk=(1, 5))
    acc = torch.empty(2, device="cuda")
    ops.topk_accuracy(dev(lg), 32, 5, acc)
    assert abs(float(acc[0]) - float(want[0])) < 1e-4 and abs(float(acc[1]) - float(want[1])) < 1e-4


def test_colsum_cast_scale():
    ops = _ops()
    for dtp in (torch.float32, torch.bfloat16):
        x = rnd(1000, 264, seed=1).to(dtp)
        out = torch.ones(264, device="cuda")
        ops.colsum(dev(x), out)
        assert rel(out, 1 + x.float().sum(0)) < 1e-5
    src = rnd(100003, seed=2)
    for n in (100003, 4096, 5):
        s = dev(src[:n].clone())
        d = torch.empty(n, device="cuda", dtype=torch.bfloat16)
        ops.cast_f32(s, d)
        assert torch.equal(d.cpu(), src[:n].to(torch.bfloat16))
    y = torch.empty(100003, device="cuda")
    ops.scale(dev(src), y, 0.25)
    assert rel(y, src * 0.25) == 0.0


@pytest.mark.parametrize("n,max_norm", [(1, 0.5), (1000, 3.0), (4 * 1024 * 1024 + 37, 10.0), (50000, 1e9)])
def test_clip_grad_norm_over_a_flat_buffer(n, max_norm):
    """ilvlm_sumsq + ilvlm_clip_by_norm = clip_grad_norm_ of the reference (prototype/utils/grad_clip.py:12-47) over one flat
    gradient buffer: the squared norm on the device, the buffer scaled by max_norm / (norm + 1e-6) only when that is below 1"""
    ops = _ops()
    g = rnd(n, seed=3).cuda() * 2.5
    raw = g.clone()
    ss = ops.clip_grad_norm_(g, max_norm)
    tn = torch.linalg.vector_norm(raw.double())
    assert abs(float(ss) - float(tn) ** 2) <= 1e-4 * float(tn) ** 2
    coef = max_norm / (float(tn) + 1e-6)
    if coef < 1:
        assert float((g - raw * coef).abs().max()) <= 1e-5 * float(raw.abs().max())
    else:
        assert torch.equal(g, raw)
    scratch = torch.full((1,), 123.0, device="cuda")             # a caller-provided accumulator is zeroed first
    ss2 = ops.clip_grad_norm_(raw.clone(), max_norm, scratch)
    assert ss2.data_ptr() == scratch.data_ptr() and abs(float(ss2) - float(tn) ** 2) <= 1e-4 * float(tn) ** 2
    # bitwise reproducible (data-parallel replicas clip the same averaged arena each on their own and must stay identical):
    # the same buffer gives the same bits launch after launch, also with other work on the device in between
    first_ss, first_g = None, None
    for it in range(6):
        gi = raw.clone()
        if it % 2:
            torch.empty(8 << 20, device="cuda").normal_()
        si = ops.clip_grad_norm_(gi, max_norm).clone()
        if first_ss is None:
            first_ss, first_g = si, gi
        else:
            assert torch.equal(si, first_ss) and torch.equal(gi, first_g), "clipped arena differs between launches"


def test_adamw_matches_torch():
    import ctypes as C
    from ilvlm_amd import lib as L
    sizes = [5000, 17, 4096, 70000]
    groups = [0, 1, 2, 1]           # group 2 inactive
    lrs, wds = [1e-3, 2e-3, 5e-1], [0.1, 0.0, 0.3]
    n = sum(sizes)
    p0, g0 = rnd(n, seed=1), rnd(n, seed=2)
    P, G = dev(p0), dev(g0)
    M, V = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    shadow = torch.zeros(n, device="cuda", dtype=torch.bfloat16)
    offs, cnts, grps = [], [], []
    o = 0
    for sz, gr in zip(sizes, groups):
        for c in range(0, sz, 4096):
            offs.append(o + c); cnts.append(min(4096, sz - c)); grps.append(gr)
        o += sz
    co = torch.tensor(offs, dtype=torch.int64, device="cuda")
    cc = torch.tensor(cnts, dtype=torch.int32, device="cuda")
    cg = torch.tensor(grps, dtype=torch.int32, device="cuda")
    pr, m, v = p0.clone(), torch.zeros(n), torch.zeros(n)
    for step in (1, 2, 3):
        h = L.AdamWHyper()
        for i in range(3):
            h.lr[i], h.weight_decay[i], h.active[i] = lrs[i], wds[i], int(i != 2)
        h.beta1, h.beta2, h.eps, h.step = 0.9, 0.98, 1e-8, step
        L.check(L.load().ilvlm_adamw_step(P.data_ptr(), G.data_ptr(), M.data_ptr(), V.data_ptr(), shadow.data_ptr(),
                                          co.data_ptr(), cc.data_ptr(), cg.data_ptr(), len(offs), C.byref(h),
                                          torch.cuda.current_stream().cuda_stream), "adamw")
        o = 0
        for sz, gr in zip(sizes, groups):
            if gr != 2:
                sl = slice(o, o + sz)
                O.adamw_step(pr[sl], g0[sl], m[sl], v[sl], step, lrs[gr], 0.9, 0.98, 1e-8, wds[gr])
            o += sz
    assert rel(P, pr) < 1e-6 and rel(M, m) < 1e-6 and rel(V, v) < 1e-6
    act = torch.ones(n, dtype=torch.bool)
    act[sizes[0] + sizes[1]: sizes[0] + sizes[1] + sizes[2]] = False
    assert torch.equal(shadow.cpu()[act], pr.to(torch.bfloat16)[act])
    assert torch.equal(P.cpu()[~act], p0[~act])


def test_adamw_over_packed_weight_tiles_equals_the_chunk_kernel_and_packs():
    """ilvlm_adamw_step_packed: the AdamW update of [out, in] GEMM weights tile by tile (64 x 64) with both fragment-order
    images and the row-major bf16 shadow written from the tile -- parameters, moments and shadow bit for bit those of
    ilvlm_adamw_step on the same inputs, images equal to ilvlm_gemm_pack_b of the shadow; a tile of an inactive group is left
    untouched, images included."""
    import ctypes as C
    from ilvlm_amd import lib as L
    ops = _ops()
    lib = L.load()
    shapes = [(192, 128), (128, 320), (64, 64)]
    offs, total = [], 0
    for r, c in shapes:
        offs.append(total)
        total += r * c + 64          # a gap between tensors
    g = torch.Generator().manual_seed(5)
    P0 = torch.randn(total, generator=g).cuda()
    G = (torch.randn(total, generator=g) * 0.1).cuda()
    M0 = (torch.randn(total, generator=g) * 0.01).cuda()
    V0 = (torch.rand(total, generator=g) * 0.01).cuda()
    h = L.AdamWHyper()
    h.lr[0], h.weight_decay[0], h.active[0] = 1e-2, 0.1, 1
    h.lr[1], h.weight_decay[1], h.active[1] = 3e-3, 0.0, 1
    h.active[2] = 0
    h.beta1, h.beta2, h.eps, h.step = 0.9, 0.98, 1e-8, 3
    groups = [0, 1, 2]               # third weight: inactive group
    # reference: the chunk kernel
    P, M, V = P0.clone(), M0.clone(), V0.clone()
    S = torch.zeros(total, dtype=torch.bfloat16, device="cuda")
    co, cc, cg = [], [], []
    for (r, c), o, gr in zip(shapes, offs, groups):
        for k in range(0, r * c, 4096):
            co.append(o + k); cc.append(min(4096, r * c - k)); cg.append(gr)
    coff = torch.tensor(co, dtype=torch.int64).cuda(); ccnt = torch.tensor(cc, dtype=torch.int32).cuda(); cgrp = torch.tensor(cg, dtype=torch.int32).cuda()
    L.check(lib.ilvlm_adamw_step(P.data_ptr(), G.data_ptr(), M.data_ptr(), V.data_ptr(), S.data_ptr(), coff.data_ptr(), ccnt.data_ptr(),
                                 cgrp.data_ptr(), len(co), C.byref(h), None), "adamw_step")
    # the tile kernel
    P2, M2, V2 = P0.clone(), M0.clone(), V0.clone()
    S2 = torch.zeros(total, dtype=torch.bfloat16, device="cuda")
    fwd = torch.full((total,), 7.0, dtype=torch.bfloat16, device="cuda")
    bwd = torch.full((total,), 7.0, dtype=torch.bfloat16, device="cuda")
    tiles = [(o // 64, r, c, r0, c0, gr) for (r, c), o, gr in zip(shapes, offs, groups) for r0 in range(0, r, 64) for c0 in range(0, c, 64)]
    table = torch.tensor(tiles, dtype=torch.int32).cuda()
    L.check(lib.ilvlm_adamw_step_packed(P2.data_ptr(), G.data_ptr(), M2.data_ptr(), V2.data_ptr(), S2.data_ptr(), fwd.data_ptr(),
                                        bwd.data_ptr(), table.data_ptr(), len(tiles), C.byref(h), None), "adamw_step_packed")
    torch.cuda.synchronize()
    assert torch.equal(P, P2) and torch.equal(M, M2) and torch.equal(V, V2) and torch.equal(S, S2)
    assert not torch.equal(P, P0)
    for (r, c), o, gr in zip(shapes, offs, groups):
        w = S2[o:o + r * c].view(r, c)
        if gr == 2:                  # inactive: nothing written
            assert torch.equal(P2[o:o + r * c], P0[o:o + r * c]) and float((fwd[o:o + r * c] != 7).sum()) == 0
            continue
        assert torch.equal(fwd[o:o + r * c], ops.gemm_pack_b(w))
        assert torch.equal(bwd[o:o + r * c], ops.gemm_pack_b(w, trans_b=True))
        assert float((fwd[o + r * c:o + r * c + 64] != 7).sum()) == 0          # gaps untouched


def test_small_elementwise_kernels():
    ops = _ops()
    x, y = rnd(1000, seed=1), rnd(1000, seed=2)
    Y = dev(y.clone())
    ops.add_inplace(Y, dev(x))
    assert rel(Y, x + y) == 0.0
    a = torch.tensor([0.125], device="cuda")
    assert rel(ops.scale_dev(dev(x), a), x * 0.125) == 0.0
    z = dev(torch.tensor([2.0, 3.5, 7.0]))
    ops.clamp_(z, 3, 6)
    assert z.cpu().tolist() == [3.0, 3.5, 6.0]


# ---------------------------------------------------------------------------------------------------------------------
# packed text rows (include/ilvlm_hip.h): the packed entry points against the dense ones, sequence by sequence
# ---------------------------------------------------------------------------------------------------------------------
PACK_LENS = [[77, 8, 41, 16, 17, 33, 2, 64], [5, 1, 3], [24, 24], [80, 79, 1, 48]]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("lens", PACK_LENS, ids=[str(len(l)) + "seq" for l in PACK_LENS])
@pytest.mark.parametrize("causal", [1, 0])
def test_packed_attention_matches_dense(dtype, lens, causal):
    ops = _ops()
    H, ctx = 2, (80 if max(lens) > 77 else 77)
    B, E = len(lens), 64 * H
    seq = ops.PackedSeq(lens, ctx, "cuda")
    qkv = rnd(seq.rows, 3 * E, seed=3).to(dtype).cuda()
    dout = rnd(seq.rows, E, seed=4).to(dtype).cuda()
    out = torch.full((seq.rows, E), float("nan"), device="cuda", dtype=dtype)
    lse = torch.full((B, H, ctx), float("nan"), device="cuda")
    dqkv = torch.full((seq.rows, 3 * E), float("nan"), device="cuda", dtype=dtype)
    ops.attention_fwd(qkv, out, lse, B, ctx, H, causal, seq)
    ops.attention_bwd(dout, qkv, out, lse, dqkv, B, ctx, H, causal, seq)
    torch.cuda.synchronize()
    assert not torch.isnan(out.float()).any() and not torch.isnan(dqkv.float()).any()     # every packed row written
    off = 0
    for b, n in enumerate(lens):
        q1 = qkv[off:off + n].contiguous()
        o1 = torch.empty(n, E, device="cuda", dtype=dtype)
        l1 = torch.empty(1, H, n, device="cuda")
        d1 = torch.empty(n, 3 * E, device="cuda", dtype=dtype)
        ops.attention_fwd(q1, o1, l1, 1, n, H, causal)
        ops.attention_bwd(dout[off:off + n].contiguous(), q1, o1, l1, d1, 1, n, H, causal)
        tol = 1e-6 if dtype == torch.float32 else 1e-2
        assert rel(out[off:off + n], o1.float().cpu()) < tol
        assert rel(lse[b, :, :n], l1[0].cpu()) < 1e-6
        assert rel(dqkv[off:off + n], d1.float().cpu()) < (1e-5 if dtype == torch.float32 else 2e-2)
        off += n


def test_packed_embedding_pooling_and_row_gather():
    ops = _ops()
    lens, ctx, W, V, Cn = [7, 24, 1, 13, 24], 24, 64, 500, 300
    B = len(lens)
    seq = ops.PackedSeq(lens, ctx, "cuda")
    gen = torch.Generator().manual_seed(5)
    tok = torch.randint(1, V, (B, ctx), generator=gen)
    mask = torch.full((B, ctx), float("-inf"))
    for b, n in enumerate(lens):
        tok[b, n:] = 0
        mask[b, :n] = 0
    rows = torch.cat([torch.arange(n) + b * ctx for b, n in enumerate(lens)])
    table, pos = rnd(V, W, seed=1), rnd(ctx, W, seed=2)
    xd = torch.empty(B * ctx, W, device="cuda"); xp = torch.full((seq.rows, W), float("nan"), device="cuda")
    ops.embed_fwd(dev(tok), dev(table), dev(pos), xd)
    ops.embed_fwd(dev(tok), dev(table), dev(pos), xp, seq)
    assert torch.equal(xp.cpu(), xd.cpu()[rows])
    # backward: dense gradient is zero on the padded rows, as in the step
    dxd = rnd(B * ctx, W, seed=3)
    keep = torch.zeros(B * ctx, 1); keep[rows] = 1
    dxd = dxd * keep
    dtd, dpd = torch.zeros(V, W, device="cuda"), torch.zeros(ctx, W, device="cuda")
    dtp, dpp = torch.zeros(V, W, device="cuda"), torch.zeros(ctx, W, device="cuda")
    ops.embed_bwd(dev(tok), dev(dxd), dtd, dpd)
    ops.embed_bwd(dev(tok), dev(dxd[rows].contiguous()), dtp, dpp, seq)
    assert rel(dtp, dtd.cpu()) < 1e-6 and rel(dpp, dpd.cpu()) < 1e-6
    # EOT-style row gather / scatter
    idx = torch.tensor([n - 1 for n in lens])
    yd, yp = torch.empty(B, W, device="cuda"), torch.empty(B, W, device="cuda")
    ops.gather_rows(xd, dev(idx), yd, B, ctx, W)
    ops.gather_rows(xp, dev(idx), yp, B, ctx, W, seq)
    assert torch.equal(yd.cpu(), yp.cpu())
    sd_, sp_ = torch.zeros(B * ctx, W, device="cuda"), torch.zeros(seq.rows, W, device="cuda")
    ops.scatter_rows(yd, dev(idx), sd_, B, ctx, W)
    ops.scatter_rows(yd, dev(idx), sp_, B, ctx, W, seq)
    assert torch.equal(sd_.cpu()[rows], sp_.cpu())
    # FDT pooling: scores of the padded positions are arbitrary in the dense form (they are multiplied by zero)
    sc = rnd(B * ctx, Cn, seed=7)
    sc[:, :40] -= 3.0          # some codes negative everywhere: the zero of a masked position wins the maximum
    for pool in (0, 1, 2):
        pd, pp = torch.empty(B, Cn, device="cuda"), torch.empty(B, Cn, device="cuda")
        ad = torch.empty(B, Cn, device="cuda", dtype=torch.int32); ap = torch.empty_like(ad)
        ops.fdt_pool_fwd(dev(sc), dev(mask), pd, ad if pool == 0 else None, B, ctx, Cn, 8.0, 2.0, pool)
        ops.fdt_pool_fwd(dev(sc[rows].contiguous()), None, pp, ap if pool == 0 else None, B, ctx, Cn, 8.0, 2.0, pool, seq)
        assert torch.equal(pd.cpu(), pp.cpu())
        if pool == 0:
            assert torch.equal(ad.cpu(), ap.cpu())
        dp = rnd(B, Cn, seed=9)
        dsd = torch.empty(B * ctx, Cn, device="cuda"); dsp = torch.full((seq.rows, Cn), float("nan"), device="cuda")
        ops.fdt_pool_bwd(dev(dp), ad if pool == 0 else None, dev(mask), dsd, B, ctx, Cn, 8.0, 2.0, pool)
        ops.fdt_pool_bwd(dev(dp), ap if pool == 0 else None, None, dsp, B, ctx, Cn, 8.0, 2.0, pool, seq)
        assert torch.equal(dsd.cpu()[rows], dsp.cpu())
        assert float(dsd.cpu()[keep[:, 0] == 0].abs().max()) == 0.0


@pytest.mark.parametrize("K", [10873, 77, 4097])
def test_weight_gradient_gemm_with_ragged_reduction(K):
    """both operands K-strided: the reduction length need not be a multiple of the K-tile (packed text rows)"""
    ops = _ops()
    M, N = 256, 192
    a, b = rnd(K, M, seed=1).to(torch.bfloat16), rnd(K, N, seed=2).to(torch.bfloat16)
    # neighbours in memory that must NOT leak into the sum: allocate the operands inside larger poisoned buffers
    abuf = torch.full((K + 70, M), float("nan"), dtype=torch.bfloat16, device="cuda"); abuf[:K] = a.cuda()
    bbuf = torch.full((K + 70, N), float("nan"), dtype=torch.bfloat16, device="cuda"); bbuf[:K] = b.cuda()
    out = torch.zeros(M, N, device="cuda")
    bias = torch.zeros(M, device="cuda")
    ops.gemm(abuf[:K], bbuf[:K], out, trans_a=True, trans_b=True, accumulate=True, split_k=5, a_rowsum=bias)
    ref = a.float().t() @ b.float()
    assert rel(out, ref) < 2e-3
    assert rel(bias, a.float().sum(0)) < 2e-3


@pytest.mark.parametrize("variant", [5, 16, 18, 1696, 1664])
@pytest.mark.parametrize("M,N,K,tb", [(19712, 2048, 512, 0), (12800, 3072, 768, 0), (12800, 768, 3072, 1), (11319, 1536, 512, 0)])
def test_forward_gemm_is_deterministic_and_right_at_full_size(variant, M, N, K, tb):
    """Chip-filling launches of the step's shapes, repeated: every launch must reproduce the first bit for bit (no atomics in
    these kernels) and match an fp32 reference.  Guards the LDS race fixed at ILVLM_WG_BARRIER (csrc/gemm.hip): fragment reads
    still queued at the end-of-K-tile barrier were overtaken by the next tile's DMA in about one launch out of ten -- only at
    sizes where four workgroups per CU keep the LDS pipeline busy, never at unit-test sizes.  Variant 16 runs the streaming
    kernel (packed B operand, hand-placed waits around inline-asm loads): same guard for its two-stage ring, and its result
    must equal the direct-to-LDS kernel's bit for bit (same MFMA sequence per output element)."""
    ops = _ops()
    a = rnd(M, K, seed=1).to(torch.bfloat16).cuda()
    w = (rnd(K, N, seed=2) if tb else rnd(N, K, seed=2)).to(torch.bfloat16).cuda()
    ref = a.float() @ (w.float() if tb else w.float().t())
    wp = ops.gemm_pack_b(w, trans_b=bool(tb)) if variant >= 16 else None       # 18: the persistent streaming kernel (2 workgroups per CU walk 1.4 .. 4.8 tiles each)
    rows = -1
    if variant > 100:                # 1696 / 1664: the streaming kernel with 96- / 64-row tiles
        variant, rows = 16, variant % 100
    try:
        ops.gemm_set_tile_rows(rows)
        ops.gemm_set_variant(5)
        base = torch.empty(M, N, device="cuda", dtype=torch.float32)
        ops.gemm(a, w, base, trans_b=bool(tb))
        ops.gemm_set_variant(variant)
        first = None
        for it in range(25):
            out = torch.empty(M, N, device="cuda", dtype=torch.float32)
            ops.gemm(a, w, out, trans_b=bool(tb), b_packed=wp)
            if first is None:
                first = out
                assert float((out - ref).abs().max()) < 2e-3 * float(ref.abs().max())
                assert torch.equal(out, base), "streaming kernel differs from the direct-to-LDS kernel"
            else:
                assert torch.equal(out, first), "launch %d differs from launch 0" % it
    finally:
        ops.gemm_set_variant(15)
        ops.gemm_set_tile_rows(-1)


@pytest.mark.parametrize("M,N,K,tb", [(12800, 768, 3072, 0), (12800, 768, 2304, 1), (11319, 512, 2048, 0), (11319, 512, 1536, 1),
                                      (1000, 256, 768, 0), (300, 144, 1536, 1)])
def test_streaming_gemm_store_type_split_k(M, N, K, tb):
    """deep-K / narrow-N store-type products (down-projection forward, up- and in-projection input gradients) split K over
    2-4 workgroups per tile when the caller offers a slab workspace: all but the last arriver publish their partial tile, the
    last one adds the slabs in slice order and runs the epilogue (opt-in: measured slower than the unsplit kernel on the
    step's shapes; selector 17 forces it here).  Equal to the fp32 reference through the bias / residual
    epilogue, bit-identical from launch to launch (no atomics on C, a sum that does not depend on arrival order), counters left
    at zero for the next launch, and the workspace untouched when none is offered."""
    ops = _ops()
    a = rnd(M, K, seed=1).to(torch.bfloat16).cuda()
    w = (rnd(K, N, seed=2) if tb else rnd(N, K, seed=2)).to(torch.bfloat16).cuda()
    wp = ops.gemm_pack_b(w, trans_b=bool(tb))
    ref = a.float() @ (w.float() if tb else w.float().t())
    bias, res = rnd(N, seed=3).cuda(), rnd(M, N, seed=4).cuda()
    want = ref + bias + res
    slab = (torch.empty(160 << 20, dtype=torch.uint8, device="cuda"), torch.zeros(8192, dtype=torch.int32, device="cuda"))
    slab[0].fill_(0xff)                       # NaN patterns: a slab read before it was written would poison the result
    try:
        ops.gemm_set_variant(16)
        base = torch.full((M, N), float("nan"), device="cuda")
        ops.gemm(a, w, base, trans_b=bool(tb), bias=bias, residual=res, b_packed=wp, slab=slab)     # 16: never splits
        assert int((slab[0][:4 << 20] != 0xff).sum()) == 0
        ops.gemm_set_variant(17)                                                                     # 17: splits where offered
        assert float((base - want).abs().max()) < 2e-5 * float(want.abs().max())
        first = None
        for it in range(8):
            out = torch.full((M, N), float("nan"), device="cuda")
            ops.gemm(a, w, out, trans_b=bool(tb), bias=bias, residual=res, b_packed=wp, slab=slab)
            assert float((out - want).abs().max()) < 2e-5 * float(want.abs().max()), "launch %d" % it
            if first is None:
                first = out
            else:
                assert torch.equal(out, first), "launch %d differs from launch 0" % it
        assert int(slab[1].abs().sum()) == 0
        # the split really happened: slabs were written somewhere in the first tiles' slots (WHICH K-slice of a tile publishes
        # its slab depends on who arrives last, so the very first bytes may stay untouched) ...
        assert int((slab[0][:4 << 20] != 0xff).sum()) > 0
        # ... and changes the summation order: equal to the unsplit kernel to fp32 rounding, not bit for bit
        assert float((first - base).abs().max()) < 1e-5 * float(want.abs().max())
    finally:
        ops.gemm_set_variant(15)


def _unpack_b(packed, n, k):
    """inverse of the fragment order of ilvlm_gemm_pack_b (include/ilvlm_hip.h): [n, k] from the packed image"""
    p = packed.view(n // 16, k // 32, 4, 16, 8)           # block (n/16, k/32), lane = 16 * (l >> 4) + (l & 15), 8 elements
    return p.permute(0, 3, 1, 2, 4).reshape(n, k)


@pytest.mark.parametrize("N,K", [(16, 32), (64, 64), (768, 3072), (2304, 768), (80, 96)])
def test_gemm_pack_b_layout(N, K):
    """the packed B operand is exactly the documented permutation of B (both storage orders)"""
    ops = _ops()
    b = rnd(N, K, seed=5).to(torch.bfloat16)
    for tb in (False, True):
        src = dev(b.t()) if tb else dev(b)
        packed = ops.gemm_pack_b(src, trans_b=tb).cpu()
        assert torch.equal(_unpack_b(packed, N, K), b), "trans_b=%s" % tb


def test_pack_weights_table_equals_single_matrix_packs():
    """ilvlm_pack_weights (all GEMM weights of an arena in one launch) = ilvlm_gemm_pack_b per weight, both images"""
    ops = _ops()
    shapes = [(192, 64), (64, 256), (128, 128)]
    offs, total = [], 0
    for r, c in shapes:
        offs.append(total)
        total += r * c + 64                     # gaps between the weights (other parameters live there)
    arena = rnd(total, seed=9).to(torch.bfloat16).cuda()
    table = [(o // 64, r, c, r0, c0) for o, (r, c) in zip(offs, shapes) for r0 in range(0, r, 64) for c0 in range(0, c, 64)]
    fwd = torch.zeros(total, dtype=torch.bfloat16, device="cuda")
    bwd = torch.zeros(total, dtype=torch.bfloat16, device="cuda")
    ops.pack_weights(arena, fwd, bwd, torch.tensor(table, dtype=torch.int32).cuda())
    for o, (r, c) in zip(offs, shapes):
        w = arena[o:o + r * c].view(r, c)
        assert torch.equal(fwd[o:o + r * c], ops.gemm_pack_b(w))                      # Bop[n][k] = W[n][k]
        assert torch.equal(bwd[o:o + r * c], ops.gemm_pack_b(w, trans_b=True))        # Bop[n'][k'] = W[k'][n']
        assert float(fwd[o + r * c:o + r * c + 64].abs().sum()) == 0.0               # gaps untouched


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (1000, 768, 512), (640, 400, 256), (136, 2304, 768), (8, 16, 128), (512, 256, 192),
                                   (776, 528, 320), (300, 64, 64), (129, 144, 448)])
@pytest.mark.parametrize("tb", [0, 1])
@pytest.mark.parametrize("rows", [128, 96, 64])
def test_streaming_gemm_equals_direct_to_lds_kernel(M, N, K, tb, rows, monkeypatch):
    """gemm_bf16_pk_kernel (A through a two-stage LDS ring, B from the packed copy straight into registers): odd and even
    K-tile counts, ragged row tiles, column counts that are not multiples of the 128-column tile, every store epilogue --
    against an fp32 reference AND bit for bit against the direct-to-LDS kernel"""
    ops = _ops()
    a = rnd(M, K, seed=1).to(torch.bfloat16).cuda()
    w = (rnd(K, N, seed=2) if tb else rnd(N, K, seed=2)).to(torch.bfloat16).cuda()
    wp = ops.gemm_pack_b(w, trans_b=bool(tb))
    ref = a.float() @ (w.float() if tb else w.float().t())
    bias, res = rnd(N, seed=3).cuda(), rnd(M, N, seed=4).cuda()
    pre = rnd(M, N, seed=6).to(torch.bfloat16).cuda()

    def run(packed, **kw):
        outs = []
        out = torch.full((M, N), float("nan"), device="cuda")
        ops.gemm(a, w, out, trans_b=bool(tb), b_packed=packed, **kw)
        outs.append(out)
        outb = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        ops.gemm(a, w, outb, trans_b=bool(tb), b_packed=packed, bias=bias)
        outs.append(outb)
        outr = torch.full((M, N), float("nan"), device="cuda")
        ops.gemm(a, w, outr, trans_b=bool(tb), b_packed=packed, bias=bias, residual=res)
        outs.append(outr)
        aux = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        outg = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        ops.gemm(a, w, outg, trans_b=bool(tb), b_packed=packed, bias=bias, aux=aux, act=1)          # QuickGELU forward
        outs += [outg, aux]
        outd = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        ops.gemm(a, w, outd, trans_b=bool(tb), b_packed=packed, aux=pre, act=3)                     # QuickGELU backward
        outs.append(outd)
        return outs

    try:
        ops.gemm_set_variant(16)         # the streaming kernel for every eligible shape (15 keeps short K-loops on the other kernel)
        ops.gemm_set_tile_rows(rows)     # every tile height of the kernel (round 4: 96- and 64-row tiles beside the 128-row one)
        got = run(wp)
        ops.gemm_set_variant(5)
        base = run(None)
    finally:
        ops.gemm_set_variant(15)
        ops.gemm_set_tile_rows(-1)
    assert rel(got[0], ref) < 2e-5
    assert rel(got[2], ref + bias + res) < 2e-5
    for i, (g, b) in enumerate(zip(got, base)):
        assert torch.equal(g, b), "epilogue case %d differs from the direct-to-LDS kernel" % i


@pytest.mark.parametrize("epi_sep", [1, 0, 2])
@pytest.mark.parametrize("slots", [8, 24, 0])
@pytest.mark.parametrize("M,N,K,tb", [(1000, 768, 512, 0), (640, 400, 256, 1), (136, 2304, 768, 0), (776, 528, 128, 1), (2000, 1024, 384, 0),
                                      (129, 144, 1536, 1), (3000, 512, 2048, 0)])
def test_persistent_streaming_gemm_equals_direct_to_lds_kernel(M, N, K, tb, slots, epi_sep):
    """gemm_bf16_pkp_kernel: a workgroup walks several tiles, the operand stream (three-stage A ring, two B register sets) runs on
    across tile boundaries and the epilogue of a tile runs under the next tile's first loads.  With 8 / 24 workgroups every
    problem here has 2 .. 48 tiles per workgroup (and workgroups with one tile fewer than others); 0 = two per CU, one tile each.
    Ragged row tiles, widths that are not multiples of the 256- / 128-column tile, 2 .. 32 K-tiles, both places where the epilogue
    may transpose; every store epilogue bit for bit against the direct-to-LDS kernel and against an fp32 reference."""
    ops = _ops()
    a = rnd(M, K, seed=1).to(torch.bfloat16).cuda()
    w = (rnd(K, N, seed=2) if tb else rnd(N, K, seed=2)).to(torch.bfloat16).cuda()
    wp = ops.gemm_pack_b(w, trans_b=bool(tb))
    ref = a.float() @ (w.float() if tb else w.float().t())
    bias, res = rnd(N, seed=3).cuda(), rnd(M, N, seed=4).cuda()
    pre = rnd(M, N, seed=6).to(torch.bfloat16).cuda()

    def run(packed):
        outs = []
        out = torch.full((M, N), float("nan"), device="cuda")
        ops.gemm(a, w, out, trans_b=bool(tb), b_packed=packed)
        outs.append(out)
        outb = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        ops.gemm(a, w, outb, trans_b=bool(tb), b_packed=packed, bias=bias)
        outs.append(outb)
        outr = torch.full((M, N), float("nan"), device="cuda")
        ops.gemm(a, w, outr, trans_b=bool(tb), b_packed=packed, bias=bias, residual=res)
        outs.append(outr)
        aux = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        outg = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        ops.gemm(a, w, outg, trans_b=bool(tb), b_packed=packed, bias=bias, aux=aux, act=1)          # QuickGELU forward
        outs += [outg, aux]
        outd = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        ops.gemm(a, w, outd, trans_b=bool(tb), b_packed=packed, aux=pre, act=3)                     # QuickGELU backward
        outs.append(outd)
        return outs

    try:
        ops.gemm_set_persistent(slots, epi_sep, 700 if slots == 24 else 0)      # a start stagger changes timing, never results
        ops.gemm_set_variant(18)
        got = run(wp)
        again = run(wp)
        ops.gemm_set_variant(5)
        base = run(None)
    finally:
        ops.gemm_set_variant(15)
        ops.gemm_set_persistent(0, -1, -1)
    assert rel(got[0], ref) < 2e-5
    assert rel(got[2], ref + bias + res) < 2e-5
    for i, (g, b, g2) in enumerate(zip(got, base, again)):
        assert torch.equal(g, b), "epilogue case %d differs from the direct-to-LDS kernel" % i
        assert torch.equal(g, g2), "epilogue case %d differs between two launches" % i


@pytest.mark.parametrize("M,N,K,split", [(2048, 512, 19712, 6), (3072, 768, 12800, 3), (512, 512, 11319, 16)])
def test_weight_gradient_gemm_at_full_size(M, N, K, split):
    """split-K weight-gradient launches of the step's shapes, repeated: fp32 atomics make the sum order vary, the result may
    not (same race guard as above for the accumulating form of the kernel, incl. a ragged reduction length)."""
    ops = _ops()
    a, b = rnd(K, M, seed=1).to(torch.bfloat16).cuda(), rnd(K, N, seed=2).to(torch.bfloat16).cuda()
    ref = a.float().t() @ b.float()
    rs_ref = a.float().sum(0)
    scale = float(ref.abs().max())
    for it in range(12):
        out = torch.zeros(M, N, device="cuda")
        rs = torch.zeros(M, device="cuda")
        ops.gemm(a, b, out, trans_a=True, trans_b=True, accumulate=True, split_k=split, a_rowsum=rs)
        assert float((out - ref).abs().max()) < 1e-3 * scale, "launch %d" % it
        assert float((rs - rs_ref).abs().max()) < 1e-3 * float(rs_ref.abs().max())


@pytest.mark.parametrize("tile", [256, 257])
@pytest.mark.parametrize("M,N,K,split", [(3072, 768, 12800, 3), (768, 3072, 12800, 1), (2304, 768, 12800, 4), (2048, 512, 11319, 6),
                                         (256, 128, 256, 1), (512, 200, 1000, 5), (1536, 512, 11319, 1)])
def test_weight_gradient_on_256_row_tiles_equals_the_128_row_tiles(M, N, K, split, tile):
    """the 256 x 128 workgroup tile of the weight-gradient kernel (128 x 64 per wave; two stages / one stage): same K-slices and
    the same K-tile order per output element as the 128 x 128 tile -- bit-identical at one K-slice and through the slab
    workspace, within the atomics' reordering otherwise; ragged reduction lengths and N; row sums (bias gradient) included"""
    ops = _ops()
    a, b = rnd(K, M, seed=1).to(torch.bfloat16).cuda(), rnd(K, N, seed=2).to(torch.bfloat16).cuda()
    ref = a.float().t() @ b.float()
    rs_ref = a.float().sum(0)
    scale = float(ref.abs().max())
    slab = (torch.empty(64 << 20, dtype=torch.uint8, device="cuda"), torch.zeros(4096, dtype=torch.int32, device="cuda"))
    slab[0].fill_(0xff)
    base = rnd(M, N, seed=5).cuda() * scale
    res = {}
    try:
        for t in (128, tile):
            ops.gemm_set_wgrad_tile(t)
            for use_slab in (False, True):
                out, rs = base.clone(), torch.zeros(M, device="cuda")
                ops.gemm(a, b, out, trans_a=True, trans_b=True, accumulate=True, split_k=split, a_rowsum=rs, slab=slab if use_slab else None)
                assert float((out - base - ref).abs().max()) < 1e-3 * scale
                assert float((rs - rs_ref).abs().max()) < 1e-3 * float(rs_ref.abs().max())
                res[(t, use_slab)] = out
        # the regime hint (several GEMM streams in flight: what the engine declares for its concurrent towers) selects the
        # single-stage wide tile by itself; without it the default stays the 128 x 128 tile
        if tile == 257:
            ops.gemm_set_wgrad_tile(-1)
            for hint in (True, False):
                ops.gemm_set_concurrent(hint)
                out = base.clone()
                ops.gemm(a, b, out, trans_a=True, trans_b=True, accumulate=True, split_k=split, slab=slab)
                assert torch.equal(out, res[(257 if hint else 128, True)])
    finally:
        ops.gemm_set_wgrad_tile(-1)
        ops.gemm_set_concurrent(False)
    assert int(slab[1].abs().sum()) == 0
    if split == 1:
        assert torch.equal(res[(128, False)], res[(tile, False)])
    assert torch.equal(res[(128, True)], res[(tile, True)])


@pytest.mark.parametrize("M,N,K,split", [(3072, 768, 12800, 3), (768, 3072, 12800, 3), (2304, 768, 12800, 4), (2048, 512, 11319, 6),
                                         (304, 200, 1000, 5), (128, 128, 256, 4), (1536, 512, 11319, 8)])
def test_slab_split_k_weight_gradient_is_right_and_bit_reproducible(M, N, K, split):
    """split-K through slab workspaces (every K-slice stores its tile, the last arriver adds the slabs in slice order and
    alone writes C): equal to the reference like the atomic form, bit-identical from launch to launch -- which the atomic
    form is not --, accumulates into what C already holds, and leaves the ticket counters zero for the next launch (the
    same workspace serves all 14 launches, of two shapes)."""
    ops = _ops()
    a, b = rnd(K, M, seed=1).to(torch.bfloat16).cuda(), rnd(K, N, seed=2).to(torch.bfloat16).cuda()
    ref = a.float().t() @ b.float()
    rs_ref = a.float().sum(0)
    scale = float(ref.abs().max())
    slab = (torch.empty(48 << 20, dtype=torch.uint8, device="cuda"), torch.zeros(4096, dtype=torch.int32, device="cuda"))
    slab[0].fill_(0xff)                                  # NaN patterns: an unwritten slab word would show
    a2, b2 = rnd(512, 256, seed=3).to(torch.bfloat16).cuda(), rnd(512, 384, seed=4).to(torch.bfloat16).cuda()
    base = rnd(M, N, seed=5).cuda() * scale
    first = None
    for it in range(12):
        out = base.clone()
        rs = torch.zeros(M, device="cuda")
        ops.gemm(a, b, out, trans_a=True, trans_b=True, accumulate=True, split_k=split, a_rowsum=rs if M % 8 == 0 else None, slab=slab)
        assert float((out - base - ref).abs().max()) < 1e-3 * scale, "launch %d" % it
        if M % 8 == 0:
            assert float((rs - rs_ref).abs().max()) < 1e-3 * float(rs_ref.abs().max())
        if first is None:
            first = out
        else:
            assert torch.equal(out, first), "launch %d differs from the first" % it
        if it % 5 == 4:                                  # another shape through the same workspace in between
            o2 = torch.zeros(256, 384, device="cuda")
            ops.gemm(a2, b2, o2, trans_a=True, trans_b=True, accumulate=True, split_k=2, slab=slab)
            assert float((o2 - a2.float().t() @ b2.float()).abs().max()) < 1e-3 * float((a2.float().t() @ b2.float()).abs().max())
    assert int(slab[1].abs().sum()) == 0                 # every ticket counter is back at zero


@pytest.mark.parametrize("rows,E,target", [(12800, 768, 512), (11319, 512, 512), (11319, 512, 2000), (1000, 192, 512), (77, 64, 512)])
def test_grouped_weight_gradients(rows, E, target):
    """the four weight gradients of a block as ONE launch (ilvlm_wgrad_group): every product and bias gradient equal to the
    fp32 reference, accumulating into what the gradient slots already hold.  At one K-slice (the ViT-B/32 block: 432 tiles)
    every tile has a single writer: plain load-add-store instead of atomics, so repeated launches are bit-identical and equal,
    bit for bit, to the single launches at split_k = 1 (same MFMA sequence); with K-slices (the text block) atomics again."""
    ops = _ops()
    dims = ((3 * E, E), (E, E), (4 * E, E), (E, 4 * E))
    prob, refs = [], []
    for i, (n, k) in enumerate(dims):
        dy = rnd(rows, n, seed=10 + i).to(torch.bfloat16).cuda()
        x = rnd(rows, k, seed=20 + i).to(torch.bfloat16).cuda()
        base = rnd(n, k, seed=30 + i).cuda() * 50
        bb = rnd(n, seed=40 + i).cuda() * 50
        prob.append((dy, x, base.clone(), bb.clone()))
        refs.append((base, bb, dy.float().t() @ x.float(), dy.float().sum(0)))
    ops.wgrad_group(prob, rows, target=target)
    for (dy, x, gw, gb), (base, bb, ref, rs) in zip(prob, refs):
        assert float((gw - base - ref).abs().max()) < 1e-3 * float(ref.abs().max())
        assert float((gb - bb - rs).abs().max()) < 1e-3 * float(rs.abs().max())
    tiles = sum(-(-n // 128) * -(-k // 128) for n, k in dims)
    assert ops.wgrad_group_split(432, 12800, 512) == 1 and ops.wgrad_group_split(192, 11319, 512) == 2
    assert ops.wgrad_group_split(768, 32896, 512) == 2
    if ops.wgrad_group_split(tiles, rows, target) == 1:    # single-writer form
        again = [(dy, x, base.clone(), bb.clone()) for (dy, x, _, _), (base, bb, _, _) in zip(prob, refs)]
        ops.wgrad_group(again, rows, target=target)
        for (_, _, gw, _), (_, _, gw2, _) in zip(prob, again):
            assert torch.equal(gw, gw2)
        for (dy, x, gw, _), (base, _, _, _) in zip(prob, refs):
            one = base.clone()
            ops.gemm(dy, x, one, trans_a=True, trans_b=True, accumulate=True, split_k=1)
            assert torch.equal(gw, one)
    with pytest.raises(RuntimeError):
        ops.wgrad_group(prob + prob[:1], rows)             # more than ILVLM_WGRAD_GROUP_MAX problems


@pytest.mark.parametrize("B,L,H,causal,packed", [(256, 77, 8, 1, True), (256, 50, 12, 0, False), (64, 257, 16, 0, False)])
def test_attention_at_full_size_is_deterministic_and_right(B, L, H, causal, packed):
    """the step's attention launches at full size: repeated launches bit-identical (no atomics), a sample of sequences equal to
    the single-sequence launch, outputs finite"""
    ops = _ops()
    E = 64 * H
    gen = torch.Generator().manual_seed(7)
    lens = torch.randint(8, L + 1, (B,), generator=gen).tolist() if packed else [L] * B
    seq = ops.PackedSeq(lens, L, "cuda") if packed else None
    rows = sum(lens)
    qkv = rnd(rows, 3 * E, seed=3).to(torch.bfloat16).cuda()
    dout = rnd(rows, E, seed=4).to(torch.bfloat16).cuda()
    first = None
    for it in range(6):
        out = torch.full((rows, E), float("nan"), device="cuda", dtype=torch.bfloat16)
        lse = torch.zeros(B, H, L, device="cuda")
        dqkv = torch.full((rows, 3 * E), float("nan"), device="cuda", dtype=torch.bfloat16)
        ops.attention_fwd(qkv, out, lse, B, L, H, causal, seq)
        ops.attention_bwd(dout, qkv, out, lse, dqkv, B, L, H, causal, seq)
        if first is None:
            first = (out, dqkv)
            assert torch.isfinite(out.float()).all() and torch.isfinite(dqkv.float()).all()
        else:
            assert torch.equal(out, first[0]) and torch.equal(dqkv, first[1]), "launch %d differs" % it
    offs = [0]
    for n in lens:
        offs.append(offs[-1] + n)
    for b in (0, 1, B // 2, B - 1):
        n, o = lens[b], offs[b]
        q1 = qkv[o:o + n].contiguous()
        o1 = torch.empty(n, E, device="cuda", dtype=torch.bfloat16); l1 = torch.empty(1, H, n, device="cuda")
        d1 = torch.empty(n, 3 * E, device="cuda", dtype=torch.bfloat16)
        ops.attention_fwd(q1, o1, l1, 1, n, H, causal)
        ops.attention_bwd(dout[o:o + n].contiguous(), q1, o1, l1, d1, 1, n, H, causal)
        assert rel(first[0][o:o + n], o1.float().cpu()) < 1e-2 and rel(first[1][o:o + n], d1.float().cpu()) < 2e-2


@pytest.mark.parametrize("B,T,C,d,packed", [(5, 49, 320, 64, False), (7, 24, 256, 128, True), (64, 49, 4096, 512, False),
                                             (48, 77, 4096, 512, True)])
def test_fused_fdt_score_pool_equals_scores_then_pool(B, T, C, d, packed):
    """the GEMM epilogue that max-pools the codebook scores over the tokens of each sequence (no [rows, C] score matrix)
    against the two-kernel form: same pooled values bit for bit (same MFMA sums, same two scalings), same argmax; packed
    rows with captions shorter than the context (their masked positions contribute exactly 0, argmax = length)"""
    ops = _ops()
    gen = torch.Generator().manual_seed(5)
    lens = torch.randint(3, T + 1, (B,), generator=gen).tolist() if packed else [T] * B
    if packed:
        lens[0], lens[1] = T, 3
    seq = ops.PackedSeq(lens, T, "cuda") if packed else None
    rows = sum(lens)
    q = (rnd(rows, d, seed=1) - (0.3 if packed else 0.0)).to(torch.bfloat16).cuda()
    sd = rnd(C, d, seed=2).to(torch.bfloat16).cuda()
    sqrt_d, temp = math.sqrt(d), 1000.0
    scores = torch.empty(rows, C, device="cuda")
    ops.gemm(q, sd, scores)
    want_p = torch.empty(B, C, device="cuda"); want_a = torch.empty(B, C, device="cuda", dtype=torch.int32)
    ops.fdt_pool_fwd(scores, None, want_p, want_a, B, T, C, sqrt_d, temp, 0, seq)
    got_p = torch.full((B, C), float("nan"), device="cuda"); got_a = torch.full((B, C), -1, device="cuda", dtype=torch.int32)
    ops.fdt_score_pool_fwd(q, sd, got_p, got_a, B, T, sqrt_d, temp, seq)
    torch.cuda.synchronize()
    assert torch.equal(got_p, want_p)
    same = got_a == want_a
    if not bool(same.all()):      # a tie between two tokens of a sequence (equal fp32 scores): either index is a maximum
        offs = torch.tensor([0] + list(np.cumsum(lens)), device="cuda")
        b_idx, c_idx = torch.nonzero(~same, as_tuple=True)
        for b, c_ in zip(b_idx.tolist()[:50], c_idx.tolist()[:50]):
            assert float(scores[offs[b] + got_a[b, c_], c_]) == float(scores[offs[b] + want_a[b, c_], c_])
    if packed:
        assert bool((got_a[1] <= 3).all()) and bool((got_p[1] >= 0).all())      # the 3-token caption: zeros can win

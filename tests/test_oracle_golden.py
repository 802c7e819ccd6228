"""Pin the CPU oracle (oracle/clip_oracle.py) against fixtures produced by the unmodified
reference (tests/golden/make_golden.py).  Tolerance: 1e-5 relative (fp32 vs fp32)."""
import json
import os

import numpy as np
import pytest
import torch

from configs import CFG, FDT_VARIANTS, variant_key, oracle_cfg, state_shapes, VITB32
from detfill import det_state, det_param, det_images, det_tokens, probe
from oracle import clip_oracle as O

SEED = 11
RTOL = 1e-5


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def params(c, fdt, logit_scale=None, seed=SEED):
    st = det_state(state_shapes(c, fdt=fdt), seed, logit_scale)
    p = {k: torch.from_numpy(v).requires_grad_(True) for k, v in st.items()}
    p["visual.conv1.weight"].requires_grad_(False)      # frozen in train() (visual_transformer.py:40-52)
    return p


def close(a, b, rtol=RTOL, atol=None, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    tol = rtol * scale if atol is None else atol
    err = np.abs(a - b).max()
    assert err <= tol, "%s: max err %.3e > tol %.3e (scale %.3e)" % (what, err, tol, scale)


def check_grad_probes(g, prefix, p, rtol=2e-5):
    seen = 0
    for name, t in p.items():
        if (prefix + "gradnone." + name) in g:
            assert t.grad is None or float(t.grad.abs().max()) == 0.0, name
            seen += 1
            continue
        key = prefix + "grad." + name
        if key not in g:
            continue
        assert t.grad is not None, name
        want = g[key]
        got = probe(name, t.grad.numpy())
        scale = max(np.abs(want[2:]).max(), np.abs(want[1]) / t.numel(), 1e-30)
        # 1e-8 floor: fp32 cancellation noise of near-uniform attention (T=1000) gradients
        # (logit_scale: a sum of B*B*2 O(1) terms that cancel when all logits tie, noise ~1e-6)
        floor = 2e-6 if name == "logit_scale" else 1e-8
        assert np.abs(got[2:] - want[2:]).max() <= rtol * scale * 4 + floor, "grad %s" % name
        assert abs(got[1] - want[1]) <= 1e-4 * max(want[1], 1e-30) + floor * t.numel(), "grad L1 %s" % name
        seen += 1
    assert seen >= len(p) - 1


@pytest.mark.parametrize("ck", list(CFG))
def test_g1_towers(golden_dir, ck):
    g = load(golden_dir, "g1_fdt_step_%s.npz" % ck)
    c = CFG[ck]
    p = params(c, True)
    img, tok, mask = torch.from_numpy(g["images"]), torch.from_numpy(g["tokens"]), torch.from_numpy(g["pad_mask"])
    # inputs are reproducible from seeds alone
    np.testing.assert_array_equal(det_images(c["batch"], c["res"], SEED), g["images"])
    t2, m2 = det_tokens(c["batch"], c["ctx"], SEED)
    np.testing.assert_array_equal(t2, g["tokens"])
    np.testing.assert_array_equal(m2, g["pad_mask"])
    with torch.no_grad():
        proj, dense, _ = O.vit_forward(img, p, c["heads"])
        tproj, words, _ = O.text_forward(tok, p, c["t_heads"])
        close(dense, g["patch_ft"], what="patch_ft")
        close(words, g["word_ft"], what="word_ft")
        close(proj, g["img_proj"], what="img_proj")
        close(tproj, g["txt_proj"], what="txt_proj")
        close(O.q_map(dense, p, "img_query_model."), g["img_q"], what="img_q")
        close(O.q_map(words, p, "txt_query_model."), g["txt_q"], what="txt_q")


@pytest.mark.parametrize("ck", list(CFG))
@pytest.mark.parametrize("v", FDT_VARIANTS, ids=variant_key)
def test_g1_fdt_step(golden_dir, ck, v):
    g = load(golden_dir, "g1_fdt_step_%s.npz" % ck)
    c = CFG[ck]
    vk = variant_key(v)
    p = params(c, True, logit_scale=v[3])
    img, tok, mask = torch.from_numpy(g["images"]), torch.from_numpy(g["tokens"]), torch.from_numpy(g["pad_mask"])
    o = O.clip_fdt_forward(p, img, tok, mask, oracle_cfg(c, v))
    loss, labels = O.info_nce(o["logits_i"], o["logits_t"])
    loss.backward()
    with torch.no_grad():
        close(o["img_q"]["att_w"], g[vk + ".img_att_w"], what="img_att_w")
        close(o["txt_q"]["att_w"], g[vk + ".txt_att_w"], what="txt_att_w")
        close(o["img_q"]["att_ft"], g[vk + ".img_att_ft"], what="img_att_ft")
        close(o["txt_q"]["att_ft"], g[vk + ".txt_att_ft"], what="txt_att_ft")
        close(o["logits_i"], g[vk + ".logits_i"], rtol=2e-5, what="logits_i")
        close(o["logits_t"], g[vk + ".logits_t"], rtol=2e-5, what="logits_t")
        assert abs(loss.item() - float(g[vk + ".loss"])) <= 1e-5 * abs(float(g[vk + ".loss"]))
        np.testing.assert_array_equal(labels.numpy(), g[vk + ".labels"])
        prec = O.accuracy(o["logits_i"], labels, topk=(1, min(5, o["logits_i"].shape[1])))
        np.testing.assert_allclose([x.item() for x in prec], g[vk + ".prec"], rtol=1e-6)
    check_grad_probes(g, vk + ".", p)
    # parameters the FDT loss never reaches
    for name in ("logit_scale_sd", "visual.proj", "visual.ln_post.weight", "encode_text.text_projection.weight"):
        assert (vk + ".gradnone." + name) in g


@pytest.mark.parametrize("ck", list(CFG))
def test_g2_clip_step(golden_dir, ck):
    g = load(golden_dir, "g2_clip_step_%s.npz" % ck)
    c = CFG[ck]
    p = params(c, False)
    img = torch.from_numpy(det_images(c["batch"], c["res"], SEED))
    tok, _ = det_tokens(c["batch"], c["ctx"], SEED)
    o = O.clip_forward(p, img, torch.from_numpy(tok), oracle_cfg(c))
    loss, labels = O.info_nce(o["logits_i"], o["logits_t"])
    loss.backward()
    close(o["logits_i"].detach(), g["logits_i"], rtol=2e-5, what="logits_i")
    close(o["logits_t"].detach(), g["logits_t"], rtol=2e-5, what="logits_t")
    assert abs(loss.item() - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    check_grad_probes(g, "", p)


def test_g3_blocks(golden_dir):
    g = load(golden_dir, "g3_ops.npz")
    for tag, L, E, heads, causal in (("vit", 50, 768, 12, False), ("txt", 77, 512, 8, True)):
        names = ["attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj.weight", "attn.out_proj.bias",
                 "ln_1.weight", "ln_1.bias", "mlp.c_fc.weight", "mlp.c_fc.bias", "mlp.c_proj.weight",
                 "mlp.c_proj.bias", "ln_2.weight", "ln_2.bias"]
        shapes = {"attn.in_proj_weight": (3 * E, E), "attn.in_proj_bias": (3 * E,), "attn.out_proj.weight": (E, E),
                  "attn.out_proj.bias": (E,), "ln_1.weight": (E,), "ln_1.bias": (E,), "mlp.c_fc.weight": (4 * E, E),
                  "mlp.c_fc.bias": (4 * E,), "mlp.c_proj.weight": (E, 4 * E), "mlp.c_proj.bias": (E,),
                  "ln_2.weight": (E,), "ln_2.bias": (E,)}
        p = {"b." + n: torch.from_numpy(det_param("blk." + n, shapes[n], SEED)).requires_grad_(True) for n in names}
        x = torch.from_numpy(det_param("x." + tag, (2, L, E), SEED) * (E ** 0.5)).requires_grad_(True)
        gy = torch.from_numpy(det_param("gy." + tag, (2, L, E), SEED) * (E ** 0.5))
        y = O.resblock(x, p, "b.", heads, causal)
        y.backward(gy)
        close(probe(tag + ".y", y.detach().numpy(), 4096)[2:], g[tag + ".y"][2:], rtol=2e-5, what=tag + ".y")
        close(probe(tag + ".dx", x.grad.numpy(), 4096)[2:], g[tag + ".dx"][2:], rtol=2e-5, what=tag + ".dx")
        for n in names:
            close(probe(n, p["b." + n].grad.numpy(), 256)[2:], g[tag + ".grad." + n][2:], rtol=5e-5,
                  what=tag + ".grad." + n)


def test_g3_rows(golden_dir):
    g = load(golden_dir, "g3_ops.npz")
    for E in (768, 512):
        w = torch.from_numpy(det_param("ln.weight", (E,), SEED)).requires_grad_(True)
        b = torch.from_numpy(det_param("ln.bias", (E,), SEED)).requires_grad_(True)
        x = torch.from_numpy(det_param("lnx", (16, E), SEED) * (E ** 0.5) * 3 + 0.5).requires_grad_(True)
        gy = torch.from_numpy(det_param("lngy", (16, E), SEED) * (E ** 0.5))
        y = O.layer_norm(x, w, b)
        y.backward(gy)
        close(y.detach(), g["ln%d.y" % E], what="ln.y")
        close(x.grad, g["ln%d.dx" % E], rtol=2e-5, what="ln.dx")
        close(w.grad, g["ln%d.dw" % E], what="ln.dw")
        close(b.grad, g["ln%d.db" % E], what="ln.db")
    x = torch.from_numpy(g["act.x"])
    close(O.quick_gelu(x), g["act.quick_gelu"], rtol=1e-6)
    close(O.gelu_erf(x), g["act.gelu_erf"], rtol=1e-6)
    z = torch.from_numpy(g["sparsemax.z"]).requires_grad_(True)
    out = O.sparsemax(z)
    out.backward(torch.from_numpy(g["sparsemax.g"]))
    close(out.detach(), g["sparsemax.out"], rtol=1e-6, what="sparsemax")
    close(z.grad, g["sparsemax.dz"], rtol=1e-5, what="sparsemax.dz")
    # the closed form the HIP kernel uses: dz = (g - mean_S g) on the support S
    o = out.detach()
    S = (o > 0).float()
    gg = torch.from_numpy(g["sparsemax.g"])
    dz = S * (gg - (gg * S).sum(-1, keepdim=True) / S.sum(-1, keepdim=True))
    close(dz, g["sparsemax.dz"], rtol=5e-5, what="sparsemax closed-form bwd")  # summation-order noise over 4096 terms
    li = torch.from_numpy(g["ce.li"]).requires_grad_(True)
    lt = torch.from_numpy(g["ce.lt"]).requires_grad_(True)
    loss, labels = O.info_nce(li, lt, rank=2)
    loss.backward()
    assert abs(loss.item() - float(g["ce.loss"])) < 1e-5 * abs(float(g["ce.loss"]))
    np.testing.assert_array_equal(labels.numpy(), g["ce.labels"])
    close(li.grad, g["ce.dli"], what="ce.dli")
    close(lt.grad, g["ce.dlt"], what="ce.dlt")


def test_g4_two_rank(golden_dir):
    g = load(golden_dir, "g4_two_rank_a.npz")
    meta = json.loads(str(g["meta"]))
    c = CFG[meta["cfg"]]
    v = FDT_VARIANTS[0]
    p = params(c, True)
    per_rank = []
    for s in meta["input_seeds"]:
        tok, mask = det_tokens(c["batch"], c["ctx"], s)
        per_rank.append((torch.from_numpy(det_images(c["batch"], c["res"], s)), torch.from_numpy(tok),
                         torch.from_numpy(mask)))
    cfg = oracle_cfg(c, v)
    outs, losses, total = O.simulate_ranks(lambda p_, i, t, m, gather: O.clip_fdt_forward(p_, i, t, m, cfg, gather),
                                           p, per_rank)
    total.backward()
    W = len(per_rank)
    for r in range(W):
        close(outs[r]["logits_i"].detach(), g["r%d.logits_i" % r], rtol=2e-5, what="r%d.logits_i" % r)
        close(outs[r]["logits_t"].detach(), g["r%d.logits_t" % r], rtol=2e-5, what="r%d.logits_t" % r)
        np.testing.assert_array_equal(outs[r]["labels"].numpy(), g["r%d.labels" % r])
        assert abs(losses[r].item() - float(g["r%d.loss" % r])) < 1e-5 * abs(float(g["r%d.loss" % r]))
    with torch.no_grad():
        for t in p.values():
            if t.grad is not None:
                t.grad /= W          # DDP mean
    check_grad_probes(g, "", p, rtol=5e-5)


def test_g5_trajectory(golden_dir):
    g = load(golden_dir, "g5_trajectory.npz")
    with open(os.path.join(golden_dir, "g7_param_groups.json")) as f:
        pass
    c = CFG["a"]
    v = FDT_VARIANTS[0]
    p = params(c, True)
    cfg = oracle_cfg(c, v)
    ln_w = {k for k in p if k.endswith(".weight") and p[k].dim() == 1}
    bias = {k for k in p if k.endswith("bias") and "in_proj_bias" not in k}
    m = {k: torch.zeros_like(t) for k, t in p.items()}
    vv = {k: torch.zeros_like(t) for k, t in p.items()}
    losses, scales, lrs = [], [], []
    for step in range(1, 6):
        lr = O.cosine_lr(step, 5e-5, 5e-4, 3, 20, 0.0, reset_steps=8)
        lrs.append(lr)
        tok, mask = det_tokens(c["batch"], c["ctx"], SEED + 200 + step)
        img = torch.from_numpy(det_images(c["batch"], c["res"], SEED + 200 + step))
        o = O.clip_fdt_forward(p, img, torch.from_numpy(tok), torch.from_numpy(mask), cfg)
        loss, _ = O.info_nce(o["logits_i"], o["logits_t"])
        for t in p.values():
            t.grad = None
        with torch.no_grad():
            p["logit_scale"].clamp_(3, 6)
        loss.backward()
        with torch.no_grad():
            for k, t in p.items():
                if t.grad is None:
                    continue
                wd = 0.0 if (k in ln_w or k in bias or "logit_scale" in k) else 0.1
                O.adamw_step(t, t.grad, m[k], vv[k], step, lr, 0.9, 0.98, 1e-8, wd)
            p["logit_scale"].clamp_(3, 6)
        losses.append(loss.item())
        scales.append(p["logit_scale"].item())
    np.testing.assert_allclose(lrs, g["lrs"], rtol=1e-12)
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-5)
    np.testing.assert_allclose(scales, g["logit_scale"], rtol=1e-6)
    from detfill import probe_index
    for k, t in p.items():
        want = g["final." + k][2:]
        got = probe(k, t.detach().numpy())[2:]
        if k.endswith("in_proj_bias"):
            # the key-bias third has a mathematically ZERO gradient (softmax shift invariance): its fp32
            # gradient is rounding noise whose sign Adam amplifies to +-lr per step -> not comparable
            E = t.numel() // 3
            idx = probe_index(k, t.numel())
            keep = (idx < E) | (idx >= 2 * E)
            want, got = want[keep], got[keep]
        scale = max(np.abs(want).max(), 1e-30)
        assert np.abs(got - want).max() <= 1e-4 * scale, k


def test_g6_lr_table(golden_dir):
    g = load(golden_dir, "g6_lr_table.npz")
    got = [O.cosine_lr(int(s), 5e-5, 5e-4, 500, 80000, 0.0, reset_steps=6000) for s in g["steps"]]
    np.testing.assert_allclose(got, g["lrs"], rtol=1e-12, atol=1e-20)


def test_g7_shapes_and_groups(golden_dir):
    with open(os.path.join(golden_dir, "g7_param_groups.json")) as f:
        g = json.load(f)
    for mtype, fdt in (("clip_fdt_vitb32", True), ("clip_vitb32", False)):
        want = [(k, tuple(s)) for k, s in g[mtype]["state_dict"]]
        got = list(state_shapes(VITB32, fdt=fdt).items())
        assert sorted(want) == sorted(got)
        groups = g[mtype]["groups"]
        assert len(groups) == 10
        ln_names = {k for k, s in want if len(s) == 1 and k.endswith(".weight")}
        for gi, grp in enumerate(groups):
            for name in grp["names"]:
                shape = dict(want)[name]
                is_ln_w = name in ln_names
                is_bias = name.endswith(".bias") and "in_proj_bias" not in name
                assert O.param_group_of(name, shape, is_ln_w, is_bias) == gi, name

"""Generate the golden fixtures under tests/golden/ by running the UNMODIFIED reference
(/root/reference) on CPU.  Build-container only (the reference does not travel).

    python tests/golden/make_golden.py            # all fixtures
    python tests/golden/make_golden.py g1 g6      # a subset

Fixtures hold seeds + inputs + expected outputs (+ gradient probes), never reference source.
Parameters are filled deterministically (detfill.py) so the tests can rebuild them.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_harness as H  # noqa: E402
from configs import CFG, FDT_VARIANTS, variant_key, model_kwargs  # noqa: E402
from detfill import det_param, det_images, det_tokens, probe  # noqa: E402

SEED = 11


def fill(model, seed=SEED, logit_scale=None):
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.copy_(torch.from_numpy(det_param(name, tuple(p.shape), seed, logit_scale)))


def inputs(c, seed=SEED, batch=None):
    b = batch or c["batch"]
    img = torch.from_numpy(det_images(b, c["res"], seed))
    tok, mask = det_tokens(b, c["ctx"], seed)
    return img, torch.from_numpy(tok), torch.from_numpy(mask)


def grad_probes(model):
    out = {}
    for name, p in model.named_parameters():
        if p.grad is None:
            out["gradnone." + name] = np.zeros(1)
        else:
            out["grad." + name] = probe(name, p.grad.numpy())
    return out


def save(name, d):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **d)
    print("wrote %s (%.1f KB)" % (name, os.path.getsize(path) / 1024))


# --------------------------------------------------------------------------------------
def g1_fdt_step():
    """Tiny Clip_FDT forward + loss + backward, all FDT variants."""
    from prototype.loss_functions.loss import ClipInfoCELoss
    from prototype.utils.misc import accuracy
    for ck, c in CFG.items():
        d = {"meta": json.dumps(dict(cfg=ck, seed=SEED))}
        img, tok, mask = inputs(c)
        d["images"], d["tokens"], d["pad_mask"] = img.numpy(), tok.numpy(), mask.numpy()
        for v in FDT_VARIANTS:
            vk = variant_key(v)
            model = H.build("clip_fdt_vitb32", model_kwargs(c, v, H.BPE))
            fill(model, logit_scale=v[3])
            model.train()
            hooks = {}

            def grab(nm):
                def hook(mod, inp, out):
                    hooks[nm] = out
                return hook
            h1 = model.img_query_model.register_forward_hook(grab("img"))
            h2 = model.txt_query_model.register_forward_hook(grab("txt"))
            h3 = model.img_query_model.q_map.register_forward_hook(grab("img_q"))
            h4 = model.txt_query_model.q_map.register_forward_hook(grab("txt_q"))
            img_out = {}
            h5 = model.visual.register_forward_hook(lambda m, i, o: img_out.__setitem__("v", o))
            h6 = model.encode_text.register_forward_hook(lambda m, i, o: img_out.__setitem__("t", o))
            (li, lt), _ = model(img, (tok, mask))
            loss, labels = ClipInfoCELoss()(li, lt)
            prec1, prec5 = accuracy(li, labels, topk=(1, min(5, li.shape[1])))
            model.zero_grad()
            loss.backward()
            for h in (h1, h2, h3, h4, h5, h6):
                h.remove()
            if v is FDT_VARIANTS[0]:
                d["patch_ft"] = img_out["v"][1].detach().numpy()
                d["word_ft"] = img_out["t"][1].detach().numpy()
                d["img_proj"] = img_out["v"][0].detach().numpy()
                d["txt_proj"] = img_out["t"][0].detach().numpy()
                d["img_q"] = hooks["img_q"].detach().numpy()
                d["txt_q"] = hooks["txt_q"].detach().numpy()
            d[vk + ".img_att_w"] = hooks["img"][0].detach().numpy()
            d[vk + ".txt_att_w"] = hooks["txt"][0].detach().numpy()
            d[vk + ".img_att_ft"] = hooks["img"][1].detach().numpy()
            d[vk + ".txt_att_ft"] = hooks["txt"][1].detach().numpy()
            d[vk + ".logits_i"] = li.detach().numpy()
            d[vk + ".logits_t"] = lt.detach().numpy()
            d[vk + ".loss"] = np.array(loss.item(), dtype=np.float64)
            d[vk + ".labels"] = labels.numpy()
            d[vk + ".prec"] = np.array([prec1.item(), prec5.item()])
            for k, val in grad_probes(model).items():
                d[vk + "." + k] = val
        save("g1_fdt_step_%s.npz" % ck, d)


def g2_clip_step():
    """Tiny baseline CLIP forward + loss + backward (G2)."""
    from prototype.loss_functions.loss import ClipInfoCELoss
    for ck, c in CFG.items():
        d = {"meta": json.dumps(dict(cfg=ck, seed=SEED))}
        img, tok, mask = inputs(c)
        model = H.build("clip_vitb32", model_kwargs(c, None, H.BPE))
        fill(model)
        model.train()
        li, lt = model(img, (tok, mask))
        loss, labels = ClipInfoCELoss()(li, lt)
        model.zero_grad()
        loss.backward()
        d.update(logits_i=li.detach().numpy(), logits_t=lt.detach().numpy(),
                 loss=np.array(loss.item(), dtype=np.float64), labels=labels.numpy())
        d.update(grad_probes(model))
        save("g2_clip_step_%s.npz" % ck, d)


def g3_ops():
    """Per-op vectors at real widths (G3): one ViT block, one causal text block, LayerNorm,
    QuickGELU / erf-GELU, sparsemax edge rows, cross-entropy with rank offset."""
    from prototype.model.image_encoder.base_transformer import ResidualAttentionBlock as VBlock, QuickGELU
    from prototype.model.text_encoder.base_transformer import ResidualAttentionBlock as TBlock
    from prototype.model.sparsemax import Sparsemax
    d = {}
    for tag, Block, L, E, heads, causal in (("vit", VBlock, 50, 768, 12, False), ("txt", TBlock, 77, 512, 8, True)):
        mask = None
        if causal:
            mask = torch.triu(torch.full((L, L), float("-inf")), 1)
        blk = Block(E, heads, mask)
        with torch.no_grad():
            for name, p in blk.named_parameters():
                p.copy_(torch.from_numpy(det_param("blk." + name, tuple(p.shape), SEED)))
        x = torch.from_numpy(det_param("x." + tag, (2, L, E), SEED) * (E ** 0.5)).requires_grad_(True)  # ~N(0,1)
        y = blk(x.permute(1, 0, 2))            # reference is [L,N,E]
        if isinstance(y, tuple):
            y = y[0]
        y = y.permute(1, 0, 2)
        gy = torch.from_numpy(det_param("gy." + tag, (2, L, E), SEED) * (E ** 0.5))
        y.backward(gy)
        d[tag + ".y"] = probe(tag + ".y", y.detach().numpy(), 4096)
        d[tag + ".dx"] = probe(tag + ".dx", x.grad.numpy(), 4096)
        for name, p in blk.named_parameters():
            d[tag + ".grad." + name] = probe(name, p.grad.numpy(), 256)
    # LayerNorm rows + activations
    for E in (768, 512):
        ln = torch.nn.LayerNorm(E)
        with torch.no_grad():
            ln.weight.copy_(torch.from_numpy(det_param("ln.weight", (E,), SEED)))
            ln.bias.copy_(torch.from_numpy(det_param("ln.bias", (E,), SEED)))
        x = torch.from_numpy(det_param("lnx", (16, E), SEED) * (E ** 0.5) * 3 + 0.5).requires_grad_(True)
        y = ln(x)
        gy = torch.from_numpy(det_param("lngy", (16, E), SEED) * (E ** 0.5))
        y.backward(gy)
        d["ln%d.y" % E] = y.detach().numpy()
        d["ln%d.dx" % E] = x.grad.numpy()
        d["ln%d.dw" % E] = ln.weight.grad.numpy()
        d["ln%d.db" % E] = ln.bias.grad.numpy()
    x = torch.linspace(-8, 8, 257)
    d["act.x"] = x.numpy()
    d["act.quick_gelu"] = QuickGELU()(x).numpy()
    d["act.gelu_erf"] = torch.nn.GELU()(x).numpy()
    # sparsemax edge cases: random at two scales, ties, one-hot-ish, uniform
    rows = [det_param("sp%d" % i, (4096,), SEED) * 64.0 * s for i, s in enumerate((1.0, 1e-3, 10.0, 0.05))]
    tie = np.zeros(4096, np.float32); tie[:7] = 1.0
    onehot = np.full(4096, -5.0, np.float32); onehot[123] = 9.0
    uni = np.full(4096, 0.25, np.float32)
    two = np.full(4096, -3.0, np.float32); two[5] = 0.3; two[4000] = 0.1
    z = torch.from_numpy(np.stack(rows + [tie, onehot, uni, two])).requires_grad_(True)
    out = Sparsemax(dim=-1)(z)
    g = torch.from_numpy(det_param("spg", tuple(z.shape), SEED) * 64.0)
    out.backward(g)
    d["sparsemax.z"] = z.detach().numpy()
    d["sparsemax.out"] = out.detach().numpy()
    d["sparsemax.g"] = g.numpy()
    d["sparsemax.dz"] = z.grad.numpy()
    # cross entropy with rank offset: local batch 4, world 4 -> [4,16], rank 2
    from prototype.loss_functions.loss import ClipInfoCELoss
    os.environ["RANK"] = "2"
    li = torch.from_numpy(det_param("ce.li", (4, 16), SEED) * 4 * 10).requires_grad_(True)
    lt = torch.from_numpy(det_param("ce.lt", (4, 16), SEED) * 4 * 10).requires_grad_(True)
    loss, labels = ClipInfoCELoss()(li, lt)
    loss.backward()
    os.environ["RANK"] = "0"
    d.update({"ce.li": li.detach().numpy(), "ce.lt": lt.detach().numpy(), "ce.loss": np.array(loss.item()),
              "ce.labels": labels.numpy(), "ce.dli": li.grad.numpy(), "ce.dlt": lt.grad.numpy()})
    save("g3_ops.npz", d)


def _g4_worker(rank, world, port, ck, ret):
    H.install(rank=rank, world_size=world, port=port)
    from prototype.loss_functions.loss import ClipInfoCELoss
    from torch.nn.parallel import DistributedDataParallel as DDP
    c = CFG[ck]
    v = FDT_VARIANTS[0]
    model = H.build("clip_fdt_vitb32", model_kwargs(c, v, H.BPE))
    fill(model)
    model.train()
    model.find_always_freeze_weight()
    ddp = DDP(model, find_unused_parameters=True)
    img, tok, mask = inputs(c, seed=SEED + 100 + rank)
    (li, lt), _ = ddp(img, (tok, mask))
    loss, labels = ClipInfoCELoss()(li, lt)
    loss = loss / world
    ddp.zero_grad()
    loss.backward()
    d = {"r%d.logits_i" % rank: li.detach().numpy(), "r%d.logits_t" % rank: lt.detach().numpy(),
         "r%d.loss" % rank: np.array(loss.item()), "r%d.labels" % rank: labels.numpy()}
    if rank == 0:
        for k, val in grad_probes(model).items():
            d[k] = val
    ret[rank] = d
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def g4_two_rank():
    """2-rank gloo step through the reference AllGather + DDP (G4): pins the gathered logits,
    rank-offset labels, loss/W and the post-all-reduce gradients."""
    import torch.multiprocessing as mp
    for ck in ("a",):
        mgr = mp.Manager()
        ret = mgr.dict()
        mp.spawn(_g4_worker, args=(2, 29533, ck, ret), nprocs=2, join=True)
        d = {"meta": json.dumps(dict(cfg=ck, seed=SEED, world=2, input_seeds=[SEED + 100, SEED + 101]))}
        for r in range(2):
            d.update(ret[r])
        save("g4_two_rank_%s.npz" % ck, d)


def g5_trajectory():
    """5 optimisation steps of the tiny FDT model re-enacting train_solver.py:348-439 with the
    reference model / loss / param_group_all / AdamW / Cosine scheduler (G5)."""
    from prototype.loss_functions.loss import ClipInfoCELoss
    from prototype.utils.misc import param_group_all
    from prototype.optimizer import optim_entry
    from prototype.lr_scheduler import scheduler_entry
    c = CFG["a"]
    v = FDT_VARIANTS[0]
    model = H.build("clip_fdt_vitb32", model_kwargs(c, v, H.BPE))
    fill(model)
    model.train()
    pconfig = dict(bn_w=dict(weight_decay=0), bn_b=dict(weight_decay=0), ln_w=dict(weight_decay=0),
                   ln_b=dict(weight_decay=0), bias=dict(weight_decay=0), logit_scale=dict(weight_decay=0))
    groups = param_group_all(model, pconfig)[0]
    opt = optim_entry(dict(type="AdamW", kwargs=dict(params=groups, lr=5e-5, weight_decay=0.1, betas=[0.9, 0.98],
                                                     amsgrad=False, eps=1e-8)))
    sch = scheduler_entry(H.EasyDict(type="Cosine", kwargs=dict(
        optimizer=opt, base_lr=5e-5, warmup_lr=5e-4, min_lr=0.0, warmup_steps=3, max_iter=20, last_iter=0,
        reset_steps=8)))
    crit = ClipInfoCELoss()
    d = {"meta": json.dumps(dict(cfg="a", seed=SEED, steps=5))}
    losses, scales, lrs = [], [], []
    for step in range(1, 6):
        sch.step(step)
        lrs.append(sch.get_lr()[0])
        img, tok, mask = inputs(c, seed=SEED + 200 + step)
        (li, lt), _ = model(img, (tok, mask))
        loss, _ = crit(li, lt)
        opt.zero_grad()
        model.logit_scale.data.clamp_(min=3, max=6)
        loss.backward()
        opt.step()
        model.logit_scale.data.clamp_(min=3, max=6)
        losses.append(loss.item())
        scales.append(model.logit_scale.item())
    d["losses"] = np.array(losses, dtype=np.float64)
    d["logit_scale"] = np.array(scales, dtype=np.float64)
    d["lrs"] = np.array(lrs, dtype=np.float64)
    for name, p in model.named_parameters():
        d["final." + name] = probe(name, p.detach().numpy())
    save("g5_trajectory.npz", d)


def g6_lr_table():
    from prototype.lr_scheduler import scheduler_entry
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=5e-5)
    sch = scheduler_entry(H.EasyDict(type="Cosine", kwargs=dict(
        optimizer=opt, base_lr=5e-5, warmup_lr=5e-4, min_lr=0.0, warmup_steps=500, max_iter=80000, last_iter=0,
        reset_steps=6000)))
    steps = [1, 2, 250, 499, 500, 501, 5999, 6000, 6001, 6250, 6499, 6500, 12000, 12001, 40000, 79999, 80000]
    lrs = []
    for s in steps:
        sch.step(s)
        lrs.append(sch.get_lr()[0])
    save("g6_lr_table.npz", dict(steps=np.array(steps), lrs=np.array(lrs, dtype=np.float64)))


def g7_param_groups():
    """param_group_all partition + state_dict manifest of the real clip_fdt_vitb32 / clip_vitb32 (G7)."""
    from prototype.utils.misc import param_group_all
    pconfig = dict(bn_w=dict(weight_decay=0), bn_b=dict(weight_decay=0), ln_w=dict(weight_decay=0),
                   ln_b=dict(weight_decay=0), bias=dict(weight_decay=0), logit_scale=dict(weight_decay=0))
    out = {}
    for mtype, extra in (("clip_fdt_vitb32", dict(fdt=dict(sd_temperature=1000, att_func_type="sparsemax",
                                                             pool_type="max", use_allgather=True, sd_num=4096,
                                                             sd_dim=512, raw_img_ft_dim=768, raw_txt_ft_dim=512))),
                         ("clip_vitb32", dict(clip=dict(use_allgather=True)))):
        kw = dict(image_encode=dict(embed_dim=512),
                  text_encode=dict(bpe_path=H.BPE, text_encode_type="Transformer",
                                   text_model_utils=dict(random=False, freeze=False), embed_dim=512))
        kw.update(extra)
        model = H.build(mtype, kw)
        model.train()
        groups = param_group_all(model, pconfig)[0]
        id2name = {id(p): n for n, p in model.named_parameters()}
        out[mtype] = dict(
            state_dict=[[k, list(v.shape)] for k, v in model.state_dict().items()],
            groups=[dict(weight_decay=g.get("weight_decay", None), names=[id2name[id(p)] for p in g["params"]])
                    for g in groups])
    with open(os.path.join(HERE, "g7_param_groups.json"), "w") as f:
        json.dump(out, f)
    print("wrote g7_param_groups.json")


def g8_reset():
    """reset_text_encoder(seed) on the tiny model: which keys change, and their new values
    (probes) for seed 6000 (G8)."""
    c = CFG["a"]
    model = H.build("clip_fdt_vitb32", model_kwargs(c, FDT_VARIANTS[0], H.BPE))
    fill(model)
    before = {k: v.clone() for k, v in model.state_dict().items()}
    model.reset_text_encoder(6000)
    d = {"meta": json.dumps(dict(cfg="a", seed=SEED, reset_seed=6000))}
    changed = []
    for k, v in model.state_dict().items():
        if not torch.equal(v, before[k]):
            changed.append(k)
            d["after." + k] = probe(k, v.numpy())
    d["changed"] = np.array(json.dumps(changed))
    save("g8_reset.npz", d)


def g9_tokenizer():
    from prototype.model.text_encoder.text_transformer import TextTransformer
    caps = ["a photo of a cat", "A man riding a wave on top of a surfboard.", "two dogs, one brown & one white!",
            "the quick brown fox jumps over the lazy dog", "hello", "", "  extra   spaces   here  ",
            "it's a dog's life; isn't it?", "1234 numbers 56 and 7.89", "CAPS LOCK TEXT",
            "a very long caption " + "with many repeated words " * 30, "semi-colon; colon: dash- slash/",
            "an image of a red apple on a wooden table next to a glass of water",
            "supercalifragilisticexpialidocious", "e-mail user@example.com now", "<|startoftext|> literal"]
    tt = TextTransformer(embed_dim=8, context_length=77, transformer_width=64, transformer_heads=1,
                         transformer_layers=1, positional_embedding_flag=True, checkpoint=False, bpe_path=H.BPE,
                         text_encode_type="Transformer", text_model_utils=dict(random=False, freeze=False))
    tok, mask = tt.tokenize(caps, context_length=77)
    with open(os.path.join(HERE, "g9_tokenizer.json"), "w") as f:
        json.dump(dict(captions=caps, tokens=tok.tolist(), pad_mask_valid=(mask == 0).int().tolist()), f)
    print("wrote g9_tokenizer.json")


ALL = dict(g1=g1_fdt_step, g2=g2_clip_step, g3=g3_ops, g4=g4_two_rank, g5=g5_trajectory, g6=g6_lr_table,
           g7=g7_param_groups, g8=g8_reset, g9=g9_tokenizer)

if __name__ == "__main__":
    which = sys.argv[1:] or list(ALL)
    if which != ["g4"]:
        H.install()
    torch.manual_seed(0)
    for k in which:
        if k == "g4" and len(which) > 1:
            # g4 spawns its own process group; run it in a clean interpreter
            import subprocess
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "g4"])
            continue
        ALL[k]()

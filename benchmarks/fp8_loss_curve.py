"""Loss curve of the fp8 mode against bf16 (BASELINE.json configs[4]: "loss-scale validated vs bf16"): the shipped
ViT-B/32 + FDT geometry, AdamW + cosine schedule as bench.py, the same initial weights and the same stream of synthetic
batches (a fixed set cycled, so the loss has something to learn).

    python benchmarks/fp8_loss_curve.py [--steps 200] [--batch 512] [--out profiles/round2/fp8_loss_curve.json]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def concept_batches(batch, steps, concepts=512, noise=0.5, seed=1234, device="cuda"):
    """Learnable synthetic pairs with EVERY batch distinct: `concepts` latent ids, each a fixed low-resolution image pattern
    (3 x 7 x 7, upsampled to 224 x 224) and a fixed caption (bench.py's row format, random length in 8..77, random token ids);
    a batch is a fresh sample of `batch` distinct concepts plus fresh Gaussian pixel noise (drawn on the device from a seeded
    generator: both precisions see the same stream).  Random unrelated pairs (bench.py's generator) cannot be learnt in a
    few hundred steps by either precision -- the loss sits at ln(batch) -- so they validate nothing."""
    import torch
    from ilvlm_amd import ops
    g = torch.Generator().manual_seed(seed)
    pat = torch.randn(concepts, 3, 7, 7, generator=g).to(device)
    lens = torch.randint(8, 78, (concepts,), generator=g)
    toks = torch.randint(0, 49406, (concepts, 77), generator=g)
    gd = torch.Generator(device=device).manual_seed(seed + 1)
    for _ in range(steps):
        ids = torch.randperm(concepts, generator=g)[:batch]
        img = pat[ids.to(device)].repeat_interleave(32, 2).repeat_interleave(32, 3)
        img = img + noise * torch.randn(img.shape, generator=gd, device=device)
        tokens = torch.zeros(batch, 77, dtype=torch.int64)
        pad = torch.full((batch, 77), float("-inf"))
        ln = lens[ids].tolist()
        for r, (i, n) in enumerate(zip(ids.tolist(), ln)):
            tokens[r, 0] = 49407
            tokens[r, 1:n - 1] = toks[i, 1:n - 1]
            tokens[r, n - 1] = 49408
            pad[r, :n] = 0.0
        yield img.contiguous(), (tokens.to(device), pad.to(device), ops.PackedSeq(ln, 77, device))


def run(precision, steps=200, batch=512, n_batches=2, seed=0, peak_lr=5e-4, structured=False, warmup_steps=50):
    import torch
    import bench as BN
    from ilvlm_amd import ops
    from ilvlm_amd.prototype.model import model_entry
    from ilvlm_amd.prototype.loss_functions import ClipInfoCELoss
    from ilvlm_amd.prototype.optimizer import optim_entry
    from ilvlm_amd.prototype.lr_scheduler import scheduler_entry
    from ilvlm_amd.prototype.utils.misc import param_group_all
    from ilvlm_amd.prototype.utils import torch_ddp_dist as D
    D.set_random_seed(seed)
    model = model_entry(dict(type="clip_fdt_vitb32", kwargs=BN.fdt_kwargs(precision)))
    model.cuda().train()
    opt = optim_entry(dict(type="AdamW", kwargs=dict(params=param_group_all(model, BN.PCONFIG)[0], lr=peak_lr / 10, weight_decay=0.1,
                                                     betas=[0.9, 0.98], amsgrad=False, eps=1e-8)))
    sched = scheduler_entry(dict(type="Cosine", kwargs=dict(optimizer=opt, base_lr=peak_lr / 10, warmup_lr=peak_lr, min_lr=0.0,
                                                            warmup_steps=warmup_steps, max_iter=max(steps, warmup_steps + 1), last_iter=0, reset_steps=0)))
    crit = ClipInfoCELoss()
    data = []
    stream = concept_batches(batch, steps) if structured else None
    for i in range(0 if structured else n_batches):
        images, tokens, pad, lens = BN.synthetic_batch(batch, 100 + i, "cuda")
        data.append((images, (tokens, pad, ops.PackedSeq(lens, tokens.shape[1], "cuda"))))
    losses = []
    for step in range(1, steps + 1):
        sched.step(step)
        images, texts = next(stream) if structured else data[(step - 1) % n_batches]
        (li, lt), _ = model(images, texts)
        loss, _ = crit(li, lt)
        opt.zero_grad()
        ops.clamp_(model.logit_scale.data, 3, 6)
        loss.backward()
        opt.step()
        ops.clamp_(model.logit_scale.data, 3, 6)
        losses.append(loss.detach())
    torch.cuda.synchronize()
    out = [float(l) for l in losses]
    del model, opt
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--n-batches", type=int, default=2, help="distinct synthetic batches cycled (random pairs: a large set is "
                    "not learnable within a few hundred steps and the curve stays at ln(batch))")
    ap.add_argument("--peak-lr", type=float, default=1e-4, help="warm-up target (config_cc3m.yaml uses 5e-4 at global batch 1024; "
                    "on a couple of random batches that schedule is chaotic in BOTH precisions, which makes a pointwise "
                    "comparison meaningless)")
    ap.add_argument("--structured", action="store_true", help="learnable concept pairs, every batch distinct (concept_batches)")
    ap.add_argument("--warmup-steps", type=int, default=50, help="config_cc3m.yaml: 500")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    l8 = run("fp8", a.steps, a.batch, a.n_batches, peak_lr=a.peak_lr, structured=a.structured, warmup_steps=a.warmup_steps)
    lb = run("bf16", a.steps, a.batch, a.n_batches, peak_lr=a.peak_lr, structured=a.structured, warmup_steps=a.warmup_steps)
    gap = [abs(x - y) / max(abs(y), 1e-3) for x, y in zip(l8, lb)]
    res = dict(steps=a.steps, per_gpu_batch=a.batch, peak_lr=a.peak_lr, warmup_steps=a.warmup_steps,
               data=("%d distinct batches of learnable concept pairs (512 concepts, fresh sample + noise per batch)" % a.steps) if a.structured
               else "%d synthetic batch(es) cycled (bench.py generator, seeds 100..)" % a.n_batches,
               fp8=l8, bf16=lb, max_relative_gap=max(gap), mean_relative_gap=sum(gap) / len(gap),
               final=dict(fp8=l8[-1], bf16=lb[-1]))
    print(json.dumps({k: v for k, v in res.items() if k not in ("fp8", "bf16")}))
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)

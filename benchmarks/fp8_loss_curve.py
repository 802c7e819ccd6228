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


def run(precision, steps=200, batch=512, n_batches=2, seed=0, peak_lr=5e-4):
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
                                                            warmup_steps=50, max_iter=max(steps, 51), last_iter=0, reset_steps=0)))
    crit = ClipInfoCELoss()
    data = []
    for i in range(n_batches):
        images, tokens, pad, lens = BN.synthetic_batch(batch, 100 + i, "cuda")
        data.append((images, (tokens, pad, ops.PackedSeq(lens, tokens.shape[1], "cuda"))))
    losses = []
    for step in range(1, steps + 1):
        sched.step(step)
        images, texts = data[(step - 1) % n_batches]
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
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    l8 = run("fp8", a.steps, a.batch, a.n_batches, peak_lr=a.peak_lr)
    lb = run("bf16", a.steps, a.batch, a.n_batches, peak_lr=a.peak_lr)
    gap = [abs(x - y) / max(abs(y), 1e-3) for x, y in zip(l8, lb)]
    res = dict(steps=a.steps, per_gpu_batch=a.batch, peak_lr=a.peak_lr, data="%d synthetic batch(es) cycled (bench.py generator, seeds 100..)" % a.n_batches,
               fp8=l8, bf16=lb, max_relative_gap=max(gap), mean_relative_gap=sum(gap) / len(gap),
               final=dict(fp8=l8[-1], bf16=lb[-1]))
    print(json.dumps({k: v for k, v in res.items() if k not in ("fp8", "bf16")}))
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)

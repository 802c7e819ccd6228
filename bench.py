"""Headline benchmark: image-text pairs / second of the full CLIP+FDT training step (ViT-B/32 + FDT codebook,
bf16 MFMA compute with fp32 master weights, per-GPU batch 256 -> global batch 256 x ngpu, synthetic 224x224 + 77-token
pairs, random-init weights), i.e. BASELINE.json configs[1] (1 GPU) / configs[2] (N GPUs, embedding all-gather).

A "step" is one full optimisation step on one batch already resident in HBM: LR schedule, forward of both towers +
FDT heads + gathered logits, InfoNCE loss / world, top-1/5 accuracy, zero_grad, logit-scale clamp, backward
(embedding-gradient reduce-scatter + overlapped gradient all-reduce), fused AdamW, clamp.

    python bench.py [--gpus N] [--steps K] [--warmup W]
For N > 1 run under torchrun (the driver does); a bare `python bench.py --gpus N` re-launches itself that way.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# fwd+bwd algorithmic FLOPs per pair
FLOPS_PER_PAIR = {"vitb32": 45.9e9, "vitl14": 531.1e9}
PEAK_BF16 = 2500.0             # TFLOP/s dense bf16 MFMA, MI355X (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (BASELINE configs: 256 for vitb32, 128 for vitl14)")
    ap.add_argument("--model", default="vitb32", choices=["vitb32", "vitl14"],
                    help="vitb32 = the headline workload (BASELINE.json configs[1]/[2]); vitl14 = configs[3], ViT-L/14 + FDT, "
                         "an extra data point that is never the default")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--all-text-positions", action="store_true",
                    help="compute all 77 positions of every caption as the reference does; default: the text tower runs on the "
                         "valid tokens only (rows behind <|endoftext|> never reach the loss; same logits and gradients)")
    ap.add_argument("--phase-times", action="store_true", help="diagnostic: GPU time between the phase boundaries of a step")
    ap.add_argument("--serial-towers", action="store_true",
                    help="run both towers on one stream (used for per-kernel profiles; the headline run overlaps them)")
    return ap.parse_args()


def fdt_kwargs(precision, model="vitb32"):
    embed, img_w, txt_w = (512, 768, 512) if model == "vitb32" else (768, 1024, 768)
    return dict(
        image_encode=dict(embed_dim=embed),
        text_encode=dict(bpe_path=None, text_encode_type="Transformer", text_model_utils=dict(random=False, freeze=False),
                         embed_dim=embed),
        fdt=dict(sd_temperature=1000, att_func_type="sparsemax", pool_type="max", use_allgather=True, sd_num=4096,
                 sd_dim=512, raw_img_ft_dim=img_w, raw_txt_ft_dim=txt_w),
        precision=precision)


PCONFIG = dict(bn_w=dict(weight_decay=0), bn_b=dict(weight_decay=0), ln_w=dict(weight_decay=0), ln_b=dict(weight_decay=0),
               bias=dict(weight_decay=0), logit_scale=dict(weight_decay=0))


def synthetic_batch(batch, rank, device):
    """randn images (seed 1234+rank); rows [SOT, U{0..49405} x (n-2), EOT, 0...], n ~ U{8..77}."""
    import torch
    g = torch.Generator().manual_seed(1234 + rank)
    images = torch.randn(batch, 3, 224, 224, generator=g)
    tokens = torch.zeros(batch, 77, dtype=torch.int64)
    pad = torch.full((batch, 77), float("-inf"))
    lens = torch.randint(8, 78, (batch,), generator=g)
    for b in range(batch):
        n = int(lens[b])
        tokens[b, 0] = 49407
        tokens[b, 1:n - 1] = torch.randint(0, 49406, (n - 2,), generator=g)
        tokens[b, n - 1] = 49408
        pad[b, :n] = 0
    return images.to(device), tokens.to(device), pad.to(device), [int(n) for n in lens]


def cpu_baseline(batch=16, steps=2):
    """The CPU oracle (a port of the reference's algorithm, parity-pinned to it) timed on the host cores: full-size
    clip_fdt_vitb32, fp32, forward + loss + backward + AdamW.  Bounded sample; reported baseline only."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from configs import VITB32, state_shapes, oracle_cfg, FDT_VARIANTS
    from oracle import clip_oracle as O
    torch.manual_seed(0)
    shapes = state_shapes(VITB32, fdt=True)
    p = {}
    for k, s in shapes.items():
        t = torch.randn(*s) * (0.02 if len(s) > 1 else 0.05)
        if len(s) == 1 and k.endswith(".weight"):
            t = 1 + t
        if "logit_scale" in k:
            t = torch.full(s, 2.659)
        p[k] = t.requires_grad_(k != "visual.conv1.weight")
    images, tokens, pad, _ = synthetic_batch(batch, 0, "cpu")
    cfg = oracle_cfg(VITB32, FDT_VARIANTS[0])
    m = {k: torch.zeros_like(t) for k, t in p.items()}
    v = {k: torch.zeros_like(t) for k, t in p.items()}
    times = []
    for step in range(1, steps + 2):
        t0 = time.time()
        o = O.clip_fdt_forward(p, images, tokens, pad, cfg)
        loss, _ = O.info_nce(o["logits_i"], o["logits_t"])
        for t in p.values():
            t.grad = None
        loss.backward()
        with torch.no_grad():
            for k, t in p.items():
                if t.grad is not None:
                    O.adamw_step(t, t.grad, m[k], v[k], step, 5e-5, 0.9, 0.98, 1e-8, 0.1 if t.dim() > 1 else 0.0)
        times.append(time.time() - t0)
    dt = sorted(times[1:])[len(times[1:]) // 2]
    return dict(value=batch / dt, unit="pairs/s", cores=torch.get_num_threads(), kind="port",
                sample="oracle/clip_oracle.py, clip_fdt_vitb32 fp32 fwd+loss+bwd+AdamW, batch %d, median of %d steps after 1 warm-up "
                       "(%.2f s/step)" % (batch, steps, dt))


def main():
    args = parse()
    if args.batch is None:
        args.batch = 256 if args.model == "vitb32" else 128
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        port = os.environ.get("MASTER_PORT", "29517")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import torch
    import torch.distributed as dist
    from ilvlm_amd import ops
    from ilvlm_amd.prototype.model import model_entry
    from ilvlm_amd.prototype.loss_functions import ClipInfoCELoss
    from ilvlm_amd.prototype.optimizer import optim_entry
    from ilvlm_amd.prototype.lr_scheduler import scheduler_entry
    from ilvlm_amd.prototype.utils.misc import param_group_all, accuracy
    from ilvlm_amd.prototype.utils import torch_ddp_dist as D

    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal hooks (one-GPU boxes): ILVLM_BENCH_ONE_DEVICE=1 puts every rank on cuda:0, ILVLM_DIST_BACKEND=gloo
    # replaces RCCL (which refuses two ranks on one device).  The driver's real runs use neither.
    if os.environ.get("ILVLM_BENCH_ONE_DEVICE") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(os.environ.get("ILVLM_DIST_BACKEND", "nccl"), rank=rank, world_size=world)

    D.set_random_seed(0)
    model = model_entry(dict(type="clip_fdt_vitb32" if args.model == "vitb32" else "clip_fdt_vitL14",
                             kwargs=fdt_kwargs(args.precision, args.model)))
    model.cuda()
    ddp = D.convert_to_ddp_model(model, local)
    groups = param_group_all(ddp, PCONFIG)[0]
    opt = optim_entry(dict(type="AdamW", kwargs=dict(params=groups, lr=5e-5, weight_decay=0.1, betas=[0.9, 0.98],
                                                     amsgrad=False, eps=1e-8)))
    sched = scheduler_entry(dict(type="Cosine", kwargs=dict(optimizer=opt, base_lr=5e-5, warmup_lr=5e-4, min_lr=0.0,
                                                            warmup_steps=500, max_iter=80000, last_iter=0, reset_steps=6000)))
    ddp.train()
    if os.environ.get("ILVLM_GEMM_VARIANT"):          # tuning hook (benchmarks): force one bf16 GEMM kernel
        ops.gemm_set_variant(int(os.environ["ILVLM_GEMM_VARIANT"]))
    model.engine.concurrent_towers = not args.serial_towers
    crit = ClipInfoCELoss()
    images, tokens, pad, lens = synthetic_batch(args.batch, rank, dev)
    # the resident batch carries its caption lengths as host metadata (what a tokenising data loader knows); with them the
    # text tower runs on the valid tokens only
    texts = (tokens, pad) if args.all_text_positions else (tokens, pad, ops.PackedSeq(lens, tokens.shape[1], dev))
    state = dict(step=0)

    def one_step():
        state["step"] += 1
        sched.step(state["step"])
        (li, lt), _ = ddp(images, texts)
        loss, target = crit(li, lt)
        loss = loss / world
        prec1, prec5 = accuracy(li, target, topk=(1, 5))
        opt.zero_grad()
        ops.clamp_(model.logit_scale.data, 3, 6)
        loss.backward()
        opt.step()
        ops.clamp_(model.logit_scale.data, 3, 6)
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = one_step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
    host_dt = time.perf_counter() - t0          # time the host needed to ENQUEUE the steps (no sync yet)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.item()) * world
    ms_per_step = 1000.0 * dt / args.steps
    value = world * args.batch * args.steps / dt

    roofline = None
    if rank == 0 and not args.no_roofline and args.precision == "bf16":
        # per-launch durations are only meaningful when kernels do not share the chip: serialise the towers here
        # (the timed region above overlaps them on two streams)
        model.engine.concurrent_towers = False
        torch.cuda.synchronize()
        prof = ops.GemmProfiler()
        ops.set_gemm_profiler(prof)
        nprof = 2
        for _ in range(nprof):
            one_step()
        ops.set_gemm_profiler(None)
        s = prof.summary()
        model.engine.concurrent_towers = not args.serial_towers
        achieved = s["flops"] / (s["ms"] * 1e-3) / 1e12
        traffic = None      # HBM bytes per launch of this kernel family from the committed PMC pass (profiles/round1)
        try:
            if args.model != "vitb32":
                raise LookupError("the committed counter pass is of the headline workload")
            with open(os.path.join(ROOT, "profiles", "round1", "hbm_traffic.json")) as f:
                hb = json.load(f)
            rows = [v for k, v in hb.items() if "gemm_bf16_dma_kernel" in k]
            n = sum(v["launches"] for v in rows)
            traffic = round(sum((v["read_mb_per_launch"] + v["write_mb_per_launch"]) * 1e6 * v["launches"] for v in rows) / n)
        except Exception:
            pass
        roofline = dict(bound="mfma", achieved=round(achieved, 2), peak=PEAK_BF16, unit="TFLOP/s",
                        frac=round(achieved / PEAK_BF16, 4), traffic=traffic,
                        traffic_source="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE x2 (gfx950), L2-miss bytes per "
                                       "launch (profiles/round1/hbm_traffic.json); algorithmic_bytes_per_launch = operands + "
                                       "output + epilogue operands once each, from the launches timed here",
                        algorithmic_bytes_per_launch=round(s["bytes"] / max(s["launches"], 1)),
                        kernel="gemm_bf16_dma_kernel (all bf16 MFMA GEMM launches of a step, towers serialised for timing)",
                        launches_per_step=s["launches"] // nprof,
                        gemm_ms_per_step=round(s["ms"] / nprof, 3),
                        algorithmic_gflop_per_step=round(s["flops"] / nprof / 1e9, 1))
    elif world > 1:
        pass
    if world > 1 and rank != 0:
        # other ranks run the same extra profiling steps so collectives stay matched
        if not args.no_roofline and args.precision == "bf16":
            for _ in range(2):
                one_step()
    if args.phase_times and rank == 0:
        model._phase_marks = []
        for _ in range(5):
            one_step()
            ev = torch.cuda.Event(enable_timing=True); ev.record(); model._phase_marks.append(("optimizer_done", ev))
        torch.cuda.synchronize()
        marks, model._phase_marks = model._phase_marks, None
        acc = {}
        for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
            acc.setdefault("%s -> %s" % (n0, n1), []).append(e0.elapsed_time(e1))
        for k, v in acc.items():
            print("phase %-40s %7.3f ms" % (k, sum(v) / len(v)), file=sys.stderr, flush=True)
        hs = []
        for _ in range(5):              # host cost of enqueueing one step into an idle GPU (no queue back-pressure)
            torch.cuda.synchronize()
            t0h = time.perf_counter()
            one_step()
            hs.append(1000.0 * (time.perf_counter() - t0h))
        print("phase %-40s %7.3f ms" % ("host enqueue of one step, GPU idle", sorted(hs)[2]), file=sys.stderr, flush=True)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.model == "vitb32":
        cpu = cpu_baseline()
    if world > 1:
        dist.barrier()
    if rank == 0:
        out = {
            "metric": "image-text pairs/sec (whole node), %s + FDT, global batch %d x ngpu" % (
                "ViT-B/32" if args.model == "vitb32" else "ViT-L/14", args.batch),
            "value": round(value, 1), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "towers": "serial" if args.serial_towers else "concurrent (one HIP stream per tower + a weight-gradient companion stream each)",
            "config": {"workload": "example/clip_fdt %s + FDT codebook (4096x512, sparsemax, max-pool, T=1000), "
                                   "%s compute / fp32 master weights, full train step incl. AdamW" % (
                                       "ViT-B/32" if args.model == "vitb32" else "ViT-L/14 (BASELINE configs[3])", args.precision),
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world, "image": "3x224x224",
                       "context_length": 77, "parallelism": "dp%d" % world,
                       "caption_tokens": "n ~ U{8..77} per caption (SURVEY.md 8d), %d valid of %d positions on rank 0" % (
                           sum(lens), len(lens) * 77),
                       "text_rows": "all positions" if args.all_text_positions else
                                    "valid tokens only (packed rows; positions behind <|endoftext|> never reach the loss)"},
            "step_mfma_frac": round(value / world * FLOPS_PER_PAIR[args.model] / (PEAK_BF16 * 1e12), 4),
            "final_loss": round(final_loss, 4), "host_enqueue_ms_per_step": round(1000.0 * host_dt / args.steps, 3),
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

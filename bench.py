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
PEAK_FP8 = 5000.0              # TFLOP/s dense fp8 MFMA (block-scaled forms; the non-scaled fp8 MFMA used here issues at the bf16 rate)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (BASELINE configs: 256 for vitb32, 128 for vitl14)")
    ap.add_argument("--model", default="vitb32", choices=["vitb32", "vitl14"],
                    help="vitb32 = the headline workload (BASELINE.json configs[1]/[2]); vitl14 = configs[3], ViT-L/14 + FDT, "
                         "an extra data point that is never the default")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "fp8"],
                    help="bf16 = the headline (BASELINE configs[1]/[2]); fp8 = configs[4]: QKV / out / MLP GEMMs forward + input "
                         "gradient on OCP fp8 operands with per-tensor delayed scaling, everything else as bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--all-text-positions", action="store_true",
                    help="compute all 77 positions of every caption as the reference does; default: the text tower runs on the "
                         "valid tokens only (rows behind <|endoftext|> never reach the loss; same logits and gradients)")
    ap.add_argument("--no-dense-leg", action="store_true", help="skip the extra all-text-positions timing leg")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the driver-timed legs for BASELINE configs[3] (ViT-L/14 + FDT, bf16, batch 128) and configs[4] "
                         "(ViT-B/32 fp8) that follow the headline in the default run")
    ap.add_argument("--overlap-adamw", action="store_true",
                    help="per-block AdamW launches issued from inside backward instead of one launch in optimizer.step() "
                         "(bit-identical; measured +-0 on one GPU: the chip is already full during backward)")
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


def host_cpu():
    """(threads to use, description): physical cores from lscpu, capped by what this process may run on"""
    model, cores = "unknown CPU", None
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        kv = dict((l.split(":", 1)[0].strip(), l.split(":", 1)[1].strip()) for l in txt.splitlines() if ":" in l)
        model = kv.get("Model name", model)
        cores = int(kv.get("Core(s) per socket", "0")) * int(kv.get("Socket(s)", "1")) or None
    except Exception:
        pass
    allowed = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n = min(cores or allowed, allowed)
    return n, "%s, %s physical cores, %d usable by this process" % (model, cores if cores else "?", allowed)


def cpu_baseline(batch=32, steps=10):
    """the CPU oracle (a port of the reference's algorithm, parity-pinned to it) timed on the
    host cores: full-size clip_fdt_vitb32, fp32, forward + loss + backward + AdamW, batch 32, 10 steps, median of the last 9,
    torch threads = physical cores.  Reported baseline only."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from configs import VITB32, state_shapes, oracle_cfg, FDT_VARIANTS
    from oracle import clip_oracle as O
    phys, desc = host_cpu()
    prev = torch.get_num_threads()
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
    def one(step):
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
        return time.time() - t0

    # all physical cores is what SURVEY 8(d) prescribes, but on a two-socket host that many threads run these layer sizes
    # SLOWER than one socket's worth; the baseline should be the host's best, so one step each at a few thread counts first
    sweep = {}
    step = 0
    for n in sorted(set(c for c in (16, 32, 64, phys) if c <= phys)):
        torch.set_num_threads(n)
        step += 1
        sweep[n] = one(step)
    threads = min(sweep, key=sweep.get)
    torch.set_num_threads(threads)
    times = []
    t_begin = time.time()
    for _ in range(steps):
        step += 1
        times.append(one(step))
        if time.time() - t_begin > 60 and len(times) >= 4:      # bounded sample: stop early on a slow host, say so below
            break
    torch.set_num_threads(prev)
    rest = sorted(times[1:])
    dt = rest[len(rest) // 2]
    return dict(value=batch / dt, unit="pairs/s", cores=threads, kind="port",
                sample="oracle/clip_oracle.py, clip_fdt_vitb32 fp32 fwd+loss+bwd+AdamW, batch %d, %d steps, median of the last %d "
                       "(%.2f s/step), torch.set_num_threads(%d) = fastest of %s (s/step); host: %s" % (
                           batch, len(times), len(rest), dt, threads, {k: round(x, 2) for k, x in sweep.items()}, desc))


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

    # who takes part: the world as the process group reports it and every rank's device (a SCALE run proves itself)
    prop = torch.cuda.get_device_properties(local)
    me = dict(rank=rank, local_rank=local, device=prop.name, pci_bus_id=getattr(prop, "pci_bus_id", None),
              uuid=str(getattr(prop, "uuid", "")), cus=prop.multi_processor_count, host=os.uname().nodename)
    if world > 1:
        devices = [None] * world
        dist.all_gather_object(devices, me)
        rccl_world = dist.get_world_size()
        backend = dist.get_backend()
    else:
        devices, rccl_world, backend = [me], 1, None

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
    opt.prezero_grads = os.environ.get("ILVLM_PREZERO", "1") == "1"      # as the solver does (solver.build_optimizer)
    if args.precision == "bf16" and not args.overlap_adamw:
        opt.defer_late_blocks(int(os.environ.get("ILVLM_ADAMW_DEFER", "0")))   # as the solver does
    if args.overlap_adamw:
        # reference order (zero_grad, one backward, step): each block's AdamW goes out as soon as its gradients are final
        opt.overlap_backward(True)
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
        if world > 1:                   # (the reference divides unconditionally: train_solver.py:420; a no-op kernel at one rank)
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
    reducer = model.engine.arena.reducer
    sent0, calls0 = reducer.bytes_sent, reducer.calls
    reducer.timing = []                         # exposed communication: comm-stream end vs the optimizer's arrival, per step
    # HIP events at the step boundaries beside the wall clock (on the stream a step begins and ends on; every
    # side stream has been joined into it when the optimizer's launch is enqueued)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        loss = one_step()
        marks[i + 1].record()
    host_dt = time.perf_counter() - t0          # time the host needed to ENQUEUE the steps (no sync yet)
    fence()
    dt = time.perf_counter() - t0
    ev_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    ev_median = ev_ms[len(ev_ms) // 2] if len(ev_ms) % 2 else 0.5 * (ev_ms[len(ev_ms) // 2 - 1] + ev_ms[len(ev_ms) // 2])
    comm_stats = dict(grad_bucket=reducer.bucket, grad_algo=reducer.algo,
                      grad_bytes_per_step=(reducer.bytes_sent - sent0) // args.steps,
                      grad_collectives_per_step=(reducer.calls - calls0) // args.steps,
                      exposed_grad_comm_ms_per_step=round(reducer.exposed_ms(), 4),
                      embedding_exchange="one all_gather_into_tensor of [2,B,512] fp32 forward + one reduce_scatter_tensor of "
                                         "[W,2,B,512] backward per step" if world > 1 else "none (one rank)")
    reducer.timing = None
    # host cost of ENQUEUEING one step, measured into an idle GPU (inside the timed loop the host runs ahead of the GPU and
    # ends up waiting for room in the hardware queues: that figure, host_loop_ms_per_step, is not a host cost)
    hs = []
    for _ in range(3):
        fence()
        th = time.perf_counter()
        one_step()
        hs.append(1000.0 * (time.perf_counter() - th))
    fence()
    host_idle_ms = sorted(hs)[1]
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.item()) * world
    ms_per_step = 1000.0 * dt / args.steps
    value = world * args.batch * args.steps / dt

    # ---- the same step with all 77 text positions computed, as the reference does (every rank: collectives stay matched)
    dense = None
    if not args.all_text_positions and not args.no_dense_leg:
        texts_packed = texts
        texts = (tokens, pad)
        for _ in range(2):
            one_step()
        fence()
        t0d = time.perf_counter()
        for _ in range(args.steps):
            one_step()
        fence()
        dtd = time.perf_counter() - t0d
        if world > 1:
            t = torch.tensor([dtd], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtd = float(t.item())
        dense = dict(ms_per_step=round(1000.0 * dtd / args.steps, 3), value=round(world * args.batch * args.steps / dtd, 1),
                     unit="pairs/s", note="text tower on all %d positions of every caption (reference behaviour); same logits "
                                          "and gradients as the packed rows" % tokens.shape[1],
                     # SURVEY 8(d)'s FLOPs per pair count all 77 text positions: only this leg executes them
                     step_mfma_frac_survey_flops=round(args.batch * args.steps / dtd * FLOPS_PER_PAIR[args.model] / (PEAK_BF16 * 1e12), 4))
        texts = texts_packed

    # ---- roofline legs (every rank runs the same steps; rank 0 reports)
    roofline = None
    executed_flops = None
    if not args.no_roofline and args.precision in ("bf16", "fp8"):
        nprof = 2
        legs = {}
        for leg, conc in (("serial", False), ("in_step", True)):
            # "serial": both towers on one stream, so a launch owns the chip and its duration is the kernel's own (the regime
            # of the committed rocprofv3 summary); "in_step": the headline regime, towers and weight gradients on their own
            # streams -- launches overlap, so the family's busy time is the UNION of the launch intervals
            model.engine.concurrent_towers = conc
            one_step()
            torch.cuda.synchronize()
            prof = ops.GemmProfiler()
            ops.set_gemm_profiler(prof)
            for _ in range(nprof):
                one_step()
            ops.set_gemm_profiler(None)
            legs[leg] = prof.summary()
        model.engine.concurrent_towers = not args.serial_towers
        s = legs["serial"]
        c = legs["in_step"]
        achieved = s["flops"] / (s["ms"] * 1e-3) / 1e12
        # attention products (2 forward + 5 backward, 2 L^2 64 each per head) on the rows actually computed
        v_tok, v_heads, v_layers, t_heads, t_layers = (50, 12, 12, 8, 12) if args.model == "vitb32" else (257, 16, 24, 12, 12)
        text_l2 = sum(n * n for n in lens) if not args.all_text_positions else len(lens) * 77 * 77
        attn_flops = 7 * 2.0 * 64 * (args.batch * v_tok * v_tok * v_heads * v_layers + text_l2 * t_heads * t_layers)
        executed_flops = s["flops"] / nprof + s["f32_flops"] / nprof + attn_flops
        traffic, traffic_src = None, None      # HBM-side bytes per launch of this kernel family from the committed PMC passes
        for rnd in ("round4", "round3", "round2", "round1"):
            try:
                if args.model != "vitb32":
                    break
                with open(os.path.join(ROOT, "profiles", rnd, "hbm_traffic.json")) as f:
                    hb = json.load(f)
                rows = [v for k, v in hb.items() if "gemm_bf16" in k]
                n = sum(v["launches"] for v in rows)
                traffic = round(sum((v["read_mb_per_launch"] + v["write_mb_per_launch"]) * 1e6 * v["launches"] for v in rows) / n)
                traffic_src = "profiles/%s/hbm_traffic.json" % rnd
                break
            except Exception:
                continue
        if rank == 0:
            roofline = dict(bound="mfma", achieved=round(achieved, 2), peak=PEAK_BF16, unit="TFLOP/s",
                            frac=round(achieved / PEAK_BF16, 4), traffic=traffic,
                            traffic_source="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE x2 (gfx950), L2-miss bytes "
                                           "per launch incl. Infinity-Cache hits (%s); algorithmic_bytes_per_launch = operands + "
                                           "output + epilogue operands once each, from the launches timed here" % traffic_src,
                            algorithmic_bytes_per_launch=round(s["bytes"] / max(s["launches"], 1)),
                            kernel="gemm_bf16_pk_kernel + gemm_bf16_dma_kernel (all bf16 MFMA GEMM launches of a step: the streaming "
                                   "kernel on fragment-packed weights for forward / input gradients with K >= 512, the "
                                   "direct-to-LDS kernels for the rest and the weight gradients); achieved = their FLOPs / the "
                                   "sum of their durations with the towers serialised (HIP events on the launch stream).  The weight "
                                   "gradients run the kernel the library picks for the regime: two-stage 128x128 tiles in this serial leg, "
                                   "single-stage 256x128 tiles (slower alone, faster steps) when the towers run concurrently, i.e. in the "
                                   "timed step and in in_step below",
                            launches_per_step=s["launches"] // nprof,
                            gemm_ms_per_step=round(s["ms"] / nprof, 3),
                            algorithmic_gflop_per_step=round(s["flops"] / nprof / 1e9, 1),
                            in_step=dict(note="headline regime (towers and weight gradients on their own streams, kernel-by-kernel "
                                              "launches): launches overlap, busy time = union of the launch intervals",
                                         gemm_busy_ms_per_step=round(c["union_ms"] / nprof, 3),
                                         gemm_sum_of_durations_ms_per_step=round(c["ms"] / nprof, 3),
                                         achieved=round(c["flops"] / (c["union_ms"] * 1e-3) / 1e12, 2),
                                         frac=round(c["flops"] / (c["union_ms"] * 1e-3) / 1e12 / PEAK_BF16, 4)))
            if args.precision == "fp8" and s["fp8"]["launches"]:
                f8 = s["fp8"]
                a8 = f8["flops"] / (f8["ms"] * 1e-3) / 1e12
                roofline["fp8_family"] = dict(
                    kernel="gemm_bf16_dma_kernel<..., FP8> (forward + input-gradient GEMMs of the transformer blocks on OCP fp8 "
                           "operands, v_mfma_f32_16x16x32_fp8_*); the other launches counted above are bf16",
                    achieved=round(a8, 2), peak=PEAK_FP8, frac=round(a8 / PEAK_FP8, 4), launches_per_step=f8["launches"] // nprof,
                    ms_per_step=round(f8["ms"] / nprof, 3), algorithmic_gflop_per_step=round(f8["flops"] / nprof / 1e9, 1),
                    algorithmic_bytes_per_launch=round(f8["bytes"] / f8["launches"]))
    if args.phase_times:                 # every rank runs the steps (matched collectives); rank 0 prints
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
            if rank == 0:
                print("phase %-40s %7.3f ms" % (k, sum(v) / len(v)), file=sys.stderr, flush=True)
        hs = []
        for _ in range(5):              # host cost of enqueueing one step into an idle GPU (no queue back-pressure)
            torch.cuda.synchronize()
            t0h = time.perf_counter()
            one_step()
            hs.append(1000.0 * (time.perf_counter() - t0h))
        if rank == 0:
            print("phase %-40s %7.3f ms" % ("host enqueue of one step, GPU idle", sorted(hs)[2]), file=sys.stderr, flush=True)
    # ---- driver-timed legs for the other single-GPU forms of BASELINE configs: [3] ViT-L/14 + FDT bf16 at per-GPU batch 128
    #      and [4] ViT-B/32 fp8 at its stated per-GPU batch 512 (and at the headline's 256).  Fresh model / optimizer each, the same step and timing contract as
    #      the headline (every rank runs them: collectives stay matched), own roofline from the GEMM launches of that leg.
    def extra_leg(model_name, precision, batch):
        D.set_random_seed(0)
        m = model_entry(dict(type="clip_fdt_vitb32" if model_name == "vitb32" else "clip_fdt_vitL14",
                             kwargs=fdt_kwargs(precision, model_name)))
        m.cuda()
        dd = D.convert_to_ddp_model(m, local)
        o = optim_entry(dict(type="AdamW", kwargs=dict(params=param_group_all(dd, PCONFIG)[0], lr=5e-5, weight_decay=0.1,
                                                       betas=[0.9, 0.98], amsgrad=False, eps=1e-8)))
        sc = scheduler_entry(dict(type="Cosine", kwargs=dict(optimizer=o, base_lr=5e-5, warmup_lr=5e-4, min_lr=0.0,
                                                             warmup_steps=500, max_iter=80000, last_iter=0, reset_steps=6000)))
        o.prezero_grads = os.environ.get("ILVLM_PREZERO", "1") == "1"
        if precision == "bf16":
            o.defer_late_blocks(int(os.environ.get("ILVLM_ADAMW_DEFER", "0")))
        dd.train()
        im, tk, pd, ln = synthetic_batch(batch, rank, dev)
        tx = (tk, pd, ops.PackedSeq(ln, tk.shape[1], dev))
        st = dict(step=0)

        def step():
            st["step"] += 1
            sc.step(st["step"])
            (li_, lt_), _ = dd(im, tx)
            ls, tg = crit(li_, lt_)
            if world > 1:
                ls = ls / world
            accuracy(li_, tg, topk=(1, 5))
            o.zero_grad()
            ops.clamp_(m.logit_scale.data, 3, 6)
            ls.backward()
            o.step()
            ops.clamp_(m.logit_scale.data, 3, 6)
            return ls
        for _ in range(max(args.warmup, 3)):        # fp8: one observing step + one step to fill the gradient history
            step()
        fence()
        t0x = time.perf_counter()
        for _ in range(args.steps):
            ls = step()
        fence()
        dtx = time.perf_counter() - t0x
        if world > 1:
            t = torch.tensor([dtx], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtx = float(t.item())
        res = dict(ms_per_step=round(1000.0 * dtx / args.steps, 3), value=round(world * batch * args.steps / dtx, 1), unit="pairs/s",
                   dtype=precision, per_gpu_batch=batch, steps=args.steps, final_loss=round(float(ls.item()) * world, 4),
                   workload="example/clip_fdt %s + FDT, %s compute / fp32 master weights, full train step incl. AdamW, packed text rows" % (
                       "ViT-B/32" if model_name == "vitb32" else "ViT-L/14", precision))
        if not args.no_roofline:
            m.engine.concurrent_towers = False
            step()
            torch.cuda.synchronize()
            prof = ops.GemmProfiler()
            ops.set_gemm_profiler(prof)
            for _ in range(2):
                step()
            ops.set_gemm_profiler(None)
            sm = prof.summary()
            peak = PEAK_BF16
            res["roofline"] = dict(bound="mfma", unit="TFLOP/s", peak=peak, achieved=round(sm["flops"] / (sm["ms"] * 1e-3) / 1e12, 2),
                                   frac=round(sm["flops"] / (sm["ms"] * 1e-3) / 1e12 / peak, 4), launches_per_step=sm["launches"] // 2,
                                   gemm_ms_per_step=round(sm["ms"] / 2, 3), algorithmic_gflop_per_step=round(sm["flops"] / 2 / 1e9, 1),
                                   kernel="all MFMA GEMM launches of a step (towers serialised, HIP events on the launch stream), "
                                          "priced against the dense bf16 peak")
            if precision == "fp8" and sm["fp8"]["launches"]:
                f8 = sm["fp8"]
                a8 = f8["flops"] / (f8["ms"] * 1e-3) / 1e12
                res["roofline"]["fp8_family"] = dict(achieved=round(a8, 2), peak=PEAK_FP8, frac=round(a8 / PEAK_FP8, 4),
                                                     launches_per_step=f8["launches"] // 2, ms_per_step=round(f8["ms"] / 2, 3),
                                                     kernel="the transformer-block GEMMs on OCP fp8 operands, priced against the dense "
                                                            "fp8 peak")
        del m, dd, o, sc
        torch.cuda.empty_cache()
        return res

    legs_extra = {}
    if not args.no_extra_legs and args.model == "vitb32" and args.precision == "bf16" and not args.all_text_positions:
        # BASELINE configs[4] is quoted at global batch 4096 on 8 GPUs = 512 per GPU; the 256 leg is the like-for-like
        # comparison with the bf16 headline
        legs_extra["fp8"] = extra_leg("vitb32", "fp8", 512)
        legs_extra["fp8_b256"] = extra_leg("vitb32", "fp8", args.batch)
        legs_extra["vitl14"] = extra_leg("vitl14", "bf16", 128)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.model == "vitb32" and args.precision != "fp8":
        cpu = cpu_baseline()
    if world > 1:
        dist.barrier()
    if rank == 0:
        out = {
            "metric": "image-text pairs/sec (whole node), %s + FDT, global batch %d x ngpu" % (
                "ViT-B/32" if args.model == "vitb32" else "ViT-L/14", args.batch),
            "value": round(value, 1), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "ms_per_step_event_median": round(ev_median, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "towers": "serial" if args.serial_towers else "concurrent (one HIP stream per tower + a weight-gradient companion stream each)",
            "optimizer": "fused AdamW, one launch in step()" if not args.overlap_adamw else
                         "fused AdamW inside the timed step: per-block launches on their own stream as soon as a block's gradients are final, the rest in step()",
            "config": {"workload": "example/clip_fdt %s + FDT codebook (4096x512, sparsemax, max-pool, T=1000), "
                                   "%s compute / fp32 master weights, full train step incl. AdamW" % (
                                       "ViT-B/32" if args.model == "vitb32" else "ViT-L/14 (BASELINE configs[3])", args.precision),
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world, "image": "3x224x224",
                       "context_length": 77, "parallelism": "dp%d" % world,
                       "caption_tokens": "n ~ U{8..77} per caption (SURVEY.md 8d), %d valid of %d positions on rank 0" % (
                           sum(lens), len(lens) * 77),
                       "text_rows": "all positions" if args.all_text_positions else
                                    "valid tokens only (packed rows; positions behind <|endoftext|> never reach the loss)"},
            "timing": "value / ms_per_step: wall clock around %d steps between a barrier + device synchronise on both sides, max "
                      "over ranks (driver contract); ms_per_step_event_median: median of the per-step HIP-event intervals of the "
                      "same steps on rank 0 (SURVEY 8d; it asks for >= 50 steps: pass --steps 50)" % args.steps,
            "step_mfma_frac": (round(executed_flops / (ms_per_step * 1e-3) / (PEAK_BF16 * 1e12), 4) if executed_flops else None),
            "step_mfma_frac_note": "FLOPs EXECUTED per step (bf16 GEMM launches as timed + fp32 GEMMs + attention products on "
                                   "the rows computed) / step time / dense bf16 peak",
            "executed_gflop_per_step": (round(executed_flops / 1e9, 1) if executed_flops else None),
            "all_text_positions": dense,
            "fp8": legs_extra.get("fp8"), "fp8_b256": legs_extra.get("fp8_b256"), "vitl14": legs_extra.get("vitl14"),
            "rccl_world": rccl_world, "dist_backend": backend, "devices": devices, "comm": comm_stats,
            "final_loss": round(final_loss, 4), "host_enqueue_ms_per_step": round(host_idle_ms, 3),
            "host_loop_ms_per_step": round(1000.0 * host_dt / args.steps, 3),
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""The production RCCL branch of comm.py / NativeDDP on the hardware the test box has.

RCCL refuses two ranks on one device, so on a one-GPU box the `nccl` process group has world size 1 -- and comm.py's
collectives are skipped at W = 1 unless `comm.force_collectives(True)`: with the switch a world of one runs the SAME RCCL
calls (all_gather_into_tensor, reduce_scatter_tensor, all_reduce(AVG), reduce-scatter + all-gather buckets, bf16 buckets)
a world of eight does, and every result must equal the no-collective path.  With >= 2 GPUs visible the two-rank test
reproduces the reference's 2-rank golden (G4) over RCCL itself."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(rank, world, port, dev):
    for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(dev), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    return dist


def _step(precision, ddp_wrap, seed_w=11):
    from configs import CFG, FDT_VARIANTS, model_kwargs, state_shapes
    from detfill import det_state, det_images, det_tokens
    from ilvlm_amd.prototype.model import model_entry
    from ilvlm_amd.prototype.loss_functions import ClipInfoCELoss
    from ilvlm_amd.prototype.utils.torch_ddp_dist import convert_to_ddp_model
    c, v = CFG["a"], FDT_VARIANTS[0]
    kw = model_kwargs(c, v)
    kw["precision"] = precision
    model = model_entry(dict(type="clip_fdt_vitb32", kwargs=kw))
    model.load_state_dict({k: torch.from_numpy(a) for k, a in det_state(state_shapes(c, True), seed_w).items()})
    model.cuda().train()
    net = convert_to_ddp_model(model, 0) if ddp_wrap else model
    img = torch.from_numpy(det_images(c["batch"], c["res"], 111)).cuda()
    tok, mask = det_tokens(c["batch"], c["ctx"], 111)
    (li, lt), _ = net(img, (torch.from_numpy(tok), torch.from_numpy(mask)))
    loss, _ = ClipInfoCELoss()(li, lt)
    model.zero_grad()
    loss.backward()
    model.engine.arena.wait_grads()
    torch.cuda.synchronize()
    return model, li.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters()}


def _w1_worker(rank, world, port, ret):
    dist = _setup(rank, world, port, 0)
    from ilvlm_amd import comm
    try:
        out = {}
        comm.force_collectives(True)
        # --- the embedding exchange: one fused all-gather, one reduce-scatter
        g = torch.Generator().manual_seed(3)
        img, txt = torch.randn(6, 32, generator=g).cuda(), torch.randn(6, 32, generator=g).cuda()
        gi, gt = comm.gather_pair(img, txt)
        out["gather"] = bool(torch.equal(gi, img) and torch.equal(gt, txt) and gi.data_ptr() != img.data_ptr())
        di, dt = comm.reduce_gathered(gi * 2, gt * 3, 6)
        out["reduce_scatter"] = bool(torch.equal(di, img * 2) and torch.equal(dt, txt * 3))
        # --- gradient buckets: fp32 / bf16, all-reduce / reduce-scatter + all-gather, ragged range (not a multiple of W or 4)
        flat0 = torch.randn(3 * 64 * 1000 + 64, generator=g).cuda()
        for bucket in ("fp32", "bf16"):
            for algo in ("allreduce", "rs_ag"):
                flat = flat0.clone()
                r = comm.GradReducer(flat, bucket=bucket, algo=algo)
                r.reduce_range(64, 64 + 64 * 999, chunk_elems=64 * 400)
                r.reduce_range(64 * 1000, flat.numel())
                r.wait()
                torch.cuda.synchronize()
                want = flat0.clone()
                if bucket == "bf16":
                    want[64:64 + 64 * 999] = want[64:64 + 64 * 999].bfloat16().float()
                    want[64 * 1000:] = want[64 * 1000:].bfloat16().float()
                out["bucket_%s_%s" % (bucket, algo)] = bool(torch.equal(flat, want)) and r.bytes_sent == (
                    (64 * 999 + flat.numel() - 64 * 1000) * (2 if bucket == "bf16" else 4))
        # --- a full data-parallel step through NativeDDP (init broadcast, per-block reductions from two tower streams, the
        #     embedding exchange inside forward / backward) equals the step without the wrapper
        for precision in ("fp32", "bf16", "bf16_optin"):
            # bf16 mode reduces fp32 buckets by default (the reference's arithmetic); ILVLM_GRAD_BUCKET=bf16 opts in
            os.environ.pop("ILVLM_GRAD_BUCKET", None)
            if precision == "bf16_optin":
                os.environ["ILVLM_GRAD_BUCKET"] = "bf16"
            prec = precision.split("_")[0]
            comm.force_collectives(False)
            _, li0, g0 = _step(prec, False)
            comm.force_collectives(True)
            model, li1, g1 = _step(prec, True)
            os.environ.pop("ILVLM_GRAD_BUCKET", None)
            same = bool(torch.equal(li0, li1))
            worst = 0.0
            for n in g0:
                scale = max(float(g0[n].abs().max()), 1e-12)
                worst = max(worst, float((g1[n] - g0[n]).abs().max()) / scale)
            out["ddp_%s" % precision] = (same, worst, model.engine.arena.reducer.bucket, model.engine.arena.reducer.bytes_sent)
        ret[rank] = out
    finally:
        comm.force_collectives(False)
        dist.destroy_process_group()


def test_rccl_collective_branch_at_world_size_one():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_w1_worker, args=(1, 29571, ret), nprocs=1, join=True)
    out = dict(ret)[0]
    assert out["gather"] and out["reduce_scatter"]
    for k in ("bucket_fp32_allreduce", "bucket_fp32_rs_ag", "bucket_bf16_allreduce", "bucket_bf16_rs_ag"):
        assert out[k], k
    same, worst, bucket, sent = out["ddp_fp32"]
    assert same and bucket == "fp32" and sent > 0
    assert worst < 1e-4, worst            # fp32 atomics order only (the mean over one rank is the identity)
    same, worst, bucket, sent32 = out["ddp_bf16"]
    assert same and bucket == "fp32" and sent32 > 0     # default in bf16 mode: fp32 buckets
    assert worst < 2e-2, worst            # atomics order of bf16-rounded products only
    same, worst, bucket, sent = out["ddp_bf16_optin"]
    assert same and bucket == "bf16" and 0 < sent == sent32 // 2
    assert worst < 2.4e-2, worst          # + one bf16 rounding of every gradient element (2^-8 relative)


def _w2_worker(rank, world, port, ret):
    dist = _setup(rank, world, port, rank)
    from configs import CFG, FDT_VARIANTS, model_kwargs, state_shapes
    from detfill import det_state, det_images, det_tokens, probe
    from ilvlm_amd.prototype.model import model_entry
    from ilvlm_amd.prototype.loss_functions import ClipInfoCELoss
    from ilvlm_amd.prototype.utils.torch_ddp_dist import convert_to_ddp_model
    c, v = CFG["a"], FDT_VARIANTS[0]
    kw = model_kwargs(c, v)
    kw["precision"] = "fp32"
    model = model_entry(dict(type="clip_fdt_vitb32", kwargs=kw))
    seed_w = 11 if rank == 0 else 12345
    model.load_state_dict({k: torch.from_numpy(a) for k, a in det_state(state_shapes(c, True), seed_w).items()})
    model.cuda().train()
    ddp = convert_to_ddp_model(model, rank)
    seed = 11 + 100 + rank
    img = torch.from_numpy(det_images(c["batch"], c["res"], seed)).cuda()
    tok, mask = det_tokens(c["batch"], c["ctx"], seed)
    (li, lt), _ = ddp(img, (torch.from_numpy(tok), torch.from_numpy(mask)))
    loss, labels = ClipInfoCELoss()(li, lt)
    loss = loss / world
    model.zero_grad()
    loss.backward()
    model.engine.arena.wait_grads()
    torch.cuda.synchronize()
    out = {"logits_i": li.detach().cpu().numpy(), "labels": labels.cpu().numpy(), "loss": float(loss)}
    for name, p in model.named_parameters():
        out["grad." + name] = probe(name, p.grad.detach().cpu().numpy())
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks per device)")
def test_two_ranks_over_rccl_match_reference_ddp(golden_dir):
    g = np.load(os.path.join(golden_dir, "g4_two_rank_a.npz"))
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_w2_worker, args=(2, 29573, ret), nprocs=2, join=True)
    ret = dict(ret)
    for r in range(2):
        ref_i = g["r%d.logits_i" % r]
        assert np.abs(ret[r]["logits_i"] - ref_i).max() < 1e-3 * np.abs(ref_i).max()
        np.testing.assert_array_equal(ret[r]["labels"], g["r%d.labels" % r])
        assert abs(ret[r]["loss"] - float(g["r%d.loss" % r])) < 1e-3 * abs(float(g["r%d.loss" % r]))
    for k in g.files:
        if k.startswith("grad."):
            want = g[k][2:]
            scale = max(np.abs(want).max(), 1e-30)
            for r in range(2):
                assert np.abs(ret[r][k][2:] - want).max() <= 1e-3 * scale + 1e-8, (k, r)

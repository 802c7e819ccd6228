"""The N>1 GPU product path (NativeDDP + comm + engine), exercised with two ranks that share the one GPU of the test
box over gloo (RCCL refuses two ranks on one device; the driver runs the real xGMI scaling).  Checked against the
reference's 2-rank AllGather + torch-DDP golden (G4): logits, rank-offset labels, loss / W, post-all-reduce gradients."""
import json
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from configs import CFG, FDT_VARIANTS, model_kwargs, state_shapes
    from detfill import det_state, det_images, det_tokens, probe
    from ilvlm_amd.prototype.model import model_entry
    from ilvlm_amd.prototype.loss_functions import ClipInfoCELoss
    from ilvlm_amd.prototype.utils.torch_ddp_dist import convert_to_ddp_model
    c, v = CFG["a"], FDT_VARIANTS[0]
    kw = model_kwargs(c, v)
    kw["precision"] = "fp32"
    model = model_entry(dict(type="clip_fdt_vitb32", kwargs=kw))
    seed_w = 11 if rank == 0 else 12345          # rank 1 starts from different weights: the init broadcast must fix that
    model.load_state_dict({k: torch.from_numpy(a) for k, a in det_state(state_shapes(c, True), seed_w).items()})
    model.cuda().train()
    ddp = convert_to_ddp_model(model, 0)
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
    out = {"logits_i": li.detach().cpu().numpy(), "logits_t": lt.detach().cpu().numpy(), "labels": labels.cpu().numpy(),
           "loss": float(loss)}
    for name, p in model.named_parameters():
        out["grad." + name] = probe(name, p.grad.detach().cpu().numpy())
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_reference_ddp(golden_dir):
    g = np.load(os.path.join(golden_dir, "g4_two_rank_a.npz"))
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, 29561, ret), nprocs=2, join=True)
    ret = dict(ret)
    for r in range(2):
        ref_i, ref_t = g["r%d.logits_i" % r], g["r%d.logits_t" % r]
        assert np.abs(ret[r]["logits_i"] - ref_i).max() < 1e-3 * np.abs(ref_i).max()
        assert np.abs(ret[r]["logits_t"] - ref_t).max() < 1e-3 * np.abs(ref_t).max()
        np.testing.assert_array_equal(ret[r]["labels"], g["r%d.labels" % r])
        assert abs(ret[r]["loss"] - float(g["r%d.loss" % r])) < 1e-3 * abs(float(g["r%d.loss" % r]))
    n = 0
    for k in g.files:
        if not k.startswith("grad."):
            continue
        want = g[k][2:]
        scale = max(np.abs(want).max(), 1e-30)
        for r in range(2):          # both ranks hold the same averaged gradient
            assert np.abs(ret[r][k][2:] - want).max() <= 1e-3 * scale + 1e-8, (k, r)
        n += 1
    assert n > 60

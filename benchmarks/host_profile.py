"""Where the HOST time of one train step goes (bench.py's step, enqueued into an idle GPU): cProfile over a few steps, top functions
by own time and by cumulative time.  usage: host_profile.py [--precision fp8]"""
import sys, os, cProfile, pstats, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as BN
from ilvlm_amd import ops
from ilvlm_amd.prototype.model import model_entry
from ilvlm_amd.prototype.loss_functions import ClipInfoCELoss
from ilvlm_amd.prototype.optimizer import optim_entry
from ilvlm_amd.prototype.utils.misc import param_group_all, accuracy

precision = "fp8" if "fp8" in sys.argv else "bf16"
torch.manual_seed(0)
model = model_entry(dict(type="clip_fdt_vitb32", kwargs=BN.fdt_kwargs(precision))).cuda().train()
opt = optim_entry(dict(type="AdamW", kwargs=dict(params=param_group_all(model, BN.PCONFIG)[0], lr=5e-5, weight_decay=0.1, betas=[0.9, 0.98],
                                                 eps=1e-8)))
opt.prezero_grads = True
crit = ClipInfoCELoss()
images, tokens, pad, lens = BN.synthetic_batch(256, 0, "cuda")
texts = (tokens, pad, ops.PackedSeq(lens, tokens.shape[1], "cuda"))


def one_step():
    (li, lt), _ = model(images, texts)
    loss, target = crit(li, lt)
    accuracy(li, target, topk=(1, 5))
    opt.zero_grad()
    ops.clamp_(model.logit_scale.data, 3, 6)
    loss.backward()
    opt.step()
    ops.clamp_(model.logit_scale.data, 3, 6)


for _ in range(5):
    one_step()
torch.cuda.synchronize()
hs = []
for _ in range(7):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); one_step(); hs.append(1e3 * (time.perf_counter() - t0))
print("host enqueue of one step into an idle GPU: median %.3f ms (%s)" % (sorted(hs)[3], " ".join("%.2f" % h for h in hs)))
pr = cProfile.Profile()
N = 10
for _ in range(N):
    torch.cuda.synchronize()
    pr.enable(); one_step(); pr.disable()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
    print("==== by %s (%d steps; divide by %d for one step; cProfile inflates python-level calls)" % (key, N, N))
    print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:6000])

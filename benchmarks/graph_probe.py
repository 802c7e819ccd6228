"""Feasibility probe for hipGraph capture of the train step, stage by stage, one process per stage:
    1  one streaming-kernel GEMM            2  LayerNorm + attention launches       3  model forward (no grad; two tower streams)
    4  forward + loss + backward (companion weight-gradient streams)               5  the whole step incl. AdamW
Each stage warms up eagerly, captures, replays and compares with the eager result."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import faulthandler; faulthandler.enable()
import torch
import bench as BN
from ilvlm_amd import ops


_KEPT_EVENTS = []


def keep_wait_stream_events():
    """--keep-events: torch's Stream.wait_stream records a temporary Event on the other stream, waits for it and drops it at once
    (hipEventDestroy while the capture is still open).  This variant keeps every such event alive until the process ends."""
    def wait_stream(self, other):
        ev = torch.cuda.Event()
        ev.record(other)
        self.wait_event(ev)
        _KEPT_EVENTS.append(ev)
    torch.cuda.Stream.wait_stream = wait_stream


def capture(fn, warm=3):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warm):
            out = fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        out = fn()
    torch.cuda.synchronize()
    return g, out


def model_and_batch(batch=64):
    from ilvlm_amd.prototype.model import model_entry
    from ilvlm_amd.prototype.utils import torch_ddp_dist as D
    D.set_random_seed(0)
    model = model_entry(dict(type="clip_fdt_vitb32", kwargs=BN.fdt_kwargs("bf16")))
    model.cuda().train()
    images, tokens, pad, lens = BN.synthetic_batch(batch, 0, "cuda")
    return model, images, (tokens, pad, ops.PackedSeq(lens, tokens.shape[1], "cuda"))


def main(stage):
    if "--keep-events" in sys.argv:
        keep_wait_stream_events()
    torch.manual_seed(0)
    if stage == 1:
        a = torch.randn(1024, 768, device="cuda").to(torch.bfloat16)
        w = torch.randn(768, 768, device="cuda").to(torch.bfloat16)
        bp = ops.gemm_pack_b(w)
        out = torch.empty(1024, 768, dtype=torch.bfloat16, device="cuda")
        g, _ = capture(lambda: ops.gemm(a, w, out, b_packed=bp))
        ref = out.clone(); out.zero_(); g.replay(); torch.cuda.synchronize()
        print("stage 1 ok, replay equals eager:", torch.equal(out, ref), flush=True)
    elif stage == 2:
        x = torch.randn(1000, 768, device="cuda")
        gam, bet = torch.ones(768, device="cuda"), torch.zeros(768, device="cuda")
        y = torch.empty(1000, 768, dtype=torch.bfloat16, device="cuda")
        mean, rstd = torch.empty(1000, device="cuda"), torch.empty(1000, device="cuda")
        qkv = torch.randn(20 * 50, 3 * 768, device="cuda").to(torch.bfloat16)
        att = torch.empty(20 * 50, 768, dtype=torch.bfloat16, device="cuda")
        lse = torch.empty(20, 12, 50, device="cuda")

        def fn():
            ops.layernorm_fwd(x, gam, bet, y, mean, rstd, 1000, 768)
            ops.attention_fwd(qkv, att, lse, 20, 50, 12, 0)
        g, _ = capture(fn)
        ref = att.clone(); att.zero_(); g.replay(); torch.cuda.synchronize()
        print("stage 2 ok, replay equals eager:", torch.equal(att, ref), flush=True)
    else:
        from ilvlm_amd.prototype.loss_functions import ClipInfoCELoss
        from ilvlm_amd.prototype.optimizer import optim_entry
        from ilvlm_amd.prototype.utils.misc import param_group_all
        model, images, texts = model_and_batch()
        crit = ClipInfoCELoss()
        opt = optim_entry(dict(type="AdamW", kwargs=dict(params=param_group_all(model, BN.PCONFIG)[0], lr=1e-4, weight_decay=0.1,
                                                         betas=[0.9, 0.98], amsgrad=False, eps=1e-8)))

        def fwd():
            with torch.no_grad():
                (li, lt), _ = model(images, texts)
            return li

        def fwd_bwd():
            (li, lt), _ = model(images, texts)
            loss, _ = crit(li, lt)
            model.zero_grad()
            loss.backward()
            return loss

        def step():
            (li, lt), _ = model(images, texts)
            loss, _ = crit(li, lt)
            opt.zero_grad()
            ops.clamp_(model.logit_scale.data, 3, 6)
            loss.backward()
            opt.step()
            ops.clamp_(model.logit_scale.data, 3, 6)
            return loss
        fn = {3: fwd, 4: fwd_bwd, 5: step}[stage]
        if stage >= 4 and model.engine.wgrad_streams and "--force" not in sys.argv:
            # Recorded in profiles/round3/graph_probe.txt: with the companion weight-gradient streams, torch's capture_end
            # (hipStreamEndCapture) segfaults.  Round 4 found why: the HIP runtime the torch
            # wheel bundles (roc-7.0.2) recurses without end in hip::Stream::EndCapture() when a stream forked from the capture's
            # origin has itself forked a third stream -- the text tower's stream and its weight-gradient companion.  Pure-HIP
            # reproducer: benchmarks/micro/graph_fork_probe.hip (passes on /opt/rocm's 7.2 runtime, crashes on the bundled one:
            # profiles/round4/graph_fork_probe_both_runtimes.txt; native backtrace: graph_probe_capture_end_backtrace.txt).
            # Graphs are not shipped (a replay costs the host what the eager step costs it), so the probe refuses this
            # configuration instead of crashing.
            print("stage %d with companion weight-gradient streams crashes hipStreamEndCapture of the HIP runtime torch bundles "
                  "(roc-7.0.2: nested stream forks, profiles/round4/graph_fork_probe_both_runtimes.txt); run with ILVLM_WGRAD_STREAMS=0, or pass --force to try anyway" % stage)
            sys.exit(3)
        g, out = capture(fn)
        vals = []
        for _ in range(3):
            g.replay()
            vals.append(float(out.detach().float().flatten()[0]))
        torch.cuda.synchronize()
        print("stage %d ok, replayed values:" % stage, vals, flush=True)
        n = 30
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            g.replay()
        host = time.perf_counter() - t0
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("   graph replay %.3f ms / iteration on the GPU, %.3f ms of host time" % (dt / n * 1e3, host / n * 1e3), flush=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            fn()
        host = time.perf_counter() - t0
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("   eager        %.3f ms / iteration on the GPU, %.3f ms of host time" % (dt / n * 1e3, host / n * 1e3), flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]))

"""Correctness + speed check of the 256x256 phased GEMM (variant 8) against the 128x128 kernel (variant 5) on the step's
shapes.  Interleaved rounds in one process, cold-ish caches (a 512 MB buffer is rewritten between launches)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ilvlm_amd import ops


def check():
    torch.manual_seed(0)
    bad = 0
    for (M, N, K) in [(512, 256, 64), (512, 512, 128), (1000, 768, 512), (776, 520, 192), (2304, 768, 1000)]:
        for ta in (0, 1):
            for tb in (0, 1):
                if K % 64 and not (ta and tb):
                    continue
                a = torch.randn(M, K).to(torch.bfloat16); b = torch.randn(N, K).to(torch.bfloat16)
                want = a.float() @ b.float().t()
                A = (a.t() if ta else a).contiguous().cuda(); B = (b.t() if tb else b).contiguous().cuda()
                ops.gemm_set_variant(8)
                out = torch.full((M, N), float("nan"), device="cuda")
                ops.gemm(A, B, out, trans_a=bool(ta), trans_b=bool(tb))
                e1 = float((out.cpu() - want).abs().max() / want.abs().max())
                acc = torch.ones(M, N, device="cuda"); rs = torch.ones(M, device="cuda")
                ops.gemm(A, B, acc, trans_a=bool(ta), trans_b=bool(tb), accumulate=True, split_k=3, a_rowsum=rs if M % 8 == 0 else None)
                e2 = float((acc.cpu() - want - 1).abs().max() / want.abs().max())
                e3 = float((rs.cpu() - 1 - a.float().sum(1)).abs().max() / a.float().sum(1).abs().max()) if M % 8 == 0 else 0.0
                ok = e1 < 2e-5 and e2 < 2e-5 and e3 < 2e-5
                bad += not ok
                print("check M=%d N=%d K=%d ta=%d tb=%d  plain %.1e  acc %.1e  rowsum %.1e %s" % (M, N, K, ta, tb, e1, e2, e3, "ok" if ok else "FAIL"), flush=True)
    ops.gemm_set_variant(5)
    return bad


def bench(rounds=6):
    flush = torch.empty(128 * 1024 * 1024, device="cuda")
    shapes = []
    for tag, M, E in (("vit", 12800, 768), ("pk", 11319, 512)):
        for name, n, k in (("qkv", 3 * E, E), ("out", E, E), ("fc", 4 * E, E), ("proj", E, 4 * E)):
            shapes.append((tag + "." + name + ".fwd", 0, 0, M, n, k, False))
            shapes.append((tag + "." + name + ".dgrad", 0, 1, M, k, n, False))
            shapes.append((tag + "." + name + ".wgrad", 1, 1, n, k, M, True))
    shapes.append(("fdt.img.scores", 0, 0, 12544, 4096, 512, False))
    tot = {5: 0.0, 8: 0.0}
    flops = 0.0
    for (tag, ta, tb, M, N, K, acc) in shapes:
        a = torch.randn((K, M) if ta else (M, K), device="cuda").to(torch.bfloat16)
        b = torch.randn((K, N) if tb else (N, K), device="cuda").to(torch.bfloat16)
        out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if acc else torch.bfloat16)
        cfgs = [(5, ops.wgrad_split(M, N, K, 128) if acc else 1)]
        if acc:
            tiles = math.ceil(M / 256) * math.ceil(N / 256)
            for tgt in (128, 256):
                cfgs.append((8, max(1, min(16, round(tgt / tiles)))))
        else:
            cfgs.append((8, 1))
        best = {}
        for r in range(rounds):
            for (v, sp) in cfgs:
                ops.gemm_set_variant(v)
                flush.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.gemm(a, b, out, trans_a=bool(ta), trans_b=bool(tb), accumulate=acc, split_k=sp)
                e1.record()
                torch.cuda.synchronize()
                best[(v, sp)] = min(best.get((v, sp), 1e9), e0.elapsed_time(e1))
        fl = 2.0 * M * N * K
        flops += fl
        tot[5] += best[cfgs[0]]
        tot[8] += min(best[c] for c in cfgs[1:])
        print("%-16s M=%6d N=%5d K=%6d  " % (tag, M, N, K) +
              "  ".join("v%d/s%-2d %6.1f us %6.0f TF" % (v, sp, best[(v, sp)] * 1e3, fl / best[(v, sp)] / 1e9) for (v, sp) in cfgs), flush=True)
    ops.gemm_set_variant(5)
    print("sum: v5 %.1f us (%.0f TF/s)   v8-best %.1f us (%.0f TF/s)" % (tot[5] * 1e3, flops / tot[5] / 1e9, tot[8] * 1e3, flops / tot[8] / 1e9))


if __name__ == "__main__":
    bad = check()
    if bad:
        print("CORRECTNESS FAILURES: %d" % bad)
        sys.exit(1)
    bench()

"""Micro-benchmark of the bf16 GEMM variants on the shapes of one ViT-B/32 + text block at per-GPU batch 256
(forward, dgrad, wgrad).  Interleaved rounds in one process; prints TFLOP/s per shape and variant."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ilvlm_amd import ops

SHAPES = []   # (tag, ta, tb, M, N, K, accumulate, split)
for tag, M, E in (("vit", 12800, 768), ("txt", 19712, 512)):
    for name, n, k in (("qkv", 3 * E, E), ("out", E, E), ("fc", 4 * E, E), ("proj", E, 4 * E)):
        SHAPES.append((tag + "." + name + ".fwd", 0, 0, M, n, k, False, 1))
        SHAPES.append((tag + "." + name + ".dgrad", 0, 1, M, k, n, False, 1))
        SHAPES.append((tag + "." + name + ".wgrad", 1, 1, n, k, M, True, ops.wgrad_split(n, k, M)))
for (tag, M, N, K) in [("pk.qkv.fwd", 11319, 1536, 512), ("pk.out.fwd", 11319, 512, 512), ("pk.fc.fwd", 11319, 2048, 512),
                       ("pk.proj.fwd", 11319, 512, 2048)]:
    SHAPES.append((tag, 0, 0, M, N, K, False, 1))
SHAPES.append(("fdt.img.scores", 0, 0, 12544, 4096, 512, False, 1))
SHAPES.append(("fdt.txt.scores", 0, 0, 19712, 4096, 512, False, 1))


FLUSH = None


def run(variants=(5, 6, 7, 9), rounds=5, only=None):
    global FLUSH
    torch.manual_seed(0)
    FLUSH = torch.empty(128 * 1024 * 1024, device="cuda")
    res = {}
    for (tag, ta, tb, M, N, K, acc, split) in SHAPES:
        if only and only not in tag:
            continue
        a = torch.randn((K, M) if ta else (M, K), device="cuda").to(torch.bfloat16)
        b = torch.randn((K, N) if tb else (N, K), device="cuda").to(torch.bfloat16)
        out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if acc else torch.bfloat16)
        for v in variants:
            ops.gemm_set_variant(v)
            ops.gemm(a, b, out, trans_a=bool(ta), trans_b=bool(tb), accumulate=acc, split_k=split)
        torch.cuda.synchronize()
        best = {v: 1e9 for v in variants}
        for r in range(rounds):
            for v in variants:
                ops.gemm_set_variant(v)
                FLUSH.zero_()            # cold caches, as inside a train step
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.gemm(a, b, out, trans_a=bool(ta), trans_b=bool(tb), accumulate=acc, split_k=split)
                e1.record()
                torch.cuda.synchronize()
                best[v] = min(best[v], e0.elapsed_time(e1))
        fl = 2.0 * M * N * K
        res[tag] = {v: fl / (best[v] * 1e-3) / 1e12 for v in variants}
        print("%-18s M=%6d N=%5d K=%6d split=%2d  " % (tag, M, N, K, split) +
              "  ".join("v%d %7.1f TF/s (%6.1f us)" % (v, res[tag][v], best[v] * 1e3) for v in variants), flush=True)
    ops.gemm_set_variant(5)
    return res


if __name__ == "__main__":
    run(only=sys.argv[1] if len(sys.argv) > 1 else None)

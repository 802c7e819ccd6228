"""fp8 vs bf16 direct-to-LDS GEMM on the forward / input-gradient shapes of the step (cold caches)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ilvlm_amd import ops

flush = torch.empty(128 * 1024 * 1024, device="cuda")
one = torch.ones(1, device="cuda")


def t_of(fn, rounds=6):
    best = 1e9
    for _ in range(rounds):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best * 1e3


# round 4: the streaming kernel on fragment-order fp8 weights against the direct-to-LDS fp8 kernel, per shape, at the fp8
# configuration's per-GPU batch 512 (rows 25600 / ~22600) and at 256; with the fc epilogue (bias + QuickGELU + pre-activation) on the
# up-projection.  usage: fp8_gemm_bench.py pk
if len(sys.argv) > 1 and sys.argv[1] == "pk":
    for batch in (512, 256):
        tot = [0.0, 0.0]
        for tag, M, E in (("vit", 50 * batch, 768), ("txt", int(44.2 * batch), 512)):
            for name, n, k in (("qkv", 3 * E, E), ("out", E, E), ("fc", 4 * E, E), ("proj", E, 4 * E), ("qkv.dgrad", E, 3 * E),
                               ("out.dgrad", E, E), ("fc.dgrad", E, 4 * E), ("proj.dgrad", 4 * E, E)):
                a8 = torch.randint(0, 120, (M, k), device="cuda", dtype=torch.uint8); w8 = torch.randint(0, 120, (n, k), device="cuda", dtype=torch.uint8)
                wp = ops.gemm_pack_b8(w8)
                out = torch.empty(M, n, device="cuda", dtype=torch.bfloat16)
                kw = dict(bias=torch.randn(n, device="cuda"), aux=torch.empty(M, n, device="cuda", dtype=torch.bfloat16), act=1) if name == "fc" else {}
                e5 = "dgrad" in name
                td = t_of(lambda: ops.gemm_fp8(a8, w8, out, one, one, a_e5m2=e5, **kw))
                tp = t_of(lambda: ops.gemm_fp8(a8, w8, out, one, one, a_e5m2=e5, b_packed=wp, **kw))
                fl = 2.0 * M * n * k
                tot[0] += td; tot[1] += tp
                print("b%d %-14s M=%6d N=%5d K=%5d  direct-to-LDS %6.1f us %5.0f TF   streaming %6.1f us %5.0f TF   x%.2f" % (
                    batch, tag + "." + name, M, n, k, td, fl / td / 1e6, tp, fl / tp / 1e6, td / tp), flush=True)
        print("batch %d: sum direct-to-LDS %.0f us, streaming %.0f us, x%.2f" % (batch, tot[0], tot[1], tot[0] / tot[1]))
    sys.exit(0)

tot = [0.0, 0.0]
for tag, M, E in (("vit", 12800, 768), ("pk", 11319, 512)):
    for name, n, k in (("qkv", 3 * E, E), ("out", E, E), ("fc", 4 * E, E), ("proj", E, 4 * E), ("qkv.dgrad", E, 3 * E), ("fc.dgrad", E, 4 * E),
                       ("proj.dgrad", 4 * E, E)):
        a = torch.randn(M, k, device="cuda").to(torch.bfloat16); w = torch.randn(n, k, device="cuda").to(torch.bfloat16)
        a8 = torch.randint(0, 120, (M, k), device="cuda", dtype=torch.uint8); w8 = torch.randint(0, 120, (n, k), device="cuda", dtype=torch.uint8)
        out = torch.empty(M, n, device="cuda", dtype=torch.bfloat16)
        tb = t_of(lambda: ops.gemm(a, w, out))
        t8 = t_of(lambda: ops.gemm_fp8(a8, w8, out, one, one))
        fl = 2.0 * M * n * k
        tot[0] += tb; tot[1] += t8
        print("%-14s M=%6d N=%5d K=%5d  bf16 %6.1f us %5.0f TF   fp8 %6.1f us %5.0f TF   x%.2f" % (
            tag + "." + name, M, n, k, tb, fl / tb / 1e6, t8, fl / t8 / 1e6, tb / t8), flush=True)
print("sum bf16 %.0f us, fp8 %.0f us, x%.2f" % (tot[0], tot[1], tot[0] / tot[1]))

# weight gradients: dW[out, in] += dY^T X, both operands K-strided, split-K atomics
tot = [0.0, 0.0]
for tag, M, E in (("vit", 12800, 768), ("pk", 11319, 512)):
    for name, n_out, n_in in (("qkv", 3 * E, E), ("out", E, E), ("fc", 4 * E, E), ("proj", E, 4 * E)):
        dy = torch.randn(M, n_out, device="cuda").to(torch.bfloat16); x = torch.randn(M, n_in, device="cuda").to(torch.bfloat16)
        dy8 = torch.randint(0, 120, (M, n_out), device="cuda", dtype=torch.uint8)
        x8 = torch.randint(0, 120, (M, n_in), device="cuda", dtype=torch.uint8)
        dw = torch.zeros(n_out, n_in, device="cuda"); db = torch.zeros(n_out, device="cuda")
        sk = ops.wgrad_split(n_out, n_in, M)
        tb = t_of(lambda: ops.gemm(dy, x, dw, trans_a=True, trans_b=True, accumulate=True, split_k=sk, a_rowsum=db))
        best = (1e9, 0)
        for s8 in sorted({sk, max(1, sk // 2), min(16, sk * 2)}):
            t = t_of(lambda: ops.gemm_fp8_wgrad(dy8, x8, dw, one, one, split_k=s8, rowsum=db))
            best = min(best, (t, s8))
        t8 = t_of(lambda: ops.gemm_fp8_wgrad(dy8, x8, dw, one, one, split_k=sk, rowsum=db))
        fl = 2.0 * M * n_out * n_in
        tot[0] += tb; tot[1] += t8
        print("%-14s out=%5d in=%5d T=%6d split %2d  bf16 %6.1f us %5.0f TF   fp8 %6.1f us %5.0f TF  x%.2f  (best fp8 %.1f us at split %d)" % (
            tag + "." + name + ".wgrad", n_out, n_in, M, sk, tb, fl / tb / 1e6, t8, fl / t8 / 1e6, tb / t8, best[0], best[1]), flush=True)
print("wgrad sum bf16 %.0f us, fp8 %.0f us, x%.2f" % (tot[0], tot[1], tot[0] / tot[1]))

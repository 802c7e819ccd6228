"""K-scaling of the GEMM variants on forward shapes: time(K) = fixed (prologue + epilogue) + per-K-tile cost."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ilvlm_amd import ops

flush = torch.empty(128 * 1024 * 1024, device="cuda")
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [5, 8]


def t_of(fn, rounds=6):
    best = 1e9
    for _ in range(rounds):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best * 1e3


for (M, N, epi) in [(12800, 3072, "bf16"), (12800, 3072, "act"), (12800, 768, "res"), (12800, 2304, "bias"), (11319, 2048, "act")]:
    for K in (64, 768, 3072):
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        if epi == "res":
            out = torch.empty(M, N, device="cuda"); res = torch.randn(M, N, device="cuda")
            fn = lambda: ops.gemm(a, w, out, bias=bias, residual=res)
        elif epi == "act":
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); aux = torch.empty_like(out)
            fn = lambda: ops.gemm(a, w, out, bias=bias, aux=aux, act=1)
        elif epi == "bias":
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            fn = lambda: ops.gemm(a, w, out, bias=bias)
        else:
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            fn = lambda: ops.gemm(a, w, out)
        line = "M=%d N=%d K=%4d %-4s " % (M, N, K, epi)
        for v in variants:
            ops.gemm_set_variant(v)
            us = t_of(fn)
            line += "  v%d %7.1f us %6.0f TF" % (v, us, 2.0 * M * N * K / us / 1e6)
        print(line, flush=True)
ops.gemm_set_variant(5)

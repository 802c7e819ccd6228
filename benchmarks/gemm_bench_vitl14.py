import sys, os
sys.path.insert(0, os.getcwd())
import torch
from ilvlm_amd import ops
torch.manual_seed(0)
flush = torch.empty(128 * 1024 * 1024, device="cuda")
M, E = 32896, 1024
for name, N, K, tb in (("qkv.fwd", 3 * E, E, 0), ("out.fwd", E, E, 0), ("fc.fwd", 4 * E, E, 0), ("proj.fwd", E, 4 * E, 0),
                       ("qkv.dgrad", E, 3 * E, 1), ("fc.dgrad", E, 4 * E, 1), ("proj.dgrad", 4 * E, E, 1)):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    b = torch.randn((K, N) if tb else (N, K), device="cuda").to(torch.bfloat16)
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    packed = ops.gemm_pack_b(b, trans_b=bool(tb))
    best = {}
    for r in range(6):
        for v in (5, 15):
            ops.gemm_set_variant(v)
            flush.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.gemm(a, b, out, trans_b=bool(tb), b_packed=packed if v == 15 else None)
            e1.record(); torch.cuda.synchronize()
            if r: best[v] = min(best.get(v, 1e9), e0.elapsed_time(e1))
    fl = 2.0 * M * N * K
    print("vitl14.%-10s N=%5d K=%5d  " % (name, N, K) + "  ".join("v%d %7.1f TF/s (%6.1f us)" % (v, fl / (t * 1e-3) / 1e12, t * 1e3) for v, t in best.items()), flush=True)
ops.gemm_set_variant(15)

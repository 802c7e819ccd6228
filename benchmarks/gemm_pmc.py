"""Runs one GEMM shape a few times per variant; meant to be wrapped by rocprofv3 --pmc (counter collection)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ilvlm_amd import ops
M, N, K = 12800, 3072, 768
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
for v in [int(x) for x in (sys.argv[1:] or ["5", "7"])]:
    ops.gemm_set_variant(v)
    for _ in range(3):
        ops.gemm(a, b, out)
torch.cuda.synchronize()

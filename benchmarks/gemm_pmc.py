"""Runs one GEMM shape a few times; meant to be wrapped by rocprofv3 --pmc (counter collection).
usage: gemm_pmc.py [fwd|wgrad] [variant ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ilvlm_amd import ops
kind = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].isdigit() else "fwd"
variants = [int(x) for x in sys.argv[1:] if x.isdigit()] or [5]
flush = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
if kind == "fwd":
    M, N, K = 12800, 3072, 768
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    run = lambda: ops.gemm(a, b, out)
else:       # weight gradient of the MLP up-projection: dW[3072, 768] += dY^T X over 12800 token rows
    T, O, I = 12800, 3072, 768
    dy = torch.randn(T, O, device="cuda").to(torch.bfloat16)
    x = torch.randn(T, I, device="cuda").to(torch.bfloat16)
    dw = torch.zeros(O, I, device="cuda")
    sk = int(os.environ.get("SPLIT", ops.wgrad_split(O, I, T)))
    run = lambda: ops.gemm(dy, x, dw, trans_a=True, trans_b=True, accumulate=True, split_k=sk)
for v in variants:
    ops.gemm_set_variant(v)
    for _ in range(3):
        flush.zero_()
        run()
torch.cuda.synchronize()

"""Launches the same forward GEMM 100 times per shape and counts outputs that differ from the first launch (must be 0:
the forward kernels have no atomics).  Caught the LDS race described at ILVLM_WG_BARRIER in csrc/gemm.hip."""
import sys, torch
sys.path.insert(0, "/root/repo")
import os
import ilvlm_amd.lib as L
if len(sys.argv) > 1:
    L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "libilvlm_hip%s.so" % sys.argv[1])
from ilvlm_amd import ops
from ilvlm_amd.lib import ACT_QUICKGELU
torch.manual_seed(0)
for (M, N, K) in [(19712, 2048, 512), (12800, 3072, 768), (11319, 2048, 512), (12800, 768, 3072)]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    ref_g = ref_u = None
    nbad = 0
    for it in range(100):
        u = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); g = torch.empty_like(u)
        ops.gemm(a, w, g, bias=bias, aux=u, act=ACT_QUICKGELU)
        torch.cuda.synchronize()
        if ref_g is None:
            ref_g, ref_u = g.clone(), u.clone()
        else:
            d = (g.float() - ref_g.float()).abs().max(1).values
            bad = torch.nonzero(d > 0).flatten()
            if bad.numel():
                nbad += 1
                if nbad <= 2: print("  run", it, "rows differ:", bad[:6].tolist(), "n", bad.numel(), "max", float(d.max()))
    print("M=%d N=%d K=%d: %d of 99 runs differ from the first" % (M, N, K, nbad), flush=True)

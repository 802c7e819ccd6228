"""Diagnostic: phase cycle breakdown of the pipelined GEMM main loop (needs the -DILVLM_GEMM_STAMPS build)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ilvlm_amd.lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "libilvlm_hip_stamps.so")
from ilvlm_amd import ops
import numpy as np
WN = int(os.environ.get("ILVLM_PK_WN", "2"))
CASES = [("fc.fwd", 0, 0, 12800, 3072, 768, False, 1, 5, 128, 128, 4), ("fc.fwd", 0, 0, 12800, 3072, 768, False, 1, 15, 128, 64 * WN, WN),
         ("fc.dgrad", 0, 1, 12800, 768, 3072, False, 1, 5, 128, 128, 4), ("fc.dgrad", 0, 1, 12800, 768, 3072, False, 1, 15, 128, 64 * WN, WN),
         ("pk.fc.fwd", 0, 0, 11319, 2048, 512, False, 1, 5, 128, 128, 4), ("pk.fc.fwd", 0, 0, 11319, 2048, 512, False, 1, 15, 128, 64 * WN, WN),
         ("fc.wgrad", 1, 1, 3072, 768, 12800, True, 3, 5, 128, 128, 4), ("fc.wgrad", 1, 1, 3072, 768, 12800, True, 3, 15, 128, 128, 4),
         ("fc.wgrad", 1, 1, 3072, 768, 12800, True, 2, 15, 128, 128, 4), ("pk.fc.wgrad", 1, 1, 2048, 512, 11319, True, 6, 15, 128, 128, 4)]
if len(sys.argv) > 1:
    CASES = [c for c in CASES if sys.argv[1] in c[0]]
for (tag, ta, tb, M, N, K, acc, split, v, bm, bn, nw) in CASES:
    a = torch.randn((K, M) if ta else (M, K), device="cuda").to(torch.bfloat16)
    b = torch.randn((K, N) if tb else (N, K), device="cuda").to(torch.bfloat16)
    out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if acc else torch.bfloat16)
    ops.gemm_set_variant(v)
    packed = ops.gemm_pack_b(b, trans_b=bool(tb)) if (v == 15 and not acc) else None
    slab = None
    if os.environ.get("STAMP_SK") and v == 15 and not acc:
        slab = (torch.empty(160 << 20, dtype=torch.uint8, device="cuda"), torch.zeros(8192, dtype=torch.int32, device="cuda"))
    flush = torch.empty(128 * 1024 * 1024, device="cuda")
    for _ in range(3):
        flush.zero_()
        ops.gemm(a, b, out, trans_a=bool(ta), trans_b=bool(tb), accumulate=acc, split_k=split, b_packed=packed, slab=slab)
    torch.cuda.synchronize()
    nb = min(4096, ((M + bm - 1) // bm) * ((N + bn - 1) // bn) * split)
    print(tag)
    buf = (ctypes.c_ulonglong * (nb * 8 * 6))()
    rc = L.load().ilvlm_debug_read_stamps(buf, nb * 8 * 6)
    arr = np.array(buf, dtype=np.float64).reshape(nb, 8, 6)[:, :nw]
    names = ["vmcnt wait", "barrier", "glds issue", "compute", "loop total", "epilogue"]
    print("variant", v, "tile %dx%d" % (bm, bn), "blocks", nb, "K-tiles per block", K // 64 // split)
    for i, n in enumerate(names):
        print("  %-12s mean %9.0f  p10 %9.0f  p90 %9.0f cycles/wave" % (n, arr[:, :, i].mean(), np.percentile(arr[:, :, i], 10), np.percentile(arr[:, :, i], 90)))
    print("  per wave index (mean loop total):", arr[:, :, 4].mean(0).round())
    print("  per wave index (mean barrier):", arr[:, :, 1].mean(0).round())

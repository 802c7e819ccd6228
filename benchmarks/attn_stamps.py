"""Diagnostic: where a workgroup of the attention backward spends its cycles (needs the -DILVLM_ATTN_STAMPS build:
hipcc ... -DILVLM_ATTN_STAMPS -c attention.hip, linked as libilvlm_hip_astamps.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ILVLM_LIB_SUFFIX"] = "_astamps"
import numpy as np
import torch
import ilvlm_amd.lib as L
from ilvlm_amd import ops

for (tag, B, Lq, H, causal) in [("vision", 256, 50, 12, 0), ("text", 256, 77, 8, 1)]:
    E = 64 * H
    rows = B * Lq
    qkv = torch.randn(rows, 3 * E, device="cuda").to(torch.bfloat16)
    dout = torch.randn(rows, E, device="cuda").to(torch.bfloat16)
    out = torch.empty(rows, E, device="cuda", dtype=torch.bfloat16)
    dqkv = torch.empty_like(qkv)
    lse = torch.zeros(B, H, Lq, device="cuda")
    ops.attention_fwd(qkv, out, lse, B, Lq, H, causal)
    for _ in range(3):
        ops.attention_bwd(dout, qkv, out, lse, dqkv, B, Lq, H, causal)
    torch.cuda.synchronize()
    nb = min(4096, B * H)
    nw = (Lq + 15) // 16
    buf = (ctypes.c_ulonglong * (nb * 8 * 6))()
    L.load().ilvlm_debug_read_attn_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    rc = L.load().ilvlm_debug_read_attn_stamps(buf, nb * 8 * 6)
    arr = np.array(buf, dtype=np.float64).reshape(nb, 8, 6)[:, :nw]
    names = ["issue loads + stage", "barrier", "pass A (dQ)", "pass B (dK, dV)", "store drain"]
    print(tag, "workgroups", nb, "waves", nw, "(s_memtime ticks of 100 MHz = 10 ns)")
    for i, n in enumerate(names):
        print("  %-20s mean %8.1f  p10 %8.1f  p90 %8.1f" % (n, arr[:, :, i].mean(), np.percentile(arr[:, :, i], 10), np.percentile(arr[:, :, i], 90)))
    t0 = arr[:, 0, 5]
    print("  launch span of workgroup starts: %.1f ticks; total per workgroup %.1f" % (t0.max() - t0.min(), arr[:, :, :5].sum(2).mean()))

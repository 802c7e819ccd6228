"""Diagnostic (needs `make stamps`): in-kernel cycle stamps of the persistent streaming kernel next to the one-tile form on the
step's store-type shapes.  Per wave: counted-wait, barrier, the wait in front of the epilogue, multiply phase (with the
interleaved load issue), epilogue; per workgroup the number of tiles it walked."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ilvlm_amd.lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "libilvlm_hip_stamps.so")
from ilvlm_amd import ops
import numpy as np

# tag, tb, M, N, K, epilogue
CASES = [("vit.fc.fwd", 0, 12800, 3072, 768, "gelu"), ("vit.qkv.fwd", 0, 12800, 2304, 768, "bias"), ("vit.proj.fwd", 0, 12800, 768, 3072, "res"),
         ("vit.fc.dgrad", 1, 12800, 768, 3072, "plain"), ("pk.fc.fwd", 0, 11319, 2048, 512, "gelu"), ("pk.qkv.fwd", 0, 11319, 1536, 512, "bias")]
if len(sys.argv) > 1:
    CASES = [c for c in CASES if sys.argv[1] in c[0]]
SLOTS = 512
for (tag, tb, M, N, K, epi) in CASES:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    b = torch.randn((K, N) if tb else (N, K), device="cuda").to(torch.bfloat16)
    packed = ops.gemm_pack_b(b, trans_b=bool(tb))
    kw = {}
    if epi in ("gelu", "bias", "res"):
        kw["bias"] = torch.randn(N, device="cuda")
    if epi == "gelu":
        kw.update(aux=torch.empty(M, N, device="cuda", dtype=torch.bfloat16), act=1)
    out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if epi == "res" else torch.bfloat16)
    if epi == "res":
        kw["residual"] = torch.randn(M, N, device="cuda")
    flush = torch.empty(128 * 1024 * 1024, device="cuda")
    tiles = ((M + 127) // 128) * ((N + 255) // 256)
    for v in (19, 18):
        ops.gemm_set_variant(v)
        ops.gemm_set_persistent(0, int(os.environ.get("STAMP_EPI_SEP", "-1")), int(os.environ.get("STAMP_STAGGER", "-1")))
        ts = []
        for _ in range(4):
            flush.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.gemm(a, b, out, trans_b=bool(tb), b_packed=packed, **kw)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        nb = min(4096, tiles if v == 19 else min(tiles, SLOTS))
        buf = (ctypes.c_ulonglong * (nb * 8 * 6))()
        L.load().ilvlm_debug_read_stamps(buf, nb * 8 * 6)
        arr = np.array(buf, dtype=np.float64).reshape(nb, 8, 6)[:, :4]
        print("%s  %s  M=%d N=%d K=%d  tiles %d  workgroups %d  K-tiles/tile %d   %.1f us (stamped build)" % (
            tag, "one tile per workgroup" if v == 19 else "persistent", M, N, K, tiles, nb, K // 64, min(ts)))
        if v == 19:
            names = ["vmcnt wait", "barrier", "-", "multiply", "loop total", "epilogue"]
        else:
            names = ["vmcnt wait", "barrier", "pre-epi wait", "multiply", "all but epi", "epilogue"]
        tpw = 1.0 if v == 19 else tiles / nb
        for i, n in enumerate(names):
            if n == "-":
                continue
            x = arr[:, :, i]
            print("    %-13s per workgroup: mean %9.0f  p10 %9.0f  p90 %9.0f   per tile: %8.0f cycles/wave" % (
                n, x.mean(), np.percentile(x, 10), np.percentile(x, 90), x.mean() / tpw))
        life = arr[:, :, 4] + arr[:, :, 5]
        print("    workgroup lifetime mean %.0f  max %.0f cycles; per tile %.0f" % (life.mean(), life.max(), life.mean() / tpw), flush=True)
ops.gemm_set_variant(15)

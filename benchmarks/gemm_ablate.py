"""Diagnostic: time the direct-to-LDS GEMM with one phase removed (needs `make -C .../csrc ablate`).
usage: python benchmarks/gemm_ablate.py          -> runs itself once per library (full, no-math, no-load, no-epilogue)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = {"": "full", "_abl1": "no MFMA/ds_read", "_abl2": "no operand DMA", "_abl3": "no epilogue", "_abl4": "no MFMA", "_abl5": "no ds_read"}
SHAPES = [("big.fwd", 0, 0, 12800, 3072, 3072, False, 1), ("fc.fwd", 0, 0, 12800, 3072, 768, False, 1), ("fc2.fwd", 0, 0, 12800, 768, 3072, False, 1),
          ("fc.dgrad", 0, 1, 12800, 768, 3072, False, 1), ("fc.wgrad", 1, 1, 3072, 768, 12800, True, 3),
          ("qkv.fwd", 0, 0, 12800, 2304, 768, False, 1), ("txt.fc.fwd", 0, 0, 19712, 2048, 512, False, 1),
          ("qkv.wgrad", 1, 1, 2304, 768, 12800, True, 4), ("out.wgrad", 1, 1, 768, 768, 12800, True, 11),
          ("txt.fc.wgrad", 1, 1, 2048, 512, 19712, True, 6), ("txt.out.wgrad", 1, 1, 512, 512, 19712, True, 16)]


def child(suffix):
    import torch
    import ilvlm_amd.lib as L
    L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "libilvlm_hip%s.so" % suffix)
    from ilvlm_amd import ops
    ops.gemm_set_variant(int(os.environ.get("ABL_VARIANT", "5")))
    for (tag, ta, tb, M, N, K, acc, split) in SHAPES:
        a = torch.randn((K, M) if ta else (M, K), device="cuda").to(torch.bfloat16)
        b = torch.randn((K, N) if tb else (N, K), device="cuda").to(torch.bfloat16)
        out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if acc else torch.bfloat16)
        flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
        ts = []
        for it in range(8):
            flush.fill_(it)           # cold L2 / Infinity Cache, as inside a train step
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.gemm(a, b, out, trans_a=bool(ta), trans_b=bool(tb), accumulate=acc, split_k=split)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        hot = []
        for it in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.gemm(a, b, out, trans_a=bool(ta), trans_b=bool(tb), accumulate=acc, split_k=split)
            e1.record()
            torch.cuda.synchronize()
            hot.append(e0.elapsed_time(e1) * 1e2)
        hot.sort()
        fl = 2.0 * M * N * K
        print("%-16s %-12s cold %7.1f us (%6.0f TF/s)   back-to-back %7.1f us (%6.0f TF/s)" % (
            NAMES[suffix], tag, ts[len(ts) // 2], fl / ts[len(ts) // 2] / 1e6, hot[len(hot) // 2], fl / hot[len(hot) // 2] / 1e6), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1] if sys.argv[1] != "full" else "")
    else:
        for sfx in NAMES:
            subprocess.run([sys.executable, os.path.abspath(__file__), sfx or "full"], check=True)

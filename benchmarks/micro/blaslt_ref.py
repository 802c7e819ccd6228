"""Reference point only (never used by the product path): what the vendor GEMM library reaches on the step's shapes,
measured the same way as benchmarks/gemm_ablate.py (after a 512 MB flush, and back to back)."""
import torch
SHAPES = [("fc.fwd", 0, 0, 12800, 3072, 768), ("fc2.fwd", 0, 0, 12800, 768, 3072), ("fc.dgrad", 0, 1, 12800, 768, 3072),
          ("fc.wgrad", 1, 1, 3072, 768, 12800), ("qkv.fwd", 0, 0, 12800, 2304, 768), ("txt.fc.fwd", 0, 0, 19712, 2048, 512),
          ("big", 0, 0, 8192, 8192, 8192)]
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
for (tag, ta, tb, M, N, K) in SHAPES:
    a = torch.randn((K, M) if ta else (M, K), device="cuda").to(torch.bfloat16)
    b = torch.randn((K, N) if tb else (N, K), device="cuda").to(torch.bfloat16)
    A = a.t() if ta else a
    B = b if tb else b.t()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        torch.matmul(A, B, out=out)
    cold, hot = [], []
    for it in range(8):
        flush.fill_(it)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); torch.matmul(A, B, out=out); e1.record(); torch.cuda.synchronize()
        cold.append(e0.elapsed_time(e1) * 1e3)
    for it in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            torch.matmul(A, B, out=out)
        e1.record(); torch.cuda.synchronize()
        hot.append(e0.elapsed_time(e1) * 1e2)
    cold.sort(); hot.sort()
    fl = 2.0 * M * N * K
    print("%-12s cold %7.1f us (%6.0f TF/s)   back-to-back %7.1f us (%6.0f TF/s)" % (tag, cold[4], fl / cold[4] / 1e6, hot[4], fl / hot[4] / 1e6), flush=True)

"""Times the LayerNorm kernels at the step's shapes (ViT-B/32 rows 12800 x 768, packed text rows 11319 x 512, ViT-L/14
16448 x 1024): forward (fp32 in, bf16 out) and backward (bf16 dy, fp32 x, residual gradient added, fp32 + bf16 outputs),
cold caches and back to back."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ilvlm_amd import ops, lib as L

flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")


def t_of(fn, cold):
    ts = []
    for it in range(12):
        if cold:
            flush.fill_(it)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for rows, cols in ((12800, 768), (11319, 512), (16448, 1024)):
    x = torch.randn(rows, cols, device="cuda"); g = torch.randn(cols, device="cuda"); b = torch.randn(cols, device="cuda")
    y = torch.empty(rows, cols, device="cuda", dtype=torch.bfloat16)
    mean = torch.empty(rows, device="cuda"); rstd = torch.empty_like(mean)
    dy = torch.randn(rows, cols, device="cuda").to(torch.bfloat16); dres = torch.randn(rows, cols, device="cuda")
    dx = torch.empty_like(x); dxl = torch.empty_like(y); dg = torch.zeros(cols, device="cuda"); db = torch.zeros(cols, device="cuda")
    f = lambda: ops.layernorm_fwd(x, g, b, y, mean, rstd, rows, cols)
    bw = lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dg, db, rows, cols, dres=dres, dx_f32=dx, dx_lp=dxl)
    f(); bw(); torch.cuda.synchronize()
    mbf = rows * cols * 6 / 1e6
    mbb = rows * cols * (2 + 4 + 4 + 4 + 2) / 1e6
    for cold in (True, False):
        tf, tb = t_of(f, cold), t_of(bw, cold)
        print("%6d x %4d %-5s fwd %6.1f us (%.2f TB/s)   bwd %6.1f us (%.2f TB/s)" % (
            rows, cols, "cold" if cold else "warm", tf, mbf / tf, tb, mbb / tb), flush=True)

"""Times the attention kernels at the two step shapes (ViT-B/32: B=256, L=50, 12 heads; text: L=77, 8 heads, causal)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ilvlm_amd import ops, lib as L

g = torch.Generator().manual_seed(0)
LENS = [int(v) for v in torch.randint(8, 78, (256,), generator=g)]        # bench.py's caption lengths
for (tag, B, L, H, causal, packed) in [("vision", 256, 50, 12, 0, False), ("text", 256, 77, 8, 1, False),
                                       ("text.packed", 256, 77, 8, 1, True), ("vit-l/14", 64, 257, 16, 0, False)]:
    E = 64 * H
    seq = ops.PackedSeq(LENS, L, "cuda") if packed else None
    rows = seq.rows if packed else B * L
    qkv = torch.randn(rows, 3 * E, device="cuda").to(torch.bfloat16)
    dout = torch.randn(rows, E, device="cuda").to(torch.bfloat16)
    out = torch.empty(rows, E, device="cuda", dtype=torch.bfloat16)
    dqkv = torch.empty_like(qkv)
    lse = torch.zeros(B, H, L, device="cuda")
    flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    res = {}
    for name, fn in [("fwd", lambda: ops.attention_fwd(qkv, out, lse, B, L, H, causal, seq)),
                     ("bwd", lambda: ops.attention_bwd(dout, qkv, out, lse, dqkv, B, L, H, causal, seq))]:
        fn(); torch.cuda.synchronize()
        ts = []
        for it in range(10):
            flush.fill_(it)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        res[name] = ts[len(ts) // 2]
    mb_f = (qkv.numel() + out.numel()) * 2 / 1e6
    mb_b = (2 * qkv.numel() + 2 * out.numel()) * 2 / 1e6
    print("%-9s B=%d L=%d H=%d causal=%d  fwd %6.1f us (%.2f TB/s algorithmic)  bwd %6.1f us (%.2f TB/s)" % (
        tag, B, L, H, causal, res["fwd"], mb_f / res["fwd"], res["bwd"], mb_b / res["bwd"]), flush=True)

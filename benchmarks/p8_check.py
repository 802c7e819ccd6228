import sys, os
sys.path.insert(0, os.getcwd())
import torch
from ilvlm_amd import ops
torch.manual_seed(0)
bad = 0
for (M, N, K, ta, tb, acc) in [(12800, 3072, 768, 0, 0, 0), (12800, 768, 3072, 0, 1, 0), (1000, 520, 256, 0, 0, 0), (512, 256, 64, 0, 0, 0),
                               (3072, 768, 12800, 1, 1, 1), (768, 768, 1000, 1, 1, 1), (4096, 4096, 4096, 0, 0, 0)]:
    a = torch.randn((K, M) if ta else (M, K), device="cuda").to(torch.bfloat16)
    b = torch.randn((K, N) if tb else (N, K), device="cuda").to(torch.bfloat16)
    outs = {}
    for v in (5, 8):
        ops.gemm_set_variant(v)
        out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if acc else torch.bfloat16)
        for rep in range(3):
            if acc: out.zero_()
            ops.gemm(a, b, out, trans_a=bool(ta), trans_b=bool(tb), accumulate=bool(acc), split_k=3 if acc else 1)
        torch.cuda.synchronize()
        outs[v] = out.float()
    ref = (a.float().t() if ta else a.float()) @ (b.float() if tb else b.float().t())
    e5 = float((outs[5] - ref).abs().max() / ref.abs().max()); e8 = float((outs[8] - ref).abs().max() / ref.abs().max())
    eq = torch.equal(outs[5], outs[8])
    print("M=%d N=%d K=%d ta=%d tb=%d acc=%d  err v5 %.2e v8 %.2e  bit-equal %s" % (M, N, K, ta, tb, acc, e5, e8, eq), flush=True)
    if e8 > 1e-2: bad += 1
ops.gemm_set_variant(15)
sys.exit(bad)

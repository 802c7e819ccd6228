"""Micro-benchmark of the bf16 GEMM kernels on the shapes of one ViT-B/32 + text block at per-GPU batch 256
(forward, dgrad, wgrad).  Interleaved rounds in one process, cold caches (a 512 MB write between launches, as inside a
train step); prints TFLOP/s per shape for selector 5 (direct-to-LDS single-stage kernel everywhere) and 15 (default: the
streaming kernel on a packed B operand for the store-type shapes, the two-stage ring for the weight gradients).  Round 4:
GEMM_BENCH_VARIANTS=19,15 times the one-tile streaming kernel (19) against the persistent one (15, the default; 18 = persistent
for every eligible shape).  usage: gemm_bench.py [tag-substring] [--epi] [--wn 2|4 via ILVLM_PK_WN]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ilvlm_amd import ops

# selectors timed side by side (GEMM_BENCH_VARIANTS=5,15 by default).  A selector may carry settings of the persistent streaming
# kernel: "15:s1000:e2:g256" = selector 15 with a start stagger of 1000 cycles per K-tile, epilogue placement 2, 256 workgroups;
# "15:t96" = 96-row tiles of the streaming kernel (t128 / t64 likewise; t0 = the per-launch cost model; default 128);
# "15:w256" = 256 x 128 workgroup tiles of the weight-gradient kernel (w257: single-stage), ":m2" = twice the K-slices, ":b1" = the
# weight gradients' K-slices through the slab workspace (what the engine offers) instead of atomics
VARIANT_SPECS = os.environ.get("GEMM_BENCH_VARIANTS", "5,15").split(",")


def _parse(spec):
    parts = spec.split(":")
    d = dict(v=int(parts[0]), s=-1, e=-1, g=0, t=-1, w=-1, m=1, b=0)
    for q in parts[1:]:
        d[q[0]] = int(q[1:])
    return d


VSET = {spec: _parse(spec) for spec in VARIANT_SPECS}
VARIANTS = tuple(VARIANT_SPECS)


def select(spec):
    d = VSET[spec]
    ops.gemm_set_variant(d["v"])
    ops.gemm_set_persistent(d["g"], d["e"], d["s"])
    ops.gemm_set_tile_rows(d["t"])
    ops.gemm_set_wgrad_tile(d["w"])
    return d["v"]
SHAPES = []   # (tag, ta, tb, M, N, K, accumulate, split)
for tag, M, E in (("vit", 12800, 768), ("pk", 11319, 512)):
    for name, n, k in (("qkv", 3 * E, E), ("out", E, E), ("fc", 4 * E, E), ("proj", E, 4 * E)):
        SHAPES.append((tag + "." + name + ".fwd", 0, 0, M, n, k, False, 1))
        SHAPES.append((tag + "." + name + ".dgrad", 0, 1, M, k, n, False, 1))
        SHAPES.append((tag + "." + name + ".wgrad", 1, 1, n, k, M, True, ops.wgrad_split(n, k, M)))
SHAPES.append(("sq4096", 0, 0, 4096, 4096, 4096, False, 1))        # the guides' reference shape
SHAPES.append(("fdt.img.scores", 0, 0, 12544, 4096, 512, False, 1))
SHAPES.append(("fdt.txt.scores", 0, 0, 19712, 4096, 512, False, 1))


def run(rounds=7, only=None, epi=False, sk=True):
    torch.manual_seed(0)
    flush = torch.empty(128 * 1024 * 1024, device="cuda")
    # slab workspace for the store-type split-K of the streaming kernel (what the engine offers per stream)
    slab = (torch.empty(160 << 20, dtype=torch.uint8, device="cuda"), torch.zeros(8192, dtype=torch.int32, device="cuda")) if sk else None
    tot = {v: 0.0 for v in VARIANTS}
    for (tag, ta, tb, M, N, K, acc, split) in SHAPES:
        if only and only not in tag:
            continue
        a = torch.randn((K, M) if ta else (M, K), device="cuda").to(torch.bfloat16)
        b = torch.randn((K, N) if tb else (N, K), device="cuda").to(torch.bfloat16)
        out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if acc else torch.bfloat16)
        kw = {}
        if epi and not acc:        # the MLP up-projection's epilogue: bias + QuickGELU with the pre-activation stored
            kw = dict(bias=torch.randn(N, device="cuda"), aux=torch.empty(M, N, device="cuda", dtype=torch.bfloat16), act=1)
        packed = None if acc else ops.gemm_pack_b(b, trans_b=bool(tb))
        variants = VARIANTS
        best = {v: 1e9 for v in variants}
        for r in range(rounds + 1):
            for spec in variants:
                v = select(spec)
                flush.zero_()            # cold caches, as inside a train step
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.gemm(a, b, out, trans_a=bool(ta), trans_b=bool(tb), accumulate=acc, split_k=split * (VSET[spec]["m"] if acc else 1),
                         b_packed=packed if v >= 15 else None,
                         slab=slab if ((v in (15, 17) and not acc) or (acc and VSET[spec]["b"])) else None, **kw)
                e1.record()
                torch.cuda.synchronize()
                if r:
                    best[spec] = min(best[spec], e0.elapsed_time(e1))
        fl = 2.0 * M * N * K
        for v in variants:
            tot[v] += best[v]
        print("%-18s M=%6d N=%5d K=%6d split=%2d  " % (tag, M, N, K, split) +
              "  ".join("v%s %7.1f TF/s (%6.1f us)" % (v, fl / (best[v] * 1e-3) / 1e12, best[v] * 1e3) for v in variants) +
              "   x%.2f" % (best[VARIANTS[0]] / best[VARIANTS[-1]]), flush=True)
    ops.gemm_set_variant(15)
    ops.gemm_set_persistent(0, -1, -1)
    ops.gemm_set_tile_rows(-1)
    ops.gemm_set_wgrad_tile(-1)
    print("sum of best times: " + ", ".join("v%s %.1f us" % (v, tot[v] * 1e3) for v in VARIANTS))


def run_grouped(rounds=7, targets=(512,)):
    """the four weight gradients of a block: four launches (each split to ~384 workgroups, fp32 atomics) against ONE grouped launch"""
    torch.manual_seed(0)
    flush = torch.empty(128 * 1024 * 1024, device="cuda")
    for tag, M, E in (("vit", 12800, 768), ("pk", 11319, 512), ("vitl14", 32896, 1024)):
        dims = (("qkv", 3 * E, E), ("out", E, E), ("fc", 4 * E, E), ("proj", E, 4 * E))
        prob = []
        for name, n, k in dims:
            dy = torch.randn(M, n, device="cuda").to(torch.bfloat16)
            x = torch.randn(M, k, device="cuda").to(torch.bfloat16)
            prob.append((dy, x, torch.zeros(n, k, device="cuda"), torch.zeros(n, device="cuda")))
        best = {}
        for r in range(rounds + 1):
            for mode in ("4 launches",) + tuple("grouped@%d" % t for t in targets):
                flush.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                if mode == "4 launches":
                    for dy, x, gw, gb in prob:
                        ops.gemm(dy, x, gw, trans_a=True, trans_b=True, accumulate=True, split_k=ops.wgrad_split(gw.shape[0], gw.shape[1], M),
                                 a_rowsum=gb)
                else:
                    ops.wgrad_group(prob, M, target=int(mode.split("@")[1]))
                e1.record()
                torch.cuda.synchronize()
                if r:
                    best[mode] = min(best.get(mode, 1e9), e0.elapsed_time(e1))
        fl = sum(2.0 * M * n * k for _, n, k in dims)
        print("%-7s block weight gradients (rows %d): " % (tag, M) +
              "  ".join("%s %6.1f us %6.1f TF/s" % (m, t * 1e3, fl / (t * 1e-3) / 1e12) for m, t in best.items()), flush=True)


if __name__ == "__main__":
    if "--grouped" in sys.argv:
        run_grouped()
        sys.exit(0)
    args = [x for x in sys.argv[1:] if not x.startswith("--")]
    run(only=args[0] if args else None, epi="--epi" in sys.argv, sk="--no-sk" not in sys.argv)

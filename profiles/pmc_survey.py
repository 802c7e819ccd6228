"""Per-kernel means of the counters collected by profiles/pmc_survey.sh (one directory per rocprofv3 --pmc pass).

    python3 profiles/pmc_survey.py gpurun_out/r3/pmc

Counters are per launch (mean over the launches of a kernel); SQ_* cycle counters are summed over the chip's SQs by
rocprofv3, GRBM_GUI_ACTIVE over the 8 XCDs (MI355X_MICROARCH.md), so ratios between SQ counters of one pass are meaningful,
absolute values only against GRBM_GUI_ACTIVE / 8 x the unit count."""
import collections, csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize import short

KEEP = ("gemm_bf16_pk_kernel<4, false>", "gemm_bf16_dma_kernel<true, true, false, 128, 128, 2, 2, 2", "ln_bwd_kernel", "attn_bwd_wave<4, 4>",
        "adamw_kernel")


def main(src):
    for d in sorted(glob.glob(os.path.join(src, "*", ""))):
        files = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
        if not files:
            continue
        agg = collections.defaultdict(lambda: collections.Counter())
        disp = collections.defaultdict(set)
        for r in csv.DictReader(open(files[0])):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
        print("== pass %s" % os.path.basename(os.path.dirname(d)))
        for k in agg:
            if not any(k.startswith("void " + p) or k.startswith(p) for p in KEEP):
                continue
            n = len(disp[k])
            print("  %-72s x%-4d " % (k[:72], n) + "  ".join("%s=%.4g" % (c, v / n) for c, v in sorted(agg[k].items())))


if __name__ == "__main__":
    main(sys.argv[1])

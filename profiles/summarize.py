"""Turn rocprofv3 CSV output (kernel stats + FETCH_SIZE / WRITE_SIZE counter passes) into the committed summary.

    python profiles/summarize.py gpurun_out/r1 profiles/round1 --steps 25
"""
import argparse
import collections
import csv
import glob
import json
import os
import re
import shutil


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I?[A-Z]", name)
    return (m.group(1) if m else name).split("(")[0][:90]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--steps", type=int, required=True, help="train steps executed in the stats run (timed + warm-up)")
    a = ap.parse_args()
    os.makedirs(a.dst, exist_ok=True)
    stats = glob.glob(os.path.join(a.src, "stats", "*", "*_kernel_stats.csv"))[0]
    shutil.copy(stats, os.path.join(a.dst, "kernel_stats.csv"))
    rows = list(csv.DictReader(open(stats)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    lines = ["# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-roofline --no-dense-leg --steps 20 --warmup 5 --serial-towers",
             "# (towers serialised so per-kernel durations are those of a kernel owning the chip; the headline run overlaps them)",
             "# GPU busy per train step: %.2f ms  (%d steps)" % (total / a.steps / 1e6, a.steps), "",
             "%7s %9s %10s %10s  %s" % ("share", "calls/st", "avg us", "ms/step", "kernel")]
    for r in rows[:30]:
        lines.append("%6.2f%% %9.1f %10.1f %10.3f  %s" % (float(r["Percentage"]), int(r["Calls"]) / a.steps,
                                                          float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / a.steps / 1e6,
                                                          short(r["Name"])))
    conc = glob.glob(os.path.join(a.src, "stats_concurrent", "*", "*_kernel_stats.csv"))
    if conc:
        shutil.copy(conc[0], os.path.join(a.dst, "kernel_stats_concurrent.csv"))
        crows = list(csv.DictReader(open(conc[0])))
        ctotal = sum(float(r["TotalDurationNs"]) for r in crows)
        lines += ["", "# the same command WITHOUT --serial-towers (headline regime: towers and weight gradients on their own streams;",
                  "# launches overlap, so durations are inflated and their sum exceeds the step time).  Sum of durations per step: %.2f ms"
                  % (ctotal / a.steps / 1e6), "%7s %9s %10s %10s  %s" % ("share", "calls/st", "avg us", "ms/step", "kernel")]
        for r in crows[:12]:
            lines.append("%6.2f%% %9.1f %10.1f %10.3f  %s" % (float(r["Percentage"]), int(r["Calls"]) / a.steps,
                                                              float(r["AverageNs"]) / 1e3,
                                                              float(r["TotalDurationNs"]) / a.steps / 1e6, short(r["Name"])))
    mf = glob.glob(os.path.join(a.src, "mfma", "*", "*_counter_collection.csv"))
    if mf:
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.Counter()
        for r in csv.DictReader(open(mf[0])):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                cnt[k] += 1
        lines += ["", "# MFMA utilisation from PMC counters (one pass): MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x SIMDs),",
                  "# the rocprofv3 MfmaUtil expression; GRBM_GUI_ACTIVE is summed over the 8 XCDs by rocprofv3 (MI355X_MICROARCH.md),",
                  "# so busy cycles are set against (GRBM_GUI_ACTIVE / 8) x 1024 SIMDs.  MFMA FLOPs = SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512.",
                  "%-40s %10s %14s %10s" % ("kernel", "MfmaUtil %", "GFLOP/launch", "launches")]
        mout = {}
        # effective clock (round 3): GRBM_GUI_ACTIVE / 8 XCDs / average launch duration of the same kernel in the serial
        # kernel-stats pass (MI355X_MICROARCH.md, DVFS give-back: reads high on dispatches shorter than ~0.3 ms)
        avg_ns = {short(r["Name"]): float(r["AverageNs"]) for r in rows}
        lines[-1] += " %12s" % "clock GHz"
        for k in sorted(agg, key=lambda k: -agg[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0))[:8]:
            busy, gui = agg[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), agg[k].get("GRBM_GUI_ACTIVE", 0.0)
            if gui <= 0 or busy <= 0:
                continue
            util = 100.0 * busy / (gui / 8.0 * 1024.0)
            gfl = agg[k].get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) * 512 / max(cnt[k], 1) / 1e9
            clock = gui / 8.0 / max(cnt[k], 1) / avg_ns[k] if k in avg_ns else float("nan")
            mout[k] = dict(mfma_util_pct=util, mfma_gflop_per_launch=gfl, launches=cnt[k], effective_clock_ghz=clock)
            lines.append("%-40s %10.1f %14.2f %10d %12.2f" % (k, util, gfl, cnt[k], clock))
        json.dump(mout, open(os.path.join(a.dst, "mfma_util.json"), "w"), indent=1)
    traffic = {}
    for what in ("fetch", "write"):
        f = glob.glob(os.path.join(a.src, what, "*", "*_counter_collection.csv"))
        if not f:
            continue
        agg = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f[0])):
            k = short(r["Kernel_Name"])
            agg[k][0] += float(r["Counter_Value"])
            agg[k][1] += 1
        traffic[what] = agg
    if traffic:
        lines += ["", "# HBM traffic per launch from PMC counters (separate passes; FETCH_SIZE doubled per the gfx950",
                  "# correction in MI355X_MICROARCH.md section HBM; counters are in KiB)",
                  "%-40s %12s %12s %12s" % ("kernel", "read MB", "write MB", "launches")]
        out = {}
        for k in sorted(traffic.get("fetch", {}), key=lambda k: -traffic["fetch"][k][0])[:12]:
            fs, n = traffic["fetch"][k]
            ws = traffic.get("write", {}).get(k, [0.0, 1])[0]
            rd, wr = 2 * fs * 1024 / n / 1e6, ws * 1024 / n / 1e6
            out[k] = dict(read_mb_per_launch=rd, write_mb_per_launch=wr, launches=n)
            lines.append("%-40s %12.2f %12.2f %12d" % (k, rd, wr, n))
        tot_r = sum(2 * v[0] * 1024 for v in traffic.get("fetch", {}).values())
        tot_w = sum(v[0] * 1024 for v in traffic.get("write", {}).values())
        nsteps_pmc = max(1, traffic["fetch"].get("adamw_kernel", [0, 3])[1]) if "fetch" in traffic else 3
        lines += ["", "# whole step (all kernels): HBM read %.2f GB + write %.2f GB per train step (%d steps in the counter pass)"
                  % (tot_r / nsteps_pmc / 1e9, tot_w / nsteps_pmc / 1e9, nsteps_pmc)]
        out["_step_total"] = dict(read_gb=tot_r / nsteps_pmc / 1e9, write_gb=tot_w / nsteps_pmc / 1e9)
        json.dump(out, open(os.path.join(a.dst, "hbm_traffic.json"), "w"), indent=1)
    tc = glob.glob(os.path.join(a.src, "tcc", "*", "*_counter_collection.csv"))
    if tc:
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(tc[0])):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
        lines += ["", "# L2 (TCC) hit rate per kernel: TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum), all launches of the counter pass",
                  "%-40s %10s %16s" % ("kernel", "hit rate", "requests (M)")]
        tout = {}
        for k in sorted(agg, key=lambda k: -(agg[k].get("TCC_HIT_sum", 0) + agg[k].get("TCC_MISS_sum", 0)))[:10]:
            h, m = agg[k].get("TCC_HIT_sum", 0.0), agg[k].get("TCC_MISS_sum", 0.0)
            if h + m <= 0:
                continue
            tout[k] = dict(hit_rate=h / (h + m), requests=h + m)
            lines.append("%-40s %9.1f%% %16.1f" % (k, 100.0 * h / (h + m), (h + m) / 1e6))
        json.dump(tout, open(os.path.join(a.dst, "tcc_hit_rate.json"), "w"), indent=1)
    open(os.path.join(a.dst, "summary.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()

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
    lines = ["# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --serial-towers",
             "# (towers serialised so per-kernel durations are those of a kernel owning the chip; the headline run overlaps them)",
             "# GPU busy per train step: %.2f ms  (%d steps)" % (total / a.steps / 1e6, a.steps), "",
             "%7s %9s %10s %10s  %s" % ("share", "calls/st", "avg us", "ms/step", "kernel")]
    for r in rows[:30]:
        lines.append("%6.2f%% %9.1f %10.1f %10.3f  %s" % (float(r["Percentage"]), int(r["Calls"]) / a.steps,
                                                          float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / a.steps / 1e6,
                                                          short(r["Name"])))
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
    open(os.path.join(a.dst, "summary.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()

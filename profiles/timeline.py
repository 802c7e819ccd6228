"""Timeline view of a rocprofv3 --kernel-trace CSV of the two-stream train step: wall time, GPU-idle gaps, and for
every kernel the time it ran ALONE (no other kernel resident) -- i.e. what the concurrent schedule fails to hide.

    python profiles/timeline.py gpurun_out/tl --skip 0.4
"""
import argparse, collections, csv, glob, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize import short


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("--skip", type=float, default=0.4, help="leading fraction of the trace to drop (warm-up)")
    a = ap.parse_args()
    f = glob.glob(os.path.join(a.src, "**", "*_kernel_trace.csv"), recursive=True)[0]
    ev = []
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    ev.sort()
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    cut = t0 + (t1 - t0) * a.skip
    ev = [e for e in ev if e[0] >= cut]
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    pts = []
    for i, (s, e, n) in enumerate(ev):
        pts.append((s, 1, i)); pts.append((e, -1, i))
    pts.sort()
    active = set()
    solo = collections.Counter(); dur = collections.Counter(); cnt = collections.Counter()
    idle = 0; prev = t0; gaps = []
    last_name = None
    for (t, d, i) in pts:
        span = t - prev
        if span > 0:
            if len(active) == 0:
                idle += span
                gaps.append((span, last_name, None, prev))
            elif len(active) == 1:
                solo[ev[next(iter(active))][2]] += span
        prev = t
        if d == 1:
            active.add(i)
            if gaps and gaps[-1][2] is None:
                gaps[-1] = gaps[-1][:2] + (ev[i][2], gaps[-1][3])
        else:
            active.discard(i); last_name = ev[i][2]
    for (s, e, n) in ev:
        dur[n] += e - s; cnt[n] += 1
    wall = t1 - t0
    print("wall %.2f ms, idle %.2f ms (%.1f%%), sum of kernel durations %.2f ms, solo total %.2f ms" % (
        wall / 1e6, idle / 1e6, 100.0 * idle / wall, sum(dur.values()) / 1e6, sum(solo.values()) / 1e6))
    print("\nkernel: share of wall spent running alone / total duration / launches")
    for n, v in solo.most_common(25):
        print("  %6.2f%%  solo %8.2f ms  of %8.2f ms  x%-6d %s" % (100.0 * v / wall, v / 1e6, dur[n] / 1e6, cnt[n], n))
    gaps.sort(reverse=True)
    agg = collections.Counter()
    for g in gaps:
        agg[(g[1], g[2])] += g[0]
    print("\nidle time by (previous kernel -> next kernel):")
    for (k, v) in agg.most_common(15):
        print("  %8.3f ms  %s -> %s" % (v / 1e6, k[0], k[1]))


if __name__ == "__main__":
    main()

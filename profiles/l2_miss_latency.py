"""L2-miss latency and fabric credit stalls per kernel from one rocprofv3 counter pass (round 3; run through gpurun):

    mkdir -p gpurun_out/r3 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
    rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum --output-format csv \
        -d gpurun_out/r3/ea -- python3 bench.py --no-cpu-baseline --no-roofline --no-dense-leg --no-extra-legs --steps 2 --warmup 1 --serial-towers
    python3 profiles/l2_miss_latency.py gpurun_out/r3/ea gpurun_out/r3/l2_miss_latency.txt

mean outstanding time of a fabric read request as the L2 sees it = TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ (L2 clock cycles)."""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:72]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "TCC_EA0_RDREQ_sum":
        n[k] += 1
out = ["# rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum (towers serialised, 3 steps)",
       "# latency of an L2 miss as the L2 sees it = RDREQ_LEVEL / RDREQ; requests are 64-byte units (MI355X_MICROARCH.md, HBM)",
       "%-74s %8s %16s %18s %20s" % ("kernel", "launches", "requests/launch", "cycles per request", "credit stalls/launch")]
for k in sorted(agg, key=lambda k: -agg[k]["TCC_EA0_RDREQ_sum"])[:12]:
    a = agg[k]
    rq = a["TCC_EA0_RDREQ_sum"]
    if rq > 0:
        out.append("%-74s %8d %16.0f %18.0f %20.0f" % (k, n[k], rq / n[k], a["TCC_EA0_RDREQ_LEVEL_sum"] / rq,
                                                     a["TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"] / n[k]))
open(sys.argv[2], "w").write("\n".join(out) + "\n")
print("\n".join(out))

#!/bin/bash
# Collects the per-round profile on the GPU box (run from the repo root through gpurun):
#   bash profiles/collect.sh r2 [pmc]
# -> gpurun_out/r2/{stats,stats_concurrent,fetch,write,mfma}; then
#    python profiles/summarize.py gpurun_out/r2 profiles/roundN --steps 25
# stats            : towers serialised (a launch owns the chip: per-kernel durations are the kernel's own)
# stats_concurrent : the headline regime (towers and weight gradients on their own streams)
# Counter passes are separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with trace domains).
# The program comes directly after `--` (no env / bash -c hop: the profiler's preload has already initialised the GPU).
set -e
root=$(pwd)
out=$root/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
B="bench.py --no-cpu-baseline --no-roofline --no-dense-leg --no-extra-legs"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $B --steps 20 --warmup 5 --serial-towers > $out/bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_concurrent -- python3 $B --steps 20 --warmup 5 > $out/bench_under_rocprof_concurrent.log 2>&1
if [ "$2" = "pmc" ]; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $B --steps 2 --warmup 1 --serial-towers > $out/pmc_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $B --steps 2 --warmup 1 --serial-towers > $out/pmc_write.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $out/mfma -- python3 $B --steps 2 --warmup 1 --serial-towers > $out/pmc_mfma.log 2>&1
  # L2 hit rate of the operand streams (round 3: is the GEMM over-fetch served by the Infinity Cache or by re-reads that miss?)
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/tcc -- python3 $B --steps 2 --warmup 1 --serial-towers > $out/pmc_tcc.log 2>&1
fi
python3 profiles/summarize.py $out $out/summary --steps 28 > /dev/null
python3 profiles/timeline.py $out/stats_concurrent --skip 0.4 > $out/summary/timeline_concurrent.txt 2>&1 || true
# keep only what is merged back: the CSV traces are large
rm -rf $out/stats/*/*_kernel_trace.csv $out/stats_concurrent/*/*_kernel_trace.csv
head -60 $out/summary/summary.txt

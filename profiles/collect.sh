#!/bin/bash
# Collects the per-round profile on the GPU box (run from the repo root through gpurun):
#   bash profiles/collect.sh r1 [pmc]
# -> gpurun_out/r1/{stats,fetch,write}; then `python profiles/summarize.py gpurun_out/r1 profiles/roundN --steps 25`.
# Counter passes are separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with trace domains).
set -e
root=$(pwd)
out=$root/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --serial-towers > $out/bench_under_rocprof.log 2>&1
if [ "$2" = "pmc" ]; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --serial-towers > $out/pmc_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --serial-towers > $out/pmc_write.log 2>&1
fi
python3 profiles/summarize.py $out $out/summary --steps 25 > /dev/null
# keep only what is merged back: the CSV traces are large
rm -rf $out/stats/*/*_kernel_trace.csv
head -45 $out/summary/summary.txt

set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 900 bash profiles/collect.sh r4final3 pmc > gpurun_out/r4/t45_collect.txt 2>&1 || { tail -20 gpurun_out/r4/t45_collect.txt; exit 1; }
tail -5 gpurun_out/r4/t45_collect.txt

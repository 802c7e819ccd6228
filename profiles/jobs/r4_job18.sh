set -o pipefail
mkdir -p gpurun_out/r4
B="bench.py --no-cpu-baseline --no-roofline --no-dense-leg --no-extra-legs --steps 30 --warmup 8"
timeout -k 10 200 python $B > gpurun_out/r4/t18_step_group0_a.txt 2>&1 || exit 1
ILVLM_WGRAD_GROUP=1 timeout -k 10 200 python $B > gpurun_out/r4/t18_step_group1_a.txt 2>&1 || exit 1
timeout -k 10 200 python $B > gpurun_out/r4/t18_step_group0_b.txt 2>&1 || exit 1
ILVLM_WGRAD_GROUP=1 timeout -k 10 200 python $B > gpurun_out/r4/t18_step_group1_b.txt 2>&1 || exit 1
grep -H -o '"ms_per_step": [0-9.]*' gpurun_out/r4/t18_step_*.txt
timeout -k 10 900 bash profiles/collect.sh r4final pmc > gpurun_out/r4/t18_collect.txt 2>&1 || { tail -20 gpurun_out/r4/t18_collect.txt; exit 1; }
tail -30 gpurun_out/r4/t18_collect.txt

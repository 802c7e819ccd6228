set -o pipefail
mkdir -p gpurun_out/r4
B="bench.py --no-cpu-baseline --no-roofline --no-dense-leg --no-extra-legs --steps 30 --warmup 8"
for i in a b; do
timeout -k 10 200 python $B > gpurun_out/r4/t42_step_default_$i.txt 2>&1 || exit 1
ILVLM_PK_TI=0 timeout -k 10 200 python $B > gpurun_out/r4/t42_step_ti0_$i.txt 2>&1 || exit 1
ILVLM_PK_TI=6 timeout -k 10 200 python $B > gpurun_out/r4/t42_step_ti6_$i.txt 2>&1 || exit 1
ILVLM_PKP=1 timeout -k 10 200 python $B > gpurun_out/r4/t42_step_pkp_$i.txt 2>&1 || exit 1
ILVLM_ADAMW_DEFER=2 timeout -k 10 200 python $B > gpurun_out/r4/t42_step_defer2_$i.txt 2>&1 || exit 1
done
grep -H -o '"ms_per_step": [0-9.]*' gpurun_out/r4/t42_*.txt

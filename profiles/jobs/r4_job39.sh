set -o pipefail
mkdir -p gpurun_out/r4
B="bench.py --no-cpu-baseline --no-roofline --no-dense-leg --no-extra-legs --steps 30 --warmup 8"
for i in a b; do
timeout -k 10 200 python $B > gpurun_out/r4/t39_step_default_$i.txt 2>&1 || exit 1
ILVLM_WGRAD_GROUP_LDS=81920 timeout -k 10 200 python $B > gpurun_out/r4/t39_step_lds80_$i.txt 2>&1 || exit 1
ILVLM_WGRAD_GROUP_LDS=65536 timeout -k 10 200 python $B > gpurun_out/r4/t39_step_lds64_$i.txt 2>&1 || exit 1
ILVLM_WGRAD_TARGET=256 timeout -k 10 200 python $B > gpurun_out/r4/t39_step_target256_$i.txt 2>&1 || exit 1
done
grep -H -o '"ms_per_step": [0-9.]*' gpurun_out/r4/t39_*.txt

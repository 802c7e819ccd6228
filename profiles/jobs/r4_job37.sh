set -o pipefail
mkdir -p gpurun_out/r4
B="bench.py --no-cpu-baseline --no-roofline --no-dense-leg --no-extra-legs --steps 30 --warmup 8 --precision fp8"
for i in a b; do
timeout -k 10 200 python $B > gpurun_out/r4/t37_fp8_256_default_$i.txt 2>&1 || exit 1
ILVLM_FP8_WGRAD_WIDE_MIN_ROWS=0 ILVLM_WGRAD_GROUP_SLOTS=128 timeout -k 10 200 python $B > gpurun_out/r4/t37_fp8_256_wide_s128_$i.txt 2>&1 || exit 1
ILVLM_FP8_WGRAD_WIDE_MIN_ROWS=0 ILVLM_WGRAD_GROUP_SLOTS=256 timeout -k 10 200 python $B > gpurun_out/r4/t37_fp8_256_wide_s256_$i.txt 2>&1 || exit 1
ILVLM_WGRAD_GROUP_SLOTS=256 timeout -k 10 200 python $B > gpurun_out/r4/t37_fp8_256_narrow_s256_$i.txt 2>&1 || exit 1
timeout -k 10 200 python $B --batch 512 > gpurun_out/r4/t37_fp8_512_default_$i.txt 2>&1 || exit 1
ILVLM_WGRAD_GROUP_SLOTS=128 timeout -k 10 200 python $B --batch 512 > gpurun_out/r4/t37_fp8_512_s128_$i.txt 2>&1 || exit 1
done
grep -H -o '"ms_per_step": [0-9.]*' gpurun_out/r4/t37_*.txt

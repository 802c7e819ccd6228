set -o pipefail
mkdir -p gpurun_out/r4
B="bench.py --no-cpu-baseline --no-roofline --no-dense-leg --no-extra-legs --steps 30 --warmup 8 --precision fp8 --batch 512"
for i in a b; do
timeout -k 10 200 python $B > gpurun_out/r4/t30_fp8_512_s512_$i.txt 2>&1 || exit 1
ILVLM_WGRAD_GROUP_SLOTS=384 timeout -k 10 200 python $B > gpurun_out/r4/t30_fp8_512_s384_$i.txt 2>&1 || exit 1
ILVLM_WGRAD_GROUP_SLOTS=256 timeout -k 10 200 python $B > gpurun_out/r4/t30_fp8_512_s256_$i.txt 2>&1 || exit 1
ILVLM_WGRAD_GROUP_SLOTS=768 timeout -k 10 200 python $B > gpurun_out/r4/t30_fp8_512_s768_$i.txt 2>&1 || exit 1
done
grep -H -o '"ms_per_step": [0-9.]*' gpurun_out/r4/t30_fp8_*.txt

set -o pipefail
mkdir -p gpurun_out/r4
B="bench.py --no-cpu-baseline --no-roofline --no-dense-leg --no-extra-legs --steps 30 --warmup 8"
for i in a b c; do
timeout -k 10 200 python $B > gpurun_out/r4/t32_step_w128_$i.txt 2>&1 || exit 1
ILVLM_WGRAD_TILE=257 timeout -k 10 200 python $B > gpurun_out/r4/t32_step_w257_$i.txt 2>&1 || exit 1
ILVLM_WGRAD_TILE=257 ILVLM_WGRAD_TILE_SPLIT_MUL=2 timeout -k 10 200 python $B > gpurun_out/r4/t32_step_w257m2_$i.txt 2>&1 || exit 1
done
timeout -k 10 300 python $B --model vitl14 --batch 128 --steps 10 --warmup 3 > gpurun_out/r4/t32_vitl14_w128.txt 2>&1 || exit 1
ILVLM_WGRAD_TILE=257 timeout -k 10 300 python $B --model vitl14 --batch 128 --steps 10 --warmup 3 > gpurun_out/r4/t32_vitl14_w257.txt 2>&1 || exit 1
timeout -k 10 200 python $B --serial-towers > gpurun_out/r4/t32_serial_w128.txt 2>&1 || exit 1
ILVLM_WGRAD_TILE=257 timeout -k 10 200 python $B --serial-towers > gpurun_out/r4/t32_serial_w257.txt 2>&1 || exit 1
grep -H -o '"ms_per_step": [0-9.]*' gpurun_out/r4/t32_*.txt

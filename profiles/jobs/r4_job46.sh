set -o pipefail
mkdir -p gpurun_out/r4
B="bench.py --no-cpu-baseline --no-roofline --no-dense-leg --no-extra-legs --steps 30 --warmup 8"
for i in a b; do
timeout -k 10 200 python $B > gpurun_out/r4/t46_step_default_$i.txt 2>&1 || exit 1
ILVLM_PK_MIN_K=768 timeout -k 10 200 python $B > gpurun_out/r4/t46_step_pkmink768_$i.txt 2>&1 || exit 1
ILVLM_GEMM_TILE_GROUP=0 timeout -k 10 200 python $B > gpurun_out/r4/t46_step_tilegroup0_$i.txt 2>&1 || exit 1
ILVLM_WGRAD_GROUP_SLOTS_WIDE=512 timeout -k 10 200 python $B > gpurun_out/r4/t46_step_wide512_$i.txt 2>&1 || exit 1
done
grep -H -o '"ms_per_step": [0-9.]*' gpurun_out/r4/t46_*.txt

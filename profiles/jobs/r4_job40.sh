set -o pipefail
mkdir -p gpurun_out/r4
B="bench.py --no-cpu-baseline --no-roofline --no-dense-leg --no-extra-legs --steps 30 --warmup 8"
for i in a b c; do
ILVLM_WGRAD_TAIL_SPREAD=0 timeout -k 10 200 python $B > gpurun_out/r4/t40_step_tail0_$i.txt 2>&1 || exit 1
timeout -k 10 200 python $B > gpurun_out/r4/t40_step_tail1_$i.txt 2>&1 || exit 1
done
ILVLM_WGRAD_TAIL_SPREAD=0 timeout -k 10 300 python $B --model vitl14 --batch 128 --steps 10 --warmup 3 > gpurun_out/r4/t40_vitl14_tail0.txt 2>&1 || exit 1
timeout -k 10 300 python $B --model vitl14 --batch 128 --steps 10 --warmup 3 > gpurun_out/r4/t40_vitl14_tail1.txt 2>&1 || exit 1
grep -H -o '"ms_per_step": [0-9.]*' gpurun_out/r4/t40_*.txt

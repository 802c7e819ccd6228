set -o pipefail
mkdir -p gpurun_out/r4
B="bench.py --no-cpu-baseline --no-roofline --no-dense-leg --no-extra-legs"
for i in a b c; do
ILVLM_WGRAD_TAIL_SPREAD=0 timeout -k 10 300 python $B --model vitl14 --batch 128 --steps 12 --warmup 3 > gpurun_out/r4/t41_vitl14_tail0_$i.txt 2>&1 || exit 1
timeout -k 10 300 python $B --model vitl14 --batch 128 --steps 12 --warmup 3 > gpurun_out/r4/t41_vitl14_tail1_$i.txt 2>&1 || exit 1
done
grep -H -o '"ms_per_step": [0-9.]*' gpurun_out/r4/t41_*.txt

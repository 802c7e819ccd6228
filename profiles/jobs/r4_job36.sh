set -o pipefail
mkdir -p gpurun_out/r4
B="bench.py --no-cpu-baseline --no-roofline --no-dense-leg --no-extra-legs --steps 30 --warmup 8"
for i in a b; do
for sl in 32 64 96 128 160; do
ILVLM_WGRAD_GROUP=1 ILVLM_WGRAD_GROUP_SLOTS=$sl timeout -k 10 200 python $B > gpurun_out/r4/t36_step_group_s${sl}_$i.txt 2>&1 || exit 1
done
done
for sl in 64 128; do
ILVLM_WGRAD_GROUP=1 ILVLM_WGRAD_GROUP_SLOTS=$sl timeout -k 10 300 python $B --model vitl14 --batch 128 --steps 10 --warmup 3 > gpurun_out/r4/t36_vitl14_group_s$sl.txt 2>&1 || exit 1
done
grep -H -o '"ms_per_step": [0-9.]*' gpurun_out/r4/t36_*.txt

set -o pipefail
mkdir -p gpurun_out/r4
B="bench.py --no-cpu-baseline --no-roofline --no-dense-leg --no-extra-legs --steps 30 --warmup 8"
for i in a b; do
timeout -k 10 200 python $B > gpurun_out/r4/t35_step_default_$i.txt 2>&1 || exit 1
for sl in 128 192 256 320; do
ILVLM_WGRAD_GROUP=1 ILVLM_WGRAD_GROUP_SLOTS=$sl timeout -k 10 200 python $B > gpurun_out/r4/t35_step_group_s${sl}_$i.txt 2>&1 || exit 1
done
done
ILVLM_WGRAD_GROUP=1 ILVLM_WGRAD_GROUP_SLOTS=256 timeout -k 10 300 python $B --model vitl14 --batch 128 --steps 10 --warmup 3 > gpurun_out/r4/t35_vitl14_group_s256.txt 2>&1 || exit 1
timeout -k 10 300 python $B --model vitl14 --batch 128 --steps 10 --warmup 3 > gpurun_out/r4/t35_vitl14_default.txt 2>&1 || exit 1
grep -H -o '"ms_per_step": [0-9.]*' gpurun_out/r4/t35_*.txt

set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_fp8_gpu.py -x -q -k "grouped_weight_gradients or tiny or vitl14_tracks or amax_jump or loss_curve_tracks" > gpurun_out/r4/t29_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r4/t29_tests.txt
tail -12 gpurun_out/r4/t29_tests.txt
grep -q "tests rc=0" gpurun_out/r4/t29_tests.txt || exit 1
B="bench.py --no-cpu-baseline --no-roofline --no-dense-leg --no-extra-legs --steps 30 --warmup 8 --precision fp8"
for i in a b; do
ILVLM_FP8_WGRAD_TILE=128 timeout -k 10 200 python $B --batch 512 > gpurun_out/r4/t29_fp8_512_w128_$i.txt 2>&1 || exit 1
timeout -k 10 200 python $B --batch 512 > gpurun_out/r4/t29_fp8_512_w256_$i.txt 2>&1 || exit 1
ILVLM_FP8_WGRAD_TILE=128 timeout -k 10 200 python $B > gpurun_out/r4/t29_fp8_256_w128_$i.txt 2>&1 || exit 1
timeout -k 10 200 python $B > gpurun_out/r4/t29_fp8_256_w256_$i.txt 2>&1 || exit 1
done
grep -H -o '"ms_per_step": [0-9.]*' gpurun_out/r4/t29_fp8_*.txt

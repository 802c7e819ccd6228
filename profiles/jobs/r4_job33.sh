set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "256_row_tiles" > gpurun_out/r4/t33_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r4/t33_tests.txt
tail -4 gpurun_out/r4/t33_tests.txt
grep -q "tests rc=0" gpurun_out/r4/t33_tests.txt || exit 1
B="bench.py --no-cpu-baseline --no-roofline --no-dense-leg --no-extra-legs --steps 30 --warmup 8"
for i in a b; do
ILVLM_WGRAD_TILE=128 timeout -k 10 200 python $B > gpurun_out/r4/t33_step_w128_$i.txt 2>&1 || exit 1
timeout -k 10 200 python $B > gpurun_out/r4/t33_step_default_$i.txt 2>&1 || exit 1
done
timeout -k 10 200 python $B --serial-towers > gpurun_out/r4/t33_serial_default.txt 2>&1 || exit 1
ILVLM_WGRAD_TILE=128 timeout -k 10 200 python $B --precision fp8 --batch 512 > gpurun_out/r4/t33_fp8_512_w128.txt 2>&1 || exit 1
timeout -k 10 200 python $B --precision fp8 --batch 512 > gpurun_out/r4/t33_fp8_512_default.txt 2>&1 || exit 1
grep -H -o '"ms_per_step": [0-9.]*' gpurun_out/r4/t33_*.txt

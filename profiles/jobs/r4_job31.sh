set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "256_row_tiles" > gpurun_out/r4/t31_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r4/t31_tests.txt
tail -4 gpurun_out/r4/t31_tests.txt
grep -q "tests rc=0" gpurun_out/r4/t31_tests.txt || exit 1
GEMM_BENCH_VARIANTS="15,15:w257,15:w256,15:w257:m2" timeout -k 10 500 python benchmarks/gemm_bench.py wgrad > gpurun_out/r4/t31_gemm_wgrad.txt 2>&1 || exit 1
grep -v amdgpu gpurun_out/r4/t31_gemm_wgrad.txt
B="bench.py --no-cpu-baseline --no-roofline --no-dense-leg --no-extra-legs --steps 30 --warmup 8"
for i in a b; do
timeout -k 10 200 python $B > gpurun_out/r4/t31_step_w128_$i.txt 2>&1 || exit 1
ILVLM_WGRAD_TILE=257 timeout -k 10 200 python $B > gpurun_out/r4/t31_step_w257_$i.txt 2>&1 || exit 1
ILVLM_WGRAD_TILE=256 timeout -k 10 200 python $B > gpurun_out/r4/t31_step_w256_$i.txt 2>&1 || exit 1
done
grep -H -o '"ms_per_step": [0-9.]*' gpurun_out/r4/t31_step_*.txt

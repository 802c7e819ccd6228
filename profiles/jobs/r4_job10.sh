set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "streaming or full_size or persistent" > gpurun_out/r4/t10_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r4/t10_tests.txt
tail -4 gpurun_out/r4/t10_tests.txt
grep -q "tests rc=0" gpurun_out/r4/t10_tests.txt || exit 1
V="15:t128,15:t96,15:t64,15"
GEMM_BENCH_VARIANTS=$V timeout -k 10 400 python benchmarks/gemm_bench.py fwd > gpurun_out/r4/t10_gemm_fwd.txt 2>&1 || exit 1
GEMM_BENCH_VARIANTS=$V timeout -k 10 400 python benchmarks/gemm_bench.py dgrad > gpurun_out/r4/t10_gemm_dgrad.txt 2>&1 || exit 1
B="bench.py --no-cpu-baseline --no-roofline --no-dense-leg --no-extra-legs --steps 30 --warmup 8"
for i in a b c; do
ILVLM_PK_TI=8 timeout -k 10 200 python $B > gpurun_out/r4/t10_step_t128_$i.txt 2>&1 || exit 1
timeout -k 10 200 python $B > gpurun_out/r4/t10_step_auto_$i.txt 2>&1 || exit 1
done
ILVLM_PK_TI=6 timeout -k 10 200 python $B > gpurun_out/r4/t10_step_t96.txt 2>&1 || exit 1
ILVLM_PK_TI=4 timeout -k 10 200 python $B > gpurun_out/r4/t10_step_t64.txt 2>&1 || exit 1
ILVLM_PK_TI=8 timeout -k 10 200 python $B --serial-towers > gpurun_out/r4/t10_step_serial_t128.txt 2>&1 || exit 1
timeout -k 10 200 python $B --serial-towers > gpurun_out/r4/t10_step_serial_auto.txt 2>&1 || exit 1
grep -H -o '"ms_per_step": [0-9.]*' gpurun_out/r4/t10_step_*.txt
cat gpurun_out/r4/t10_gemm_fwd.txt gpurun_out/r4/t10_gemm_dgrad.txt | grep -v amdgpu

set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -x -q > gpurun_out/r4/t4_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r4/t4_tests.txt
tail -3 gpurun_out/r4/t4_tests.txt
grep -q "tests rc=0" gpurun_out/r4/t4_tests.txt || exit 1
for sfx in _nowide ""; do
ILVLM_LIB_SUFFIX=$sfx GEMM_BENCH_VARIANTS=5,19 timeout -k 10 400 python benchmarks/gemm_bench.py fwd --epi > gpurun_out/r4/t4_gemm_fwd_epi$sfx.txt 2>&1 || exit 1
ILVLM_LIB_SUFFIX=$sfx GEMM_BENCH_VARIANTS=5,19 timeout -k 10 400 python benchmarks/gemm_bench.py dgrad > gpurun_out/r4/t4_gemm_dgrad$sfx.txt 2>&1 || exit 1
done
B="bench.py --no-cpu-baseline --no-roofline --no-dense-leg --no-extra-legs --steps 30 --warmup 8"
for i in a b c; do
ILVLM_LIB_SUFFIX=_nowide ILVLM_PKP=0 timeout -k 10 200 python $B > gpurun_out/r4/t4_step_nowide_$i.txt 2>&1 || exit 1
ILVLM_PKP=0 timeout -k 10 200 python $B > gpurun_out/r4/t4_step_wide_$i.txt 2>&1 || exit 1
done
ILVLM_LIB_SUFFIX=_nowide ILVLM_PKP=0 timeout -k 10 200 python $B --serial-towers > gpurun_out/r4/t4_step_serial_nowide.txt 2>&1 || exit 1
ILVLM_PKP=0 timeout -k 10 200 python $B --serial-towers > gpurun_out/r4/t4_step_serial_wide.txt 2>&1 || exit 1
grep -H -o '"ms_per_step": [0-9.]*' gpurun_out/r4/t4_step_*.txt
paste -d'\n' gpurun_out/r4/t4_gemm_fwd_epi_nowide.txt gpurun_out/r4/t4_gemm_fwd_epi.txt | grep -v amdgpu
paste -d'\n' gpurun_out/r4/t4_gemm_dgrad_nowide.txt gpurun_out/r4/t4_gemm_dgrad.txt | grep -v amdgpu

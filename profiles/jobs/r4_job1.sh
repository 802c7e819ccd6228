set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "persistent or streaming or full_size or fragment" > gpurun_out/r4/t1_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r4/t1_tests.txt
tail -5 gpurun_out/r4/t1_tests.txt
grep -q "tests rc=0" gpurun_out/r4/t1_tests.txt || exit 1
GEMM_BENCH_VARIANTS=19,15 timeout -k 10 300 python benchmarks/gemm_bench.py fwd --epi > gpurun_out/r4/t1_gemm_fwd_epi.txt 2>&1 && \
GEMM_BENCH_VARIANTS=19,15 timeout -k 10 300 python benchmarks/gemm_bench.py dgrad > gpurun_out/r4/t1_gemm_dgrad.txt 2>&1 && \
timeout -k 10 300 python benchmarks/gemm_stamps_pkp.py > gpurun_out/r4/t1_stamps.txt 2>&1 && \
B="bench.py --no-cpu-baseline --no-roofline --no-dense-leg --no-extra-legs --steps 30 --warmup 8" && \
ILVLM_PKP=0 timeout -k 10 200 python $B > gpurun_out/r4/t1_step_old_a.txt 2>&1 && \
timeout -k 10 200 python $B > gpurun_out/r4/t1_step_new_a.txt 2>&1 && \
ILVLM_PKP=0 timeout -k 10 200 python $B > gpurun_out/r4/t1_step_old_b.txt 2>&1 && \
timeout -k 10 200 python $B > gpurun_out/r4/t1_step_new_b.txt 2>&1
cat gpurun_out/r4/t1_gemm_fwd_epi.txt gpurun_out/r4/t1_gemm_dgrad.txt
grep -h ms_per_step gpurun_out/r4/t1_step_*.txt | cut -c1-200

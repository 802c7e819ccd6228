set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "persistent or clip_grad" > gpurun_out/r4/t3_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r4/t3_tests.txt
tail -3 gpurun_out/r4/t3_tests.txt
grep -q "tests rc=0" gpurun_out/r4/t3_tests.txt || exit 1
V="19,15:s0:e1,15:s500:e1,15:s900:e1,15:s1300:e1,15:s900:e2,15:s900:e1:g256"
GEMM_BENCH_VARIANTS=$V timeout -k 10 400 python benchmarks/gemm_bench.py fwd --epi > gpurun_out/r4/t3_gemm_fwd_epi.txt 2>&1 || exit 1
GEMM_BENCH_VARIANTS=$V timeout -k 10 400 python benchmarks/gemm_bench.py dgrad > gpurun_out/r4/t3_gemm_dgrad.txt 2>&1 || exit 1
STAMP_STAGGER=900 STAMP_EPI_SEP=1 timeout -k 10 300 python benchmarks/gemm_stamps_pkp.py vit > gpurun_out/r4/t3_stamps_s900.txt 2>&1 || exit 1
B="bench.py --no-cpu-baseline --no-roofline --no-dense-leg --no-extra-legs --steps 30 --warmup 8"
run() { tag=$1; shift; env "$@" timeout -k 10 200 python $B > gpurun_out/r4/t3_step_$tag.txt 2>&1 || exit 1; }
for i in a b; do
run old_$i ILVLM_PKP=0
run e2s0_$i ILVLM_PKP_EPI_SEP=2
run e2s900_$i ILVLM_PKP_EPI_SEP=2 ILVLM_PKP_STAGGER=900
run e1s900_$i ILVLM_PKP_EPI_SEP=1 ILVLM_PKP_STAGGER=900
run e2s1300_$i ILVLM_PKP_EPI_SEP=2 ILVLM_PKP_STAGGER=1300
run e2s900g256_$i ILVLM_PKP_EPI_SEP=2 ILVLM_PKP_STAGGER=900 ILVLM_PKP_SLOTS=256
done
ILVLM_PKP=0 timeout -k 10 200 python $B --serial-towers > gpurun_out/r4/t3_step_serial_old.txt 2>&1 || exit 1
ILVLM_PKP_EPI_SEP=2 ILVLM_PKP_STAGGER=900 timeout -k 10 200 python $B --serial-towers > gpurun_out/r4/t3_step_serial_e2s900.txt 2>&1 || exit 1
grep -H -o '"ms_per_step": [0-9.]*' gpurun_out/r4/t3_step_*.txt
cat gpurun_out/r4/t3_gemm_fwd_epi.txt gpurun_out/r4/t3_gemm_dgrad.txt

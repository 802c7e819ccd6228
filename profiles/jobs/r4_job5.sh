set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r4/t5_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r4/t5_tests.txt
tail -15 gpurun_out/r4/t5_tests.txt
grep -q "tests rc=0" gpurun_out/r4/t5_tests.txt || exit 1
timeout -k 10 600 python bench.py > gpurun_out/r4/t5_bench_default.txt 2>&1 || exit 1
tail -1 gpurun_out/r4/t5_bench_default.txt | cut -c1-1500

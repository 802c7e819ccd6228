set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r4/t16_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r4/t16_tests.txt
tail -8 gpurun_out/r4/t16_tests.txt

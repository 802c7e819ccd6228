set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_input_pipeline_gpu.py -x -q > gpurun_out/r4/t15_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r4/t15_tests.txt
tail -15 gpurun_out/r4/t15_tests.txt

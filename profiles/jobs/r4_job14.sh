set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 1000 python -m pytest tests/test_parity_bf16_gpu.py -q -s -k "batch256_matches_oracle or vitl14_fdt_bf16_batch8" > gpurun_out/r4/t14_parity.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r4/t14_parity.txt
grep -E "control|gradient cosines|passed|failed|rc=|assert|Error" gpurun_out/r4/t14_parity.txt

set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r4/t8_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r4/t8_tests.txt
tail -12 gpurun_out/r4/t8_tests.txt
grep -q "tests rc=0" gpurun_out/r4/t8_tests.txt || exit 1
B="bench.py --no-cpu-baseline --no-roofline --no-dense-leg --no-extra-legs --steps 30 --warmup 8"
ILVLM_TOWER=0 timeout -k 10 200 python $B > gpurun_out/r4/t8_bf16_notower.txt 2>&1 || exit 1
timeout -k 10 200 python $B > gpurun_out/r4/t8_bf16_tower.txt 2>&1 || exit 1
ILVLM_TOWER=0 timeout -k 10 200 python $B --precision fp8 > gpurun_out/r4/t8_fp8_notower.txt 2>&1 || exit 1
timeout -k 10 200 python $B --precision fp8 > gpurun_out/r4/t8_fp8_tower.txt 2>&1 || exit 1
grep -H -o '"ms_per_step": [0-9.]*\|"host_enqueue_ms_per_step": [0-9.]*\|"host_loop_ms_per_step": [0-9.]*' gpurun_out/r4/t8_*.txt
bash profiles/collect.sh r4prof pmc > gpurun_out/r4/t8_collect.txt 2>&1 || { tail -20 gpurun_out/r4/t8_collect.txt; exit 1; }
tail -40 gpurun_out/r4/t8_collect.txt

set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r4/t6_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r4/t6_tests.txt
tail -12 gpurun_out/r4/t6_tests.txt
grep -q "tests rc=0" gpurun_out/r4/t6_tests.txt || exit 1
B="bench.py --no-cpu-baseline --no-roofline --no-dense-leg --no-extra-legs --steps 30 --warmup 8"
for i in a b; do
ILVLM_LIB_SUFFIX=_nowide timeout -k 10 200 python $B --precision fp8 > gpurun_out/r4/t6_fp8_256_nowide_$i.txt 2>&1 || exit 1
timeout -k 10 200 python $B --precision fp8 > gpurun_out/r4/t6_fp8_256_wide_$i.txt 2>&1 || exit 1
ILVLM_TOWER=0 timeout -k 10 200 python $B --precision fp8 > gpurun_out/r4/t6_fp8_256_wide_notower_$i.txt 2>&1 || exit 1
ILVLM_LIB_SUFFIX=_nowide timeout -k 10 200 python $B --precision fp8 --batch 512 > gpurun_out/r4/t6_fp8_512_nowide_$i.txt 2>&1 || exit 1
timeout -k 10 200 python $B --precision fp8 --batch 512 > gpurun_out/r4/t6_fp8_512_wide_$i.txt 2>&1 || exit 1
ILVLM_LIB_SUFFIX=_nowide timeout -k 10 200 python $B > gpurun_out/r4/t6_bf16_nowide_$i.txt 2>&1 || exit 1
timeout -k 10 200 python $B > gpurun_out/r4/t6_bf16_wide_$i.txt 2>&1 || exit 1
ILVLM_TOWER=0 timeout -k 10 200 python $B > gpurun_out/r4/t6_bf16_wide_notower_$i.txt 2>&1 || exit 1
done
grep -H -o '"ms_per_step": [0-9.]*\|"host_enqueue_ms_per_step": [0-9.]*' gpurun_out/r4/t6_*.txt

set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_train_gpu.py -x -q -k "grouped_weight_gradients" > gpurun_out/r4/t34_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r4/t34_tests.txt
tail -4 gpurun_out/r4/t34_tests.txt
grep -q "tests rc=0" gpurun_out/r4/t34_tests.txt || exit 1
B="bench.py --no-cpu-baseline --no-roofline --no-dense-leg --no-extra-legs --steps 30 --warmup 8"
for i in a b c; do
timeout -k 10 200 python $B > gpurun_out/r4/t34_step_default_$i.txt 2>&1 || exit 1
ILVLM_WGRAD_GROUP=1 timeout -k 10 200 python $B > gpurun_out/r4/t34_step_group_$i.txt 2>&1 || exit 1
done
ILVLM_WGRAD_GROUP=1 ILVLM_WGRAD_GROUP_SLOTS=256 timeout -k 10 200 python $B > gpurun_out/r4/t34_step_group_s256.txt 2>&1 || exit 1
ILVLM_WGRAD_GROUP=1 ILVLM_WGRAD_GROUP_SLOTS=1024 timeout -k 10 200 python $B > gpurun_out/r4/t34_step_group_s1024.txt 2>&1 || exit 1
grep -H -o '"ms_per_step": [0-9.]*' gpurun_out/r4/t34_step_*.txt

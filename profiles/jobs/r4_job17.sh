set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_train_gpu.py -x -q -k "deferred_update or fragment_order_weight_copies" > gpurun_out/r4/t17_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r4/t17_tests.txt
tail -12 gpurun_out/r4/t17_tests.txt
grep -q "tests rc=0" gpurun_out/r4/t17_tests.txt || exit 1
B="bench.py --no-cpu-baseline --no-roofline --no-dense-leg --no-extra-legs --steps 30 --warmup 8"
for i in a b; do
timeout -k 10 200 python $B > gpurun_out/r4/t17_step_d0_$i.txt 2>&1 || exit 1
ILVLM_ADAMW_DEFER=1 timeout -k 10 200 python $B > gpurun_out/r4/t17_step_d1_$i.txt 2>&1 || exit 1
ILVLM_ADAMW_DEFER=2 timeout -k 10 200 python $B > gpurun_out/r4/t17_step_d2_$i.txt 2>&1 || exit 1
ILVLM_ADAMW_DEFER=4 timeout -k 10 200 python $B > gpurun_out/r4/t17_step_d4_$i.txt 2>&1 || exit 1
done
GPU_MAX_HW_QUEUES=8 timeout -k 10 200 python $B > gpurun_out/r4/t17_step_d0_q8.txt 2>&1 || exit 1
GPU_MAX_HW_QUEUES=8 ILVLM_ADAMW_DEFER=2 timeout -k 10 200 python $B > gpurun_out/r4/t17_step_d2_q8.txt 2>&1 || exit 1
GPU_MAX_HW_QUEUES=2 timeout -k 10 200 python $B > gpurun_out/r4/t17_step_d0_q2.txt 2>&1 || exit 1
ILVLM_ADAMW_DEFER=2 timeout -k 10 200 python $B --phase-times > gpurun_out/r4/t17_phase_d2.txt 2>&1 || exit 1
grep -H -o '"ms_per_step": [0-9.]*' gpurun_out/r4/t17_step_*.txt
grep "^phase" gpurun_out/r4/t17_phase_d2.txt

set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 300 python benchmarks/gemm_stamps_pkp.py > gpurun_out/r4/t2_stamps.txt 2>&1 || { tail -5 gpurun_out/r4/t2_stamps.txt; exit 1; }
B="bench.py --no-cpu-baseline --no-roofline --no-dense-leg --no-extra-legs --steps 30 --warmup 8"
for i in a b; do
ILVLM_PKP=0 timeout -k 10 200 python $B > gpurun_out/r4/t2_step_old_$i.txt 2>&1 || exit 1
timeout -k 10 200 python $B > gpurun_out/r4/t2_step_new_$i.txt 2>&1 || exit 1
ILVLM_PKP_EPI_SEP=0 timeout -k 10 200 python $B > gpurun_out/r4/t2_step_new48_$i.txt 2>&1 || exit 1
done
ILVLM_PKP=0 timeout -k 10 200 python $B --serial-towers > gpurun_out/r4/t2_serial_old.txt 2>&1 || exit 1
timeout -k 10 200 python $B --serial-towers > gpurun_out/r4/t2_serial_new.txt 2>&1 || exit 1
grep -H -o '"ms_per_step": [0-9.]*' gpurun_out/r4/t2_step_*.txt gpurun_out/r4/t2_serial_*.txt

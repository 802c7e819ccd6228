mkdir -p gpurun_out/r4
export PYTHONFAULTHANDLER=1
timeout -k 10 200 python benchmarks/graph_probe.py 4 --force --keep-events > gpurun_out/r4/t21_stage4_keep.txt 2>&1; echo "exit code $?" >> gpurun_out/r4/t21_stage4_keep.txt
tail -6 gpurun_out/r4/t21_stage4_keep.txt
grep -q "exit code 0" gpurun_out/r4/t21_stage4_keep.txt || exit 0
timeout -k 10 200 python benchmarks/graph_probe.py 5 --force --keep-events > gpurun_out/r4/t21_stage5_keep.txt 2>&1; echo "exit code $?" >> gpurun_out/r4/t21_stage5_keep.txt
tail -6 gpurun_out/r4/t21_stage5_keep.txt
timeout -k 10 200 python benchmarks/graph_probe.py 4 --force > gpurun_out/r4/t21_stage4_plain.txt 2>&1; echo "exit code $?" >> gpurun_out/r4/t21_stage4_plain.txt
tail -8 gpurun_out/r4/t21_stage4_plain.txt
exit 0

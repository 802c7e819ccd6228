mkdir -p gpurun_out/r4
timeout -k 10 400 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "set confirm off" -ex "handle SIGSEGV stop print" -ex run -ex bt -ex "info registers rip" --args python3 benchmarks/graph_probe.py 4 --force > gpurun_out/r4/t23_gdb.txt 2>&1; echo "exit code $?" >> gpurun_out/r4/t23_gdb.txt
grep -n -A40 "SIGSEGV" gpurun_out/r4/t23_gdb.txt | head -80
tail -5 gpurun_out/r4/t23_gdb.txt
exit 0

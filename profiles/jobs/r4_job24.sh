mkdir -p gpurun_out/r4
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o graph_fork_probe_bin benchmarks/micro/graph_fork_probe.hip > gpurun_out/r4/t24_build.txt 2>&1 || exit 1
TL=$(python3 -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))")
out=gpurun_out/r4/t24_graph_fork_probe.txt
: > $out
for lib in default torch; do
for v in "4 0 1 0 0" "96 64 1 1 0" "4 0 1 0 1" "96 64 1 1 1"; do
  echo "== runtime $lib; forks ring nested thread two: $v" >> $out
  if [ $lib = torch ]; then LD_LIBRARY_PATH=$TL timeout -k 5 60 ./graph_fork_probe_bin $v >> $out 2>&1; else timeout -k 5 60 ./graph_fork_probe_bin $v >> $out 2>&1; fi
  echo "exit code $?" >> $out
done
done
cat $out
strings $TL/libamdhip64.so | grep -m3 -i "HIP version\|rocm-7\|7\.0\.\|7\.2\." ; ls -la $TL/libamdhip64.so /opt/rocm/lib/libamdhip64.so*
exit 0

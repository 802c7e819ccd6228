mkdir -p gpurun_out/r4
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o graph_fork_probe_bin benchmarks/micro/graph_fork_probe.hip > gpurun_out/r4/t25_build.txt 2>&1 || exit 1
TL=$(python3 -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))")
out=gpurun_out/r4/t25_graph_fork_probe_torch_runtime.txt
: > $out
for v in "4 0 1 0 1" "96 64 1 1 1"; do
  echo "== LD_PRELOAD=torch/lib/libamdhip64.so; forks ring nested thread two: $v" >> $out
  LD_PRELOAD=$TL/libamdhip64.so LD_LIBRARY_PATH=$TL LD_DEBUG=libs timeout -k 5 60 ./graph_fork_probe_bin $v 2>&1 | grep -E "calling init.*amdhip|capture|ok|error|Segm" >> $out
  echo "exit code ${PIPESTATUS[0]}" >> $out
done
cat $out
exit 0

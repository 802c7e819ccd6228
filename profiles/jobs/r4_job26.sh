mkdir -p gpurun_out/r4
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o graph_fork_probe_bin benchmarks/micro/graph_fork_probe.hip > gpurun_out/r4/t26_build.txt 2>&1 || exit 1
TL=$(python3 -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))")
out=gpurun_out/r4/t26_graph_fork_probe_both_runtimes.txt
: > $out
for v in "1 0 0 0 0" "4 0 0 0 0" "1 0 1 0 0" "4 0 1 0 0" "96 64 1 1 0" "1 0 1 0 1" "4 0 1 0 1" "96 64 1 1 1"; do
  for lib in rocm-7.2 torch-bundled-7.0.2; do
    if [ $lib = rocm-7.2 ]; then r=$(timeout -k 5 60 ./graph_fork_probe_bin $v 2>&1 | grep -E "^ok|error" | head -1); rc=$?; else r=$(LD_PRELOAD=$TL/libamdhip64.so LD_LIBRARY_PATH=$TL timeout -k 5 60 ./graph_fork_probe_bin $v 2>&1 | grep -E "^ok|error|Segm" | head -1); fi
    if [ $lib = rocm-7.2 ]; then timeout -k 5 60 ./graph_fork_probe_bin $v > /dev/null 2>&1; rc=$?; else LD_PRELOAD=$TL/libamdhip64.so LD_LIBRARY_PATH=$TL timeout -k 5 60 ./graph_fork_probe_bin $v > /dev/null 2>&1; rc=$?; fi
    echo "forks ring nested thread two = $v   runtime $lib: exit code $rc  $r" >> $out
  done
done
cat $out
exit 0

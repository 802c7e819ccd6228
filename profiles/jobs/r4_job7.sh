set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_train_gpu.py tests/test_kernels_gpu.py tests/test_input_pipeline_gpu.py -x -q -k "adamw or shadow or trajectory or solver or full_size_step_properties or pack or augment or prefetcher or uint8" > gpurun_out/r4/t7_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r4/t7_tests.txt
tail -12 gpurun_out/r4/t7_tests.txt
grep -q "tests rc=0" gpurun_out/r4/t7_tests.txt || exit 1
B="bench.py --no-cpu-baseline --no-roofline --no-dense-leg --no-extra-legs --steps 30 --warmup 8"
for i in a b; do
ILVLM_LIB_SUFFIX=_nowide ILVLM_ADAMW_PACK=0 ILVLM_PREZERO=0 ILVLM_TOWER=0 timeout -k 10 200 python $B > gpurun_out/r4/t7_bf16_round3like_$i.txt 2>&1 || exit 1
ILVLM_ADAMW_PACK=0 ILVLM_PREZERO=0 ILVLM_TOWER=0 timeout -k 10 200 python $B > gpurun_out/r4/t7_bf16_wide_$i.txt 2>&1 || exit 1
ILVLM_ADAMW_PACK=0 ILVLM_PREZERO=0 timeout -k 10 200 python $B > gpurun_out/r4/t7_bf16_wide_tower_$i.txt 2>&1 || exit 1
ILVLM_PREZERO=0 timeout -k 10 200 python $B > gpurun_out/r4/t7_bf16_wide_tower_pack_$i.txt 2>&1 || exit 1
timeout -k 10 200 python $B > gpurun_out/r4/t7_bf16_all_$i.txt 2>&1 || exit 1
ILVLM_LIB_SUFFIX=_nowide ILVLM_TOWER=0 timeout -k 10 200 python $B --precision fp8 > gpurun_out/r4/t7_fp8_256_nowide_$i.txt 2>&1 || exit 1
ILVLM_TOWER=0 timeout -k 10 200 python $B --precision fp8 > gpurun_out/r4/t7_fp8_256_wide_$i.txt 2>&1 || exit 1
timeout -k 10 200 python $B --precision fp8 > gpurun_out/r4/t7_fp8_256_all_$i.txt 2>&1 || exit 1
ILVLM_LIB_SUFFIX=_nowide timeout -k 10 200 python $B --precision fp8 --batch 512 > gpurun_out/r4/t7_fp8_512_nowide_$i.txt 2>&1 || exit 1
timeout -k 10 200 python $B --precision fp8 --batch 512 > gpurun_out/r4/t7_fp8_512_all_$i.txt 2>&1 || exit 1
done
grep -H -o '"ms_per_step": [0-9.]*\|"host_enqueue_ms_per_step": [0-9.]*' gpurun_out/r4/t7_*.txt
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o graph_fork_probe_bin benchmarks/micro/graph_fork_probe.hip > gpurun_out/r4/t7_graph_build.txt 2>&1 || exit 1
for v in "1 0 0 0" "2 0 0 0" "96 0 0 0" "4 2 0 0" "96 64 0 0" "96 64 1 0" "96 64 1 1" "1 0 1 1"; do
  echo "== forks ring nested thread: $v" >> gpurun_out/r4/t7_graph_fork_probe.txt
  timeout -k 5 60 ./graph_fork_probe_bin $v >> gpurun_out/r4/t7_graph_fork_probe.txt 2>&1; echo "exit code $?" >> gpurun_out/r4/t7_graph_fork_probe.txt
done
cat gpurun_out/r4/t7_graph_fork_probe.txt

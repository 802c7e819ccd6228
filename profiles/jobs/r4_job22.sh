mkdir -p gpurun_out/r4
export PYTHONFAULTHANDLER=1
run() { tag=$1; shift; env "$@" timeout -k 10 200 python benchmarks/graph_probe.py 4 --force > gpurun_out/r4/t22_$tag.txt 2>&1; echo "exit code $?" >> gpurun_out/r4/t22_$tag.txt; echo "== $tag: $(grep -E 'stage 4 ok|exit code|Segmentation' gpurun_out/r4/t22_$tag.txt | tr '\n' ' ')"; }
run composite0 ILVLM_COMPOSITE=0
run tower0 ILVLM_TOWER=0
run slab0 ILVLM_SLAB_SPLITK=0
run prio ILVLM_WGRAD_PRIO=-1
exit 0

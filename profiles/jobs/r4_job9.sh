set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/r4/t9_smoke.txt 2>&1 || { tail -5 gpurun_out/r4/t9_smoke.txt; exit 1; }
tail -3 gpurun_out/r4/t9_smoke.txt
timeout -k 10 900 python bench.py > gpurun_out/r4/t9_bench_default.txt 2>&1 || { tail -5 gpurun_out/r4/t9_bench_default.txt; exit 1; }
tail -1 gpurun_out/r4/t9_bench_default.txt | cut -c1-600
timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --no-dense-leg --no-extra-legs --phase-times --steps 20 --warmup 5 > gpurun_out/r4/t9_phase_times.txt 2>&1 || exit 1
grep phase gpurun_out/r4/t9_phase_times.txt

set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 500 python benchmarks/fp8_gemm_bench.py pk > gpurun_out/r4/t20_fp8_gemm_pk.txt 2>&1 || { tail gpurun_out/r4/t20_fp8_gemm_pk.txt; exit 1; }
grep -v amdgpu gpurun_out/r4/t20_fp8_gemm_pk.txt
B="bench.py --no-cpu-baseline --no-roofline --no-dense-leg --no-extra-legs --steps 30 --warmup 8 --precision fp8 --batch 512"
for i in a b; do
timeout -k 10 200 python $B > gpurun_out/r4/t20_fp8_512_k0_$i.txt 2>&1 || exit 1
ILVLM_FP8_PK_MIN_K=1024 timeout -k 10 200 python $B > gpurun_out/r4/t20_fp8_512_k1024_$i.txt 2>&1 || exit 1
ILVLM_FP8_PK_MIN_K=1536 timeout -k 10 200 python $B > gpurun_out/r4/t20_fp8_512_k1536_$i.txt 2>&1 || exit 1
done
grep -H -o '"ms_per_step": [0-9.]*' gpurun_out/r4/t20_fp8_512_*.txt

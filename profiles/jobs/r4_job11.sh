set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_fp8_gpu.py -x -q -k "streaming or packed_weight or dequantised or tiny or weight_quant or text_encoder_reset" > gpurun_out/r4/t11_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r4/t11_tests.txt
tail -6 gpurun_out/r4/t11_tests.txt
grep -q "tests rc=0" gpurun_out/r4/t11_tests.txt || exit 1
B="bench.py --no-cpu-baseline --no-roofline --no-dense-leg --no-extra-legs --steps 30 --warmup 8 --precision fp8"
for i in a b; do
ILVLM_FP8_PACK=0 timeout -k 10 200 python $B > gpurun_out/r4/t11_fp8_256_dma_$i.txt 2>&1 || exit 1
timeout -k 10 200 python $B > gpurun_out/r4/t11_fp8_256_pk_$i.txt 2>&1 || exit 1
ILVLM_FP8_PACK=0 timeout -k 10 200 python $B --batch 512 > gpurun_out/r4/t11_fp8_512_dma_$i.txt 2>&1 || exit 1
timeout -k 10 200 python $B --batch 512 > gpurun_out/r4/t11_fp8_512_pk_$i.txt 2>&1 || exit 1
done
grep -H -o '"ms_per_step": [0-9.]*' gpurun_out/r4/t11_fp8_*.txt

set -o pipefail
root=$(pwd)
out=$root/gpurun_out/r4/fp8prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
B="bench.py --no-cpu-baseline --no-roofline --no-dense-leg --no-extra-legs --precision fp8 --batch 512"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $B --steps 10 --warmup 4 --serial-towers > $out/serial.log 2>&1 || exit 1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_concurrent -- python3 $B --steps 10 --warmup 4 > $out/conc.log 2>&1 || exit 1
python3 profiles/timeline.py $out/stats_concurrent --skip 0.4 > $out/timeline_concurrent.txt 2>&1 || true
for d in stats stats_concurrent; do f=$(ls $out/$d/*/*_kernel_stats.csv | head -1); cp $f $out/${d}_kernel_stats.csv; done
rm -rf $out/stats $out/stats_concurrent
head -40 $out/timeline_concurrent.txt

set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4/t43_smoke.txt 2>&1 || { tail -20 gpurun_out/r4/t43_smoke.txt; exit 1; }
tail -3 gpurun_out/r4/t43_smoke.txt
timeout -k 10 800 python bench.py > gpurun_out/r4/t43_bench_default.txt 2> gpurun_out/r4/t43_bench_default.err || { tail -20 gpurun_out/r4/t43_bench_default.err; exit 1; }
python - <<'PY'
import json
for l in open('gpurun_out/r4/t43_bench_default.txt'):
    if l.startswith('{'):
        d=json.loads(l)
        print({k:d[k] for k in ('value','ms_per_step','ms_per_step_event_median','host_enqueue_ms_per_step','step_mfma_frac')})
        print('roofline', {k:d['roofline'][k] for k in ('achieved','frac','traffic')}, 'in_step', d['roofline']['in_step']['frac'])
        for leg in ('fp8','fp8_b256','vitl14','all_text_positions'):
            x=d.get(leg) or {}
            print(leg, x.get('ms_per_step'), x.get('value'), (x.get('roofline') or {}).get('frac'), json.dumps((x.get('roofline') or {}).get('fp8_family'))[:120])
PY
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r4/t43_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r4/t43_tests.txt
tail -4 gpurun_out/r4/t43_tests.txt

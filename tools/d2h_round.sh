cd ${GRAFT_REPO_ROOT:-/root/repo}; out=gpurun_out
timeout 420 python bench.py --d2h --steps 16 --d2h-legs devnull_cold,devnull,gzip_device --cpu-seconds 0 > $out/r03_bench_d2h.json 2> $out/r03_bench_d2h.err
echo rc=$?
timeout 330 python bench.py --d2h --steps 16 --reads-per-step 294912 --scratch-gb 36 --d2h-legs devnull_cold,devnull --cpu-seconds 0 > $out/r03_bench_d2h_49152.json 2> $out/r03_bench_d2h_49152.err
echo rc=$?
for f in r03_bench_d2h r03_bench_d2h_49152; do python - <<PY
import json
try:
    d=json.loads([l for l in open('$out/$f.json') if l.startswith('{')][-1])
    print('$f', round(d['value']/1e9,3), {k: round(v/1e9,3) for k,v in d.items() if k.startswith('value_incl')})
    for k,v in d['driver_end_to_end'].items():
        if isinstance(v, dict): print(' ', k, round(v['seconds'],1), v['consumer_thread_seconds'])
except Exception as e:
    print('$f failed', e); print(open('$out/$f.err').read()[-1500:])
PY
done

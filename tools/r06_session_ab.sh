cd ${GRAFT_REPO_ROOT:-/root/repo}
BRX_DEBUG=1 python bench.py --workload rough --steps 1 --warmup 0 --streams 1 --reads-per-step 65536 --scratch-gb 44 --cpu-seconds 0 > gpurun_out/ab_rough.out 2>&1
grep -a "final set" gpurun_out/ab_rough.out | grep -v " 64 reads" | cut -c1-300
grep -a '^{' gpurun_out/ab_rough.out | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
print('rough one batch', round(d['value']/1e9,3), round(d['ms_per_step']))
for k,v in sorted(d['kernels_per_device_batch'].items(), key=lambda kv:-kv[1]['ms'])[:4]: print('  ',k, round(v['ms'],1), v['launches'], round(v['mbases'],1))
"
python bench.py --workload rough --scratch-gb 42 --steps 2 --warmup 1 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('rough six in flight', round(d['value']/1e9,3), d['scratch_or_output_retries'], round(d['ms_per_step']))"
python bench.py --cpu-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('human', round(d['value']/1e9,3), d['scratch_or_output_retries'])"

cd ${GRAFT_REPO_ROOT:-/root/repo}
python bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err
python bench.py --workload hifi --cpu-seconds 0 > gpurun_out/r06_bench_hifi.json 2>> gpurun_out/r06_bench.err
python bench.py --workload kpn --cpu-seconds 0 > gpurun_out/r06_bench_kpn.json 2>> gpurun_out/r06_bench.err
python -c "
import json
for f in ('r06_bench','r06_bench_hifi','r06_bench_kpn'):
    d=json.load(open('gpurun_out/'+f+'.json')); print(f, round(d['value']/1e9,3), d['roofline_alu']['frac'], d['roofline_alu']['stale'], d.get('cpu_baseline') and d['cpu_baseline'].get('value'), d['roofline']['launch_ms'])"

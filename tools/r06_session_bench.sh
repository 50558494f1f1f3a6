cd ${GRAFT_REPO_ROOT:-/root/repo}
export BRX_ROUND_TAG=r06
python bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err
python bench.py --workload hifi --cpu-seconds 0 > gpurun_out/r06_bench_hifi.json 2>> gpurun_out/r06_bench.err
python bench.py --workload kpn --cpu-seconds 0 > gpurun_out/r06_bench_kpn.json 2>> gpurun_out/r06_bench.err
sleep 20      # as tools/cli_30x.sh: the driver clears the memory the runs above gave back; a process that maps 240 GB right behind them waits for it
timeout 600 python bench.py --scaling strong --cpu-seconds 0 2>> gpurun_out/r06_bench.err | grep '^{' > gpurun_out/r06_bench_strong_n1.json
sleep 20
BRX_DEVICE=0 BRX_DIST_BACKEND=gloo timeout 900 python bench.py --scaling strong --gpus 2 --streams 3 --reads-per-step 196608 --cpu-seconds 0 2>> gpurun_out/r06_bench.err | grep '^{' > gpurun_out/r06_bench_strong_n2_one_gpu.json
for i in 1 2 3; do bash tools/cli_30x.sh 30x 2>&1 | head -1 | cut -c1-600; cp gpurun_out/r06_cli_30x.json gpurun_out/r06_cli_30x_run$i.json; done
bash tools/cli_30x.sh 30x "--error_model pacbio2021 --qscore_model pacbio2021 --identity 30,3" 2>&1 | head -1 | cut -c1-600
bash tools/cli_30x.sh 30x "--gzip-device" 2>&1 | head -1 | cut -c1-600
python -c "
import json
for f in ('r06_bench','r06_bench_hifi','r06_bench_kpn','r06_bench_strong_n1','r06_bench_strong_n2_one_gpu'):
    d=json.load(open('gpurun_out/'+f+'.json')); print(f, round(d['value']/1e9,3), d.get('roofline',{}) and {k: d['roofline'].get(k) for k in ('kernel','frac','traffic')}, d.get('roofline_alu',{}) and {k: d['roofline_alu'].get(k) for k in ('frac','valu_per_base','stale')}, d.get('cpu_baseline') and d['cpu_baseline'].get('value'), d.get('fixed_cost_s'), d.get('loop_s'))"

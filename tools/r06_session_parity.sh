cd ${GRAFT_REPO_ROOT:-/root/repo}
{
echo "== new gpu tests"; timeout 1500 python -m pytest tests/test_gpu_golden.py tests/test_gzip_device.py -m gpu -q -x 2>&1 | tail -4
echo "== rough 65536 at 40 GB: what does the library ask for"; timeout 600 python bench.py --workload rough --steps 1 --warmup 0 --streams 1 --reads-per-step 65536 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('rough', round(d['value']/1e9,3), 'retries', d['scratch_or_output_retries'], d['retry_log'])"
echo "== strong N=2 on one GPU over gloo (R = 65536)"; BRX_DEVICE=0 BRX_DIST_BACKEND=gloo timeout 900 python bench.py --scaling strong --gpus 2 --streams 3 --reads-per-step 196608 --cpu-seconds 0 2> gpurun_out/r06_bench_strong_n2.err | grep '^{' > gpurun_out/r06_bench_strong_n2_one_gpu.json; python -c "
import json; d=json.load(open('gpurun_out/r06_bench_strong_n2_one_gpu.json')); print({k: d.get(k) for k in ('value','n_gpus','fixed_cost_s','loop_s','wall_s','value_loop','startup_s_slowest_rank','projected_wall_s','job','host_throttled','reference_preparation_s_not_in_the_clock')})"
echo "== strong N=1"; timeout 600 python bench.py --scaling strong --cpu-seconds 0 2> gpurun_out/r06_bench_strong_n1.err | grep '^{' > gpurun_out/r06_bench_strong_n1.json; python -c "
import json; d=json.load(open('gpurun_out/r06_bench_strong_n1.json')); print({k: d.get(k) for k in ('value','fixed_cost_s','loop_s','wall_s','value_loop','startup_s_slowest_rank','projected_wall_s','projected_speedup_vs_1','reference_preparation_s_not_in_the_clock')})"
echo "== fullsize human+hifi"; timeout 1500 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "configs3_and_4" 2>&1 | tail -4
} > gpurun_out/r06k.log 2>&1
tail -40 gpurun_out/r06k.log | cut -c1-1500

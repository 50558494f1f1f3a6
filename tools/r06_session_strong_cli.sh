cd ${GRAFT_REPO_ROOT:-/root/repo}
export BRX_ROUND_TAG=r06
{
echo "== gzip tests"; timeout 900 python -m pytest tests/test_gzip_device.py tests/test_gpu_cli.py -m gpu -q -x 2>&1 | tail -4
echo "== strong N=1"; timeout 600 python bench.py --scaling strong --cpu-seconds 0 > gpurun_out/r06_bench_strong_n1.json 2> gpurun_out/r06_bench_strong_n1.err; python -c "
import json; d=json.load(open('gpurun_out/r06_bench_strong_n1.json')); print({k: d[k] for k in ('value','fixed_cost_s','loop_s','wall_s','value_loop','startup_s_slowest_rank','projected_wall_s','projected_speedup_vs_1','job')})"
echo "== strong N=2 on one GPU over gloo"; BRX_DEVICE=0 BRX_DIST_BACKEND=gloo timeout 900 python bench.py --scaling strong --gpus 2 --streams 3 --cpu-seconds 0 > gpurun_out/r06_bench_strong_n2_one_gpu.json 2> gpurun_out/r06_bench_strong_n2.err; tail -3 gpurun_out/r06_bench_strong_n2.err; python -c "
import json; d=json.load(open('gpurun_out/r06_bench_strong_n2_one_gpu.json')); print({k: d.get(k) for k in ('value','n_gpus','fixed_cost_s','loop_s','wall_s','value_loop','startup_s_slowest_rank','projected_wall_s','job','host_throttled')})"
echo "== cli 30x"; bash tools/cli_30x.sh 30x 2>&1 | head -1 | cut -c1-900
echo "== cli 30x hifi gzip-device"; bash tools/cli_30x.sh 30x "--error_model pacbio2021 --qscore_model pacbio2021 --identity 30,3 --gzip-device" 2>&1 | head -1 | cut -c1-900
echo "== cli 30x ranks 2"; BRX_CLI_RANKS=2 bash tools/cli_30x.sh 30x 2>&1 | head -1 | cut -c1-900
} > gpurun_out/r06j.log 2>&1
tail -30 gpurun_out/r06j.log | cut -c1-1200

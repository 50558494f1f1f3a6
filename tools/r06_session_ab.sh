cd ${GRAFT_REPO_ROOT:-/root/repo}
export BRX_ROUND_TAG=r06
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q 2>&1 | tail -4
run() { tag=$1; shift; python bench.py --cpu-seconds 0 "$@" > gpurun_out/ab_$tag.json 2>> gpurun_out/ab.err; python -c "
import json; d=json.load(open('gpurun_out/ab_$tag.json')); print('$tag', round(d['value']/1e9,3), d['scratch_or_output_retries'], {k: round(v['ms']) for k,v in d['kernels_per_device_batch'].items() if v['ms']>100})"; }
run d32
run d32hifi --workload hifi
run d32kpn --workload kpn

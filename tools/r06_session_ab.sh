cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { tag=$1; shift; python bench.py --cpu-seconds 0 "$@" > gpurun_out/ab_$tag.json 2>> gpurun_out/ab.err; python -c "
import json; d=json.load(open('gpurun_out/ab_$tag.json')); print('$tag', round(d['value']/1e9,3), d['scratch_or_output_retries'], round(d['ms_per_step']), {k: round(v['ms']) for k,v in d['kernels_per_device_batch'].items() if v['ms']>150})"; }
run s6 --streams 6 --reads-per-step 393216 --scratch-gb 30
run s7 --streams 7 --reads-per-step 458752 --scratch-gb 30
run s8 --streams 8 --reads-per-step 524288 --scratch-gb 30
run s9 --streams 9 --reads-per-step 589824 --scratch-gb 28
run s6b --streams 6 --reads-per-step 393216 --scratch-gb 30

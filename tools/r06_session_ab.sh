cd ${GRAFT_REPO_ROOT:-/root/repo}
V=$PWD/badread_amd/csrc/variants/libbrx_lanetree.so
for i in 1 2; do
BRX_LIB_PATH=$V python bench.py --workload hifi --cpu-seconds 0 > gpurun_out/ab_hifi_old$i.json 2>> gpurun_out/ab.err
python bench.py --workload hifi --cpu-seconds 0 > gpurun_out/ab_hifi_new$i.json 2>> gpurun_out/ab.err
done
python bench.py --cpu-seconds 0 > gpurun_out/ab_human_new1.json 2>> gpurun_out/ab.err
BRX_LANES_MIN_READS=0 python bench.py --cpu-seconds 0 > gpurun_out/ab_human_min0.json 2>> gpurun_out/ab.err
python bench.py --cpu-seconds 0 > gpurun_out/ab_human_new2.json 2>> gpurun_out/ab.err
python -c "
import json,glob
for f in sorted(glob.glob('gpurun_out/ab_h*_*.json')):
    d=json.load(open(f)); print(f, round(d['value']/1e9,3), {k: (round(v['ms'],1), v['launches']) for k,v in d['kernels_per_device_batch'].items() if v['ms']>12})"
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -k "configs3_and_4" 2>&1 | tail -3

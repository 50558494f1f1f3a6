#!/bin/bash
# One GPU call at the end of a round: parity subset on the tree as it is, the previous library against it, a full bench line, PMC passes.
#   bash tools/ab_round.sh TAG OLD_LIB
cd ${GRAFT_REPO_ROOT:-/root/repo}; out=gpurun_out; tag=${1:-r03e}; old=${2:-badread_amd/csrc/variants/libbrx_hip_r03.so}
timeout 110 python -m pytest tests/test_gpu_align.py tests/test_gpu_pipeline.py tests/test_gpu_golden.py -q -x > $out/${tag}_pytest_subset.log 2>&1; echo "subset rc=$?"; tail -2 $out/${tag}_pytest_subset.log
BRX_LIB_PATH=$PWD/$old timeout 100 python bench.py --steps 3 --cpu-seconds 0 > $out/${tag}_old.json 2> $out/${tag}_old.err
timeout 150 python bench.py --cpu-seconds 6 > $out/${tag}_bench.json 2> $out/${tag}_bench.err
for f in old bench; do python -c "
import json
d=json.loads([l for l in open('$out/${tag}_$f.json') if l.startswith('{')][-1])
print('$f', round(d['value']/1e9,3), {k: round(v['ms'],1) for k,v in d['kernels_per_device_batch'].items()})
"; done
bash tools/profile_round.sh $tag human "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "FETCH_SIZE" "WRITE_SIZE" > $out/${tag}_profile.log 2>&1
python -c "
import json; d=json.load(open('$out/${tag}_valu_per_base.json'))['human']; print('valu_per_base', d['valu_per_base'], d['csrc_sha16'], {k: round(v,1) for k,v in list(d['per_kernel_valu_per_base'].items())[:8]})"

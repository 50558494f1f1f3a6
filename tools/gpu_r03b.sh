#!/bin/bash
# round 3, call b: first hardware run of the persistent mutate stage (parity under a watchdog, then A/B against the pass pipeline)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export BRX_WATCHDOG_S=120
BRX_DEBUG=1 timeout 1200 python -m pytest tests/test_gpu_golden.py tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py tests/test_gpu_cli.py -m gpu -x -q 2>gpurun_out/r03b_pytest.err | tail -15
echo "pytest rc=$?"; grep -c WATCHDOG gpurun_out/r03b_pytest.err
for cfg in "" "BRX_MUTATE_PERSIST=0" "BRX_PS_LONG=48" "BRX_PS_LONG=200" "BRX_PS_WG_PER_CU=2"; do
  env $cfg timeout 600 python bench.py --cpu-seconds 0 --steps 4 2>/tmp/err.txt | python -c "
import json, sys
try:
    d = json.loads(sys.stdin.read()); s = d['stage_ms_per_device_batch']; k = d['kernels_per_device_batch']
    print('[$cfg]', round(d['value'] / 1e9, 3), 'Gbases/s  mutate', round(s['mutate'], 1), 'final', round(s['final'], 1), {n: round(v.get('ms', 0), 1) for n, v in k.items()}, 'flagged', d.get('reads_flagged_band_segs_qmiss'))
except Exception as ex:
    print('[$cfg] failed:', ex, open('/tmp/err.txt').read()[-600:])"
done

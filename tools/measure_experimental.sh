#!/bin/bash
# First GPU call of the next round: the final-stage modes that were finished after round 2's GPU budget.
#   gpurun --timeout 1500 -- 'bash tools/measure_experimental.sh > gpurun_out/experimental.log 2>&1; tail -40 gpurun_out/experimental.log'
# 1. parity of the experimental routes on the MI355X (they are bit-exact on the interpreted kernels);
# 2. A/B of the default bench command with each mode and their combination (same box, back to back; ~45 s each).
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}
cd "$root"
BRX_TEST_EXPERIMENTAL=1 python -m pytest tests/test_gpu_pipeline.py -m gpu -q -k experimental 2>&1 | tail -4
for cfg in "" "BRX_FIN_PAIR=1" "BRX_FIN_WG=1" "BRX_FIN_PAIR=1 BRX_FIN_WG=1" ""; do
  env $cfg python bench.py --cpu-seconds 0 --steps 4 2>/tmp/exp_err.txt | python -c "
import json, sys
try:
    d = json.loads(sys.stdin.read()); s = d['stage_ms_per_device_batch']; k = d['kernels_per_device_batch']
    print('[$cfg]', round(d['value'] / 1e9, 3), 'Gbases/s  mutate', round(s['mutate'], 1), 'final', round(s['final'], 1),
          ' fin1', round(k.get('k_fin_align<1,1,1>', {}).get('ms', 0), 1), 'fin4', round(k.get('k_fin_align<4,4,4>', {}).get('ms', 0), 1),
          'fin16', round(k.get('k_fin_align<16,8,65535>', {}).get('ms', 0), 1), 'flagged', d.get('reads_flagged_band_segs_qmiss'))
except Exception as ex:
    print('[$cfg] failed:', ex, open('/tmp/exp_err.txt').read()[-400:])"
done

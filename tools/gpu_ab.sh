#!/bin/bash
# A/B of bench configurations on one box: tools/gpu_ab.sh "ENV=1 ..|bench args" ...   (each run bounded by a timeout)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for spec in "$@"; do
  envs="${spec%%|*}"; args="${spec#*|}"
  env $envs timeout 200 python bench.py --cpu-seconds 0 $args 2>/tmp/err.txt | python -c "
import json, sys
try:
    d = json.loads(sys.stdin.read()); s = d['stage_ms_per_device_batch']; k = d['kernels_per_device_batch']
    print('[$envs | $args]', round(d['value'] / 1e9, 3), 'Gbases/s  mutate', round(s['mutate'], 1), 'final', round(s['final'], 1), {n: (v.get('launches'), round(v.get('ms', 0), 1)) for n, v in k.items() if v.get('ms', 0) > 15},
          'retries', d.get('scratch_or_output_retries'), 'misses', d.get('traceback_window_misses_per_step'), 'host_cores', round(d.get('host_cpu', {}).get('busy_cores_per_rank', 0), 2),
          'lanes', round(d.get('roofline_alu', {}).get('useful_lane_frac_aligner_model') or 0, 3))
except Exception as ex:
    print('[$envs | $args] failed:', ex, open('/tmp/err.txt').read()[-300:])"
done

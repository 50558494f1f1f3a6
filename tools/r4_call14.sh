#!/bin/bash
# Round 4, GPU call 14: instruction counts of configs[4] (hifi) and configs[1] (kpn) on the final tree, then their bench lines with roofline_alu.
root=${GRAFT_REPO_ROOT:-/root/repo}; cd "$root"; out=gpurun_out
for wl in hifi kpn; do
  bash tools/profile_round.sh r04_$wl $wl "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" > $out/r04_${wl}_profile.log 2>&1
  cd "$root"
done
python - <<'PY'
import json
d = json.load(open('profiles/valu_per_base.json'))
for wl in ('hifi', 'kpn'):
    d.update(json.load(open(f'gpurun_out/r04_{wl}_valu_per_base.json')))
json.dump(d, open('profiles/valu_per_base.json', 'w'), indent=1)
json.dump(d, open('gpurun_out/r04_valu_per_base_all.json', 'w'), indent=1)
PY
timeout 300 python bench.py --workload hifi --cpu-seconds 8 > $out/r04_bench_hifi.json 2> $out/r04_bench_hifi.err
timeout 300 python bench.py --workload kpn --cpu-seconds 6 > $out/r04_bench_kpn.json 2> $out/r04_bench_kpn.err
for f in bench_hifi bench_kpn; do python -c "
import json
d=json.loads([l for l in open('$out/r04_$f.json') if l.startswith('{')][-1])
print('$f', round(d['value']/1e9,3), 'Gbases/s', (d.get('roofline_alu') or {}).get('valu_per_base'), (d.get('roofline_alu') or {}).get('frac'), (d.get('roofline_alu') or {}).get('stale'))
"; done

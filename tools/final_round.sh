#!/bin/bash
# The round's final measurements on one box (gpurun): profile passes, bench lines, the GPU test suite, the 30x job through the driver.
root=${GRAFT_REPO_ROOT:-/root/repo}; cd "$root"; out=gpurun_out; tag=${1:-r03}
bash tools/profile_round.sh $tag human "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "FETCH_SIZE" "WRITE_SIZE" > $out/${tag}_profile.log 2>&1
cd "$root"
cp $out/${tag}_valu_per_base.json profiles/valu_per_base.json          # bench.py's roofline_alu reads it (same tree: not stale)
python tools/pmc_traffic.py $out/${tag}_pmc_per_kernel.csv 65536 "k_mutate_seg<true, false>" human "k_mutate_seg<true>" 4 > profiles/pmc_traffic.json 2>> $out/${tag}_profile.log
timeout 400 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
timeout 300 python bench.py --workload hifi --cpu-seconds 8 > $out/${tag}_bench_hifi.json 2> $out/${tag}_bench_hifi.err
timeout 300 python bench.py --workload kpn --cpu-seconds 6 > $out/${tag}_bench_kpn.json 2> $out/${tag}_bench_kpn.err
timeout 1300 python -m pytest tests -m gpu -q > $out/${tag}_pytest_gpu.log 2>&1
timeout 900 python bench.py --d2h --steps 16 --d2h-legs devnull_cold,devnull,gzip_device --cpu-seconds 0 > $out/${tag}_bench_d2h.json 2> $out/${tag}_bench_d2h.err
for f in bench bench_hifi bench_kpn bench_d2h; do python -c "
import json
d=json.loads([l for l in open('$out/${tag}_$f.json') if l.startswith('{')][-1])
print('$f', round(d['value']/1e9,3), 'Gbases/s', {k: round(v/1e9,3) for k,v in d.items() if k.startswith('value_incl')}, d.get('roofline',{}).get('kernel'), d.get('roofline',{}).get('frac'), d.get('roofline_alu',{}).get('frac'), d.get('roofline_alu',{}).get('stale'), d.get('cpu_baseline',{}).get('value'))
"; done
tail -3 $out/${tag}_pytest_gpu.log

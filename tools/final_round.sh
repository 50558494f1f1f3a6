#!/bin/bash
# The round's final measurements on one box (gpurun): the product command FIRST (a user's process on a GPU nobody has dirtied: the
# driver clears memory it re-maps), then the GPU test suite (it is what the round is judged on), profile passes, bench lines.
#   bash tools/final_round.sh [tag]
root=${GRAFT_REPO_ROOT:-/root/repo}; cd "$root"; out=gpurun_out; tag=${1:-r05}; export BRX_ROUND_TAG=$tag
mkdir -p $out
bash tools/cli_30x.sh 30x > $out/${tag}_cli_30x.log 2>&1
bash tools/cli_30x.sh 30x "--error_model pacbio2021 --qscore_model pacbio2021 --identity 30,3" > $out/${tag}_cli_30x_hifi.log 2>&1
cp $out/${tag}_cli_30xerrormodelpacbio2021qscoremodelpacbio2021identity303.json $out/${tag}_cli_30x_hifi.json 2>/dev/null
timeout 1700 python -m pytest tests -m gpu -q > $out/${tag}_pytest_gpu.log 2>&1
tail -3 $out/${tag}_pytest_gpu.log
bash tools/profile_round.sh $tag human "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "FETCH_SIZE" "WRITE_SIZE" \
     "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" > $out/${tag}_profile.log 2>&1
cd "$root"
cp $out/${tag}_valu_per_base.json profiles/valu_per_base.json          # bench.py's roofline_alu reads it (same tree: not stale)
python tools/pmc_traffic.py $out/${tag}_pmc_per_kernel.csv 65536 --all human > profiles/pmc_traffic.json 2>> $out/${tag}_profile.log
cp profiles/pmc_traffic.json $out/${tag}_pmc_traffic.json          # profiles/ on the box is not merged back: gpurun_out/ is
# one batch alone on the chip with timestamps: what runs beside what inside a batch
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$root/$out/${tag}_tl" -o t -- \
    python "$root/bench.py" --steps 1 --warmup 1 --streams 1 --reads-per-step 65536 --cpu-seconds 0 > /dev/null 2> "$root/$out/${tag}_tl.err" )
python tools/batch_timeline.py $out/${tag}_tl/*kernel_trace.csv > $out/${tag}_batch_timeline.json 2>> $out/${tag}_tl.err; rm -rf $out/${tag}_tl
timeout 400 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
timeout 300 python bench.py --workload hifi --cpu-seconds 8 > $out/${tag}_bench_hifi.json 2> $out/${tag}_bench_hifi.err
timeout 300 python bench.py --workload kpn --cpu-seconds 6 > $out/${tag}_bench_kpn.json 2> $out/${tag}_bench_kpn.err
for f in bench bench_hifi bench_kpn; do python -c "
import json
d=json.loads([l for l in open('$out/${tag}_$f.json') if l.startswith('{')][-1])
print('$f', round(d['value']/1e9,3), 'Gbases/s', d.get('roofline',{}).get('kernel'), d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('traffic'), d.get('roofline_alu',{}).get('frac'), d.get('roofline_alu',{}).get('stale'), d.get('cpu_baseline',{}).get('value'), d.get('host_cpu',{}).get('busy_cores_per_rank'))
"; done
tail -2 $out/${tag}_cli_30x.log | cut -c1-600
tail -2 $out/${tag}_cli_30x_hifi.log | cut -c1-600
tail -3 $out/${tag}_pytest_gpu.log
cat $out/${tag}_batch_timeline.json | head -12

"""SQ_INSTS_VALU per simulated base from a PMC pass (tools/profile_round.sh): what bench.py's `roofline_alu` multiplies by
the measured bases/s.    python tools/valu_per_base.py <pmc_per_kernel.csv> <bench json of the same run> <workload> <tag>

The PMC run is `bench.py --steps 1 --warmup 1 --streams 1 --reads-per-step 49152`: two device batches of 49152 reads run one
after the other (warm-up + timed) plus the 64-read priming call, all counted; bases = both batches (the priming call's
64 reads are < 0.3 % and ignored)."""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    path, bench_json, workload, tag = sys.argv[1:5]
    line = [ln for ln in open(bench_json).read().splitlines() if ln.startswith('{')][-1]
    d = json.loads(line)
    batches = d['steps'] + d['warmup']
    bases = d['config']['bases_per_step_per_gpu'] * batches
    per_kernel, total = {}, 0.0
    other = {}
    for row in csv.DictReader(open(path)):
        if row['counter'] == 'SQ_INSTS_VALU':
            per_kernel[row['kernel']] = float(row['sum'])
            total += float(row['sum'])
        elif row['counter'] in ('SQ_ACTIVE_INST_VALU', 'SQ_THREAD_CYCLES_VALU', 'SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES'):
            other[row['counter']] = other.get(row['counter'], 0.0) + float(row['sum'])
    from badread_amd.build import source_hash
    extra = {}
    if other.get('SQ_ACTIVE_INST_VALU') and other.get('SQ_THREAD_CYCLES_VALU'):
        # lanes the EXEC mask leaves on, averaged over VALU issue cycles (a lane that computes a band cell the aligner later discards
        # still counts: the geometric share of useful band words is bench.py's useful_lane_frac_aligner_model)
        extra['exec_lane_frac'] = other['SQ_THREAD_CYCLES_VALU'] / (64.0 * other['SQ_ACTIVE_INST_VALU'])
        extra['exec_lane_frac_source'] = 'SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU), summed over all kernels of the same runs'
    out = {workload: {'valu_per_base': total / bases, 'bases_counted': bases, 'device_batches': batches, 'csrc_sha16': source_hash(), **extra,
                      'per_kernel_valu_per_base': {k: v / bases for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1])[:12]},
                      'source': f'profiles/{tag}_pmc_per_kernel.csv: rocprofv3 --pmc SQ_INSTS_VALU ... --kernel-trace, bench.py --workload {workload} '
                                f'--steps 1 --warmup 1 --streams 1 --reads-per-step 65536'}}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == '__main__':
    main()

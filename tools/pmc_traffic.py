"""profiles/pmc_traffic.json (what bench.py reports as roofline.traffic) from a per-kernel PMC summary:

    python tools/pmc_traffic.py <pmc_per_kernel.csv> <reads_per_step> [rocprof kernel name] [workload] [bench.py kernel label] [full-size launches] > profiles/pmc_traffic.json

FETCH_SIZE and WRITE_SIZE (KB) come from separate rocprofv3 passes (tools/profile_round.sh).  Per
MI355X_MICROARCH.md (HBM section) gfx950's FETCH_SIZE tallies 128-byte requests as 64 bytes, so it is doubled;
WRITE_SIZE is taken as reported.  The correction was calibrated there on wide streaming reads; this kernel's reads
are narrow (target bytes, traceback words), so the read half is an upper estimate."""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def one(path, reads, kernel, workload, label, full):
    vals = {}
    for row in csv.DictReader(open(path)):
        if row['kernel'].replace(' ', '').replace('void', '') == kernel.replace(' ', '').replace('void', '') and row['counter'] in ('FETCH_SIZE', 'WRITE_SIZE'):
            vals[row['counter']] = (float(row['sum']), int(row['dispatches']))
    # the counted run = two device batches + the 64-read priming call: a kernel launched a few times per batch is averaged over
    # its FULL-SIZE launches only (`full`; default: every dispatch)
    fetch, nf = vals['FETCH_SIZE']
    write, nw = vals['WRITE_SIZE']
    fetch, write = fetch / (full or nf), write / (full or nw)
    return {'kernel': label, 'rocprof_name': kernel, 'workload': workload, 'reads_per_step': reads,
            'FETCH_SIZE_KB_per_launch': fetch, 'WRITE_SIZE_KB_per_launch': write,
            'hbm_bytes_per_launch': (2.0 * fetch + write) * 1024.0,
            'averaged_over': f'{nf} / {nw} dispatches, {full or nf} launches'}


def main():
    """One kernel (the old command line), or with --all: every kernel bench.py may rank first, so that the line carries the traffic of
    whichever it does rank first (two kernels can be within a few per cent of each other)."""
    path, reads = sys.argv[1], int(sys.argv[2])
    from badread_amd.build import source_hash          # run on the tree the counters were collected on
    note = (f'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/profile_round.sh: bench.py --steps 1 --warmup 1 '
            f'--streams 1 --reads-per-step {reads}; the dispatches include the 64-read priming call); FETCH_SIZE doubled per '
            'MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B; calibrated for wide streaming reads only: upper estimate here); '
            'WRITE_SIZE as reported (KB)')
    if len(sys.argv) > 3 and sys.argv[3] == '--all':
        workload = sys.argv[4] if len(sys.argv) > 4 else 'human'
        names = sorted({row['kernel'] for row in csv.DictReader(open(path)) if row['counter'] == 'FETCH_SIZE'})
        per_pass = ('k_mut_apply', 'k_mut_post', 'k_pass_lists', 'k_win_lane', 'k_win_wave')
        out = {}
        for name in names:
            bare = name.replace('void ', '')
            # names as badread_amd.engine.KERNEL_NAMES has them (bench.py looks the run's top kernel up by that name)
            label = (bare.split('<')[0] if bare.startswith(('k_mutate_seg<', 'k_mut_')) else bare.replace(' ', ''))      # k_mutate_seg, k_mut_lanes, k_mut_fill, k_mut_post ...
            if not label.startswith(('k_mutate_seg', 'k_mut_', 'k_win_lane', 'k_fin_align', 'k_fin_qscore', 'k_fin_quad', 'k_fin_lanes')) or '_Z' in label or label in out:
                continue
            # few-launches-per-batch kernels: the two batches' full-size launches (k_mutate_seg: head + tail each; the others one or two)
            # (the 64-read priming call of bench.py launches k_mutate_seg, k_fin_align<1,1,1> and <2,2,2> once each: not counted)
            primed = ('k_mutate_seg<', 'k_fin_align<1, 1, 1>', 'k_fin_align<2, 2, 2>')
            n_disp = max(int(row['dispatches']) for row in csv.DictReader(open(path)) if row['kernel'] == name and row['counter'] == 'FETCH_SIZE')
            full = None if any(x in bare for x in per_pass) else (max(n_disp - 1, 1) if any(x in bare for x in primed) else None)
            try:
                out[label] = one(path, reads, name, workload, label, full)
            except KeyError:
                continue
        json.dump({'workload': workload, 'reads_per_step': reads, 'csrc_sha16': source_hash(), 'kernels': out, 'note': note}, sys.stdout, indent=1)
        print()
        return
    kernel = sys.argv[3] if len(sys.argv) > 3 else 'k_fin_align<1, 1, 1>'
    workload = sys.argv[4] if len(sys.argv) > 4 else 'kpn'
    label = sys.argv[5] if len(sys.argv) > 5 else kernel.replace(' ', '')
    full = int(sys.argv[6]) if len(sys.argv) > 6 and sys.argv[6] else None
    rec = one(path, reads, kernel, workload, label, full)
    rec.update(csrc_sha16=source_hash(), note=note)
    json.dump(rec, sys.stdout, indent=1)
    print()


if __name__ == '__main__':
    main()

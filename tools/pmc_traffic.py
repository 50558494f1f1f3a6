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


def main():
    path, reads = sys.argv[1], int(sys.argv[2])
    kernel = sys.argv[3] if len(sys.argv) > 3 else 'k_fin_align<1, 1, 1>'
    workload = sys.argv[4] if len(sys.argv) > 4 else 'kpn'
    label = sys.argv[5] if len(sys.argv) > 5 else kernel.replace(' ', '')
    vals = {}
    for row in csv.DictReader(open(path)):
        if row['kernel'].replace(' ', '') == kernel.replace(' ', '') and row['counter'] in ('FETCH_SIZE', 'WRITE_SIZE'):
            vals[row['counter']] = (float(row['sum']), int(row['dispatches']))
    # the counted run = two device batches + the 64-read priming call: a kernel launched a few times per batch is averaged over
    # its FULL-SIZE launches only (argument 6; default: every dispatch)
    full = int(sys.argv[6]) if len(sys.argv) > 6 else None
    fetch, nf = vals['FETCH_SIZE']
    write, nw = vals['WRITE_SIZE']
    fetch, write = fetch / (full or nf), write / (full or nw)
    from badread_amd.build import source_hash          # run on the tree the counters were collected on (tools/profile_round.sh does)
    json.dump({'kernel': label, 'rocprof_name': kernel, 'workload': workload, 'reads_per_step': reads, 'csrc_sha16': source_hash(),
               'FETCH_SIZE_KB_per_launch': fetch, 'WRITE_SIZE_KB_per_launch': write,
               'hbm_bytes_per_launch': (2.0 * fetch + write) * 1024.0,
               'note': f'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/profile_round.sh: bench.py --steps 1 '
                       f'--warmup 1 --streams 1 --reads-per-step {reads}; the dispatches include the 64-read priming call), sum over {nf} / {nw} dispatches / {full or nf} full-size launches; FETCH_SIZE doubled per MI355X_MICROARCH.md '
                       f'(gfx950 counts 128-B requests as 64 B; calibrated for wide streaming reads only: upper estimate '
                       f'here); WRITE_SIZE as reported (KB)'}, sys.stdout, indent=1)
    print()


if __name__ == '__main__':
    main()

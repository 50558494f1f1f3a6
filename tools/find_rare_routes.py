"""Scan device batches of the bench workload (configs[3]) for reads that take the rare routes of the final stage at FULL
size: a traceback that leaves the 2 sqrt(ub) + 24 window (second phase with the full store) and a band beyond 16 words per
lane (the memory-resident wide path).  Needs a GPU.   python tools/find_rare_routes.py [first_batch] [n_batches] [workload]
Prints one JSON line per hit: {"read": index, "route": "window_miss" | "wide", ...}; tests/test_gpu_fullsize.py pins them."""
import io
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def main():
    first_batch = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n_batches = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    wlname = sys.argv[3] if len(sys.argv) > 3 else 'human'
    R = 65536
    from badread_amd.engine import HipEngine
    eng = bench.configure(HipEngine(0, scratch_bytes=int(bench.SCRATCH_GB_DEFAULT * (1 << 30))), bench.build_workload(io.StringIO(), wlname, bench.default_ref_dir()))
    hits = 0
    for b in range(first_batch, first_batch + n_batches):
        out, st = eng.simulate_batch_device(bench.SEED, b * R, R, expected_bytes=R * 36000)
        cyc = eng.read_cycles(R)
        miss = np.flatnonzero(cyc[:, 2])
        wide = np.flatnonzero((cyc[:, 7] & 0xFFFF) > 16)
        for route, idx in (('window_miss', miss), ('wide', wide)):
            for r in idx.tolist():
                s = st[r]
                hits += 1
                print(json.dumps({'read': b * R + r, 'route': route, 'batch': b, 'words_per_lane': int(cyc[r, 7]) & 0xFFFF, 'padded_len': int(s['padded_len']),
                                  'distance': int(s['n_cols']) - int(s['n_match']), 'changes': int(s['change_count']), 'status': int(s['status'])}), flush=True)
        print(f'# batch {b}: misses {len(miss)} wide {len(wide)} engine_misses {eng.window_misses()} max_words {int((cyc[:, 7] & 0xFFFF).max())}', file=sys.stderr, flush=True)
    print(json.dumps({'scanned_reads': n_batches * R, 'first_read': first_batch * R, 'hits': hits}))


if __name__ == '__main__':
    main()

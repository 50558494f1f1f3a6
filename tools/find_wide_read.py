"""Scan read indices of a bench workload ON THE CPU (the oracle's plan probe: fragment length and target identity of a read, no
mutation, no alignment) for reads whose final band will be beyond 16 words per lane -- expected changes n (1 - identity) above
57 344 -- so that tests/golden/rare_routes.json can pin one for the GPU (the memory-resident wide path of brx_align.h).
    python tools/find_wide_read.py [workload] [first] [count] [threshold]"""
import io
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'oracle'), os.path.join(REPO, 'tools')):
    sys.path.insert(0, p)
import bench  # noqa: E402
import pyoracle  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else 'wide'
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    count = int(sys.argv[3]) if len(sys.argv) > 3 else 200000
    thr = float(sys.argv[4]) if len(sys.argv) > 4 else 62000.0
    eng = bench.configure(pyoracle.OracleEngine(), bench.build_workload(io.StringIO(), wl, bench.default_ref_dir()))
    hits = 0
    for r in range(first, first + count):
        p = eng.plan(bench.SEED, r)
        exp = p['frag_len'] * (1.0 - p['target'])
        if exp >= thr and p['status'] == 0:
            hits += 1
            print(json.dumps({'read': r, 'frag_len': p['frag_len'], 'target': round(p['target'], 4), 'expected_changes': int(exp), 'pieces': len(p['pieces'])}), flush=True)
            if hits >= 8:
                break
    print(json.dumps({'scanned': [first, r + 1], 'hits': hits}))


if __name__ == '__main__':
    main()

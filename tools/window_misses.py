"""Which reads of a bench-workload batch leave the windowed traceback store?  (needs a GPU)

python tools/window_misses.py [n_reads] -> per missing read: fragment / read length, edit bound, distance,
band class, and the host's view of the machine (cgroup CPU quota), for DESIGN.md section 4.
"""
import io
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/proc/loadavg'):
        if os.path.isfile(path):
            print(path, open(path).read().strip())
    print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())
    from badread_amd.engine import HipEngine
    eng = bench.configure(HipEngine(0, scratch_bytes=24 << 30), bench.build_workload(io.StringIO()))
    out, st = eng.simulate_batch(bench.SEED, 0, n)
    cyc = eng.read_cycles(n)
    miss = np.nonzero(cyc[:, 2])[0]
    print('reads', n, 'misses', len(miss), 'engine says', eng.window_misses(), 'chunks', eng.final_launches())
    print('fields', st.dtype.names)
    for r in miss[:60]:
        s = st[r]
        d = int(s['n_cols']) - int(s['n_match'])
        print(f'read {r}: frag_len={int(s["frag_len"])} padded={int(s["padded_len"])} seq_len={int(s["seq_len"])} n_cols={int(s["n_cols"])} '
              f'dist={d} changes={int(s["change_count"])} G={int(cyc[r, 7])} status={int(s["status"]):#x} sqrt_d={d ** 0.5:.1f}')
    g = cyc[:, 7]
    for G in sorted(set(g.tolist())):
        sel = g == G
        print('class', G, 'reads', int(sel.sum()), 'bases', int(st['seq_len'][sel].sum()), 'misses', int((cyc[sel, 2] != 0).sum()))


if __name__ == '__main__':
    main()

"""Per-read records of the final stage of ONE device batch of a bench workload (brx_last_read_cycles: [3] shader clocks the read's
-- or its group's -- final alignment took, [7] band class, bit 16 = aligned as one of four): where the time of each final-stage
class goes.  Needs a GPU.   python tools/quad_probe.py [workload] [reads]"""
import io
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else 'human'
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    from badread_amd.engine import HipEngine
    eng = bench.configure(HipEngine(0, scratch_bytes=int(bench.SCRATCH_GB_DEFAULT * (1 << 30))), bench.build_workload(io.StringIO(), wl, bench.default_ref_dir()))
    eng.set_kernel_timing(True)
    out, st = eng.simulate_batch_device(bench.SEED, 0, R, expected_bytes=R * 36000)
    cyc = eng.read_cycles(R)
    ks = eng.kernel_stats()
    words, quad = (cyc[:, 7] & 0xFFFF).astype(np.int64), ((cyc[:, 7] >> 16) & 1).astype(bool)
    n = st['frag_len'].astype(np.float64)
    d = st['n_cols'].astype(np.float64) - st['n_match']
    res = {'kernels_ms': {k: round(v[1], 1) for k, v in ks.items() if v[1] > 1.0}}
    for name, sel in (('quad', quad), ('wave_1', ~quad & (words == 1)), ('wave_2', ~quad & (words == 2)), ('wave_4', ~quad & (words == 4)), ('wave_8+', ~quad & (words >= 8))):
        if not sel.any():
            continue
        c = cyc[sel, 3].astype(np.float64)
        order = np.argsort(-c)[:5]
        res[name] = {'reads': int(sel.sum()), 'bases': float(n[sel].sum()), 'clocks_sum': float(c.sum()), 'clocks_max': float(c.max()),
                     'clocks_per_base': float(c.sum() / max(n[sel].sum(), 1.0)), 'distance_max': float(d[sel].max()), 'length_max': float(n[sel].max()),
                     'slowest': [{'clocks': float(c[i]), 'length': float(n[sel][i]), 'distance': float(d[sel][i])} for i in order]}
    print(json.dumps(res))


if __name__ == '__main__':
    main()

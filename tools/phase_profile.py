"""
Mutate-stage phase breakdown on the bench workload (BRX_PROFILE=1 kernels): shader-clock time per phase of
k_mutate_seg summed over reads, and per read-length bucket.  Usage: python tools/phase_profile.py [n_reads]
"""
import io
import os
import sys

os.environ['BRX_PROFILE'] = '1'
import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
from badread_amd.engine import HipEngine  # noqa: E402
wl = bench.build_workload(io.StringIO())
eng = bench.configure(HipEngine(0, scratch_bytes=40 << 30), wl)
eng.simulate_batch_device(42, 0, n, expected_bytes=n * 36000)
out, st = eng.simulate_batch_device(42, n, n, expected_bytes=n * 36000)
ph = eng.phase_cycles(n).astype(np.float64)
clk = eng.read_cycles(n).astype(np.float64)
print('stage ms', eng.stage_ms(), 'passes', eng.mutate_passes())
names = ['propose', 'apply', 'park', 'inplace_align', 'other', 'ia_fwd', 'ia_tb']
tot = ph.sum(axis=0)
print('total Mcycles per phase:', ' '.join(f'{a}={v / 1e6:.1f}' for a, v in zip(names, tot[:7])), 'mut_total', clk[:, 0].sum() / 1e6)
L = st['padded_len'].astype(np.float64)
na = st['n_alignments'].astype(np.float64)
edges = [0, 1000, 3000, 8000, 15000, 30000, 60000, 10 ** 9]
for lo, hi in zip(edges[:-1], edges[1:]):
    sel = (L >= lo) & (L < hi)
    if not sel.any():
        continue
    t = ph[sel].sum(axis=0)
    a = max(na[sel].sum(), 1.0)
    print(f'{lo:>7d}-{hi:<8d} reads {int(sel.sum()):>6d} aligns {int(a):>8d}  kcycles per alignment cycle: ' +
          ' '.join(f'{nm}={v / a / 1e3:.1f}' for nm, v in zip(names, t[:7])))

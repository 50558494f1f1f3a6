"""
Per-read cost breakdown of the two heavy kernels on the bench workload: runs one batch and prints,
per read-length bucket, the shader cycles spent in k_mutate / in-loop alignments / final forward /
traceback / qscore lookup (brx_last_read_cycles).  Usage: python tools/read_profile.py [n_reads]
"""
import io
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
from badread_amd.engine import HipEngine  # noqa: E402
wl = bench.build_workload(io.StringIO())
eng = bench.configure(HipEngine(0, scratch_bytes=48 << 30), wl)
eng.simulate_batch_device(42, 0, n, expected_bytes=n * 36000)
out, st = eng.simulate_batch_device(42, n, n, expected_bytes=n * 36000)
clk = eng.read_cycles(n).astype(np.float64)
print('stage ms', eng.stage_ms())
L = st['padded_len'].astype(np.float64)
ed = (st['n_cols'] - st['n_match']).astype(np.float64)
names = ['mut_total', 'mut_fwd', 'mut_tb', 'fin_total', 'fin_fwd', 'fin_tb', 'fin_qs', 'G']
edges = [0, 1000, 3000, 8000, 15000, 30000, 60000, 100000, 10 ** 9]
print(f'{"bucket":>16s} {"reads":>6s} {"bases":>10s} ' + ' '.join(f'{x:>10s}' for x in names[:7]) + '   G   cyc/base(mut,fin)  naligns loops/base')
for lo, hi in zip(edges[:-1], edges[1:]):
    sel = (L >= lo) & (L < hi)
    if not sel.any():
        continue
    tot = clk[sel].sum(axis=0)
    b = L[sel].sum()
    print(f'{lo:>7d}-{hi:<8d} {int(sel.sum()):>6d} {int(b):>10d} ' + ' '.join(f'{x / 1e6:>10.2f}' for x in tot[:7]) +
          f'  {clk[sel][:, 7].mean():4.1f}  {tot[0] / b:8.1f} {tot[3] / b:8.1f}  {st["n_alignments"][sel].mean():7.1f} {st["loop_count"][sel].sum() / b:6.2f}')
tot = clk.sum(axis=0)
print('total Mcycles', ' '.join(f'{n_}={v / 1e6:.1f}' for n_, v in zip(names[:7], tot[:7])))
print('max per-read Mcycles: mutate', clk[:, 0].max() / 1e6, 'final', clk[:, 3].max() / 1e6, 'len of slowest final', L[clk[:, 3].argmax()], 'G', clk[clk[:, 3].argmax(), 7], 'ed', ed[clk[:, 3].argmax()])
order = np.argsort(-clk[:, 3])[:10]
for i in order:
    print(f'  read {i}: len {int(L[i])} ed {int(ed[i])} G {int(clk[i,7])} final {clk[i,3]/1e6:.1f}M (fwd {clk[i,4]/1e6:.1f} tb {clk[i,5]/1e6:.1f} qs {clk[i,6]/1e6:.1f}) mutate {clk[i,0]/1e6:.1f}M')
json.dump({'clk': clk.tolist(), 'len': L.tolist(), 'ed': ed.tolist()}, open(os.path.join(REPO, 'gpurun_out', 'read_profile.json'), 'w'))

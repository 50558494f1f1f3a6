"""
Micro-benchmark of brx_align_batch: N pairs of one length / error rate, proven band hint, no ops
output.  Prints ms and wave-cycles per column (time x 2.4 GHz x min(N, resident waves) / columns).
Usage: python tools/align_bench.py LEN RATE N [N ...]
"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'oracle'), os.path.join(REPO, 'tests')):
    sys.path.insert(0, p)
import helpers as H  # noqa: E402
from badread_amd.engine import HipEngine  # noqa: E402

L, rate = int(sys.argv[1]), float(sys.argv[2])
counts = [int(x) for x in sys.argv[3:]] or [1, 64, 1024]
rng = np.random.default_rng(1)
eng = HipEngine(0, scratch_bytes=24 << 30)
base_q, base_t = [], []
for i in range(8):
    q = H.random_dna(rng, L)
    base_q.append(q.encode())
    base_t.append(H.mutate_seq(rng, q, rate).encode())
k = int(L * rate * 1.25) + 32
for n in counts:
    qs = [base_q[i % 8] for i in range(n)]
    ts = [base_t[i % 8] for i in range(n)]
    eng.align_batch(qs[:1], ts[:1], k_hint=[k], want_ops=False)
    t0 = time.perf_counter()
    _, dist, ncols, _ = eng.align_batch(qs, ts, k_hint=[k] * n, want_ops=False)
    dt = time.perf_counter() - t0
    cols = sum(len(t) for t in ts)
    kms = eng.stage_ms()['final']
    print(f'L={L} rate={rate} k={k} pairs={n}: wall {dt * 1e3:.2f} ms, kernel {kms:.2f} ms in {eng.final_launches()} launch(es), '
          f'dist[0]={dist[0]}, {kms * 1e-3 * 2.4e9 / (cols / n):.0f} cycles/column/wave if all concurrent, '
          f'{cols / (kms * 1e-3) / 1e9:.3f} Gcolumns/s', flush=True)

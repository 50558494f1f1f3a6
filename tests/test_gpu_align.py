"""
GPU parity: brx_align_batch (the HIP Myers kernel behind every edlib.align call site,
simulate.py:330,340; qscore_model.py:37; error_model.py:202) against the CPU oracle, bit-exact on
distance AND on the canonical path.
"""
import numpy as np
import pytest

import helpers as H
import pyoracle

pytestmark = pytest.mark.gpu


def _check(queries, targets, k_hint=None):
    eng = H.hip_engine()
    ops, dist, ncols, nmatch = eng.align_batch(queries, targets, k_hint=k_hint)
    for i, (q, t) in enumerate(zip(queries, targets)):
        d, o = pyoracle.align(q, t)
        assert dist[i] == d, f'pair {i}: distance {dist[i]} != {d} (|q|={len(q)}, |t|={len(t)})'
        assert ncols[i] == len(o)
        assert nmatch[i] == int((o == 0).sum())
        fd = H.first_diff(ops[i], o)
        assert fd < 0, f'pair {i}: path differs at column {fd} (|q|={len(q)}, |t|={len(t)}, d={d})'


def test_reference_test_vectors():
    # alignments the reference's own tests pin (test_error_model.py:30-150 inner k-mers, test_qscore_model.py:324-419)
    pairs = [('CG', 'CGT'), ('CGT', 'CG'), ('C', 'CG'), ('ACGT', 'ACGT'), ('GATTACA', 'GATACA'),
             ('ACGACTAGCTACG', 'ACGACTAGCTACG'), ('ACGACTAGCTACG', 'ACGACTGCTACG'), ('A', 'C'), ('A', 'A'),
             ('AAAA', 'A'), ('A', 'AAAA'), ('ACGTACGTACGT', 'TGCATGCATGCA')]
    _check([q.encode() for q, _ in pairs], [t.encode() for _, t in pairs])


def test_random_pairs_small():
    rng = np.random.default_rng(5)
    qs, ts = [], []
    for it in range(1500):
        n = int(rng.choice([1, 2, 3, 5, 10, 31, 32, 33, 63, 64, 65, 100, 128, 129, 200, 500, 1000, 1500]))
        q = H.random_dna(rng, n, 'ACGT' if it % 3 else 'AC')
        if it % 5 == 0:
            t = H.random_dna(rng, int(rng.integers(1, 2 * n + 1)))
        else:
            t = H.mutate_seq(rng, q, float(rng.choice([0, 0.01, 0.05, 0.2, 0.5]))) or 'A'
        qs.append(q.encode())
        ts.append(t.encode())
    _check(qs, ts)


def test_non_acgt_symbols_and_empty():
    rng = np.random.default_rng(6)
    qs, ts = [], []
    for _ in range(100):
        q = H.random_dna(rng, int(rng.integers(1, 400)), 'ACGTNRYK')
        t = H.mutate_seq(rng, q, 0.1) or 'N'
        qs.append(q.encode())
        ts.append(t.encode())
    qs += [b'', b'ACGT', b'NNNNNNNN']
    ts += [b'ACG', b'', b'NNNNNNN']
    _check(qs, ts)


def test_mid_sizes_with_and_without_bound():
    rng = np.random.default_rng(7)
    qs, ts, ks = [], [], []
    for n, rate in ((3000, 0.05), (5000, 0.1), (15000, 0.05), (15000, 0.15), (30000, 0.03), (2000, 0.4)):
        q = H.random_dna(rng, n)
        t = H.mutate_seq(rng, q, rate)
        qs.append(q.encode())
        ts.append(t.encode())
        ks.append(pyoracle.align(q.encode(), t.encode(), want_ops=False)[0] + 7)
    _check(qs, ts)
    _check(qs, ts, k_hint=ks)


def test_long_reads():
    rng = np.random.default_rng(8)
    qs, ts, ks = [], [], []
    for n, rate in ((60000, 0.05), (120000, 0.08), (200000, 0.03), (50000, 0.25)):
        q = H.random_dna(rng, n)
        t = H.mutate_seq(rng, q, rate)
        qs.append(q.encode())
        ts.append(t.encode())
        ks.append(int(n * rate * 1.3) + 50)
    _check(qs, ts, k_hint=ks)

"""
Statistical parity with the running reference (gate 3): the reference's Mersenne-Twister
streams cannot be reproduced by a counter-based generator, so whole-run parity is distributional.  The fixture
tests/golden/ks_reference.npz holds >= 10 000 reads simulated by the unmodified reference CLI
(tools/make_ks_fixture.py); `check(engine)` simulates the same workload through our driver with `engine` and
requires two-sample Kolmogorov-Smirnov D < 1.63 sqrt((n+m)/(nm)) (alpha = 0.01) for read length, read identity,
error-free length - length and per-read mean qscore, plus a total-variation bound on the qscore histogram.
"""
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(REPO, 'tools'))
import make_ks_fixture as K  # noqa: E402   (workload definition + FASTQ parser; does not touch /root/reference on import)


def ks_two_sample(a, b):
    a, b = np.sort(np.asarray(a, np.float64)), np.sort(np.asarray(b, np.float64))
    grid = np.concatenate([a, b])
    fa = np.searchsorted(a, grid, side='right') / len(a)
    fb = np.searchsorted(b, grid, side='right') / len(b)
    return float(np.abs(fa - fb).max())


def check(engine, tmp_path):
    from badread_amd import simulate as S
    from test_host_simulate import Args
    ref = np.load(os.path.join(HERE, 'golden', 'ks_reference.npz'))
    n_ref = len(ref['length'])
    assert n_ref >= 10000
    fasta = os.path.join(str(tmp_path), 'ks_ref.fasta')
    K.write_fasta(fasta)
    mean, sd = (float(x) for x in K.LENGTH.split(','))
    i_mean, i_max, i_sd = (float(x) for x in K.IDENTITY.split(','))
    args = Args(reference=fasta, quantity=str(int(ref['length'].astype(np.int64).sum())), mean_frag_length=mean,
                frag_length_stdev=sd, mean_identity=i_mean, max_identity=i_max, identity_stdev=i_sd,
                error_model='nanopore2023', qscore_model='nanopore2023', seed=77)
    sink = io.BytesIO()
    S.simulate(args, output=io.StringIO(), engine=engine, stdout=sink, shard=S.Shard())
    L, I, D, Q, hist = K.parse(sink.getvalue())
    n = len(L)
    assert n >= 10000
    crit = 1.63 * np.sqrt((n + n_ref) / (n * n_ref))
    report = {}
    for name, ours, theirs in (('length', L, ref['length']), ('identity', I, ref['identity']),
                               ('trimmed', D, ref['trimmed']), ('mean_q', Q, ref['mean_q'])):
        report[name] = ks_two_sample(ours, theirs)
    # read_identity is printed with 3 decimals and `trimmed` is a small integer: ties are shared by both samples
    for name, d in report.items():
        assert d < crit, f'KS {name}: D = {d:.4f} >= {crit:.4f} (n = {n}, m = {n_ref}); all: {report}'
    p, q = hist / hist.sum(), ref['qhist'] / ref['qhist'].sum()
    tv = 0.5 * float(np.abs(p - q).sum())
    assert tv < 0.01, f'qscore histogram total variation {tv:.4f}'
    return report, crit, tv

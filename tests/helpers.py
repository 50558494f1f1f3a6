"""Shared fixtures for the parity tests: synthetic references, model tables, engines."""
import collections
import io
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)

from badread_amd.engine import SimParams  # noqa: E402
from badread_amd.error_model import ErrorModel  # noqa: E402
from badread_amd.qscore_model import QScoreModel  # noqa: E402
from badread_amd.reference import PackedReference  # noqa: E402

NULL = io.StringIO()
_cache = {}


def random_dna(rng, n, alphabet='ACGT'):
    return ''.join(np.array(list(alphabet))[rng.integers(0, len(alphabet), n)])


def small_reference(seed=1, with_n=True):
    """Three contigs: circular chromosome, circular plasmid with depth, linear contig with N/IUPAC runs and hairpins."""
    key = ('ref', seed, with_n)
    if key in _cache:
        return _cache[key]
    rng = np.random.default_rng(seed)
    chrom = random_dna(rng, 60000)
    plasmid = random_dna(rng, 4000)
    lin = list(random_dna(rng, 30000))
    if with_n:
        lin[100:400] = 'N' * 300
        lin[5000:5003] = 'RYK'
        lin[29000:29950] = 'N' * 950
    seqs = collections.OrderedDict([('chrom', chrom), ('plasmid', plasmid), ('lin1', ''.join(lin)),
                                    ('hp', random_dna(rng, 9000))])
    depths = {'chrom': 1.0, 'plasmid': 5.0, 'lin1': 1.5, 'hp': 1.0}
    circ = {'chrom': True, 'plasmid': True, 'lin1': False, 'hp': False}
    hl = {'chrom': False, 'plasmid': False, 'lin1': False, 'hp': True}
    hr = {'chrom': False, 'plasmid': False, 'lin1': False, 'hp': True}
    pref = PackedReference.from_seqs(seqs, depths, circ, hl, hr)
    _cache[key] = (pref, seqs)
    return _cache[key]


def error_tables(name):
    key = ('em', name)
    if key not in _cache:
        _cache[key] = ErrorModel(name, output=NULL).tables()
    return _cache[key]


def qscore_tables(name):
    key = ('qm', name)
    if key not in _cache:
        _cache[key] = QScoreModel(name, output=NULL).tables()
    return _cache[key]


def configure(engine, pref, em='random', qm='ideal', params=None):
    engine.set_reference(pref)
    engine.set_error_model(error_tables(em))
    engine.set_qscore_model(qscore_tables(qm))
    engine.set_params(params or SimParams())
    return engine


def oracle_engine():
    from pyoracle import OracleEngine
    return OracleEngine()


def hip_engine():
    from badread_amd.engine import default_engine
    return default_engine()


def first_diff(a, b):
    a, b = np.asarray(a), np.asarray(b)
    n = min(len(a), len(b))
    d = np.flatnonzero(a[:n] != b[:n])
    if len(d):
        return int(d[0])
    return n if len(a) != len(b) else -1


def mutate_seq(rng, s, rate):
    out = []
    for ch in s:
        r = rng.random()
        if r < rate / 3:
            continue
        if r < 2 * rate / 3:
            out.append('ACGT'[rng.integers(0, 4)])
            continue
        out.append(ch)
        if r < rate:
            out.append('ACGT'[rng.integers(0, 4)])
    return ''.join(out)


def identity_tolerance_check(engine, models=('random', 'nanopore2018', 'nanopore2020', 'nanopore2023', 'pacbio2016', 'pacbio2021'),
                             identities=(1.0, 0.9, 0.8), lengths=(3000, 1000), trials=20, seed=1234):
    """The reference's own acceptance test of sequence_fragment (test/test_simulate.py:57-163) against `engine`:
    for every error model (qscore model 'random'), target identity and read length, `trials` random fragments;
    each read's identity (fragment aligned against the read, misc.identity_from_edlib_cigar) within 0.5 x the target
    error of the target, and their mean within 0.05 x.  Returns the number of reads checked."""
    import statistics
    rng = np.random.default_rng(seed)
    engine.set_qscore_model(qscore_tables('random'))
    checked = 0
    first = 0
    for model in models:
        engine.set_error_model(error_tables(model))
        for target in identities:
            for length in lengths:
                frags = [rng.integers(0, 4, length).astype(np.uint8) for _ in range(trials)]
                res, _ = engine.sequence_fragments(seed, first, frags, [target] * trials)
                first += trials
                seqs = [r[0] for r in res]
                # test_simulate.py:85 aligns the fragment (query) against the read (target)
                _, _, ncols, nmatch = engine.align_batch([bytes(f) for f in frags], [bytes(s) for s in seqs], want_ops=False)
                ids = [nm / nc if nc else 0.0 for nm, nc in zip(nmatch.tolist(), ncols.tolist())]
                err = 1.0 - target
                for i, ident in enumerate(ids):
                    assert abs(ident - target) <= 0.5 * err + 1e-12, (model, target, length, i, ident)
                assert abs(statistics.mean(ids) - target) <= 0.05 * err + 1e-12, (model, target, length, statistics.mean(ids))
                checked += trials
    return checked


def recipe_fragment(seed, length, with_n=False):
    """The fragment of a digest case of tests/golden/sequence_fragment_bound.json.gz (tools/make_golden.py holds the same function):
    base i = splitmix64(seed << 32 | i) >> 62, one base in 256 an N if with_n.  Returns codes 0-4 (uint8)."""
    x = (np.uint64(seed) << np.uint64(32)) + np.arange(length, dtype=np.uint64)
    with np.errstate(over='ignore'):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    codes = (x >> np.uint64(62)).astype(np.uint8)
    if with_n:
        codes[((x >> np.uint64(20)) & np.uint64(255)) == 0] = 4
    return codes


def check_digest_cases(engine_of, cases):
    """Every digest case through `engine_of(em, qm)` (an engine configured with that model pair): sha256 of the sequence and of the
    qualities, the identity as the same double, the loop count."""
    import hashlib
    for c in cases:
        eng = engine_of(c['em'], c['qm'])
        res, st = eng.sequence_fragments(c['seed'], c['read'], [recipe_fragment(c['seed'], c['length'], c['with_n'])], [c['target']])
        tag = (c['em'], c['length'], c['target'], c['seed'])
        seq = ''.join('ACGTN'[x] for x in res[0][0])
        assert len(seq) == c['seq_len'] and hashlib.sha256(seq.encode()).hexdigest() == c['seq_sha256'], tag
        assert hashlib.sha256(res[0][1].tobytes()).hexdigest() == c['qual_sha256'], tag
        identity = st['n_match'][0] / st['n_cols'][0] if st['n_cols'][0] else 0.0
        assert identity == c['identity'], tag
        assert st['loop_count'][0] in (c['iterations'], c['iterations'] + 1), tag
        if c['length'] > 1000:                    # (the replay counts window draws: a fragment of up to ALIGNMENT_SIZE is aligned whole, without one)
            assert st['n_alignments'][0] == c['alignments'], tag

"""
GPU parity against the committed golden fixtures (produced by running the reference, see
tools/make_golden.py): the HIP path through the C-ABI must reproduce the reference's
sequence_fragment + get_qscores output bit for bit when both consume the same counter-based draws,
and the reference's build_fragment strings for the same planner decisions.
"""
import gzip
import io
import json
import os

import numpy as np
import pytest

import helpers as H
from badread_amd.engine import SimParams
from badread_amd.error_model import ErrorModel
from badread_amd.misc import load_fasta
from badread_amd.qscore_model import QScoreModel
from badread_amd.reference import PackedReference

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
NULL = io.StringIO()


def load(name):
    with gzip.open(os.path.join(GOLDEN, name), 'rt') as f:
        return json.load(f)


def test_sequence_fragment_golden_through_the_c_abi():
    g = load('sequence_fragment.json.gz')
    hip = H.hip_engine()
    by_model = {}
    for c in g['cases']:
        by_model.setdefault((c['em'], c['qm']), []).append(c)
    for (em, qm), cases in by_model.items():
        hip.set_error_model(ErrorModel(em, NULL).tables())
        hip.set_qscore_model(QScoreModel(qm, NULL).tables())
        for c in cases:                       # one call per case: seed and read index differ per case
            codes = np.array(['ACGTN'.index(ch) for ch in c['fragment']], dtype=np.uint8)
            res, st = hip.sequence_fragments(c['seed'], c['read'], [codes], [c['target']])
            tag = (em, qm, len(c['fragment']), c['target'])
            assert ''.join('ACGTN'[x] for x in res[0][0]) == c['seq'], tag
            assert res[0][1].tobytes().decode() == c['qual'], tag
            assert st['n_match'][0] / st['n_cols'][0] == c['identity'], tag
            assert abs(1.0 - st['qerr_sum'][0] / st['padded_len'][0] - c['identity_by_qscores']) < 1e-12, tag


def test_sequence_fragment_digest_cases_through_the_c_abi():
    """The 524 digest replays of the reference (tests/golden/sequence_fragment_bound.json.gz: incl. 24 fragments of 50 kb at
    80-90 % identity) through the HIP path: sequence, qualities, identity and loop count of every one."""
    g = load('sequence_fragment_bound.json.gz')
    hip = H.hip_engine()
    current = [None]

    def engine_of(em, qm):
        if current[0] != (em, qm):
            hip.set_error_model(ErrorModel(em, NULL).tables())
            hip.set_qscore_model(QScoreModel(qm, NULL).tables())
            current[0] = (em, qm)
        return hip
    H.check_digest_cases(engine_of, sorted(g['cases'], key=lambda c: (c['em'], c['qm'])))


def test_build_fragment_golden_through_the_c_abi():
    g = load('build_fragment.json.gz')
    pref = PackedReference.from_seqs(*load_fasta(os.path.join(GOLDEN, 'small_ref.fasta')))
    for cfg in g['configs']:
        params = SimParams(**dict(cfg['params'], identity_mode=0, id_max=1.0))     # identity 1.0: read == fragment
        hip = H.configure(H.hip_engine(), pref, 'random', 'ideal', params)
        n = len(cfg['reads'])
        out, st = hip.simulate_batch(cfg['seed'], 0, n)
        lines = bytes(out).decode().split('\n')
        recs = [lines[i:i + 4] for i in range(0, len(lines) - 1, 4)]
        ri = 0
        for rd, s in zip(cfg['reads'], st):
            assert s['frag_len'] == len(rd['fragment'])
            if s['rec_len'] == 0:
                assert len(rd['fragment']) == 0
                continue
            header, seq = recs[ri][0], recs[ri][1]
            ri += 1
            assert seq == rd['fragment'], rd['read']
            assert header.split(' ', 1)[1].rsplit(' length=', 1)[0] == rd['info'], rd['read']

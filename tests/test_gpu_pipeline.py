"""
GPU parity: the full per-read path (brx_simulate_batch, brx_sequence_fragments) against the CPU
oracle on the same seeds -- FASTQ bytes and every per-read statistic must be identical.
"""
import os

import numpy as np
import pytest

import helpers as H
from badread_amd.engine import SimParams

pytestmark = pytest.mark.gpu

STAT_FIELDS = ('status', 'frag_len', 'seq_len', 'n_cols', 'n_match', 'padded_len', 'loop_count',
               'change_count', 'n_alignments', 'rec_len', 'rec_off', 'target_identity', 'qerr_sum')


def _compare(em, qm, params, seed, first, n, with_n=True):
    pref, _ = H.small_reference(with_n=with_n)
    hip = H.configure(H.hip_engine(), pref, em, qm, params)
    orc = H.configure(H.oracle_engine(), pref, em, qm, params)
    out_h, st_h = hip.simulate_batch(seed, first, n)
    out_o, st_o = orc.simulate_batch(seed, first, n)
    for f in STAT_FIELDS:
        bad = np.flatnonzero(st_h[f] != st_o[f])
        assert len(bad) == 0, f'{f} differs for reads {bad[:8]}: hip {st_h[f][bad[:4]]} oracle {st_o[f][bad[:4]]}'
    fd = H.first_diff(out_h, out_o)
    assert fd < 0, f'FASTQ bytes differ at offset {fd}'
    return out_h, st_h


def test_random_ideal_small_reads():
    p = SimParams(frag_mean=1500, frag_stdev=1200, identity_mode=1, id_a=20.0, id_b=2.0, id_max=0.98)
    _compare('random', 'ideal', p, seed=42, first=0, n=200)


def test_random_random_models_constant_identity():
    p = SimParams(frag_mean=800, frag_stdev=0, identity_mode=0, id_max=0.9, glitch_rate=500, glitch_size=10, glitch_skip=10,
                  chimera_rate=0.2, junk_rate=0.1, random_rate=0.1)
    _compare('random', 'random', p, seed=7, first=1000, n=200)


def test_nanopore2023_default_parameters():
    p = SimParams(frag_mean=6000, frag_stdev=5000)
    _compare('nanopore2023', 'nanopore2023', p, seed=42, first=0, n=150)


def test_pacbio2021_qscore_identity():
    p = SimParams(frag_mean=5000, frag_stdev=4000, identity_mode=2, id_a=30.0, id_b=3.0)
    _compare('pacbio2021', 'pacbio2021', p, seed=3, first=50, n=100)


def test_low_identity_and_no_adapters():
    p = SimParams(frag_mean=2000, frag_stdev=1500, identity_mode=1, id_a=8.0, id_b=3.0, id_max=0.9,
                  start_adapter='', end_adapter='', glitch_rate=0)
    _compare('nanopore2018', 'nanopore2018', p, seed=11, first=0, n=80)


def test_batch_split_invariance():
    pref, _ = H.small_reference()
    p = SimParams(frag_mean=1500, frag_stdev=1200)
    hip = H.configure(H.hip_engine(), pref, 'nanopore2023', 'nanopore2023', p)
    whole, st = hip.simulate_batch(5, 0, 96)
    whole = whole.copy()
    a, _ = hip.simulate_batch(5, 0, 40)
    a = a.copy()
    b, _ = hip.simulate_batch(5, 40, 56)
    assert bytes(whole) == bytes(a) + bytes(b)


def test_sequence_fragments_matches_oracle():
    rng = np.random.default_rng(3)
    pref, _ = H.small_reference()
    hip = H.configure(H.hip_engine(), pref, 'nanopore2023', 'nanopore2023')
    orc = H.configure(H.oracle_engine(), pref, 'nanopore2023', 'nanopore2023')
    frags = [rng.integers(0, 4, int(L)).astype(np.uint8) for L in (60, 1, 999, 1000, 1001, 3000, 12000)]
    targets = [1.0, 0.9, 0.95, 0.8, 0.9, 0.97, 0.93]
    res_h, st_h = hip.sequence_fragments(9, 100, frags, targets)
    res_o, st_o = orc.sequence_fragments(9, 100, frags, targets)
    for i, ((sh, qh), (so, qo)) in enumerate(zip(res_h, res_o)):
        assert H.first_diff(sh, so) < 0, f'fragment {i}: sequence differs'
        assert H.first_diff(qh, qo) < 0, f'fragment {i}: qualities differ'
    for f in ('seq_len', 'n_cols', 'n_match', 'loop_count', 'change_count', 'n_alignments', 'qerr_sum'):
        assert (st_h[f] == st_o[f]).all(), f
    # identity 1.0 returns the fragment untouched (test_simulate.py:45-51)
    assert H.first_diff(res_h[0][0], frags[0]) < 0


@pytest.mark.parametrize('env', [{'BRX_TB_WINDOW': '0'}, {'BRX_TB_WINDOW': '-1'}, {'BRX_TB_WINDOW': '1'},
                                 {'BRX_TAIL_READS': '1000000'},                                      # every read runs to completion in place (k_mutate_seg)
                                 {'BRX_HEAD_READS': '0', 'BRX_TAIL_READS': '0'},                     # bulk passes only: k_mut_apply / k_mut_post / k_pass_lists / k_win_lane / k_win_wave
                                 {'BRX_HEAD_READS': '64', 'BRX_TAIL_READS': '32'},                   # two chains + an in-place tail that takes reads over from the passes
                                 {'BRX_HEAD_READS': '64', 'BRX_TAIL_READS': '32', 'BRX_TB_WINDOW': '-1', 'BRX_WIDE_STREAM': '0', 'BRX_FIN_SPREAD': '0'},
                                 {'BRX_HEAD_READS': '0', 'BRX_TAIL_READS': '16', 'BRX_FIN_HEAD_READS': '64', 'BRX_WAVES_PER_CU': '1'},      # 256 slab-owning waves per band class
                                 {'BRX_FIN_LANES': '0', 'BRX_QUAD_MIN_READS': '0'},                  # narrow bands four per wave (k_fin_quad) instead of one per lane
                                 {'BRX_FIN_LANES': '0', 'BRX_FIN_QUAD': '0'}])                       # every final alignment on a whole wave
def test_alternative_kernel_routes_give_the_same_bytes(env, monkeypatch):
    """The optional routes (full / 8-row / narrow traceback window of the final alignment -- the 8-row window
    makes most reads miss and repeat with the full store --, in-place mutate alignments, lane- or wave-per-window
    for every pass) are read from the environment when a context is created; all must reproduce the oracle."""
    from badread_amd.engine import HipEngine
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    pref, _ = H.small_reference()
    p = SimParams(frag_mean=5000, frag_stdev=4500)
    eng = H.configure(HipEngine(0, scratch_bytes=2 << 30), pref, 'nanopore2023', 'nanopore2023', p)
    orc = H.configure(H.oracle_engine(), pref, 'nanopore2023', 'nanopore2023', p)
    out_h, st_h = eng.simulate_batch(21, 0, 300)
    out_o, st_o = orc.simulate_batch(21, 0, 300)
    for f in STAT_FIELDS:
        assert (st_h[f] == st_o[f]).all(), f
    assert H.first_diff(out_h, out_o) < 0
    misses = eng.window_misses()
    if env.get('BRX_TB_WINDOW') == '0':
        assert misses == 0
    if env.get('BRX_TB_WINDOW') == '-1':
        assert misses > 50, misses                     # the retry pass really ran
    eng.close()

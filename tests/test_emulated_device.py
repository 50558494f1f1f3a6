"""
The product's HIP sources run on the CPU (tests/emu_engine.py, tests/native/emu/hip/hip_runtime.h: lanes as fibers,
cross-lane operations as rendezvous) and held to the same parity bar as on the GPU: bit-identical to the oracle
through the C-ABI.  This is the `-m "not gpu"` counterpart of tests/test_gpu_align.py / test_gpu_pipeline.py /
test_gpu_golden.py -- it cannot say anything about speed, occupancy or memory-ordering on real hardware, but a logic
error in a kernel (band geometry, LDS ring indexing, traceback addressing, the mutate pass state machine, the
windowed traceback store and its retry phase) fails here without a GPU.
"""
import gzip
import json
import os

import numpy as np
import pytest

import helpers as H
import pyoracle
from badread_amd.engine import SimParams

HERE = os.path.dirname(os.path.abspath(__file__))
STAT_FIELDS = ('status', 'frag_len', 'seq_len', 'n_cols', 'n_match', 'padded_len', 'loop_count', 'change_count',
               'n_alignments', 'rec_len', 'rec_off', 'target_identity', 'qerr_sum')


def emu_engine(monkeypatch=None, scratch=1 << 29, **env):
    import emu_engine as EE
    if monkeypatch is not None:
        for k, v in env.items():
            monkeypatch.setenv(k, str(v))
    return EE.EmuEngine(scratch)


def check_pairs(eng, queries, targets, k_hint=None):
    ops, dist, ncols, nmatch = eng.align_batch(queries, targets, k_hint=k_hint)
    for i, (q, t) in enumerate(zip(queries, targets)):
        d, o = pyoracle.align(q, t)
        assert dist[i] == d, (i, len(q), len(t), int(dist[i]), d)
        assert np.array_equal(ops[i], o), (i, len(q), len(t))
        assert ncols[i] == len(o) and nmatch[i] == int((o == 0).sum())


def test_aligner_small_pairs_and_edge_cases():
    eng = emu_engine()
    rng = np.random.default_rng(11)
    qs, ts = [b'', b'ACGT', b'', b'A', b'CG', b'GATTACA', b'A' * 128, b'A' * 40, b'ACGT' * 8], \
             [b'', b'', b'TTGA', b'A', b'CGT', b'GATACA', b'C' * 28, b'C' * 8, b'ACGT' * 7]
    for _ in range(60):
        n = int(rng.choice([1, 2, 5, 31, 32, 33, 64, 65, 100, 257, 700]))
        q = H.random_dna(rng, n)
        t = H.mutate_seq(rng, q, float(rng.choice([0.0, 0.03, 0.15, 0.4]))) or 'A'
        qs.append(q.encode()); ts.append(t.encode())
    check_pairs(eng, qs, ts)


def test_aligner_band_classes_iupac_and_hints():
    eng = emu_engine()
    rng = np.random.default_rng(12)
    qs, ts, hints = [], [], []
    # one word per lane with four columns per trip (long ring refills), two and four words per lane (wide bands)
    for n, rate in ((3000, 0.05), (2600, 0.30), (5200, 0.33), (1000, 0.9)):
        q = H.random_dna(rng, n)
        t = H.mutate_seq(rng, q, rate)
        qs.append(q.encode()); ts.append(t.encode()); hints.append(-1)
    # very different lengths (the band is all offset), symbols outside ACGT (rare equality path)
    qs += [H.random_dna(rng, 1000).encode(), H.random_dna(rng, 300, 'ACGTNRYK').encode()]
    ts += [H.random_dna(rng, 2794).encode(), H.random_dna(rng, 320, 'ACGTNRYK').encode()]
    hints += [-1, -1]
    check_pairs(eng, qs, ts)
    # a proven bound given by the caller: one round, same path
    d = [pyoracle.align(q, t)[0] for q, t in zip(qs[:2], ts[:2])]
    check_pairs(eng, qs[:2], ts[:2], k_hint=[d[0] + 7, d[1]])


ROUTES = [
    {},                                                        # defaults: few reads, all head -- every read run to completion in place (k_mutate_seg), windowed store
    {'BRX_TB_WINDOW': -1},                                     # 8-row traceback window: most reads repeat (phase 1)
    {'BRX_TB_WINDOW': 0, 'BRX_WIDE_STREAM': 0},                # full store, no third stream for the widest class
    {'BRX_TAIL_READS': 0, 'BRX_HEAD_READS': 0},                # every read in the bulk set: k_mut_fill + ONE launch of k_mut_lanes (a wave keeps its 64 reads to the end)
    {'BRX_TAIL_READS': 6, 'BRX_HEAD_READS': 9},                # two chains: 9 head reads run to completion in place, 31 in k_mut_lanes
    {'BRX_TAIL_READS': 0, 'BRX_HEAD_READS': 0, 'BRX_LANES_CYCLES': 3},   # ... which hands every read over to the in-place kernel after three alignment cycles (parked or hungry)
    {'BRX_TAIL_READS': 0, 'BRX_HEAD_READS': 0, 'BRX_LANES_CYCLES': 0},   # ... or keeps them to the end, whatever their cycles
    {'BRX_TAIL_READS': 6, 'BRX_HEAD_READS': 9, 'BRX_TB_WINDOW': -1},   # ... both sets with a retry phase
    {'BRX_TAIL_READS': 0, 'BRX_HEAD_READS': 0, 'BRX_MUTATE_PASSES': 1},          # the bulk set through host-driven passes: k_mut_apply / k_mut_post / k_pass_lists / k_win_lane / k_win_wave
    {'BRX_TAIL_READS': 6, 'BRX_FIN_HEAD_READS': 9, 'BRX_HEAD_READS': 0, 'BRX_MUTATE_PASSES': 1},   # passes, a 6-read in-place tail that takes reads over in any state, final stage in two sets
    {'BRX_TAIL_READS': 6, 'BRX_HEAD_READS': 9, 'BRX_MUTATE_PASSES': 1},          # head chain + passes + tail
    {'BRX_TAIL_READS': 0, 'BRX_HEAD_READS': 0, 'BRX_WAVES_PER_CU': 2},       # four slab-owning waves per band class reuse their slabs
    {'BRX_TAIL_READS': 6, 'BRX_HEAD_READS': 9, 'BRX_FIN_SPREAD': 0, 'BRX_FIN_LANES': 0},   # the bulk set's band classes one after the other on its own stream, no read aligned by lane
    {'BRX_FIN_LANES': 0, 'BRX_QUAD_MIN_READS': 0},             # narrow bands too go four per wave, a row of 16 lanes each (k_fin_quad), instead of one read per lane
    {'BRX_FIN_LANES': 0, 'BRX_TB_WINDOW': -1, 'BRX_QUAD_MIN_READS': 0},   # ... and the misses of an 8-row traceback window are repeated by k_fin_align with the full store
    {'BRX_FIN_QUAD': 0, 'BRX_FIN_LANES': 0},                   # every final alignment on a whole wave
    {'BRX_LANES_MIN_READS': 2048, 'BRX_TB_WINDOW': -1},        # the shipped rule: a set with few by-lane reads aligns them on whole waves (flag set, route not taken), with a retry phase
]


@pytest.mark.parametrize('env', ROUTES)
def test_pipeline_routes_equal_the_oracle(env, monkeypatch):
    pref, _ = H.small_reference()
    p = SimParams(frag_mean=1100, frag_stdev=900)
    eng = H.configure(emu_engine(monkeypatch, **env), pref, 'nanopore2023', 'nanopore2023', p)
    orc = H.configure(H.oracle_engine(), pref, 'nanopore2023', 'nanopore2023', p)
    n = 30
    out_h, st_h = eng.simulate_batch(42, 0, n)
    out_o, st_o = orc.simulate_batch(42, 0, n)
    for f in STAT_FIELDS:
        assert (st_h[f] == st_o[f]).all(), (env, f)
    assert H.first_diff(out_h, out_o) < 0, env
    if env.get('BRX_TB_WINDOW') == -1:
        assert eng.window_misses() >= 3                     # the retry phase ran (short reads: few windows are narrower than the band)
    if env.get('BRX_TAIL_READS') == 0 and env.get('BRX_MUTATE_PASSES'):
        assert eng.mutate_passes() > 3


MUTATE_ROUTES = {'default': {}, 'lanes': {'BRX_TAIL_READS': 0, 'BRX_HEAD_READS': 0}, 'head_lanes': {'BRX_TAIL_READS': 4, 'BRX_HEAD_READS': 3},
                 'passes_tail': {'BRX_TAIL_READS': 4, 'BRX_HEAD_READS': 3, 'BRX_MUTATE_PASSES': 1}}


@pytest.mark.parametrize('route', sorted(MUTATE_ROUTES))
def test_pipeline_other_models_and_fragment_kinds(route, monkeypatch):
    """random / ideal models (k = 1), low identity, chimeras, junk and random reads, glitches, N runs and hairpins;
    through the in-place chain (few reads: all head), k_mut_lanes for every read, a head chain beside it, and the host-driven passes with a tail."""
    for k, v in MUTATE_ROUTES[route].items():
        monkeypatch.setenv(k, str(v))
    pref, _ = H.small_reference(with_n=True)
    p = SimParams(frag_mean=700, frag_stdev=0, identity_mode=0, id_max=0.88, glitch_rate=400, glitch_size=10, glitch_skip=10,
                  chimera_rate=0.2, junk_rate=0.1, random_rate=0.1)
    for em, qm, seed in (('random', 'random', 7), ('pacbio2021', 'pacbio2021', 9)):
        eng = H.configure(emu_engine(), pref, em, qm, p)
        orc = H.configure(H.oracle_engine(), pref, em, qm, p)
        out_h, st_h = eng.simulate_batch(seed, 100, 30)
        out_o, st_o = orc.simulate_batch(seed, 100, 30)
        for f in STAT_FIELDS:
            assert (st_h[f] == st_o[f]).all(), (em, f)
        assert H.first_diff(out_h, out_o) < 0, em


@pytest.mark.parametrize('tail', [0, 5])
def test_bulk_passes_with_nearly_empty_survivor_rings(tail, monkeypatch):
    """brx_passes.h: k_mut_post proposes ahead into a ring of survivors per read and k_mut_apply consumes it; a read whose ring runs
    empty before its 25th change goes HUNGRY -- it takes part in the next pass without an alignment.  A build that stocks two
    survivors per ring makes that the common case (and, with a tail, hands hungry reads over to k_mutate_seg): same bytes and
    statistics as the oracle, and more passes than the shipped stock needs."""
    import emu_engine as EE
    monkeypatch.setenv('BRX_HEAD_READS', '0')
    monkeypatch.setenv('BRX_TAIL_READS', str(tail))
    monkeypatch.setenv('BRX_MUTATE_PASSES', '1')
    pref, _ = H.small_reference(with_n=True)
    p = SimParams(frag_mean=1400, frag_stdev=900, identity_mode=0, id_max=0.90)
    orc = H.configure(H.oracle_engine(), pref, 'nanopore2023', 'nanopore2023', p)
    out_o, st_o = orc.simulate_batch(77, 10, 24)
    passes = []
    for defines in ((), ('-DBRX_SV_STOCK=2u', '-DBRX_POST_U=1')):
        eng = H.configure(EE.EmuEngine(1 << 29, defines=defines), pref, 'nanopore2023', 'nanopore2023', p)
        out_h, st_h = eng.simulate_batch(77, 10, 24)
        for f in STAT_FIELDS:
            assert (st_h[f] == st_o[f]).all(), (defines, f)
        assert H.first_diff(out_h, out_o) < 0, defines
        passes.append(eng.mutate_passes())
    assert passes[1] > passes[0] > 3, passes


@pytest.mark.parametrize('head,cycles', [(0, 0), (3, 0), (0, 2)])
def test_one_wave_keeps_its_reads_to_the_end_with_rings_that_run_empty(head, cycles, monkeypatch):
    """k_mut_lanes (brx_passes.h): a wave keeps 64 reads from the first iteration to the last -- apply, park and align by lane, no
    launch in between -- with every survivor proposed ahead by k_mut_fill.  A build whose rings hold 72 entries whatever the read
    (n >> 12 + 72 instead of n >> 3 + 128) makes them run empty again and again, so the in-kernel refill (the wave proposes ahead
    for one read) and the wrap of the ring carry the loop; reads with N runs take the whole-wave aligner for their windows.  Same
    bytes and statistics as the oracle, with and without a head chain beside it."""
    import emu_engine as EE
    monkeypatch.setenv('BRX_HEAD_READS', str(head))
    monkeypatch.setenv('BRX_TAIL_READS', '0')
    monkeypatch.setenv('BRX_LANES_CYCLES', str(cycles))              # 0: to the end; 2: the in-place kernel takes the reads over after two cycles
    pref, _ = H.small_reference(with_n=True)
    p = SimParams(frag_mean=1600, frag_stdev=1100, identity_mode=0, id_max=0.88)
    orc = H.configure(H.oracle_engine(), pref, 'nanopore2023', 'nanopore2023', p)
    out_o, st_o = orc.simulate_batch(91, 40, 70)                  # two waves: 64 reads + 6
    # (the second build also runs both groups of 64 reads on ONE wave, one after the other: a batch beyond 1024 x 64 reads does that)
    for defines in ((), ('-DBRX_RING_SHIFT=12', '-DBRX_RING_MIN=72u', '-DBRX_POST_U=1', '-DBRX_LANES_MAX_WAVES=1u')):
        eng = H.configure(EE.EmuEngine(1 << 29, defines=defines), pref, 'nanopore2023', 'nanopore2023', p)
        out_h, st_h = eng.simulate_batch(91, 40, 70)
        for f in STAT_FIELDS:
            assert (st_h[f] == st_o[f]).all(), (defines, f)
        assert H.first_diff(out_h, out_o) < 0, defines
        assert eng.mutate_passes() <= 3                            # one launch for the bulk set (+ the head chain's, + the in-place takeover)


def test_sequence_fragment_golden_vectors_from_the_running_reference():
    """tests/golden/sequence_fragment.json.gz (the reference's own sequence_fragment replayed with our draws) through the
    emulated device: every case up to 4 kb (N-containing fragments, every model pair)."""
    with gzip.open(os.path.join(HERE, 'golden', 'sequence_fragment.json.gz'), 'rt') as f:
        g = json.load(f)
    eng = emu_engine()
    pref, _ = H.small_reference()
    H.configure(eng, pref)
    models, n = None, 0
    for c in sorted(g['cases'], key=lambda c: (c['em'], c['qm'])):
        if len(c['fragment']) > 4000:
            continue
        if models != (c['em'], c['qm']):
            models = (c['em'], c['qm'])
            eng.set_error_model(H.error_tables(c['em']))
            eng.set_qscore_model(H.qscore_tables(c['qm']))
        codes = np.array(['ACGTN'.index(ch) for ch in c['fragment']], dtype=np.uint8)
        res, st = eng.sequence_fragments(c['seed'], c['read'], [codes], [c['target']])
        tag = (c['em'], c['qm'], len(c['fragment']), c['target'])
        assert ''.join('ACGTN'[x] for x in res[0][0]) == c['seq'], tag
        assert res[0][1].tobytes().decode() == c['qual'], tag
        assert st['n_match'][0] / st['n_cols'][0] == c['identity'], tag
        n += 1
    assert n >= 15


def test_driver_on_the_emulated_device_equals_the_oracle_driver(monkeypatch):
    """`badread simulate` end to end (reference loading, depth adjustment, stop rule, batches in flight on engine clones)
    with the emulated device as the engine: the FASTQ stream of the oracle-driven run, byte for byte."""
    import io
    from badread_amd import simulate as S
    from test_host_simulate import Args
    monkeypatch.setattr(S, 'DEFAULT_MAX_BATCH', 12)
    args = dict(quantity='6x', mean_frag_length=350.0, frag_length_stdev=250.0, error_model='nanopore2023',
                qscore_model='nanopore2023', mean_identity=92.0, max_identity=98.0, identity_stdev=3.0, seed=3)
    a, b = io.BytesIO(), io.BytesIO()
    S.simulate(Args(gpu_streams=3, **args), output=io.StringIO(), engine=emu_engine(), stdout=a, shard=S.Shard())
    S.simulate(Args(gpu_streams=1, **args), output=io.StringIO(), engine=H.oracle_engine(), stdout=b, shard=S.Shard())
    assert a.getvalue() == b.getvalue() and a.getvalue().count(b'\n') >= 4 * 20


@pytest.mark.parametrize('route', ['default', 'giants'])      # the overflowing reads leave the passes at their first window: the pass route does not matter (80 s each)
def test_window_overflow_goes_through_the_whole_read_kernel(tmp_path, route, monkeypatch):
    """... 'giants': the same two reads on a build whose widest band class calls every store above 512 bytes a giant -- the class of
    its own (own queue, own few slabs) that keeps a batch's few GB-sized stores from taking the slabs of the thousands beside them."""
    for k, v in MUTATE_ROUTES['default' if route == 'giants' else route].items():
        monkeypatch.setenv(k, str(v))
    _window_overflow(tmp_path, defines=('-DBRX_GIANT_UNITS=64ull',) if route == 'giants' else ())


def _window_overflow(tmp_path, defines=()):
    """A synthetic error model whose alternatives insert 60 bases: the joined 1000-base windows outgrow their pass slots
    (BRX_WIN_TMAX), so those reads are handed to the whole-read kernel k_mutate with inline alignments -- a route no
    packaged model reaches.  Mutated reads of 17x the fragment length also push the final alignment into the widest
    band classes (8 and 16 words per lane)."""
    import io
    import itertools
    from badread_amd.error_model import ErrorModel
    rng = np.random.default_rng(4)
    lines = []
    for kmer in map(''.join, itertools.product('ACGT', repeat=3)):
        ins = ''.join(rng.choice(list('ACGT'), 60))
        lines.append(f'{kmer},0.2;{kmer[0]}{kmer[1]}{ins}{kmer[2]},0.6;{kmer[0]}{kmer[2]},0.2;\n')
    path = tmp_path / 'big_insertions_model'
    path.write_text(''.join(lines))
    tables = ErrorModel(str(path), io.StringIO(), aligner=pyoracle.oracle_align_batch, use_cache=False).tables()
    pref, _ = H.small_reference()
    import emu_engine as EE
    eng, orc = H.configure(EE.EmuEngine(1 << 29, defines=defines) if defines else emu_engine(), pref), H.configure(H.oracle_engine(), pref)
    for e in (eng, orc):
        e.set_error_model(tables)
        e.set_qscore_model(H.qscore_tables('ideal'))
    frags = [rng.integers(0, 4, n).astype(np.uint8) for n in (560, 300)]
    targets = [0.05, 0.3]
    rh, sh = eng.sequence_fragments(9, 0, frags, targets)
    ro, so = orc.sequence_fragments(9, 0, frags, targets)
    for f in STAT_FIELDS:
        assert (sh[f] == so[f]).all(), f
    for a, b in zip(rh, ro):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert int(sh["padded_len"][0]) > 8 * 560                      # the windows really overflowed
    return eng


@pytest.mark.parametrize('lanes', [1, 0])
def test_narrow_bands_one_read_per_lane(lanes, monkeypatch):
    """BASELINE.json configs[4] in small: pacbio2021 models, identities around Q30 -- a read differs from its fragment in a few
    dozen places, the band of its final alignment is two or three blocks, and k_fin_lanes aligns 64 such reads per wave, one per
    lane (csrc/brx_finlanes.h), writing the ops k_fin_qscore reads.  Same bytes and statistics as the oracle, and as the wave-systolic
    aligner (BRX_FIN_LANES=0) that k_fin_lanes takes these reads from."""
    pref, _ = H.small_reference()
    p = SimParams(frag_mean=1500, frag_stdev=900, identity_mode=2, id_a=30.0, id_b=3.0, id_max=1.0)
    orc = H.configure(H.oracle_engine(), pref, 'pacbio2021', 'pacbio2021', p)
    eng = H.configure(emu_engine(monkeypatch, BRX_FIN_LANES=lanes), pref, 'pacbio2021', 'pacbio2021', p)
    eng.set_kernel_timing(True)                                     # per-class launch and base counts (brx_last_kernel_stats)
    out_o, st_o = orc.simulate_batch(8, 0, 80)
    out_h, st_h = eng.simulate_batch(8, 0, 80)
    for f in STAT_FIELDS:
        assert (st_h[f] == st_o[f]).all(), f
    assert H.first_diff(out_h, out_o) < 0
    by_lane = eng.kernel_stats()['k_fin_lanes'][2]                  # bases of the reads the class handled
    assert (by_lane > 0.8 * float(st_h['frag_len'].sum())) if lanes else by_lane == 0


@pytest.mark.parametrize('case', ['one_word', 'long'])
def test_four_final_alignments_per_wave(case, monkeypatch):
    """k_fin_quad (csrc/brx_quad.h): reads whose band is at most 13 superblocks of one word (416 diagonals) are aligned four per
    wave, one per row of 16 lanes, from LDS rings of target bytes and query planes that are
    refilled every 32 loop trips.  Same bytes and statistics as the oracle; the class really takes the reads; the reads are long
    enough for several refills per ring (a ring holds 1024 target bytes / 64 query words)."""
    pref, _ = H.small_reference(with_n=False)
    p, n, kern = {'one_word': (SimParams(frag_mean=2500, frag_stdev=1500, identity_mode=1, id_a=20.0, id_b=2.0, id_max=0.98), 16, 'k_fin_quad<1>'),
                  'long': (SimParams(frag_mean=9000, frag_stdev=500, identity_mode=0, id_max=0.97), 5, 'k_fin_quad<1>')}[case]
    orc = H.configure(H.oracle_engine(), pref, 'nanopore2023', 'nanopore2023', p)
    eng = H.configure(emu_engine(monkeypatch, BRX_FIN_LANES=0 if case == 'one_word' else 1, BRX_QUAD_MIN_READS=0), pref, 'nanopore2023', 'nanopore2023', p)
    eng.set_kernel_timing(True)
    out_o, st_o = orc.simulate_batch(5, 0, n)
    out_h, st_h = eng.simulate_batch(5, 0, n)
    for f in STAT_FIELDS:
        assert (st_h[f] == st_o[f]).all(), f
    assert H.first_diff(out_h, out_o) < 0
    assert eng.kernel_stats()[kern][2] > 0.5 * float(st_h['frag_len'].sum())


@pytest.mark.parametrize('quad,small', [(0, 10 << 20), (1, 0)])
def test_final_stage_with_fewer_slabs_than_reads(quad, small, monkeypatch):
    """The traceback stores of the final stage are slabs owned by the waves of the align kernels, sized by queue position
    (brx_hip.hip, launch_final_phase).  An arena that holds the largest store but not one slab per read (per group of four reads
    with k_fin_quad): the set runs with fewer waves, every wave reusing its slab for several reads; same bytes.  (10 MB: the mutate
    stage's top region -- 29 window slots of 128 KB; an all-head batch has no lane stores and no rings -- and the batch's strings fit,
    16 slabs of the final stage do not: 8 are used.  Round 5's layout kept the mutate buffers beside the slabs and needed 20 MB for that.)"""
    pref, _ = H.small_reference()
    p = SimParams(frag_mean=3200, frag_stdev=1300)
    orc = H.configure(H.oracle_engine(), pref, 'nanopore2023', 'nanopore2023', p)
    out_o, st_o = orc.simulate_batch(8, 0, 28)
    slabs = []
    for scratch in (1 << 29, small):
        # (with four reads per wave the slabs fit any arena that holds the rest of the batch: there the second engine is limited to
        #  one slab-owning wave per CU of the two-CU interpreted chip instead)
        eng = H.configure(emu_engine(monkeypatch, scratch=scratch or 1 << 29, BRX_TB_WINDOW=0, BRX_WIN_KB=128, BRX_FIN_LANES=0, BRX_FIN_QUAD=quad,
                                     BRX_QUAD_MIN_READS=0, BRX_WAVES_PER_CU=16 if scratch else 1), pref, 'nanopore2023', 'nanopore2023', p)
        out_h, st_h = eng.simulate_batch(8, 0, 28)
        slabs.append(eng.final_launches())
        for f in STAT_FIELDS:
            assert (st_h[f] == st_o[f]).all(), f
        assert H.first_diff(out_h, out_o) < 0
    assert slabs[1] < slabs[0] <= 28, slabs            # the small arena runs with fewer slab-owning waves


def test_long_segment_lists_continue_in_the_overflow_lists():
    """VERDICT r1 item 8 / ADVICE: the planner's private list of base segments (BRX_MAX_BASE_SEGS) is not a limit any more --
    longer lists continue in a global overflow list, like the reference's unbounded chimera loop (simulate.py:101-110).
    Built here with FOUR private segments and run at --chimeras 50 so that most chimeric reads overflow: same bytes as
    the oracle, no status flag."""
    import emu_engine as EE
    pref, _ = H.small_reference()
    p = SimParams(frag_mean=300, frag_stdev=200, chimera_rate=0.5, glitch_rate=300, glitch_size=5, glitch_skip=5)
    eng = H.configure(EE.EmuEngine(1 << 29, defines=('-DBRX_MAX_BASE_SEGS=4',)), pref, 'random', 'ideal', p)
    orc = H.configure(H.oracle_engine(), pref, 'random', 'ideal', p)
    out_h, st_h = eng.simulate_batch(5, 0, 48)
    out_o, st_o = orc.simulate_batch(5, 0, 48)
    assert (st_h['status'] & 2 == 0).all()                        # RS_TOO_MANY_SEGS
    for f in STAT_FIELDS:
        assert (st_h[f] == st_o[f]).all(), f
    assert H.first_diff(out_h, out_o) < 0
    assert out_o.tobytes().count(b'chimera') >= 30                 # the workload really chains fragments


def test_band_wider_than_the_register_classes():
    """A pair whose band needs more than 64 lanes x 32 words (57 344 rows: the limit of round 1, BRX_RS_BAND) goes through
    the memory-resident wide path with 64 words per lane: distance and the whole canonical path equal the oracle's."""
    rng = np.random.default_rng(12)
    eng = emu_engine()
    q = H.random_dna(rng, 60500).encode()
    t = H.random_dna(rng, 700).encode()
    d, ops = pyoracle.align(q, t)
    got_ops, dist, ncols, nmatch = eng.align_batch([q], [t], k_hint=[d])
    assert int(dist[0]) == d and int(ncols[0]) == len(ops) and int(nmatch[0]) == int((ops == 0).sum())
    assert np.array_equal(got_ops[0], ops)

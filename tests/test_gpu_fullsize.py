"""
BASELINE.json configs[1] at full batch size on the GPU (16384 reads of ~15 kb from the 5.5 Mb reference of bench.py),
checked through properties that do not need the oracle at that size, and against the oracle:

  * batch-split invariance: one 16384-read call == two 8192-read calls, byte for byte;
  * the windowed traceback store is invisible: the first 2048 reads with BRX_TB_WINDOW=0 (full store) and with the
    8-row window (most reads repeat in the second phase) give the same bytes as the default window;
  * every record is well formed: four lines, len(seq) == len(qual) == stats.seq_len, the header's length=,
    error-free_length= and read_identity= fields equal the statistics (identity = n_match / n_cols to 3 decimals),
    qualities inside the model's range, no status bits other than EMPTY;
  * alignment sanity per read: n_cols >= max(fragment, read) length, distance <= number of changes applied x 58;
  * ALL 16384 reads: the oracle (one process per host core) reproduces every record and every statistic.

configs[3] and configs[4] (the 3.09 Gb reference) run at the SHIPPED launch geometry -- the bench's and the CLI's device batch
of 65536 reads with the bench's scratch arena and the default environment -- and every read of that batch is compared with
the oracle.
"""
import io
import re

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

N = 16384
SHIPPED_BATCH = 65536          # bench.py / CLI device batch (--reads-per-step 393216 as --streams 6)
SEED = 42
HEADER = re.compile(rb'length=(\d+) error-free_length=(\d+) read_identity=([0-9.]+)%$')


def compare_with_oracle_slices(wlname, ref_dir, n, st, raw, tmp_path, fields, base=0):
    """Every read of [base, base + n): one oracle process per usable host core on disjoint slices; FASTQ bytes and statistics."""
    import os
    import subprocess
    import sys
    import bench
    cores = max(1, min(bench.usable_cores(), 32))
    per = -(-n // cores)
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1', HIP_VISIBLE_DEVICES='')
    procs = []
    for i in range(cores):
        first, count = i * per, max(0, min(per, n - i * per))
        if count == 0:
            continue
        path = str(tmp_path / f'slice{i}.npz')
        procs.append((first, count, path, subprocess.Popen([sys.executable, os.path.join(here, 'oracle_slice_worker.py'), wlname, ref_dir or '-',
                                                            str(SEED), str(base + first), str(count), path], env=env,
                                                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)))
    for first, count, path, pr in procs:
        _, err = pr.communicate(timeout=1500)
        assert pr.returncode == 0, err.decode()[-2000:]
        z = np.load(path)
        so = z['stats']
        lo = int(st['rec_off'][first])
        hi = int(st['rec_off'][first + count - 1] + st['rec_len'][first + count - 1])
        assert z['data'].tobytes() == raw[lo:hi], f'{wlname}: reads {base + first}..{base + first + count - 1} differ from the oracle'
        for f in fields:
            assert (so[f] == st[f][first:first + count]).all(), (wlname, f, first)


ALL_FIELDS = ('status', 'frag_len', 'seq_len', 'n_cols', 'n_match', 'padded_len', 'loop_count', 'change_count', 'n_alignments',
              'rec_len', 'target_identity', 'qerr_sum')


@pytest.fixture(autouse=True)
def shipped_final_stage_rules(monkeypatch):
    """tests/conftest.py runs the suite with BRX_LANES_MIN_READS=0 (small sets reach k_fin_lanes); full batches run the shipped default."""
    monkeypatch.delenv('BRX_LANES_MIN_READS', raising=False)


@pytest.fixture(scope='module')
def workload():
    import bench
    return bench, bench.build_workload(io.StringIO())


def test_configs1_full_batch_properties(workload, monkeypatch, tmp_path):
    from badread_amd.engine import HipEngine, RS_EMPTY
    bench, wl = workload
    eng = bench.configure(HipEngine(0, scratch_bytes=30 << 30), wl)
    out, st = eng.simulate_batch(SEED, 0, N)
    out, st = out.copy(), st.copy()
    assert eng.final_launches() >= 1

    # ---- batch-split invariance at full size
    a, sa = eng.simulate_batch(SEED, 0, N // 2)
    a, sa = a.copy(), sa.copy()
    b, sb = eng.simulate_batch(SEED, N // 2, N // 2)
    assert len(a) + len(b) == len(out)
    assert H.first_diff(out[:len(a)], a) < 0 and H.first_diff(out[len(a):], b) < 0
    for f in ('seq_len', 'n_cols', 'n_match', 'change_count', 'n_alignments', 'qerr_sum'):
        assert (np.concatenate([sa[f], sb[f]]) == st[f]).all(), f

    # ---- record structure and header fields
    assert (st['status'] & ~np.uint32(RS_EMPTY) == 0).all()
    live = np.flatnonzero(st['rec_len'] > 0)
    assert len(live) > 0.99 * N
    raw = out.tobytes()
    total = 0
    for r in live[:: max(1, len(live) // 4096)].tolist() + live[-3:].tolist():       # every 4th record + the last ones
        rec = raw[int(st['rec_off'][r]): int(st['rec_off'][r]) + int(st['rec_len'][r])]
        lines = rec.split(b'\n')
        assert len(lines) == 5 and lines[4] == b'' and lines[0].startswith(b'@') and lines[2] == b'+'
        L = int(st['seq_len'][r])
        assert len(lines[1]) == L == len(lines[3]) and L > 0
        m = HEADER.search(lines[0])
        assert m and int(m.group(1)) == L and int(m.group(2)) == int(st['frag_len'][r])
        ident = 100.0 * int(st['n_match'][r]) / int(st['n_cols'][r])
        assert abs(float(m.group(3)) - ident) <= 0.00051
        q = np.frombuffer(lines[3], dtype=np.uint8)
        assert q.min() >= 33 and q.max() <= 33 + 93
        assert set(lines[1]) <= set(b'ACGT')                      # the reference is pure ACGT here
        total += L
    assert int(st['rec_off'][live[-1]] + st['rec_len'][live[-1]]) == len(out)
    # ---- alignment sanity
    n = st['frag_len'].astype(np.int64)[live]
    assert (st['n_cols'][live] >= np.maximum(n, st['padded_len'][live])).all()
    dist = st['n_cols'].astype(np.int64)[live] - st['n_match'][live]
    assert (dist <= st['change_count'].astype(np.int64)[live] * 58).all()
    # identities land where the identity law puts them (mean 95, max 99)
    ident = st['n_match'][live] / st['n_cols'][live]
    assert 0.93 < float(np.mean(ident)) < 0.97 and float(ident.max()) <= 1.0

    # ---- every read against the oracle
    compare_with_oracle_slices('kpn', None, N, st, raw, tmp_path, ALL_FIELDS)
    eng.close()

    # ---- the traceback window does not show in the output
    head = raw[: int(st['rec_off'][2047] + st['rec_len'][2047])]
    for window, want_misses in (('0', False), ('-1', True)):
        monkeypatch.setenv('BRX_TB_WINDOW', window)
        e2 = bench.configure(HipEngine(0, scratch_bytes=30 << 30), wl)
        o2, s2 = e2.simulate_batch(SEED, 0, 2048)
        assert bytes(o2) == head, f'BRX_TB_WINDOW={window}'
        assert (e2.window_misses() > 1000) == want_misses
        e2.close()


@pytest.mark.parametrize('wlname', ['human', 'hifi'])
def test_configs3_and_4_every_read_of_a_full_batch_equals_the_oracle(wlname, tmp_path):
    """BASELINE.json configs[3] (GRCh38-like 3.09 Gb, nanopore2023) and configs[4] (pacbio2021, --identity 30,3) at the
    SHIPPED launch geometry (65536 reads per device batch, the bench's scratch arena, default environment): ALL reads of the
    first device batch, HIP path vs the CPU oracle (one oracle process per usable host core on disjoint slices of the batch),
    FASTQ bytes and every per-read statistic.  The batch is known (oracle plan probes under seed 42) to hold reads that overlap N runs -- whose windows carry
    non-ACGT symbols and saturated edit bounds -- and reads clipped at the end of a linear contig
    (simulate.py:231-246); both are asserted from the GPU's own output."""
    import os
    import subprocess
    import sys
    import bench
    import synth_refs
    from badread_amd.engine import HipEngine, RS_EMPTY
    ref_dir = bench.default_ref_dir()
    wl = bench.build_workload(io.StringIO(), wlname, ref_dir)          # FASTA -> brx_fasta_pack -> sidecar (first use on this box)
    pref = wl[0]
    assert pref.n_bases == 3088269832 and len(pref.names) == 24
    assert len(pref.exceptions) == 25 + 3        # the end run of one contig and the start run of the next are one run in packed coordinates
    n = SHIPPED_BATCH
    scratch_gb = bench.SCRATCH_GB_DEFAULT
    eng = bench.configure(HipEngine(0, scratch_bytes=int(scratch_gb * (1 << 30))), wl)
    out, st = eng.simulate_batch(SEED, 0, n)
    out, st = out.copy(), st.copy()
    assert getattr(eng, 'retries', 0) == 0, 'the shipped arena must hold a shipped batch without a retry'
    route = eng.read_cycles(n)[:, 7]
    eng.close()
    # the routes of the final stage carry reads at full size: four per wave (bit 16), one per lane (bit 17) -- a change
    # of a threshold that sends everything back to k_fin_align must not pass silently
    live_reads = int((st['rec_len'] > 0).sum())
    by_quad, by_lane = int(((route >> 16) & 1).sum()), int(((route >> 17) & 1).sum())
    if wlname == 'human':
        assert by_quad >= 4096 and by_lane >= 1000, (by_quad, by_lane)
    else:
        assert by_lane >= 0.9 * live_reads, (by_lane, live_reads)
    assert (st['status'] & ~np.uint32(RS_EMPTY) == 0).all()
    raw = out.tobytes()
    compare_with_oracle_slices(wlname, ref_dir, n, st, raw, tmp_path, ALL_FIELDS)

    # the batch really exercises the non-ACGT path and the clipping of linear contigs
    lengths = dict(synth_refs.GRCH38_LENGTHS)
    with_n = clipped = 0
    for r in np.flatnonzero(st['rec_len'] > 0).tolist():
        rec = raw[int(st['rec_off'][r]): int(st['rec_off'][r]) + int(st['rec_len'][r])]
        head, seq = rec.split(b'\n', 2)[:2]
        with_n += b'N' in seq
        for m in re.finditer(rb'(chr\w+),[+-]strand,(\d+)-(\d+)', head):
            clipped += int(m.group(3)) == lengths[m.group(1).decode()]
    assert with_n >= 10 and clipped >= 1, (with_n, clipped)
    ident = st['n_match'][st['n_cols'] > 0] / st['n_cols'][st['n_cols'] > 0]
    if wlname == 'hifi':
        assert float(np.mean(ident)) > 0.995                      # qscore-distributed identities around Q30
    else:
        assert 0.93 < float(np.mean(ident)) < 0.97


def test_configs3_a_device_batch_deep_in_the_index_space_equals_the_oracle(tmp_path):
    """the driver's bench walks read indices up to ~9.8 M, the full-size parity above only [0, 65536).
    One more device batch of configs[3] at the shipped geometry, read indices 5 000 000 ..., every read against the oracle."""
    import bench
    from badread_amd.engine import HipEngine, RS_EMPTY
    ref_dir = bench.default_ref_dir()
    wl = bench.build_workload(io.StringIO(), 'human', ref_dir)
    base = 5_000_000
    eng = bench.configure(HipEngine(0, scratch_bytes=int(bench.SCRATCH_GB_DEFAULT * (1 << 30))), wl)
    out, st = eng.simulate_batch(SEED, base, SHIPPED_BATCH)
    out, st = out.copy(), st.copy()
    assert getattr(eng, 'retries', 0) == 0
    eng.close()
    assert (st['status'] & ~np.uint32(RS_EMPTY) == 0).all()
    compare_with_oracle_slices('human', ref_dir, SHIPPED_BATCH, st, out.tobytes(), tmp_path, ALL_FIELDS, base=base)


def _rare_routes():
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'rare_routes.json')
    return json.load(open(path)) if os.path.isfile(path) else {'pins': []}


@pytest.mark.parametrize('pin', _rare_routes()['pins'], ids=lambda p: f"{p['route']}@{p['read']}")
def test_pinned_reads_on_the_rare_routes_of_the_final_stage(pin, tmp_path):
    """reads of configs[3] found on the GPU (tools/find_rare_routes.py, scanned range in
    tests/golden/rare_routes.json) whose final alignment leaves the 2 sqrt(ub) + 24 traceback window -- the second phase with
    the full store -- or whose band is beyond 16 words per lane (the memory-resident wide path): a 64-read batch around
    each, DEFAULT environment, against the oracle; and the route is asserted to be taken."""
    import bench
    from badread_amd.engine import HipEngine
    ref_dir = bench.default_ref_dir()
    wlname, nb = pin.get('workload', 'human'), int(pin.get('batch', 64))
    wl = bench.build_workload(io.StringIO(), wlname, ref_dir)
    first = (int(pin['read']) // nb) * nb
    eng = bench.configure(HipEngine(0, scratch_bytes=8 << 30), wl)
    out, st = eng.simulate_batch(SEED, first, nb)
    out, st = out.copy(), st.copy()
    cyc = eng.read_cycles(nb)
    r = int(pin['read']) - first
    if pin['route'] == 'window_miss':
        assert eng.window_misses() >= 1 and cyc[r, 2] != 0
    else:
        assert (int(cyc[r, 7]) & 0xFFFF) > 16
        assert int(st['n_cols'][r]) - int(st['n_match'][r]) > 57344          # more band rows than 64 lanes x 16 words hold
    eng.close()
    compare_with_oracle_slices(wlname, ref_dir, nb, st, out.tobytes(), tmp_path, ALL_FIELDS, base=first)


ROUGH_CHECKED = 32768          # reads of the shipped batch the oracle re-computes (its 16 host processes need 440 s for all 65536 at these edit rates)


def test_off_default_parameters_full_batch_equals_the_oracle(tmp_path):
    """full-size parity away from the default parameters.  A SHIPPED device batch (65536 reads, the
    CLI's arena for the job, default environment) of configs[3]'s reference with --identity 85,95,5 --chimeras 25 --glitches 1000,100,100
    (bench.py workload 'rough'): three times the edits per base of the defaults, so most bases leave the one-word band class -- the
    2- and 4-word classes and k_fin_align<16,8,...> carry what is a few percent at the defaults -- and a quarter of the reads are
    chimeras.  The batch runs WITHOUT a retry (round 5's retry was the OUTPUT buffer, not the arena: chimeras make reads a third
    longer than the 34 kB per read every job was given; the engine now sizes it from the job's parameters,
    HipEngine.expected_record_bytes); the band classes and the four-per-wave route are asserted from the kernels' own per-read
    records; the first 32768 reads are compared with the oracle, read by read (a prefix of a batch is the smaller batch:
    test_configs1_full_batch_properties pins that)."""
    import bench
    from badread_amd.engine import HipEngine, RS_EMPTY
    ref_dir = bench.default_ref_dir()
    wl = bench.build_workload(io.StringIO(), 'rough', ref_dir)
    # the arena the CLI gives this job (HipEngine.presize from its identity law and chimera rate: 42 GiB; the bench's 32 GiB hold the
    # batch too, without a retry, but squeeze the head set's widest class onto one wave: 150 s for the batch instead of 50)
    eng = bench.configure(HipEngine(0, scratch_bytes=1 << 30), wl)
    eng.presize(SHIPPED_BATCH, 15000.0, 0.10)
    out, st = eng.simulate_batch(SEED, 0, SHIPPED_BATCH)
    out, st = out.copy(), st.copy()
    cyc = eng.read_cycles(SHIPPED_BATCH)
    retries = getattr(eng, 'retries', 0)
    eng.close()
    assert retries == 0, 'a shipped batch of this workload must fit the arena and the output buffer sized from its parameters'
    assert (st['status'] & ~np.uint32(RS_EMPTY) == 0).all()
    words = (cyc[:, 7] & 0xFFFF).astype(np.int64)
    bases = st['frag_len'].astype(np.float64)
    share = {g: float(bases[words == g].sum() / bases.sum()) for g in (1, 2, 4, 8, 16)}
    assert int((words >= 8).sum()) >= 200, share                         # k_fin_align<16,8,...> really carries reads here
    assert share[1] < 0.5 and share[2] + share[4] > 0.3, share           # ... and the bulk has left the one-word class
    assert int(((cyc[:, 7] >> 16) & 1).sum()) >= 1000                    # ... while k_fin_quad<1> still takes the reads with narrow bands
    raw = out.tobytes()
    assert raw.count(b'chimera ') >= 5000                                  # a quarter of the reads join two fragments
    compare_with_oracle_slices('rough', ref_dir, ROUGH_CHECKED, st, raw, tmp_path, ALL_FIELDS)


@pytest.mark.parametrize('wlname,error_rate', [('rough', 0.10), ('human', 0.05), ('hifi', 0.001)])
def test_the_cli_sizes_arena_and_output_for_the_job(wlname, error_rate, tmp_path):
    """what a USER gets is the CLI's sizing -- HipEngine.presize from the job's identity law and the output buffer
    from its chimera rate -- not the bench's 40 GB.  A shipped batch through an engine sized that way: NO retry (an arena that is
    short is grown and the batch repeated: correct, but six engines growing 20 GB arenas once filled the device -- round 6, the
    configs[4] job after the survivor rings were added), and the same bytes as the engine with the bench's arena (48 GiB for the rough job) gives."""
    import bench
    from badread_amd.engine import HipEngine
    ref_dir = bench.default_ref_dir()
    wl = bench.build_workload(io.StringIO(), wlname, ref_dir)
    eng = bench.configure(HipEngine(0, scratch_bytes=1 << 30), wl)
    eng.presize(SHIPPED_BATCH, 15000.0, error_rate)                        # e.g. --identity 85,95,5: ten per cent of errors on average
    assert eng.scratch_bytes() <= int(48 * (1 << 30))
    out, st = eng.simulate_batch(SEED, 0, SHIPPED_BATCH)
    digest = (len(out), int(st['seq_len'].sum()), int(st['n_match'].astype(np.int64).sum()))
    assert getattr(eng, 'retries', 0) == 0
    eng.close()
    big = bench.configure(HipEngine(0, scratch_bytes=int((48.0 if wlname == 'rough' else bench.SCRATCH_GB_DEFAULT) * (1 << 30))), wl)
    out2, st2 = big.simulate_batch(SEED, 0, SHIPPED_BATCH)
    assert digest == (len(out2), int(st2['seq_len'].sum()), int(st2['n_match'].astype(np.int64).sum()))
    assert bytes(out[:1 << 24]) == bytes(out2[:1 << 24])
    big.close()

"""
Host logic of badread_amd.simulate on CPU: start-up pieces, the stop rule, batch-size and
rank-count independence of the output (world_size 2 over gloo), fatal-exit behaviour.  The engine
behind the driver here is the tests' oracle-backed checker (same EngineBase interface as HipEngine),
so no GPU is needed; the product never takes this route.
"""
import io
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers as H
from badread_amd import simulate as S
from badread_amd.fragment_lengths import FragmentLengths

HERE = os.path.dirname(os.path.abspath(__file__))
SMALL_REF = os.path.join(HERE, 'golden', 'small_ref.fasta')


class Args(object):
    def __init__(self, **kw):
        d = dict(reference=SMALL_REF, quantity='20x', mean_frag_length=400, frag_length_stdev=300,
                 mean_identity=90, max_identity=98, identity_stdev=4, error_model='random', qscore_model='ideal',
                 seed=5, start_adapter='90,60', end_adapter='50,20', start_adapter_seq='AATGTACTTCGTTCAGTTACGTATTGCT',
                 end_adapter_seq='GCAATACGTAACTGAACGAAGT', junk_reads=1, random_reads=1, chimeras=1,
                 glitch_rate=10000, glitch_size=25, glitch_skip=25, small_plasmid_bias=False)
        d.update(kw)
        self.__dict__.update(d)


def run(args, max_batch=None, monkeypatch=None):
    out, err = io.BytesIO(), io.StringIO()
    if max_batch is not None:
        monkeypatch.setattr(S, 'DEFAULT_MAX_BATCH', max_batch)
    count, total = S.simulate(args, output=err, engine=H.oracle_engine(), stdout=out, shard=S.Shard())
    return out.getvalue(), err.getvalue(), count, total


def parse_fastq(data):
    lines = data.decode().split('\n')
    assert lines[-1] == ''
    recs = [lines[i:i + 4] for i in range(0, len(lines) - 1, 4)]
    for h, s, p, q in recs:
        assert h.startswith('@') and p == '+' and len(s) == len(q) and len(s) > 0
    return recs


def test_stop_rule_and_format():
    data, err, count, total = run(Args())
    recs = parse_fastq(data)
    target = 20 * 3621
    assert len(recs) == count and sum(len(r[1]) for r in recs) == total
    assert total >= target and total - len(recs[-1][1]) < target          # stops at the first read reaching the target
    for h, s, _, _ in recs:
        assert f'length={len(s)} ' in h and 'error-free_length=' in h and h.endswith('%')
    assert f'Target read set size: {target:,} bp' in err and 'Simulating:' in err
    assert 'Badread v' in err and 'Read glitches:' in err and 'Start adapter:' in err


def test_output_independent_of_batch_size(monkeypatch):
    a, _, _, _ = run(Args())
    b, _, _, _ = run(Args(), max_batch=7, monkeypatch=monkeypatch)
    c, _, _, _ = run(Args(), max_batch=64, monkeypatch=monkeypatch)
    assert a == b == c


def test_seed_changes_output_and_none_is_random():
    a, _, _, _ = run(Args(seed=1, quantity='3x'))
    b, _, _, _ = run(Args(seed=2, quantity='3x'))
    c, _, _, _ = run(Args(seed=1, quantity='3x'))
    assert a != b and a == c
    d, _, _, _ = run(Args(seed=None, quantity='3x'))
    e, _, _, _ = run(Args(seed=None, quantity='3x'))
    assert d != e


def test_quantity_forms():
    for q, target in (('2500', 2500), ('3k', 3000), ('1x', 3621)):
        _, err, _, total = run(Args(quantity=q))
        assert f'Target read set size: {target:,} bp' in err and total >= target


def test_incompatible_lengths_exit_like_the_reference(tmp_path):
    # three 30 bp circular contigs with default lengths: adjust_depths cannot help -> fatal (simulate.py:159-165,526)
    ref = tmp_path / 'tiny.fasta'
    ref.write_text('>a circular=true\n' + 'ACGT' * 8 + '\n>b circular=true\n' + 'GGCA' * 8 + '\n')
    with pytest.raises(SystemExit) as ex:
        run(Args(reference=str(ref), mean_frag_length=15000, frag_length_stdev=13000))
    assert 'fragment length' in str(ex.value) or 'failed to generate' in str(ex.value)
    # --small_plasmid_bias skips the depth adjustment; a read whose length never fits still fails 1000 times
    with pytest.raises(SystemExit) as ex:
        run(Args(reference=str(ref), mean_frag_length=15000, frag_length_stdev=0, small_plasmid_bias=True,
                 junk_reads=0, random_reads=0, chimeras=0))
    assert str(ex.value) == S.NOFRAG_MESSAGE
    # fragments that fit are fine
    _, _, count, _ = run(Args(reference=str(ref), mean_frag_length=20, frag_length_stdev=5, quantity='2x',
                              junk_reads=0, random_reads=0, chimeras=0))
    assert count >= 1


def test_adjust_depths_matches_the_reference_formula():
    pref = H.small_reference()[0]
    fl = FragmentLengths(3000, 2500, io.StringIO())
    rng = np.random.RandomState(3)
    lengths = fl.sample_many(S.ADJUST_SAMPLES, np.random.RandomState(3))
    got = S.adjust_depths(pref, fl, False, rng)
    total = lengths.sum()
    for name, L in zip(pref.names, pref.lengths):
        if pref.circular[name]:
            expect = pref.depths[name] * total / lengths[lengths <= L].sum()
        else:
            expect = pref.depths[name] * total / np.minimum(lengths, L).sum()
        assert abs(got[name] - expect) <= 1e-12 * expect
    same = S.adjust_depths(pref, fl, True, np.random.RandomState(3))
    for name in pref.names:
        if pref.circular[name]:
            assert same[name] == pref.depths[name]


def test_cut_point_and_plan_batch():
    assert S.cut_point([10, 0, 5, 7], 0, 15) == 2
    assert S.cut_point([10, 0, 5, 7], 0, 16) == 3
    assert S.cut_point([10, 0, 5, 7], 0, 23) is None
    assert S.cut_point([0, 0], 100, 50) is None            # empty reads never stop the loop
    assert S.plan_batch(1500000, 15000.0, 1, 16384) % 64 == 0
    assert S.plan_batch(10 ** 12, 15000.0, 8, 16384) == 8 * 16384
    sh = S.Shard(1, 2)
    assert sh.slice_of(100, 10) == (105, 5) and S.Shard(0, 2).slice_of(100, 9) == (100, 5) and sh.slice_of(100, 9) == (105, 4)


WORKER = r'''
import io, os, sys
sys.path[:0] = [{repo!r}, {repo!r} + '/oracle', {repo!r} + '/tests']
import helpers as H
from badread_amd import simulate as S
from test_host_simulate import Args
S.DEFAULT_MAX_BATCH = 24
shard = S.Shard.from_env()
out = io.BytesIO()
S.simulate(Args(), output=io.StringIO(), engine=H.oracle_engine(), stdout=out, shard=shard)
if shard.rank == 0:
    open({outfile!r}, 'wb').write(out.getvalue())
else:
    assert out.getvalue() == b''
'''


def _free_port():
    import socket
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        return sock.getsockname()[1]


@pytest.mark.parametrize('world', [2, 3])
def test_ranks_over_gloo_give_the_single_process_bytes(tmp_path, monkeypatch, world):
    """SURVEY.md 8d gate 5 / 8e: read indices sharded over `world` processes (one per GPU in production, gloo and
    the CPU checker here), no collective on the data path, output byte-identical to the single-process run --
    also for a world size that does not divide the batch evenly."""
    port = _free_port()
    single, _, _, _ = run(Args(), max_batch=24, monkeypatch=monkeypatch)
    outfile = str(tmp_path / 'ranks.fastq')
    script = tmp_path / 'worker.py'
    script.write_text(WORKER.format(repo=os.path.dirname(HERE), outfile=outfile))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}',
                        '--master-addr', '127.0.0.1', '--master-port', str(port), str(script)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert open(outfile, 'rb').read() == single


SHARD_WORKER = r'''
import io, os, sys
sys.path[:0] = [{repo!r}, {repo!r} + '/oracle', {repo!r} + '/tests']
import helpers as H
from badread_amd import simulate as S
from test_host_simulate import Args
S.DEFAULT_MAX_BATCH = 24
shard = S.Shard.from_env()
out = io.BytesIO()
S.simulate(Args(output_shards={prefix!r}, gzip_level={gzip!r}, gpu_streams=3), output=io.StringIO(), engine=H.oracle_engine(), stdout=out, shard=shard)
assert out.getvalue() == b''          # nothing goes through rank 0's stdout
'''


def _torchrun(script, world, port, timeout=600):
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    return subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}',
                           '--master-addr', '127.0.0.1', '--master-port', str(port), str(script)],
                          env=env, capture_output=True, text=True, timeout=timeout)


def reassemble(prefix, world, gz=False):
    """The ranks' files put back into read order: batch by batch, rank after rank (PREFIX.<rank>.parts = bytes per batch)."""
    import gzip
    datas, parts = [], []
    for r in range(world):
        raw = open(f'{prefix}.{r}.fastq' + ('.gz' if gz else ''), 'rb').read()
        parts.append([int(x) for x in open(f'{prefix}.{r}.parts').read().split()])
        datas.append(raw)
    if gz:        # members are per write: whole files decompress to the rank's text; order across ranks needs the text sizes
        return [gzip.decompress(d) for d in datas], parts
    out, at = b'', [0] * world
    for b in range(max(len(p) for p in parts)):
        for r in range(world):
            if b < len(parts[r]):
                out += datas[r][at[r]:at[r] + parts[r][b]]
                at[r] += parts[r][b]
    assert all(at[r] == len(datas[r]) for r in range(world))
    return out, parts


@pytest.mark.parametrize('world', [2, 3])
def test_output_shards_over_gloo_reassemble_to_the_single_process_bytes(tmp_path, monkeypatch, world):
    """--output-shards PREFIX -- every rank writes the records of its own reads (its own ring, its own
    file); the 4 B/read exchange and the stop rule are unchanged, so the files, put back batch by batch and rank after rank,
    are the single-process bytes; nothing travels to rank 0."""
    single, _, _, _ = run(Args(), max_batch=24, monkeypatch=monkeypatch)
    prefix = str(tmp_path / 'shard')
    script = tmp_path / 'worker.py'
    script.write_text(SHARD_WORKER.format(repo=os.path.dirname(HERE), prefix=prefix, gzip=None))
    r = _torchrun(script, world, _free_port())
    assert r.returncode == 0, r.stderr[-3000:]
    got, parts = reassemble(prefix, world)
    assert got == single
    assert all(len(p) == len(parts[0]) for p in parts)           # every rank saw the same batches
    # the same records whatever the order: what `cat PREFIX.*.fastq` gives a user
    cat = b''.join(open(f'{prefix}.{r}.fastq', 'rb').read() for r in range(world))
    assert sorted(tuple(x) for x in parse_fastq(cat)) == sorted(tuple(x) for x in parse_fastq(single))


def test_output_shards_with_host_gzip(tmp_path, monkeypatch):
    single, _, _, _ = run(Args(), max_batch=24, monkeypatch=monkeypatch)
    prefix = str(tmp_path / 'shardz')
    script = tmp_path / 'worker.py'
    script.write_text(SHARD_WORKER.format(repo=os.path.dirname(HERE), prefix=prefix, gzip=1))
    r = _torchrun(script, 2, _free_port())
    assert r.returncode == 0, r.stderr[-3000:]
    texts, _ = reassemble(prefix, 2, gz=True)
    assert sorted(tuple(x) for x in parse_fastq(b''.join(texts))) == sorted(tuple(x) for x in parse_fastq(single))
    # .parts counts FASTQ text; .zparts (--gzip only) counts the compressed bytes of the same batches, so the .gz files
    # go back into read order batch by batch, rank after rank, WITHOUT being decompressed first
    import gzip
    gz = [open(f'{prefix}.{r}.fastq.gz', 'rb').read() for r in range(2)]
    zparts = [[int(x) for x in open(f'{prefix}.{r}.zparts').read().split()] for r in range(2)]
    parts = [[int(x) for x in open(f'{prefix}.{r}.parts').read().split()] for r in range(2)]
    assert [len(z) for z in zparts] == [len(p) for p in parts] and all(sum(z) == len(g) for z, g in zip(zparts, gz))
    joined, at = b'', [0, 0]
    for b in range(len(zparts[0])):
        for r in range(2):
            member = gz[r][at[r]:at[r] + zparts[r][b]]
            assert len(gzip.decompress(member)) == parts[r][b] if member else parts[r][b] == 0
            joined += member
            at[r] += zparts[r][b]
    assert gzip.decompress(joined) == single


def test_single_process_output_shards(tmp_path, monkeypatch):
    single, _, _, _ = run(Args(), max_batch=24, monkeypatch=monkeypatch)
    prefix = str(tmp_path / 'one')
    out, _, _, _ = run(Args(output_shards=prefix), max_batch=24, monkeypatch=monkeypatch)
    assert out == b'' and open(prefix + '.0.fastq', 'rb').read() == single


FAIL_WORKER = r'''
import io, os, sys
sys.path[:0] = [{repo!r}, {repo!r} + '/oracle', {repo!r} + '/tests']
import helpers as H
from badread_amd import simulate as S
from test_host_simulate import Args
S.DEFAULT_MAX_BATCH = 8
shard = S.Shard.from_env()

class Full(object):
    n = 0
    def write(self, part):
        Full.n += 1
        if Full.n >= 2:
            raise OSError(28, 'No space left on device')

try:
    S.simulate(Args(quantity='60x'), output=io.StringIO(), engine=H.oracle_engine(), stdout=Full(), shard=shard)
except OSError as ex:
    assert shard.rank == 0 and ex.errno == 28
    open({marker!r} + '.0', 'w').write('oserror')
except SystemExit as ex:
    assert shard.rank != 0 and 'rank 0' in str(ex.code)
    open({marker!r} + '.%d' % shard.rank, 'w').write('exit')
else:
    raise AssertionError('the run went on after its sink had failed')
'''


def test_a_failed_sink_stops_every_rank_at_the_same_batch(tmp_path):
    """a sink that fails on rank 0 mid-run (full disk, closed pipe) used to leave the other ranks blocked in the
    next collective.  The per-batch exchange now carries one "my sink failed" word per rank: rank 0 raises its error, the
    others exit with a message, and the launcher returns instead of timing out."""
    marker = str(tmp_path / 'stopped')
    script = tmp_path / 'worker.py'
    script.write_text(FAIL_WORKER.format(repo=os.path.dirname(HERE), marker=marker))
    r = _torchrun(script, 2, _free_port(), timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    assert open(marker + '.0').read() == 'oserror' and open(marker + '.1').read() == 'exit'


def test_clones_are_made_only_for_batches_the_job_issues(monkeypatch):
    """_BatchPool started a maker thread per clone whatever the job needed -- a one-batch job mapped
    ~200 GB of clone arenas it never used.  Clone i is now started with the (i + 1)-th submitted batch; a clone that cannot
    be created (out of memory) leaves the job on fewer engines instead of failing the batch that drew its slot."""
    made = []

    class Eng(object):
        stats_dtype = np.dtype([('x', 'u4')])
        def __init__(self, fail=False):
            self.fail = fail
        def clone(self):
            if len(made) == 1:                       # the second clone fails
                made.append('failed')
                raise MemoryError('arena')
            made.append('ok')
            return Eng()
        def simulate_batch(self, seed, first, n, allow_nofrag=True):
            return np.zeros(4, dtype=np.uint8), np.zeros(n, dtype=self.stats_dtype)
        def close(self):
            pass

    pool = S._BatchPool(Eng(), 4)
    assert pool.started == 0 and made == []
    pool.submit(1, 0, 2).result()
    assert pool.started == 0 and made == []          # a one-batch job: no clone
    futs = [pool.submit(1, 2 * i, 2) for i in range(1, 6)]
    for f in futs:
        f.result()                                   # every batch ran although a clone failed
    pool.close()
    assert pool.started == 3 and sorted(made) == ['failed', 'ok', 'ok'] and len(pool.create_errors) == 1


def test_batches_in_flight_do_not_change_the_output(monkeypatch):
    """--gpu-streams: several super-batches run at once on engine clones and are consumed in index order; the bytes,
    the read count and the stop point are those of the one-batch-at-a-time run, also when speculative batches
    overshoot the target."""
    base, _, count, total = run(Args(quantity='40x'), max_batch=16, monkeypatch=monkeypatch)
    for streams, mb in ((3, 16), (4, 5), (2, 4096)):
        got, _, c2, t2 = run(Args(quantity='40x', gpu_streams=streams), max_batch=mb, monkeypatch=monkeypatch)
        assert got == base and (c2, t2) == (count, total), (streams, mb)


def test_batches_in_flight_follow_the_free_device_memory(monkeypatch):
    """_BatchPool.engines_that_fit: clones map an arena as large as the first engine's, every engine owns an output buffer,
    in_flight + 2 copies of a batch may wait for the consumer, and a reserve stays free for the runtime's own allocations."""
    GB = 1 << 30

    class FakeCuda(object):
        def __init__(self, free):
            self.free = free
        def empty_cache(self):
            pass
        def mem_get_info(self, device):
            return self.free, 288 * GB

    class FakeTorch(object):
        pass

    class FakeEngine(object):
        device = 'cuda:0'
        def scratch_bytes(self):
            return 40 * GB

    t = FakeTorch()
    monkeypatch.delenv('BRX_DRIVER_RESERVE_GB', raising=False)
    fit = S._BatchPool.engines_that_fit
    t.cuda = FakeCuda(247 * GB)          # a fresh process after the first 40 GB arena and the reference: five clones + 14 buffers of 2 GB + 24 GB
    assert fit(t, FakeEngine(), 6, 2 * GB) == 5
    t.cuda = FakeCuda(260 * GB)
    assert fit(t, FakeEngine(), 6, 2 * GB) == 6
    t.cuda = FakeCuda(10 * GB)           # never below one engine: the first one exists already
    assert fit(t, FakeEngine(), 6, 2 * GB) == 1
    monkeypatch.setenv('BRX_DRIVER_RESERVE_GB', '0')
    t.cuda = FakeCuda(229 * GB)
    assert fit(t, FakeEngine(), 6, 2 * GB) == 6


def test_device_copies_of_a_batch_come_in_size_steps(monkeypatch):
    """simulate._BatchPool._copy_of: the bytes of a finished batch leave the engine's buffer in a block whose capacity is a multiple of
    COPY_STEP, so that torch's caching allocator can hand a freed block to the next batch (exact sizes -- 2.05-2.10 GB, never the same twice
    -- made the cache creep until the 30x job died with HSA_STATUS_ERROR_OUT_OF_RESOURCES at 88 %).  The bytes are those of the source."""
    import torch
    from badread_amd.simulate import _BatchPool
    monkeypatch.setattr(_BatchPool, 'COPY_STEP', 1 << 12)
    caps = set()
    for n in (4097, 5000, 8191, 8192):
        src = torch.arange(n, dtype=torch.int64).to(torch.uint8)
        got = _BatchPool._copy_of(torch, src)
        assert got.numel() == n and bool((got == src).all())
        caps.add(got.untyped_storage().nbytes())
    assert caps == {8192}
    small = torch.arange(100, dtype=torch.uint8)
    assert _BatchPool._copy_of(torch, small).untyped_storage().nbytes() == 100      # small batches: a plain clone

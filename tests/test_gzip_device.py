"""
brx_gzip_device (include/brx.h; kernels in badread_amd/csrc/brx_gzip_dev.h): the FASTQ bytes of a batch as gzip members made
on the device.  Whatever the kernels write must be what every gzip reader accepts and must decompress to the input, byte
for byte -- the readers also verify the CRC-32 and the length in each member's trailer.  CPU: the kernels interpreted
(tests/emu_engine.py); GPU: the same cases on the MI355X.
"""
import gzip
import zlib

import numpy as np
import pytest

import helpers as H


def fastq_like(rng, n):
    """Text with the statistics of the simulator's output: headers, bases, '+', qualities."""
    parts = []
    size = 0
    while size < n:
        L = int(rng.integers(50, 4000))
        seq = rng.choice(np.frombuffer(b'ACGT', dtype=np.uint8), size=L, p=[0.3, 0.2, 0.2, 0.3]).tobytes()
        qual = (33 + np.clip(rng.normal(20, 8, size=L), 1, 50).astype(np.uint8)).tobytes()
        rec = b'@' + bytes(rng.integers(48, 58, size=36).astype(np.uint8)) + b' chr1,+strand,1-' + str(L).encode() + b' length=' + str(L).encode() + b'\n' + seq + b'\n+\n' + qual + b'\n'
        parts.append(rec)
        size += len(rec)
    return b''.join(parts)[:n]


CASES = [('fastq', 1), ('fastq', 3), ('fastq', 1023), ('fastq', 1024), ('fastq', 1025), ('fastq', 65535), ('fastq', 65536), ('fastq', 65537),
         ('fastq', 200001), ('bytes', 70000), ('one', 5000), ('two', 66000), ('skew', 131072)]


def make_case(kind, n):
    rng = np.random.default_rng(n)
    if kind == 'fastq':
        return fastq_like(rng, n)
    if kind == 'bytes':                       # all 256 values: a flat code of 8-9 bits
        return rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
    if kind == 'one':
        return b'A' * n
    if kind == 'two':
        return (b'AC' * n)[:n]
    # a Fibonacci-like histogram: an unconstrained Huffman code would need more than 15 bits
    counts = [1, 1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233, 377, 610, 987, 1597, 2584, 4181, 6765, 10946, 17711]
    data = np.concatenate([np.full(c, 40 + i, dtype=np.uint8) for i, c in enumerate(counts)])
    rng.shuffle(data)
    return np.resize(data, n).tobytes()


def check(engine):
    import torch
    for kind, n in CASES:
        data = make_case(kind, n)
        src = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(engine.device)
        out = engine.gzip_device(src)
        blob = bytes(out.cpu().numpy().tobytes())
        assert gzip.decompress(blob) == data, (kind, n)
        # member by member with zlib (wbits 31 = gzip container, CRC and length checked): one member per 64 KB
        members, rest = 0, blob
        while rest:
            d = zlib.decompressobj(31)
            d.decompress(rest)
            assert d.eof
            rest = d.unused_data
            members += 1
        assert members == -(-n // 65536)
        if kind == 'fastq' and n > 60000:
            assert len(blob) < 1.02 * len(zlib.compress(data, 1)), (len(blob), n)       # an order-0 code per 64 KB: what gzip -1 reaches on this text
    assert int(engine.gzip_device(torch.zeros(0, dtype=torch.uint8, device=engine.device)).numel()) == 0


def test_interpreted_kernels_write_valid_gzip_members():
    import emu_engine as EE
    check(EE.EmuEngine(1 << 26))


def test_crc_folding_operator_matches_zlib():
    """The x^n mod P algebra of the CRC fold, restated in Python, against zlib.crc32 on split buffers."""
    poly = 0xEDB88320

    def mulmod(a, b):
        p = 0
        for i in range(32):
            if a & (0x80000000 >> i):
                p ^= b
            b = (b >> 1) ^ poly if b & 1 else b >> 1
        return p
    x2n = [0x40000000]
    for _ in range(31):
        x2n.append(mulmod(x2n[-1], x2n[-1]))

    def x2nmodp(n, k):
        p = 0x80000000
        while n:
            if n & 1:
                p = mulmod(x2n[k & 31], p)
            n >>= 1
            k += 1
        return p
    rng = np.random.default_rng(5)
    for la, lb in ((0, 7), (1024, 1024), (5, 0), (4096, 333), (70000, 65536)):
        a, b = rng.integers(0, 256, la, dtype=np.uint8).tobytes(), rng.integers(0, 256, lb, dtype=np.uint8).tobytes()
        assert mulmod(x2nmodp(lb, 3), zlib.crc32(a)) ^ zlib.crc32(b) == zlib.crc32(a + b)
    assert x2n[13] == x2nmodp(1024, 3)


@pytest.mark.gpu
def test_hip_kernels_write_valid_gzip_members():
    check(H.hip_engine())


def simulated_text(n_reads=120, frag_mean=3000.0):
    from badread_amd.engine import SimParams
    pref, _ = H.small_reference()
    orc = H.configure(H.oracle_engine(), pref, 'nanopore2023', 'nanopore2023', SimParams(frag_mean=frag_mean, frag_stdev=frag_mean))
    out, st = orc.simulate_batch(11, 0, n_reads, allow_nofrag=True)
    return bytes(out), st.copy()


def test_blocks_are_cut_at_the_lines_and_cover_the_text():
    from badread_amd.output import fastq_blocks, LONG_READ
    data, st = simulated_text()
    n = len(data)
    cuts = fastq_blocks(st['rec_off'], st['rec_len'], st['seq_len'], n).astype(np.int64)
    assert cuts[0] == 0 and cuts[-1] == n and (np.diff(cuts) > 0).all()
    for c in cuts[1:-1]:
        assert data[c - 1:c] == b'\n'
    # a long read is two blocks: '@...' up to the end of the sequence line, '+' to the end of the record
    long_reads = np.flatnonzero(st['seq_len'] >= LONG_READ)
    assert len(long_reads) > 10
    for r in long_reads[:20]:
        a = int(st['rec_off'][r])
        i = int(np.searchsorted(cuts, a))
        assert cuts[i] == a and data[a:a + 1] == b'@' and data[cuts[i + 1]:cuts[i + 1] + 2] == b'+\n' and cuts[i + 2] == a + int(st['rec_len'][r])
    # a prefix of the batch (the stop rule keeps reads 0..k): same rule, ends at its last record
    k = len(st) // 2
    n_k = int(st['rec_off'][k] + st['rec_len'][k])
    cuts_k = fastq_blocks(st['rec_off'][:k + 1], st['rec_len'][:k + 1], st['seq_len'][:k + 1], n_k).astype(np.int64)
    assert cuts_k[-1] == n_k and set(cuts_k) <= set(cuts)
    # short reads only: merged into blocks of about 64 KB
    data_s, st_s = simulated_text(400, 300.0)
    cuts_s = fastq_blocks(st_s['rec_off'], st_s['rec_len'], st_s['seq_len'], len(data_s)).astype(np.int64)
    sizes = np.diff(cuts_s)
    assert (st_s['seq_len'] < LONG_READ).mean() > 0.9 and len(sizes) < len(st_s) / 10 and sizes.max() < 3 * 65536


def test_line_cut_blocks_beat_one_code_per_64_kb_on_simulated_reads():
    import emu_engine as EE
    import torch
    from badread_amd.output import fastq_blocks
    data, st = simulated_text()
    eng = EE.EmuEngine(1 << 26)
    src = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy())
    blob = bytes(eng.gzip_device(src, fastq_blocks(st['rec_off'], st['rec_len'], st['seq_len'], len(data))).numpy().tobytes())
    assert gzip.decompress(blob) == data
    flat = bytes(eng.gzip_device(src).numpy().tobytes())
    assert gzip.decompress(flat) == data
    # (the test's reference is 50 kb, so reads repeat each other inside gzip's 32 KB window: level 6 is level with the
    # line-cut codes here; on a genome there is nothing to match and it stays at 0.55 against 0.51: DESIGN.md section 7)
    assert len(blob) < 0.95 * len(flat) and len(blob) < len(zlib.compress(data, 1)) and len(blob) < 1.02 * len(zlib.compress(data, 6)), (len(blob), len(flat))


def test_driver_gzip_device_option_equals_plain_output():
    """`--gzip-device` through the driver on the interpreted engine: batches in flight, stop rule, then the kept bytes of
    every batch as gzip members."""
    import io
    import emu_engine as EE
    from badread_amd import simulate as S
    from test_host_simulate import Args
    plain, packed = io.BytesIO(), io.BytesIO()
    S.simulate(Args(quantity='12x'), output=io.StringIO(), engine=EE.EmuEngine(1 << 26), stdout=plain, shard=S.Shard())
    S.simulate(Args(quantity='12x', gzip_device=True), output=io.StringIO(), engine=EE.EmuEngine(1 << 26), stdout=packed, shard=S.Shard())
    assert gzip.decompress(packed.getvalue()) == plain.getvalue() and 0 < len(packed.getvalue()) < len(plain.getvalue())
    with pytest.raises(SystemExit):
        S.simulate(Args(gzip_device=True, gzip_level=3), output=io.StringIO(), engine=EE.EmuEngine(1 << 26), stdout=io.BytesIO(), shard=S.Shard())


@pytest.mark.gpu
def test_command_line_gzip_device_equals_plain_output():
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    base = [sys.executable, '-m', 'badread_amd', 'simulate', '--reference', os.path.join(here, 'golden', 'small_ref.fasta'), '--quantity', '40x',
            '--length', '2500,2000', '--seed', '11']
    plain = subprocess.run(base, cwd=os.path.dirname(here), capture_output=True, timeout=600)
    packed = subprocess.run(base + ['--gzip-device'], cwd=os.path.dirname(here), capture_output=True, timeout=600)
    assert plain.returncode == 0 and packed.returncode == 0, packed.stderr.decode()[-2000:]
    assert gzip.decompress(packed.stdout) == plain.stdout and len(packed.stdout) < 0.62 * len(plain.stdout)


@pytest.mark.gpu
def test_a_full_batch_of_simulated_reads_round_trips():
    """2048 reads of the bench's configs[1] workload: line-cut members, 30 MB of text, decompressed by zlib."""
    import io
    import bench
    from badread_amd.engine import HipEngine
    from badread_amd.output import fastq_blocks
    eng = bench.configure(HipEngine(0, scratch_bytes=8 << 30), bench.build_workload(io.StringIO(), 'kpn', bench.default_ref_dir()))
    out, st = eng.simulate_batch_device(42, 0, 2048)
    n = int(st['rec_off'][-1] + st['rec_len'][-1])
    blob = eng.gzip_device(out[:n], fastq_blocks(st['rec_off'], st['rec_len'], st['seq_len'], n))
    text = bytes(out[:n].cpu().numpy().tobytes())
    packed = bytes(blob.cpu().numpy().tobytes())
    assert gzip.decompress(packed) == text
    assert len(packed) < 0.53 * n and len(packed) < len(zlib.compress(text[:4000000], 6)) * (n / 4000000.0)
    eng.close()


@pytest.mark.gpu
def test_batches_the_stop_rule_cannot_cut_are_packed_by_their_own_worker(monkeypatch):
    """--gzip-device, round 6: the worker of a batch that has several batches' worth of bases still to come behind it packs the
    batch on its own engine's stream (one gzip stage per engine instead of one consumer thread); the job's last batches are
    packed by the consumer after the stop rule.  The stream decompresses to the plain run's bytes, and most batches took the
    worker's route."""
    import io
    from badread_amd import simulate as S
    from badread_amd.engine import HipEngine
    from test_host_simulate import Args
    monkeypatch.setattr(S, 'DEFAULT_MAX_BATCH', 128)
    outs, timing = {}, None
    for flag in (False, True):
        buf = io.BytesIO()
        eng = HipEngine(0, scratch_bytes=1 << 30)
        S.simulate(Args(quantity='400x', gzip_device=flag, gpu_streams=3), output=io.StringIO(), engine=eng, stdout=buf, shard=S.Shard())
        outs[flag] = buf.getvalue()
        if flag:
            timing = dict(S.run_batches.last_timing)
    assert gzip.decompress(outs[True]) == outs[False]
    assert timing['batches'] >= 10 and timing['batches_packed_by_their_worker'] >= timing['batches'] - 6, timing


GZ_WORKER = '''
import io, os, sys
sys.path.insert(0, {repo!r}); sys.path.insert(0, os.path.join({repo!r}, 'tests')); sys.path.insert(0, os.path.join({repo!r}, 'oracle'))
import emu_engine as EE
from badread_amd import simulate as S
from test_host_simulate import Args
S.DEFAULT_MAX_BATCH = 24
shard = S.Shard.from_env()
out = io.BytesIO()
S.simulate(Args(quantity='12x', gzip_device=True), output=io.StringIO(), engine=EE.EmuEngine(1 << 26), stdout=out, shard=shard)
if shard.rank == 0:
    open({outfile!r}, 'wb').write(out.getvalue())
else:
    assert out.getvalue() == b''
'''


def test_two_ranks_compress_their_own_bytes_and_rank_0_writes_one_stream(tmp_path, monkeypatch):
    """--gzip-device over two ranks (gloo, interpreted kernels): every rank packs the records it keeps, the members travel
    to rank 0 in read order, and the stream decompresses to the single-process text."""
    import io
    import os
    import subprocess
    import sys
    import emu_engine as EE
    from badread_amd import simulate as S
    from test_host_simulate import Args, _free_port
    here = os.path.dirname(os.path.abspath(__file__))
    monkeypatch.setattr(S, 'DEFAULT_MAX_BATCH', 24)
    plain = io.BytesIO()
    S.simulate(Args(quantity='12x'), output=io.StringIO(), engine=EE.EmuEngine(1 << 26), stdout=plain, shard=S.Shard())
    port = _free_port()
    outfile = str(tmp_path / 'ranks.fastq.gz')
    script = tmp_path / 'worker.py'
    script.write_text(GZ_WORKER.format(repo=os.path.dirname(here), outfile=outfile))
    # (gloo and one device index for both ranks: the test also passes when it is run, unfiltered, on a box that has ONE GPU)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), BRX_DIST_BACKEND='gloo', BRX_DEVICE='0')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
                        '--master-addr', '127.0.0.1', '--master-port', str(port), str(script)],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert gzip.decompress(open(outfile, 'rb').read()) == plain.getvalue()

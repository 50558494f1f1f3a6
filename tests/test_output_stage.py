"""
Host output stage (libbrx_host.so, csrc/brx_gzip.cpp): the multi-member gzip stream decompresses
to exactly the plain FASTQ, for every level / thread count / block size, for empty and incompressible input, and
through the driver (`--gzip`).
"""
import gzip
import io
import subprocess

import numpy as np
import pytest

import helpers as H
from badread_amd import simulate as S
from badread_amd.output import GzipSink
from test_host_simulate import Args


@pytest.mark.parametrize('level,threads,block', [(1, 1, 1 << 20), (6, 4, 1 << 16), (9, 3, 1000), (0, 2, 1 << 12)])
def test_members_concatenate_to_the_input(level, threads, block):
    rng = np.random.default_rng(level)
    text = b''.join(b'@r%d\n' % i + bytes(rng.choice(list(b'ACGT'), 200 + i % 97).astype(np.uint8)) + b'\n+\n' + b'I' * (200 + i % 97) + b'\n'
                    for i in range(3000))
    sink = io.BytesIO()
    gz = GzipSink(sink, level, threads, block)
    cut = len(text) // 3
    gz.write(text[:cut]); gz.write(b''); gz.write(memoryview(text[cut:]))
    out = sink.getvalue()
    assert gzip.decompress(out) == text
    assert gz.bytes_in == len(text) and gz.bytes_out == len(out)
    if level > 0:
        assert len(out) < len(text) // 2
    noise = rng.integers(0, 256, 300000, dtype=np.uint8).tobytes()          # incompressible: bound must hold
    sink = io.BytesIO()
    GzipSink(sink, level, threads, block).write(noise)
    assert gzip.decompress(sink.getvalue()) == noise


def test_system_gzip_reads_the_stream(tmp_path):
    sink = io.BytesIO()
    GzipSink(sink, 6, 4, 5000).write(b'ACGT' * 20000)
    p = tmp_path / 'x.gz'
    p.write_bytes(sink.getvalue())
    r = subprocess.run(['gzip', '-dc', str(p)], capture_output=True)
    assert r.returncode == 0 and r.stdout == b'ACGT' * 20000


def test_driver_gzip_option_equals_plain_output():
    plain, packed = io.BytesIO(), io.BytesIO()
    S.simulate(Args(), output=io.StringIO(), engine=H.oracle_engine(), stdout=plain, shard=S.Shard())
    S.simulate(Args(gzip_level=4), output=io.StringIO(), engine=H.oracle_engine(), stdout=packed, shard=S.Shard())
    assert gzip.decompress(packed.getvalue()) == plain.getvalue() and len(packed.getvalue()) < len(plain.getvalue())

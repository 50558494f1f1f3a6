"""
CLI parity (flags, derived fields, validation messages and exit codes of
/root/reference/badread/__main__.py:239-336, as pinned by the reference's test/test_cli.py) and the
C-ABI surface: libbrx_hip.so loads, exports every symbol include/brx.h declares, and the product
refuses to run without a ROCm device instead of falling back to anything.  CPU only, no compute calls.
"""
import ctypes
import os
import re
import sys

import pytest

from badread_amd import __main__ as cli
from badread_amd import engine

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.path.join(HERE, 'golden', 'small_ref.fasta')


def parse(*extra):
    args = cli.parse_args(['simulate', '--reference', REF, '--quantity', '1x'] + list(extra))
    cli.check_simulate_args(args)
    return args


def test_defaults_and_derived_fields():
    a = parse()
    assert (a.mean_frag_length, a.frag_length_stdev) == (15000.0, 13000.0)
    assert (a.mean_identity, a.max_identity, a.identity_stdev) == (95.0, 99.0, 2.5)
    assert (a.glitch_rate, a.glitch_size, a.glitch_skip) == (10000.0, 25.0, 25.0)
    assert a.error_model == a.qscore_model == 'nanopore2023' and a.seed is None
    assert a.start_adapter == '90,60' and a.end_adapter == '50,20'
    assert a.start_adapter_seq == 'AATGTACTTCGTTCAGTTACGTATTGCT' and a.end_adapter_seq == 'GCAATACGTAACTGAACGAAGT'
    assert (a.junk_reads, a.random_reads, a.chimeras, a.small_plasmid_bias) == (1, 1, 1, False)
    b = parse('--identity', '30,3', '--start_adapter_seq', 'acgt', '--end_adapter_seq', '12')
    assert (b.mean_identity, b.max_identity, b.identity_stdev) == (30.0, None, 3.0)
    assert b.start_adapter_seq == 'ACGT' and b.end_adapter_seq == '12'


@pytest.mark.parametrize('extra,message', [
    (['--length', '100,10'], 'mean read length must be at least 100'),
    (['--length=1000,-1'], 'read length stdev cannot be negative'),
    (['--length', 'abc'], 'could not parse --length values'),
    (['--length', '1000'], 'could not parse --length values'),
    (['--identity', '101,102,5'], 'mean read identity cannot be more than 100'),
    (['--identity', '90,101,5'], 'max read identity cannot be more than 100'),
    (['--identity', '50,90,5'], 'mean read identity must be at least 50'),
    (['--identity', '90,80,5'], 'cannot be larger than max identity'),
    (['--identity=90,95,-1'], 'read identity stdev cannot be negative'),
    (['--identity', '5,3'], 'mean read identity must be at least 5'),
    (['--identity=20,-3'], 'read qscore stdev cannot be negative'),
    (['--identity', '90'], 'could not parse --identity values'),
    (['--identity', '1,2,3,4'], 'could not parse --identity values'),
    (['--glitches', '1,2'], 'could not parse --glitches values'),
    (['--glitches=-1,2,3'], '--glitches must contain non-negative values'),
    (['--chimeras', '51'], '--chimeras cannot be greater than 50'),
    (['--junk_reads', '101'], '--junk_reads cannot be greater than 100'),
    (['--random_reads', '101'], '--random_reads cannot be greater than 100'),
    (['--junk_reads', '60', '--random_reads', '60'], 'cannot sum to more than 100'),
    (['--error_model', 'nope'], '--error_model must be from'),
    (['--qscore_model', 'nope'], '--qscore_model must be from'),
    (['--start_adapter_seq', 'ACGX'], '--start_adapter_seq must be a DNA sequence or a number'),
    (['--end_adapter_seq', 'hello'], '--end_adapter_seq must be a DNA sequence or a number'),
])
def test_validation_messages(extra, message):
    with pytest.raises(SystemExit) as ex:
        parse(*extra)
    assert message in str(ex.value)


def test_missing_reference_and_usage_errors(capsys):
    args = cli.parse_args(['simulate', '--reference', '/no/such/file', '--quantity', '1x'])
    with pytest.raises(SystemExit) as ex:
        cli.check_simulate_args(args)
    assert str(ex.value) == 'Error: /no/such/file is not a file'
    with pytest.raises(SystemExit) as ex:
        cli.parse_args([])
    assert ex.value.code == 1
    with pytest.raises(SystemExit) as ex:
        cli.parse_args(['simulate', '--quantity', '1x'])
    assert ex.value.code == 2
    with pytest.raises(SystemExit) as ex:
        cli.parse_args(['--version'])
    assert ex.value.code == 0 and 'Badread v' in capsys.readouterr().out


# ------------------------------------------------------------------------------------------------
def declared_symbols():
    text = open(os.path.join(REPO, 'include', 'brx.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(brx_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    lib = engine.load_library()
    names = declared_symbols()
    assert {'brx_create', 'brx_simulate_batch', 'brx_sequence_fragments', 'brx_align_batch'} <= set(names)
    for name in names:
        assert getattr(lib, name) is not None, name
    assert b'gfx950' in lib.brx_version()


def test_struct_layouts_match_the_header():
    # sizes the C side static-asserts implicitly by being read through these ctypes mirrors
    assert ctypes.sizeof(engine.BrxContig) == 24 and ctypes.sizeof(engine.BrxException) == 24
    assert engine.READ_STATS_DTYPE.itemsize == 64
    assert ctypes.sizeof(engine.BrxSimParams) % 8 == 0 and ctypes.sizeof(engine.BrxReference) % 8 == 0


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a ROCm device is present')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        engine.HipEngine(0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, 'badread_amd')
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                text = open(os.path.join(root, f), errors='replace').read()
                assert 'pyoracle' not in text and 'liboracle' not in text and 'brx_oracle' not in text, f

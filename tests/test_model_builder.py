"""
SURVEY.md section 8f row f4: `badread error_model` / `badread qscore_model` on the device (badread_amd/model_builder.py,
csrc/brx_model.h, C-ABI brx_model_count) against tests/golden/model_builder.json.gz -- the text the REFERENCE's own
make_error_model (error_model.py:31-83) and make_qscore_model (qscore_model.py:78-162) print for the committed inputs
(tests/golden/model_builder/: reference, reads and PAF alignments made by tools/make_golden.py), seven argument sets.

  not gpu:  the oracle's plain-Python restatement (oracle/model_builder_ref.py) == the reference's text;
            the product kernels, interpreted on the CPU, == the reference's text; loader filters; spilled windows
  gpu:      the HIP kernels through the C-ABI == the reference's text, and the command line writes it to stdout
"""
import gzip
import io
import json
import os
import subprocess
import sys
import types

import pytest

import helpers as H  # noqa: F401  (sys.path)

HERE = os.path.dirname(os.path.abspath(__file__))
FOLDER = os.path.join(HERE, 'golden', 'model_builder')
with gzip.open(os.path.join(HERE, 'golden', 'model_builder.json.gz'), 'rt') as _f:
    GOLDEN = json.load(_f)


def namespace(case):
    return types.SimpleNamespace(reference=os.path.join(FOLDER, 'ref.fasta'), reads=os.path.join(FOLDER, 'reads.fastq'),
                                 alignment=os.path.join(FOLDER, 'aln.paf'), **case)


def run_product(kind, case, engine):
    from badread_amd import model_builder as MB
    out = io.StringIO()
    fn = MB.make_error_model if kind == 'error' else MB.make_qscore_model
    fn(namespace(case), output=io.StringIO(), engine=engine, stdout=out)
    return out.getvalue()


def test_loader_keeps_the_best_alignment_per_read_and_drops_short_and_poor_ones():
    from badread_amd.alignment import load_alignments
    al = load_alignments(os.path.join(FOLDER, 'aln.paf'), output=io.StringIO())
    names = [a.read_name for a in al]
    assert 25 <= len(names) == len(set(names)) < 48 and 'read_short' not in names and 'read_bad' not in names      # the 17 %-error reads fall below 80 % identity
    assert all(a.num_bases > 100 and a.percent_identity > 80.0 for a in al)
    minus = [a for a in al if a.strand == '-']
    assert minus and all(a.cigar_parts == [(int(n), t) for n, t in __import__('re').findall(r'(\d+)(\w)', a.cigar)][::-1] for a in minus)
    assert len(load_alignments(os.path.join(FOLDER, 'aln.paf'), 5, output=io.StringIO())) <= 5


def test_oracle_restatement_reproduces_the_reference_text():
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'oracle'))
    import model_builder_ref as R
    from badread_amd.alignment import load_alignments
    from badread_amd.misc import load_fasta, load_fastq, reverse_complement
    refs = load_fasta(os.path.join(FOLDER, 'ref.fasta'))[0]
    reads = load_fastq(os.path.join(FOLDER, 'reads.fastq'), output=io.StringIO())
    for entry in GOLDEN['error']:
        a = entry['args']
        al = load_alignments(os.path.join(FOLDER, 'aln.paf'), a['max_alignments'], output=io.StringIO())
        assert R.error_model_text(refs, reads, al, a['k_size'], a['max_alt'], reverse_complement) == entry['text'], a
    for entry in GOLDEN['qscore']:
        a = entry['args']
        al = load_alignments(os.path.join(FOLDER, 'aln.paf'), a['max_alignments'], output=io.StringIO())
        assert R.qscore_model_text(refs, reads, al, a['k_size'], a['max_del'], a['min_occur'], a['max_output'],
                                   reverse_complement) == entry['text'], a


@pytest.mark.parametrize('kind', ['error', 'qscore'])
def test_interpreted_kernels_reproduce_the_reference_text(kind):
    import emu_engine as EE
    eng = EE.EmuEngine(1 << 26)
    for entry in GOLDEN[kind]:
        assert run_product(kind, entry['args'], eng) == entry['text'], entry['args']


def test_windows_a_key_cannot_hold_are_counted_on_the_host():
    """The inputs hold 30-base insertions: read k-mers of more than 21 bases leave the 64-bit key (error model), and
    nothing may be lost or counted twice -- the text equality above covers it; here: the spill path really ran."""
    import emu_engine as EE
    from badread_amd import model_builder as MB
    from badread_amd.alignment import load_alignments
    from badread_amd.misc import load_fasta, load_fastq
    refs = load_fasta(os.path.join(FOLDER, 'ref.fasta'))[0]
    reads = load_fastq(os.path.join(FOLDER, 'reads.fastq'), output=io.StringIO())
    al = load_alignments(os.path.join(FOLDER, 'aln.paf'), output=io.StringIO())
    job = MB.Job(refs, reads, al)
    keys, counts, first, spill = EE.EmuEngine(1 << 26).model_count(0, job, 7, 0, 1, 18)
    assert len(spill) > 0 and int(counts.sum()) > 10000
    assert EE.EmuEngine(1 << 26).model_count(0, job, 7, 0, 1, 8) is None          # a 256-slot table is full: the caller grows it


def test_quality_characters_outside_the_phred_range_are_counted_like_the_reference(tmp_path):
    """A quality character below '!' (q < 0) or above chr(160) (q > 127) does not fit the 7 bits a key gives the score: k_mb_qscore
    hands such windows to the host with their size index in the spill word, and the host counts them with the reference's own
    arithmetic (qscore_model.py:137-138: ord(c) - 33, whatever it is).  Product (interpreted kernels) == the oracle's
    restatement == the reference itself where it is installed."""
    import shutil
    import emu_engine as EE
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'oracle'))
    import model_builder_ref as R
    from badread_amd.alignment import load_alignments
    from badread_amd.misc import load_fasta, load_fastq, reverse_complement
    folder = tmp_path / 'mb'
    shutil.copytree(FOLDER, folder)
    lines = open(folder / 'reads.fastq').read().split('\n')
    changed = 0
    for i in range(3, len(lines), 4):                # every quality line: a control character and a DEL every 97 / 131 bases
        q = list(lines[i])
        for j in range(40, len(q), 97):
            q[j] = '\x05'; changed += 1              # q = -28
        for j in range(71, len(q), 131):
            q[j] = '\x7f'                            # q = 94: inside the key's range, above the phred range
        lines[i] = ''.join(q)
    assert changed > 200
    open(folder / 'reads.fastq', 'w').write('\n'.join(lines))
    case = dict(GOLDEN['qscore'][0]['args'])
    ns = types.SimpleNamespace(reference=str(folder / 'ref.fasta'), reads=str(folder / 'reads.fastq'), alignment=str(folder / 'aln.paf'), **case)
    from badread_amd import model_builder as MB
    out = io.StringIO()
    MB.make_qscore_model(ns, output=io.StringIO(), engine=EE.EmuEngine(1 << 26), stdout=out)
    refs = load_fasta(ns.reference)[0]
    reads = load_fastq(ns.reads, output=io.StringIO())
    al = load_alignments(ns.alignment, case['max_alignments'], output=io.StringIO())
    want = R.qscore_model_text(refs, reads, al, case['k_size'], case['max_del'], case['min_occur'], case['max_output'], reverse_complement)
    assert out.getvalue() == want
    assert '-28:' in want and '94:' in want
    if os.path.isdir('/root/reference/badread'):      # the reference itself on the same files (CPU container only)
        code = ('import sys, types, io, contextlib; sys.path.insert(0, "/root/reference"); sys.path.insert(0, %r);'
                'from badread.qscore_model import make_qscore_model;'
                'ns = types.SimpleNamespace(**%r);'
                'make_qscore_model(ns, output=io.StringIO())') % (os.path.join(os.path.dirname(HERE), 'oracle', 'shim'), vars(ns))
        got = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
        assert got.returncode == 0, got.stderr[-2000:]
        assert got.stdout == want


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['error', 'qscore'])
def test_hip_kernels_reproduce_the_reference_text(kind):
    eng = H.hip_engine()
    for entry in GOLDEN[kind]:
        assert run_product(kind, entry['args'], eng) == entry['text'], entry['args']


@pytest.mark.gpu
def test_command_line_writes_the_model_to_stdout():
    repo = os.path.dirname(HERE)
    for cmd, entry, extra in (('error_model', GOLDEN['error'][1], ['--k_size', '4', '--max_alt', '3']),
                              ('qscore_model', GOLDEN['qscore'][2], ['--k_size', '5', '--max_del', '2', '--min_occur', '2', '--max_output', '20'])):
        r = subprocess.run([sys.executable, '-m', 'badread_amd', cmd, '--reference', os.path.join(FOLDER, 'ref.fasta'),
                            '--reads', os.path.join(FOLDER, 'reads.fastq'), '--alignment', os.path.join(FOLDER, 'aln.paf')] + extra,
                           cwd=repo, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert r.stdout == entry['text']
        assert 'Loading alignments' in r.stderr and 'Processing alignments' in r.stderr

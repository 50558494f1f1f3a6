"""
The native FASTA packer (libbrx_host.so, badread_amd/csrc/brx_fasta.cpp) against the
Python route that restates misc.load_fasta (/root/reference/badread/misc.py:122-153): identical packed words,
contig table, exception runs, alphabet and header attributes, on the golden reference and on awkward files
(gzip, CRLF / lone CR line ends, blank and indented lines, lower case, IUPAC symbols, interior blanks, duplicate
names, depth= spellings, a one-line 300 kb contig, sequence before the first header), plus the sidecar cache.
"""
import gzip
import os

import numpy as np
import pytest

from badread_amd.reference import PackedReference, host_library

HERE = os.path.dirname(os.path.abspath(__file__))


def same(a, b):
    assert a.names == b.names and a.lengths == b.lengths
    assert a.depths == b.depths and a.circular == b.circular
    assert a.hairpin_left == b.hairpin_left and a.hairpin_right == b.hairpin_right
    assert a.n_bases == b.n_bases
    assert np.array_equal(a.packed, b.packed)
    assert np.array_equal(a.contigs, b.contigs)
    assert np.array_equal(a.exceptions, b.exceptions)
    assert a.names_pool == b.names_pool
    assert np.array_equal(a.sym, b.sym) and np.array_equal(a.comp, b.comp)
    assert a.code_of == b.code_of


def check(path):
    nat, py = PackedReference.from_fasta(path, cache=False), PackedReference.from_fasta_python(path)
    same(nat, py)
    return nat


def test_golden_reference_file():
    pref = check(os.path.join(HERE, 'golden', 'small_ref.fasta'))
    assert len(pref.names) >= 3


AWKWARD = (
    '  \n'
    '>chrA some text Depth=2.5 CIRCULAR=TRUE\r\n'
    'acgtnnnnACGTRYKM\r\n'
    '\r\n'
    '   ACGT ACGT  \n'                  # indented, interior blank kept as a symbol, trailing blanks dropped
    '>chrB depth=abc hairpin_left=true depth=7\n'
    'NNNNNNNNNNNNNNNNNNNNACGT\n'
    '>chrC depth=1.2.3 hairpin_right=true\n'
    'ACGTSWBDHV.-?\n'
    '>chrA second copy wins but keeps the first position depth=.5\n'
    'TTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTNNNN\n'
    '>tail\n'
    'NNNNACGT'                            # no final newline; N run touching the previous contig's N run
)


def test_awkward_files(tmp_path):
    p = tmp_path / 'awkward.fasta'
    p.write_bytes(AWKWARD.encode())
    pref = check(str(p))
    assert pref.names == ['chrA', 'chrB', 'chrC', 'tail']
    assert pref.depths == {'chrA': 0.5, 'chrB': 7.0, 'chrC': 1.0, 'tail': 1.0}
    assert pref.circular['chrA'] is False and pref.hairpin_left['chrB'] and pref.hairpin_right['chrC']
    gz = tmp_path / 'awkward.fasta.gz'
    with gzip.open(gz, 'wb') as f:
        f.write(AWKWARD.encode())
    same(check(str(gz)), pref)
    cr = tmp_path / 'cr.fasta'
    cr.write_bytes(b'>old_mac\rACGT\rNNAC\r>two\rGG')
    assert check(str(cr)).lengths == [8, 2]


def test_sequence_before_the_first_header_joins_the_first_contig(tmp_path):
    p = tmp_path / 'orphan.fasta'
    p.write_bytes(b'ACGTN\nAC\n>first\nGGGG\n>second\nTT\n')
    pref = check(str(p))
    assert pref.lengths == [11, 2] and pref.decode(0, '+', 0, 11) == 'ACGTNACGGGG'


def test_long_single_line_contig_and_many_contigs(tmp_path):
    rng = np.random.default_rng(3)
    big = np.frombuffer(b'ACGTN', dtype=np.uint8)[rng.choice(5, 300001, p=[.24, .24, .24, .24, .04])].tobytes()
    p = tmp_path / 'big.fasta'
    with open(p, 'wb') as f:
        f.write(b'>one_line circular=true\n' + big + b'\n')
        for i in range(300):
            n = int(rng.integers(1, 200))
            f.write(b'>c%d depth=%d\n' % (i, i + 1) + np.frombuffer(b'ACGT', dtype=np.uint8)[rng.integers(0, 4, n)].tobytes() + b'\n')
    pref = check(str(p))
    assert pref.lengths[0] == 300001 and len(pref.names) == 301


def test_errors(tmp_path):
    with pytest.raises(ValueError):
        PackedReference.from_fasta(str(tmp_path / 'missing.fasta'), cache=False)
    empty = tmp_path / 'empty.fasta'
    empty.write_bytes(b'')
    with pytest.raises(ValueError):
        PackedReference.from_fasta(str(empty), cache=False)
    many = tmp_path / 'many.fasta'
    many.write_bytes(b'>x\nACGTNRYKMSWBDHVXZQJ\n')          # more than 16 symbols with complements
    with pytest.raises(ValueError):
        PackedReference.from_fasta(str(many), cache=False)


def test_sidecar_cache(tmp_path):
    p = tmp_path / 'ref.fasta'
    p.write_bytes(b'>a circular=true depth=3\nACGTNNNNACGTAC\n>b\nGGGGRRRR\n')
    first = PackedReference.from_fasta(str(p), cache=True)
    side = str(p) + '.brx2bit'
    assert os.path.isfile(side)
    again = PackedReference.from_fasta(str(p), cache=True)             # loaded from the sidecar
    same(first, again)
    # prove the second call really read the sidecar: corrupt the FASTA's content but keep size and mtime
    st = os.stat(p)
    p.write_bytes(b'>a circular=true depth=3\nTTTTNNNNACGTAC\n>b\nGGGGRRRR\n')
    os.utime(p, ns=(st.st_atime_ns, st.st_mtime_ns))
    same(PackedReference.from_fasta(str(p), cache=True), first)
    # a changed mtime invalidates it
    os.utime(p, ns=(st.st_atime_ns, st.st_mtime_ns + 10 ** 9))
    fresh = PackedReference.from_fasta(str(p), cache=True)
    assert fresh.decode(0, '+', 0, 4) == 'TTTT'


def test_library_exports_every_symbol_of_brx_host_h():
    import re
    text = open(os.path.join(os.path.dirname(HERE), 'include', 'brx_host.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    names = sorted(set(re.findall(r'\b(brx_(?:fasta|gzip)_[a-z0-9_]+)\s*\(', text)))
    assert {'brx_fasta_pack', 'brx_fasta_view_of', 'brx_fasta_free', 'brx_fasta_save', 'brx_fasta_load',
            'brx_gzip_bound', 'brx_gzip_parallel'} <= set(names)
    lib = host_library()
    for name in names:
        assert getattr(lib, name) is not None, name


def test_the_packed_words_outlive_the_reference_object(tmp_path):
    """The packed words are a VIEW of the native object (the mapped sidecar, the packer's vector): an engine that was handed
    `pref.packed` keeps reading them after the PackedReference is gone (tests/oracle_slice_worker.py builds its workload in an
    expression) -- the view must keep the mapping alive.  Found on the GPU box as a crash of the oracle workers."""
    import gc
    rng = np.random.default_rng(3)
    p = tmp_path / 'keep.fasta'
    p.write_text('>a\n' + ''.join(rng.choice(list('ACGT'), 100000)) + '\n>b circular=true\n' + ''.join(rng.choice(list('ACGTN'), 5000)) + '\n')
    for cache in (False, True, True):                     # packed, packed + saved, mapped from the sidecar
        pref = PackedReference.from_fasta(str(p), cache=cache)
        want = pref.packed.copy()
        view = pref.packed
        del pref
        gc.collect()
        junk = [np.zeros(1 << 20, dtype=np.uint8) for _ in range(8)]      # give freed memory a chance to be reused
        assert np.array_equal(view, want)
        del junk


def test_ranks_saving_the_same_sidecar_at_once_do_not_clobber_each_other(tmp_path):
    """on a first multi-rank run every rank packs and saves the sidecar at the same time.  The save goes through a
    temporary file of its own (mkstemp) and a rename, so whichever rename lands last the file is whole, and no temporary is
    left behind."""
    import threading
    p = tmp_path / 'ref.fasta'
    p.write_bytes(b'>a circular=true\n' + b'ACGTTGCA' * 40000 + b'\n>b\nGGGGRRRRNNNN\n')
    want = PackedReference.from_fasta(str(p), cache=False)
    errors = []

    def rank():
        try:
            same(PackedReference.from_fasta(str(p), cache=True), want)
        except BaseException as ex:                      # noqa: BLE001 -- reported by the assertion below
            errors.append(ex)
    threads = [threading.Thread(target=rank) for _ in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert sorted(os.listdir(tmp_path)) == ['ref.fasta', 'ref.fasta.brx2bit']       # no '.tmp' / '.XXXXXX' leftovers
    assert oct(os.stat(str(p) + '.brx2bit').st_mode & 0o777) == oct(0o644)
    same(PackedReference.from_fasta(str(p), cache=True), want)                       # and what is there loads


def test_a_file_without_a_single_base_has_an_empty_packed_array(tmp_path):
    """`packed` is a null pointer then; the view must not dereference it."""
    p = tmp_path / 'empty.fasta'
    p.write_bytes(b'>only_a_header\n')
    try:
        ref = PackedReference.from_fasta(str(p), cache=False)
    except ValueError:
        return                                            # refusing the file is as good
    assert ref.n_bases == 0 and len(ref.packed) <= 1 and not ref.packed.any()      # the packer keeps one zero word

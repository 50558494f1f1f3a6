"""
Host-side helpers with the same names, arguments and error behaviour as the reference's
badread/misc.py, so that the reference's own tests read unchanged against this package.
None of this is on the accelerated path; it is the plumbing either side of it.

Reference lines mirrored: get_compression_type misc.py:26-45, get_open_func :48-52,
reverse_complement :56-71, load_fasta :122-153, random helpers :156-182, float_to_str :193-202,
identity_from_edlib_cigar :228-240, captured_output :243-251.
"""
import collections
import contextlib
import gzip
import io
import random
import re
import sys

_MAGIC = (('gz', b'\x1f\x8b\x08'), ('bz2', b'\x42\x5a\x68'), ('zip', b'\x50\x4b\x03\x04'))


def get_compression_type(filename):
    """Sniff the first bytes; bzip2 and zip are refused with the reference's messages."""
    with open(str(filename), 'rb') as handle:
        head = handle.read(max(len(m) for _, m in _MAGIC))
    kind = 'plain'
    for name, magic in _MAGIC:
        if head.startswith(magic):
            kind = name
    if kind == 'bz2':
        sys.exit('Error: cannot use bzip2 format - use gzip instead')
    if kind == 'zip':
        sys.exit('Error: cannot use zip format - use gzip instead')
    return kind


def get_open_func(filename):
    return gzip.open if get_compression_type(filename) == 'gz' else open


_COMPLEMENT = dict(zip('ATGCatgcRYSWKMBVDHNryswkmbvdhn.-?',
                       'TACGtacgYRSWMKVBHDNyrswmkvbhdn.-?'))
REV_COMP_DICT = _COMPLEMENT


def complement_base(base):
    return _COMPLEMENT.get(base, 'N')


def reverse_complement(seq):
    return ''.join(_COMPLEMENT.get(b, 'N') for b in reversed(seq))


_DEPTH_RE = re.compile(r'depth=([\d.]+)')


def load_fasta(filename):
    """
    Returns (seqs, depths, circular, hairpin_left, hairpin_right) keyed by the first header token.
    Sequences are upper-cased (IUPAC/N kept); depth=X, circular=true, hairpin_left/right=true are
    read from the lower-cased header exactly as the reference does (misc.py:135-147).
    """
    seqs = collections.OrderedDict()
    depths, circular, hp_left, hp_right = {}, {}, {}, {}
    header, chunks = '', []

    def flush():
        if header:
            seqs[header.split()[0]] = ''.join(chunks).upper()

    with get_open_func(filename)(filename, 'rt') as fasta:
        for raw in fasta:
            line = raw.strip()
            if not line:
                continue
            if line[0] != '>':
                chunks.append(line)
                continue
            flush()
            if header:                    # lines before the first header stay in the list: they start the first contig
                chunks = []
            header = line[1:]
            short = header.split()[0]
            lowered = header.lower()
            depth = 1.0
            if 'depth=' in lowered:
                found = _DEPTH_RE.search(lowered)
                try:
                    depth = float(found.group(1))
                except (ValueError, AttributeError):
                    depth = 1.0
            depths[short] = depth
            circular[short] = 'circular=true' in lowered
            hp_left[short] = 'hairpin_left=true' in lowered
            hp_right[short] = 'hairpin_right=true' in lowered
        flush()
    return seqs, depths, circular, hp_left, hp_right


RANDOM_SEQ_DICT = {0: 'A', 1: 'C', 2: 'G', 3: 'T'}


def load_fastq(filename, output=sys.stderr, dot_interval=1000):
    """{read name: (bases upper-cased, qualities)} with the record rule of the reference's loader (misc.py:97-119): a line
    whose first non-blank byte is '@' opens a record and the three lines behind it belong to it whatever they hold (a
    quality string may itself start with '@').  Streamed record by record -- the read sets the model builders take are
    gigabytes -- and a dot per `dot_interval` RECORDS (a repeated name overwrites its entry but still counts)."""
    if get_sequence_file_type(filename) != 'FASTQ':
        sys.exit('Error: {} is not FASTQ format'.format(filename))
    print('Loading reads', end='', file=output, flush=True)
    reads, records = {}, 0
    with get_open_func(filename)(filename, 'rb') as handle:
        rows = iter(handle)
        for row in rows:
            head = row.strip()
            if head[:1] != b'@':
                continue
            body = [next(rows, None) for _ in range(3)]
            if body[2] is None:
                raise EOFError('{}: the last record is cut short'.format(filename))
            reads[head[1:].split()[0].decode()] = (body[0].strip().upper().decode(), body[2].strip().decode())
            records += 1
            if records % dot_interval == 0:
                print('.', end='', file=output, flush=True)
    print('', file=output, flush=True)
    return reads


def get_random_base():
    return 'ACGT'[random.randint(0, 3)]


def get_random_different_base(b):
    while True:
        candidate = get_random_base()
        if candidate != b:
            return candidate


def get_random_sequence(length):
    return ''.join(get_random_base() for _ in range(length))


def random_chance(chance):
    assert 0.0 <= chance <= 1.0
    return random.random() < chance


def float_to_str(v, decimals=1, trim_zeros=False):
    if float(int(v)) == v:
        return str(int(v))
    text = '%.*f' % (decimals, v)
    if trim_zeros:
        text = text.rstrip('0')
    return text


def print_in_two_columns(l1p1, l2p1, l3p1, l1p2, l2p2, l3p2, output, space_between=6):
    width = max(len(l1p1), len(l2p1), len(l3p1)) + space_between
    for left, right in ((l1p1, l1p2), (l2p1, l2p2), (l3p1, l3p2)):
        print(left.ljust(width) + right, file=output)


def str_is_int(s):
    try:
        int(s)
    except ValueError:
        return False
    return True


def str_is_dna_sequence(s):
    return set(s) <= set('ACGT')


def only_acgt(s):
    """True when every character is an upper-case A, C, G or T (misc.py:205-206); the empty string passes."""
    return not set(s) - set('ACGT')


def get_sequence_file_type(filename):
    """'FASTA' or 'FASTQ' from the first character of a plain or gzipped file (misc.py:74-94)."""
    import os
    if not os.path.isfile(filename):
        sys.exit('Error: could not find {}'.format(filename))
    with get_open_func(filename)(filename, 'rt') as handle:
        try:
            first = handle.read(1)
        except UnicodeDecodeError:
            first = ''
    kinds = {'>': 'FASTA', '@': 'FASTQ'}
    if first not in kinds:
        raise ValueError('File is neither FASTA or FASTQ')
    return kinds[first]


_CIGAR_RE = re.compile(r'(\d+)([IDX=])')


def identity_from_edlib_cigar(cigar):
    """matches / alignment columns; 0.0 for an empty cigar (misc.py:228-240)."""
    matches = columns = 0
    for size, op in _CIGAR_RE.findall(cigar):
        columns += int(size)
        if op == '=':
            matches += int(size)
    return matches / columns if columns else 0.0


@contextlib.contextmanager
def captured_output():
    new_out, new_err = io.StringIO(), io.StringIO()
    old = sys.stdout, sys.stderr
    try:
        sys.stdout, sys.stderr = new_out, new_err
        yield sys.stdout, sys.stderr
    finally:
        sys.stdout, sys.stderr = old

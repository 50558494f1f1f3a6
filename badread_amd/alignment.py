"""
PAF alignments for the model builders: the reference's `Alignment` / `load_alignments`
(/root/reference/badread/alignment.py:24-98) -- same fields, same filters (best alignment per read by AS:i, more than
100 aligned bases, identity above 80 %), same messages and exits.  CIGAR parts are kept as (length, letter) pairs,
reversed for '-' strand alignments (alignment.py:61-64).
"""
import re
import sys

from .misc import get_open_func

_PART = re.compile(r'(\d+)(\w)')


class Alignment(object):

    def __init__(self, paf_line):
        fields = paf_line.strip().split('\t')
        if len(fields) < 11:
            sys.exit('Error: alignment file does not seem to be in PAF format')
        self.read_name = fields[0]
        self.read_start, self.read_end = int(fields[2]), int(fields[3])
        self.strand = fields[4]
        self.ref_name = fields[5]
        self.ref_start, self.ref_end = int(fields[7]), int(fields[8])
        self.matching_bases, self.num_bases = int(fields[9]), int(fields[10])
        self.percent_identity = 100.0 * self.matching_bases / self.num_bases
        self.cigar = self.alignment_score = None
        for field in fields:
            if field.startswith('cg:Z:'):
                self.cigar = field[5:]
            if field.startswith('AS:i:'):
                self.alignment_score = int(field[5:])
        if self.cigar is None:
            sys.exit('Error: no CIGAR string found')
        if self.alignment_score is None:
            sys.exit('Error: no alignment score')
        self.cigar_parts = [(int(n), letter) for n, letter in _PART.findall(self.cigar)]
        self.max_indel = max([n for n, letter in self.cigar_parts if letter in 'ID'], default=0)
        if self.strand == '-':
            self.cigar_parts = self.cigar_parts[::-1]

    def __repr__(self):
        return (f'{self.read_name}:{self.read_start}-{self.read_end}({self.strand}),'
                f'{self.ref_name}:{self.ref_start}-{self.ref_end}({self.percent_identity:.3f}%)')


def _paf_records(filename, limit):
    """Alignment objects of the first `limit` PAF lines (all of them when limit is None)."""
    with get_open_func(filename)(filename, 'rt') as paf:
        for n, line in enumerate(paf):
            if limit is not None and limit > 0 and n >= limit:
                return
            yield Alignment(line)


def load_alignments(filename, max_alignments=None, output=sys.stderr, dot_interval=1000):
    """One alignment per read -- its highest AS:i, the LAST of equals (the reference sorts a read's alignments by score and
    takes the final one, alignment.py:89) -- in order of the reads' first appearance, filtered like alignment.py:90.  The PAF is
    streamed: a read keeps one candidate at a time instead of the list of all its alignments."""
    def progress(count):
        if count % dot_interval == 0:
            print('.', end='', file=output, flush=True)

    print('Loading alignments', end='', file=output, flush=True)
    candidate = {}                                        # read name -> best so far; dicts keep first-insertion order
    for seen, a in enumerate(_paf_records(filename, max_alignments), 1):
        held = candidate.get(a.read_name)
        if held is None or a.alignment_score >= held.alignment_score:
            candidate[a.read_name] = a
        progress(seen)
    print('', file=output, flush=True)
    print('Choosing best alignment per read', end='', file=output, flush=True)
    kept = []
    for a in candidate.values():
        if a.num_bases <= 100 or a.percent_identity <= 80.0:
            continue
        kept.append(a)
        progress(len(kept))
    print('', file=output, flush=True)
    return kept

"""
`badread error_model` and `badread qscore_model` on the GPU: the reference's
make_error_model (/root/reference/badread/error_model.py:31-83) and make_qscore_model (qscore_model.py:78-162) with the
per-alignment Python loops replaced by brx_model_count (include/brx.h; kernels in csrc/brx_model.h).

Host side, as the reference: load_fasta, load_fastq, load_alignments (best alignment per read, > 100 bases, > 80 %),
check_alignment_matches_read_and_refs, the slices and the strand flip of every alignment.  What goes to the device: the
slices as bytes and the CIGAR parts with the prefix sums of their column / read / reference offsets.  What comes back: a
hash table of (key, count, earliest window) and the few windows a 64-bit key cannot hold, which are counted here.  The
text written to stdout is the reference's, byte for byte: alternatives and cigars with equal counts keep the order in
which the reference's loops would first have met them (Python dict order + stable sort), reproduced from the earliest
window recorded per key.
"""
import collections
import itertools
import sys

import numpy as np

from .alignment import load_alignments
from .misc import float_to_str, load_fasta, load_fastq, only_acgt, reverse_complement

_TYPE = {'M': 0, 'I': 1, 'D': 2}
EMPTY = np.uint64(0xFFFFFFFFFFFFFFFF)


def check_alignment_matches_read_and_refs(a, reads, refs):
    if a.read_name not in reads:
        sys.exit(f'\nError: could not find read {a.read_name}\nare you sure your read file and alignment file match?')
    if a.ref_name not in refs:
        sys.exit(f'\nError: could not find reference {a.ref_name}\nare you sure your reference file and alignment file match?')


class Job(object):
    """The alignments as flat arrays (what brx_model_job points at) + the per-alignment strings for the spilled windows."""

    def __init__(self, refs, reads, alignments):
        seqs, quals, refsl = [], [], []
        ptype, plen, pcol, pread, pref = [], [], [], [], []
        a_part, a_col = [0], [0]
        col = rpos = fpos = 0
        self.slices = []
        for a in alignments:
            check_alignment_matches_read_and_refs(a, reads, refs)
            read_seq, read_qual = (x[a.read_start:a.read_end] for x in reads[a.read_name])
            ref_seq = refs[a.ref_name][a.ref_start:a.ref_end]
            if a.strand == '-':
                ref_seq = reverse_complement(ref_seq)
            parts = [(n, t) for n, t in a.cigar_parts if t in _TYPE]       # align_sequences ignores every other letter
            used_read = sum(n for n, t in parts if t != 'D')
            used_ref = sum(n for n, t in parts if t != 'I')
            if used_read > len(read_seq) or used_ref > len(ref_seq) or len(read_qual) < len(read_seq):
                sys.exit(f'Error: the CIGAR of {a!r} runs past its read or reference range')
            r0, f0 = rpos, fpos
            for n, t in parts:
                ptype.append(_TYPE[t]); plen.append(n); pcol.append(col); pread.append(rpos); pref.append(fpos)
                col += n
                if t != 'D':
                    rpos += n
                if t != 'I':
                    fpos += n
            rpos, fpos = r0 + len(read_seq), f0 + len(ref_seq)
            seqs.append(read_seq); quals.append(read_qual[:len(read_seq)]); refsl.append(ref_seq)
            a_part.append(len(ptype)); a_col.append(col)
            self.slices.append((read_seq, read_qual, ref_seq, parts))
        enc = lambda parts_: np.frombuffer(''.join(parts_).encode('latin-1') or b'\0', dtype=np.uint8)
        self.seq, self.qual, self.ref = enc(seqs), enc(quals), enc(refsl)
        self.part_type = np.array(ptype or [0], dtype=np.uint8)
        self.part_len = np.array(plen or [0], dtype=np.uint32)
        self.part_col, self.part_read, self.part_ref = (np.array(x or [0], dtype=np.uint64) for x in (pcol, pread, pref))
        self.align_part_off = np.array(a_part, dtype=np.uint64)
        self.align_col_off = np.array(a_col, dtype=np.uint64)
        self.n_align, self.n_cols = len(alignments), col

    def gapped(self, a, gap):
        """align_sequences (alignment.py:101-132) for alignment a: gapped read, quality, reference strings."""
        read_seq, read_qual, ref_seq, parts = self.slices[a]
        read, qual, ref = [], [], []
        rp = fp = 0
        for n, t in parts:
            read.append(read_seq[rp:rp + n] if t != 'D' else gap * n)
            qual.append(read_qual[rp:rp + n] if t != 'D' else gap * n)
            ref.append(ref_seq[fp:fp + n] if t != 'I' else gap * n)
            if t != 'D':
                rp += n
            if t != 'I':
                fp += n
        return ''.join(read), ''.join(qual), ''.join(ref)


def _table(engine, kind, job, k, max_del, n_ksizes):
    bits = 16
    need = max(job.n_cols * (n_ksizes if kind else 1) // 2, 1)
    while (1 << bits) < 4 * need and bits < 28:
        bits += 1
    while True:
        res = engine.model_count(kind, job, k, max_del, n_ksizes, bits)
        if res is not None:
            return res
        bits += 2                                   # table full: the engine returned None
        if bits > 30:
            sys.exit('Error: model builder hash table does not fit')


def make_error_model(args, output=sys.stderr, dot_interval=1000, engine=None, stdout=None):
    stdout = stdout or sys.stdout
    refs, _, _, _, _ = load_fasta(args.reference)
    reads = load_fastq(args.reads, output=output)
    alignments = load_alignments(args.alignment, args.max_alignments, output=output)
    if len(alignments) == 0:
        sys.exit('Error: no usable alignments')
    k = args.k_size
    if not 2 <= k <= 8:
        sys.exit('Error: the GPU error-model builder supports --k_size 2 to 8')
    if engine is None:
        from .engine import default_engine
        engine = default_engine()
    print('Processing alignments', end='', file=output, flush=True)
    job = Job(refs, reads, alignments)
    keys, counts, first, spill = _table(engine, 0, job, k, 0, 1)
    print('.' * (len(alignments) // dot_interval), file=output, flush=True)
    # (ref k-mer, read k-mer) -> [count, earliest window]
    found = {}
    mask_k = (1 << (2 * k)) - 1
    for key, n, rank in zip(keys.tolist(), counts.tolist(), first.tolist()):
        refk = key & mask_k
        length = (key >> (2 * k)) & 31
        bits = key >> (2 * k + 5)
        ref_kmer = ''.join('ACGT'[(refk >> (2 * (k - 1 - i))) & 3] for i in range(k))
        read_kmer = ''.join('ACGT'[(bits >> (2 * i)) & 3] for i in range(length))
        found[(ref_kmer, read_kmer)] = [n, rank]
    for word in sorted(set(spill.tolist())):        # read k-mers of more than 21 bases: counted here
        a, start = word >> 32, word & 0xFFFFFFFF
        read_g, _, ref_g = job.gapped(a, '-')
        pair = _error_window(read_g, ref_g, start, k)
        if pair is not None:
            entry = found.setdefault(pair, [0, word])
            entry[0] += 1
            entry[1] = min(entry[1], word)
    by_kmer = collections.defaultdict(list)
    for (ref_kmer, read_kmer), (n, rank) in found.items():
        by_kmer[ref_kmer].append((rank, read_kmer, n))
    for kmer in (''.join(x) for x in itertools.product('ACGT', repeat=k)):
        alts = sorted(by_kmer.get(kmer, ()))        # first-insertion order of the reference's dictionary
        if not alts:
            continue
        total = sum(n for _, _, n in alts)
        same = sum(n for _, alt, n in alts if alt == kmer)
        print(f'{kmer},{same / total:.6f}', end=';', file=stdout)
        fracs = sorted(((alt, n / total) for _, alt, n in alts if alt != kmer), reverse=True, key=lambda x: x[1])
        for alt, frac in fracs[:args.max_alt]:
            print(f'{alt},{frac:.6f}', end=';', file=stdout)
        print(file=stdout)


def _error_window(read_g, ref_g, start, k):
    """One window of error_model.py:47-62 on the gapped strings: (ref k-mer, read k-mer) or None."""
    end = start
    while len(ref_g[start:end].replace('-', '')) < k:
        end += 1
        if end > len(ref_g):
            return None
    ref_kmer, read_kmer = ref_g[start:end].replace('-', ''), read_g[start:end].replace('-', '')
    if len(read_kmer) > 1 and ref_kmer[0] == read_kmer[0] and ref_kmer[-1] == read_kmer[-1] and \
            only_acgt(ref_kmer) and only_acgt(read_kmer):
        return ref_kmer, read_kmer
    return None


def make_qscore_model(args, output=sys.stderr, dot_interval=1000, engine=None, stdout=None):
    stdout = stdout or sys.stdout
    refs, _, _, _, _ = load_fasta(args.reference)
    reads = load_fastq(args.reads, output=output)
    alignments = load_alignments(args.alignment, args.max_alignments, output=output)
    if len(alignments) == 0:
        sys.exit('Error: no usable alignments')
    assert args.k_size % 2 == 1
    n_ksizes = (args.k_size + 1) // 2
    if n_ksizes > 16 or args.max_del > 15:
        sys.exit('Error: the GPU qscore-model builder supports --k_size up to 31 and --max_del up to 15')
    if engine is None:
        from .engine import default_engine
        engine = default_engine()
    print('Processing alignments', end='', file=output, flush=True)
    job = Job(refs, reads, alignments)
    keys, counts, first, spill = _table(engine, 1, job, args.k_size, args.max_del, n_ksizes)
    print('.' * (len(alignments) // dot_interval), file=output, flush=True)
    per_cigar = {}                                   # cigar -> [earliest window, {q: count}]
    for key, n, rank in zip(keys.tolist(), counts.tolist(), first.tolist()):
        q, ki, ops = key & 127, (key >> 7) & 15, key >> 11
        cigar, shift = [], 0
        for i in range(2 * ki + 1):
            if i:
                cigar.append('D' * ((ops >> shift) & 15)); shift += 4
            cigar.append('=XI'[(ops >> shift) & 3]); shift += 2
        entry = per_cigar.setdefault(''.join(cigar), [rank, collections.Counter()])
        entry[0] = min(entry[0], rank)
        entry[1][q] += n
    for word in sorted(set(spill.tolist())):         # windows a key cannot hold (k_mb_qscore names each: alignment, size index, start)
        a, ki, start = word >> 36, (word >> 32) & 15, word & 0xFFFFFFFF
        read_g, qual_g, ref_g = job.gapped(a, ' ')
        hit = _qscore_window(read_g, qual_g, ref_g, start, 2 * ki + 1, args.max_del)
        if hit is None:
            continue
        rank = (a << 40) | (ki << 36) | start
        entry = per_cigar.setdefault(hit[0], [rank, collections.Counter()])
        entry[0] = min(entry[0], rank)
        entry[1][hit[1]] += 1
    overall = collections.Counter()
    for cigar, (_, qs) in per_cigar.items():
        if len(cigar.replace('D', '')) == 1:         # every window of one read base (qscore_model.py:141-142)
            overall.update(qs)
    print_qscore_fractions('overall', overall, 0, stdout)
    ordered = sorted(per_cigar, key=lambda c: per_cigar[c][0])                       # first-insertion order
    ordered = sorted(ordered, reverse=True, key=lambda c: sum(per_cigar[c][1].values()))
    for i, cigar in enumerate(ordered, 1):
        print_qscore_fractions(cigar, per_cigar[cigar][1], args.min_occur, stdout)
        if i >= args.max_output:
            break


def _qscore_window(read_g, qual_g, ref_g, start, k_size, max_del):
    """One window of qscore_model.py:104-143 on the gapped strings: (cigar, qscore) or None."""
    import re
    end = start
    while len(read_g[start:end].replace(' ', '')) < k_size:
        end += 1
        if end > len(read_g):
            return None
    cigar = ''.join('=' if r == f else 'D' if r == ' ' else 'I' if f == ' ' else 'X'
                    for r, f in zip(read_g[start:end], ref_g[start:end]))
    cigar = re.sub('D{' + str(max_del) + ',}', 'D' * max_del, cigar)
    qual = qual_g[start:end].replace(' ', '')
    return cigar, ord(qual[(k_size - 1) // 2]) - 33


def print_qscore_fractions(cigar, qscores, min_occur, stdout):
    total = sum(qscores.values())
    if total < min_occur:
        return
    print(f'{cigar};{total};', end='', file=stdout)
    for q in sorted(qscores.keys()):
        print(f'{q}:{float_to_str(qscores[q] / total, decimals=6, trim_zeros=True)},', end='', file=stdout)
    print(file=stdout)

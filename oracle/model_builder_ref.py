"""
oracle/model_builder_ref.py -- TEST INFRASTRUCTURE ONLY.

Plain-Python restatement of the counting loops of the reference's model builders, the checker of the GPU builders
(badread_amd/model_builder.py):
    error_model_text    /root/reference/badread/error_model.py:31-83    (make_error_model)
    qscore_model_text   /root/reference/badread/qscore_model.py:78-175  (make_qscore_model, print_qscore_fractions)
    gapped              /root/reference/badread/alignment.py:101-132    (align_sequences)
Pinned against outputs of the reference itself: tests/golden/model_builder.json (tools/make_golden.py runs the reference's
own functions on the committed inputs).  Inputs are what the product's loaders return (refs dict, reads dict, Alignment
list); nothing here is imported by badread_amd/.
"""
import collections
import itertools
import re


def gapped(read_seq, read_qual, ref_seq, parts, gap):
    read, qual, ref = [], [], []
    rp = fp = 0
    for n, t in parts:
        if t == 'M':
            read.append(read_seq[rp:rp + n]); qual.append(read_qual[rp:rp + n]); ref.append(ref_seq[fp:fp + n])
            rp += n; fp += n
        elif t == 'I':
            read.append(read_seq[rp:rp + n]); qual.append(read_qual[rp:rp + n]); ref.append(gap * n)
            rp += n
        elif t == 'D':
            read.append(gap * n); qual.append(gap * n); ref.append(ref_seq[fp:fp + n])
            fp += n
    return ''.join(read), ''.join(qual), ''.join(ref)


def _slices(a, refs, reads, revcomp):
    read_seq, read_qual = (x[a.read_start:a.read_end] for x in reads[a.read_name])
    ref_seq = refs[a.ref_name][a.ref_start:a.ref_end]
    return read_seq, read_qual, (revcomp(ref_seq) if a.strand == '-' else ref_seq)


def error_model_text(refs, reads, alignments, k, max_alt, revcomp):
    table = {''.join(x): collections.defaultdict(int) for x in itertools.product('ACGT', repeat=k)}
    for a in alignments:
        read_g, _, ref_g = gapped(*_slices(a, refs, reads, revcomp), a.cigar_parts, '-')
        start = end = 0
        while end <= len(ref_g):
            ref_kmer = ref_g[start:end].replace('-', '')
            if len(ref_kmer) < k:
                end += 1
                continue
            read_kmer = read_g[start:end].replace('-', '')
            if len(read_kmer) > 1 and ref_kmer[0] == read_kmer[0] and ref_kmer[-1] == read_kmer[-1] and \
                    not (set(ref_kmer) | set(read_kmer)) - set('ACGT'):
                table[ref_kmer][read_kmer] += 1
            start += 1
            while ref_g[start] == '-':
                start += 1
            end += 1
    lines = []
    for kmer, alts in table.items():
        if not alts:
            continue
        total = sum(alts.values())
        fields = [f'{kmer},{alts.get(kmer, 0) / total:.6f}']
        others = sorted(((alt, n / total) for alt, n in alts.items() if alt != kmer), reverse=True, key=lambda x: x[1])
        fields += [f'{alt},{frac:.6f}' for alt, frac in others[:max_alt]]
        lines.append(';'.join(fields) + ';\n')
    return ''.join(lines)


def _fraction(v):
    if float(int(v)) == v:
        return str(int(v))
    return ('%.6f' % v).rstrip('0')


def _qscore_line(cigar, qs, min_occur):
    total = sum(qs.values())
    if total < min_occur:
        return ''
    return f'{cigar};{total};' + ''.join(f'{q}:{_fraction(qs[q] / total)},' for q in sorted(qs)) + '\n'


def qscore_model_text(refs, reads, alignments, k_size, max_del, min_occur, max_output, revcomp):
    overall = collections.defaultdict(int)
    per_cigar = collections.defaultdict(lambda: collections.defaultdict(int))
    long_run = re.compile('D{' + str(max_del) + ',}')
    for a in alignments:
        read_g, qual_g, ref_g = gapped(*_slices(a, refs, reads, revcomp), a.cigar_parts, ' ')
        for ks in range(1, k_size + 2, 2):
            start = end = 0
            while end <= len(read_g):
                window = read_g[start:end]
                if len(window.replace(' ', '')) < ks:
                    end += 1
                    continue
                quals = qual_g[start:end].replace(' ', '')
                cigar = ''.join('=' if r == f else 'D' if r == ' ' else 'I' if f == ' ' else 'X'
                                for r, f in zip(window, ref_g[start:end]))
                cigar = long_run.sub('D' * max_del, cigar)
                q = ord(quals[(ks - 1) // 2]) - 33
                if ks == 1:
                    overall[q] += 1
                per_cigar[cigar][q] += 1
                start += 1
                if start >= len(read_g):
                    break
                while read_g[start] == ' ':
                    start += 1
                end += 1
    out = [_qscore_line('overall', overall, 0)]
    ranked = sorted(per_cigar, reverse=True, key=lambda c: sum(per_cigar[c].values()))
    for i, cigar in enumerate(ranked, 1):
        out.append(_qscore_line(cigar, per_cigar[cigar], min_occur))
        if i >= max_output:
            break
    return ''.join(out)

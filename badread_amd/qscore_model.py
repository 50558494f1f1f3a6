"""
QScoreModel: same constructor, attributes and lookup rule as the reference's class
(/root/reference/badread/qscore_model.py:178-287) plus `tables()`, which turns the
{cigar string -> score distribution} dict into an open-addressing hash of packed cigar keys that
the HIP qscore kernel probes (layout: brx_qscore_model in include/brx.h).

Also here: the pure helpers the reference keeps in this module (align_sequences_from_edlib_cigar
:290-311, uniform_dist_scores_and_probs :314-318, qscore <-> char <-> error probability :321-334).
get_qscores itself (:32-75) is the accelerated path and lives in badread_amd/simulate.py.
"""
import os
import random
import re
import sys

import numpy as np

from . import settings
from .error_model import find_builtin_file, packaged_cache
from .misc import get_open_func

BUILTIN_QSCORE_MODELS = ('nanopore2018', 'nanopore2020', 'nanopore2023', 'pacbio2016', 'pacbio2021')
_OP_CODE = {'=': 0, 'X': 1, 'I': 2}
_HASH_MULT = 0x9E3779B97F4A7C15
_MASK64 = (1 << 64) - 1


def cigar_key(cigar, gap_bits):
    """
    Pack a window cigar (non-D ops separated by D-runs) exactly as the kernels do: op i in 2 bits,
    the D-run between op i and op i+1 in gap_bits bits (saturating to all-ones), window length in
    the top byte.  Returns None for cigars no window can produce (leading/trailing D, empty).
    """
    if not cigar or cigar[0] == 'D' or cigar[-1] == 'D':
        return None
    max_run = (1 << gap_bits) - 1
    key, shift, run, idx = 0, 0, 0, 0
    for ch in cigar:
        if ch == 'D':
            run += 1
            continue
        if idx > 0:
            key |= min(run, max_run) << shift
            shift += gap_bits
        key |= _OP_CODE[ch] << shift
        shift += 2
        run = 0
        idx += 1
    if shift > 56:
        return None
    return key | (idx << 56)


class QScoreModel(object):

    def __init__(self, model_type_or_filename, output=sys.stderr):
        self.scores, self.probabilities = {}, {}
        self.kmer_size = 1
        self.type = None
        self._tables = None
        name = model_type_or_filename
        if name == 'random':
            self.set_up_random_model(output)
        elif name == 'ideal':
            self.set_up_ideal_model(output)
        elif name in BUILTIN_QSCORE_MODELS:
            cache = packaged_cache(name, 'qscore')
            if cache:
                self._load_npz(cache, name, output)
            else:
                path = find_builtin_file('qscore_models', name)
                if path is None:
                    sys.exit(f'Error: could not find the built-in qscore model {name}; set '
                             f'BADREAD_AMD_MODEL_DIR to a directory holding qscore_models/{name}.gz')
                self.load_from_file(path, output)
        else:
            self.load_from_file(name, output)
        # get_qscore's fallback ends at these three 1-op cigars (qscore_model.py:203-207)
        assert '=' in self.scores
        assert 'X' in self.scores
        assert 'I' in self.scores

    def set_up_random_model(self, output):
        print('\nUsing a random qscore model', file=output)
        self.type = 'random'
        self.kmer_size = 1
        for c in '=XI':
            self.scores[c], self.probabilities[c] = uniform_dist_scores_and_probs(
                settings.RANDOM_QSCORE_MIN, settings.RANDOM_QSCORE_MAX)

    def set_up_ideal_model(self, output):
        print('\nUsing an ideal qscore model', file=output)
        self.type = 'ideal'
        self.kmer_size = 9
        ranks = [('X', 1), ('I', 1), ('=', 2), ('===', 3), ('=====', 4), ('=======', 5), ('=========', 6)]
        for cigar, rank in ranks:
            lo = getattr(settings, f'IDEAL_QSCORE_RANK_{rank}_MIN')
            hi = getattr(settings, f'IDEAL_QSCORE_RANK_{rank}_MAX')
            self.scores[cigar], self.probabilities[cigar] = uniform_dist_scores_and_probs(lo, hi)

    def load_from_file(self, filename, output=sys.stderr):
        print('\nLoading qscore model from {}'.format(filename), file=output)
        self.type = 'model'
        count = 0
        with get_open_func(filename)(filename, 'rt') as model_file:
            for line in model_file:
                parts = line.strip().split(';')
                try:
                    if parts[0] == 'overall':
                        continue
                    cigar = parts[0]
                    k = len(cigar.replace('D', ''))
                    self.kmer_size = max(self.kmer_size, k)
                    pairs = [x.split(':') for x in parts[2].split(',') if x]
                    self.scores[cigar] = [int(x[0]) for x in pairs]
                    self.probabilities[cigar] = [float(x[1]) for x in pairs]
                    count += 1
                except (IndexError, ValueError):
                    sys.exit(f'Error: {filename} does not seem to be a valid qscore model file')
        print(f'\r  done: loaded qscore distributions for {count} alignments', file=output)

    def get_qscore(self, cigar):
        """Host-side lookup + sample with the reference's centre-preserving trim (qscore_model.py:273-287)."""
        while True:
            assert len(cigar.replace('D', '')) % 2 == 1
            if cigar in self.scores:
                q = random.choices(self.scores[cigar], weights=self.probabilities[cigar])[0]
                return qscore_val_to_char(q)
            cigar = cigar[1:-1].strip('D')

    # ------------------------------------------------------------------ flattened device tables
    def tables(self):
        if self._tables is not None:
            return self._tables
        k = self.kmer_size
        max_run = 0
        for cigar in self.scores:
            for run in re.findall(r'D+', cigar):
                max_run = max(max_run, len(run))
        # the D-run between two ops is stored in gap_bits bits, the all-ones code being reserved for "longer than any
        # row of the table": every run of the model must stay below it, or rows would silently drop out of the table
        if max_run > 14:
            sys.exit(f'Error: qscore model holds a cigar with a run of {max_run} deletions; the HIP path encodes runs of up '
                     f'to 14 (the built-in models stop at 6)')
        gap_bits = 3 if max_run <= 6 else 4
        if 2 * k + gap_bits * (k - 1) > 56:
            sys.exit(f'Error: qscore model with {k}-op windows is too wide for the HIP path')
        rows = []
        for cigar in self.scores:
            key = cigar_key(cigar, gap_bits)
            if key is None:
                continue
            if any(len(run) >= (1 << gap_bits) - 1 for run in re.findall(r'D+', cigar)):
                continue                       # unreachable: the saturated code is reserved for misses
            if len(self.scores[cigar]) == 0:
                continue
            rows.append((key, cigar))
        hash_size = 16
        while hash_size < 2 * len(rows):
            hash_size *= 2
        hash_key = np.full(hash_size, _MASK64, dtype=np.uint64)
        hash_row = np.zeros(hash_size, dtype=np.uint32)
        row_off = np.zeros(len(rows) + 1, dtype=np.uint32)
        thr, score = [], []
        for r, (key, cigar) in enumerate(rows):
            slot = (((key * _HASH_MULT) & _MASK64) >> 32) & (hash_size - 1)
            while int(hash_key[slot]) != _MASK64:
                slot = (slot + 1) & (hash_size - 1)
            hash_key[slot] = key
            hash_row[slot] = r
            probs = self.probabilities[cigar]
            acc, cum = 0.0, []
            for p in probs:
                acc = acc + p
                cum.append(acc)
            total = acc if acc > 0.0 else 1.0
            for q, c in zip(self.scores[cigar], cum):
                if not 0 <= q <= 93:
                    sys.exit('Error: qscore model holds a score outside 0-93')
                thr.append(min(int(c / total * 4294967296.0), 0xFFFFFFFF))
                score.append(q)
            row_off[r + 1] = len(thr)
        self._tables = dict(k=k, gap_bits=gap_bits, hash_size=hash_size, n_rows=len(rows), n_entries=len(thr),
                            hash_key=hash_key, hash_row=hash_row, row_off=row_off,
                            thr=np.array(thr, dtype=np.uint32), score=np.array(score, dtype=np.uint8),
                            cigars=[c for _, c in rows])
        return self._tables

    # ------------------------------------------------------------------ cache
    def save_npz(self, path):
        cigars = list(self.scores.keys())
        n = np.array([len(self.scores[c]) for c in cigars], dtype=np.int32)
        np.savez_compressed(path, k=np.int32(self.kmer_size), cigars=np.array('\n'.join(cigars)), n=n,
                            scores=np.array([s for c in cigars for s in self.scores[c]], dtype=np.int16),
                            probs=np.array([p for c in cigars for p in self.probabilities[c]], dtype=np.float64))

    def _load_npz(self, path, name, output):
        print(f'\nLoading qscore model {name} (packed tables)', file=output)
        z = np.load(path, allow_pickle=False)
        self.type = 'model'
        self.kmer_size = int(z['k'])
        cigars = str(z['cigars']).split('\n')
        scores, probs, i = z['scores'], z['probs'], 0
        for cigar, n in zip(cigars, z['n']):
            n = int(n)
            self.scores[cigar] = [int(s) for s in scores[i:i + n]]
            self.probabilities[cigar] = [float(p) for p in probs[i:i + n]]
            i += n
        print(f'  done: loaded qscore distributions for {len(cigars)} alignments', file=output)


_CIGAR_PART = re.compile(r'(\d+)([IDX=])')


def align_sequences_from_edlib_cigar(seq, frag, cigar, gap_char='-'):
    """Expand a run-length cigar into gapped strings and a per-column op string (qscore_model.py:290-311)."""
    a_seq, a_frag, full = [], [], []
    si = fi = 0
    for size, op in _CIGAR_PART.findall(cigar):
        size = int(size)
        if op in '=X':
            a_seq.append(seq[si:si + size])
            a_frag.append(frag[fi:fi + size])
            si += size
            fi += size
        elif op == 'I':
            a_seq.append(seq[si:si + size])
            a_frag.append(gap_char * size)
            si += size
        else:
            a_seq.append(gap_char * size)
            a_frag.append(frag[fi:fi + size])
            fi += size
        full.append(op * size)
    return ''.join(a_seq), ''.join(a_frag), ''.join(full)


def uniform_dist_scores_and_probs(bottom_q, top_q):
    count = top_q - bottom_q + 1
    return list(range(bottom_q, top_q + 1)), [1 / count] * count


def qscore_char_to_val(q):
    return ord(q) - 33


def qscore_val_to_char(q):
    return chr(q + 33)


def qscore_val_to_error_prob(q):
    return 10.0 ** (-q / 10.0)


def qscore_char_to_error_prob(q):
    return qscore_val_to_error_prob(qscore_char_to_val(q))

"""
ErrorModel: same constructor, attributes and sampling law as the reference's class
(/root/reference/badread/error_model.py:86-160), plus `tables()` which flattens the model into the
arrays the HIP kernels index (layout documented at brx_error_model in include/brx.h).

Load-time work (error_model.py:111-133): every `alt` of every k-mer row is aligned to the k-mer by
align_kmers (:179-229) to place it on the k-mer's positions.  The reference makes 425,984 edlib
calls for that; here the inner alignments of ALL alts are submitted as one batch to the HIP Myers
kernel (brx_align_batch, the same kernel the simulate path uses), and the flattened result is
cached as .npz keyed by the model file's hash.

Python-facing compatibility:
  .kmer_size, .type, .alternatives[kmer] -> list of k-lists of strings, .probabilities[kmer] ->
  list of floats, .add_errors_to_kmer(kmer) -> list of k strings (host-side, uses `random` like
  the reference; the accelerated path samples from the flattened tables on the GPU instead).
"""
import hashlib
import os
import pathlib
import random
import sys

import numpy as np

from .misc import get_open_func, get_random_base, get_random_different_base, random_chance

BUILTIN_ERROR_MODELS = ('nanopore2018', 'nanopore2020', 'nanopore2023', 'pacbio2016', 'pacbio2021')
POOL_PREAMBLE = 528
_BASES = 'ACGT'
_CODE = {b: i for i, b in enumerate(_BASES)}
OP_EQ, OP_X, OP_I, OP_D = 0, 1, 2, 3


# ---------------------------------------------------------------------------------------------
# locating model files and caches
# ---------------------------------------------------------------------------------------------
def model_search_dirs(kind):
    """Directories searched for built-in `<name>.gz` model files (kind: 'error_models' | 'qscore_models')."""
    here = pathlib.Path(os.path.dirname(os.path.realpath(__file__)))
    dirs = []
    env = os.environ.get('BADREAD_AMD_MODEL_DIR')
    if env:
        dirs += [pathlib.Path(env) / kind, pathlib.Path(env)]
    dirs.append(here / kind)
    try:
        import importlib.util
        spec = importlib.util.find_spec('badread')
        if spec and spec.origin:
            dirs.append(pathlib.Path(spec.origin).parent / kind)
    except (ImportError, ValueError):
        pass
    dirs.append(pathlib.Path('/root/reference/badread') / kind)
    return dirs


def find_builtin_file(kind, name):
    for d in model_search_dirs(kind):
        candidate = d / (name + '.gz')
        if candidate.is_file():
            return str(candidate)
    return None


def packaged_cache(name, suffix):
    here = pathlib.Path(os.path.dirname(os.path.realpath(__file__)))
    p = here / 'model_cache' / f'{name}.{suffix}.npz'
    return str(p) if p.is_file() else None


def user_cache_dir():
    d = os.environ.get('BADREAD_AMD_CACHE', os.path.join(os.path.expanduser('~'), '.cache', 'badread_amd'))
    try:
        os.makedirs(d, exist_ok=True)
    except OSError:
        return None
    return d


def file_digest(filename):
    h = hashlib.sha256()
    with open(filename, 'rb') as f:
        for block in iter(lambda: f.read(1 << 20), b''):
            h.update(block)
    return h.hexdigest()[:24]


# ---------------------------------------------------------------------------------------------
# align_kmers
# ---------------------------------------------------------------------------------------------
def strings_from_ops(kmer, alt, ops):
    """
    Place `alt` on the positions of `kmer` given the alignment ops of alt[1:-1] (query) against
    kmer[1:-1] (target): the loop of error_model.py:204-228.  First and last base stay fixed; an
    insertion that lands on the first base moves to the second when it is a single base (:225-228).
    """
    k = len(kmer)
    result = [kmer[0]] + [None] * (k - 2) + [kmer[-1]]
    inner_alt = alt[1:-1]
    kpos = apos = 0
    for op in ops:
        if op == OP_EQ or op == OP_X:
            result[kpos + 1] = inner_alt[apos]
            apos += 1
            kpos += 1
        elif op == OP_D:
            result[kpos + 1] = ''
            kpos += 1
        else:
            result[kpos] += inner_alt[apos]
            apos += 1
    if len(result[0]) == 2:
        first, inserted = result[0]
        result[0] = first
        result[1] = inserted + result[1]
    return result


def align_kmers(kmer, alt, aligner=None):
    """
    align_kmers('ACGT', 'ACGTT') -> ['A', 'C', 'GT', 'T'];  align_kmers('ACGT', 'ACT') -> ['A', 'C', '', 'T']
    (error_model.py:179-229).  `aligner` is a callable (queries, targets) -> list of op arrays;
    default: the HIP batch aligner.
    """
    assert len(kmer) > 2
    assert len(alt) > 1
    assert kmer[0] == alt[0] and kmer[-1] == alt[-1]
    inner_k, inner_a = kmer[1:-1], alt[1:-1]
    if len(inner_a) == 0:
        ops = [OP_D] * len(inner_k)
    else:
        aligner = aligner or default_aligner()
        ops = aligner([inner_a.encode()], [inner_k.encode()])[0]
    return strings_from_ops(kmer, alt, ops)


def default_aligner():
    from .engine import hip_align_batch
    return hip_align_batch


def add_one_random_change(kmer):
    """One substitution, insertion (before or after) or deletion at a random position (error_model.py:163-176)."""
    result = list(kmer)
    kind = random.choice(['s', 'i', 'd'])
    pos = random.randint(0, len(kmer) - 1)
    if kind == 's':
        result[pos] = get_random_different_base(result[pos])
    elif kind == 'i':
        if random_chance(0.5):
            result[pos] = result[pos] + get_random_base()
        else:
            result[pos] = get_random_base() + result[pos]
    else:
        result[pos] = ''
    return result


# ---------------------------------------------------------------------------------------------
def derive_lookup_tables(t):
    """The lookup-order layout of an error-model table (include/brx.h, d_rowx / d_altx): one load per step of
    add_errors_to_kmer's choice (error_model.py:135-160) instead of a chain of ten.  Returns rowx, altx and thr padded by
    eight zero entries (thresholds are scanned in blocks of eight)."""
    k, n_rows, n_alts = int(t['k']), int(t['n_rows']), int(t['n_alts'])
    rowx = np.zeros(2 * (n_rows + 1) + 4, dtype=np.uint32)
    rowx[0:2 * n_rows:2] = t['self_thr'][:n_rows]
    rowx[1:2 * (n_rows + 1):2] = t['row_off'][:n_rows + 1]
    pool = t['pool']
    desc = t['desc'][:n_alts].astype(np.int64)
    altx = np.zeros(4 * max(n_alts, 1), dtype=np.uint32)
    if n_alts:
        diff = pool[desc].astype(np.uint32) | (pool[desc + 1].astype(np.uint32) << 8)
        lens = np.stack([pool[desc + 2 + j].astype(np.uint32) if j < k else np.zeros(n_alts, np.uint32) for j in range(8)], axis=1)
        long_ = (lens.max(axis=1) > 15) | (k > 8)
        packed = np.zeros(n_alts, dtype=np.uint32)
        for j in range(8):
            packed |= (lens[:, j] & 15) << np.uint32(4 * j)
        altx[0::4] = t['desc'][:n_alts]
        altx[1::4] = diff | (long_.astype(np.uint32) << 16)
        altx[2::4] = np.where(long_, 0, packed)
    thr = np.concatenate([t['thr'][:max(n_alts, 1)], np.zeros(8, dtype=np.uint32)])
    return dict(rowx=rowx, altx=altx, thr=thr)


def _rows_field(name):
    """A per-row list of the model (k-mers, alternatives, probabilities, inner alignment ops): parsed out of the cache file
    only when something asks for it -- `badread simulate` needs the flattened device tables and nothing else, and turning
    425 984 alternatives into Python lists at every start was 0.4 s of the command's fixed cost."""
    private = '_rows' + name

    def get(self):
        if self._npz_pending is not None:
            self._parse_rows()
        return getattr(self, private)

    def put(self, value):
        setattr(self, private, value)
    return property(get, put)


class ErrorModel(object):
    _kmers = _rows_field('_kmers')                 # row order as in the file
    _probs = _rows_field('_probs')                 # list of lists of float
    _alt_strings = _rows_field('_alt_strings')     # list of lists of alt strings (un-aligned)
    _ops = _rows_field('_ops')                     # per alt: op arrays for the inner alignment

    def __init__(self, model_type_or_filename, output=sys.stderr, aligner=None, use_cache=True):
        self.kmer_size = None
        self.type = None
        self._npz_pending = None    # cache file whose per-row lists have not been parsed yet
        self._n_rows_in_file = None
        self._kmers = []
        self._probs = []
        self._alt_strings = []
        self._positions = None      # list of lists of k-lists (built lazily from ops)
        self._ops = None
        self._tables = None
        self._alternatives = None
        self._probabilities = None
        self._aligner = aligner
        self._use_cache = use_cache
        self._appended = set()

        name = model_type_or_filename
        if name == 'random':
            print('\nUsing a random error model', file=output)
            self.type = 'random'
            self.kmer_size = 1
            self._alternatives, self._probabilities = {}, {}
        elif name in BUILTIN_ERROR_MODELS:
            self._load_builtin(name, output)
        else:
            self.load_from_file(name, output)

    # ------------------------------------------------------------------ loading
    def _load_builtin(self, name, output):
        cache = packaged_cache(name, 'error') if self._use_cache else None
        if cache:
            print(f'\nLoading error model {name} (packed tables)', file=output)
            self._load_npz(cache)
            print(f'  done: loaded error distributions for {self._n_rows_in_file} {self.kmer_size}-mers',
                  file=output)
            return
        path = find_builtin_file('error_models', name)
        if path is None:
            sys.exit(f'Error: could not find the built-in error model {name}; set BADREAD_AMD_MODEL_DIR '
                     f'to a directory holding error_models/{name}.gz')
        self.load_from_file(path, output)

    def load_from_file(self, filename, output=sys.stderr):
        print('\nLoading error model from {}'.format(filename), file=output)
        self.type = 'model'
        cache_path = None
        if self._use_cache:
            d = user_cache_dir()
            if d:
                cache_path = os.path.join(d, f'error-{file_digest(filename)}.npz')
                if os.path.isfile(cache_path):
                    try:
                        self._load_npz(cache_path)
                        print(f'  done: loaded error distributions for {self._n_rows_in_file} '
                              f'{self.kmer_size}-mers', file=output)
                        return
                    except Exception:          # truncated / foreign file (zipfile.BadZipFile, KeyError ...): re-parse the model
                        pass
        with get_open_func(filename)(filename, 'rt') as model_file:
            for line in model_file:
                kmer = line.split(',', 1)[0]
                if self.kmer_size is None:
                    self.kmer_size = len(kmer)
                else:
                    assert self.kmer_size == len(kmer)
                entries = [x.split(',') for x in line.strip().split(';') if x]
                assert entries[0][0] == kmer
                self._kmers.append(kmer)
                self._alt_strings.append([e[0] for e in entries])
                self._probs.append([float(e[1]) for e in entries])
        self._align_all()
        print(f'\r  done: loaded error distributions for {len(self._kmers)} {self.kmer_size}-mers',
              file=output)
        if cache_path and os.environ.get('RANK', '0') in ('', '0'):     # one writer per launch, and atomically:
            tmp = f'{cache_path}.{os.getpid()}.tmp.npz'                 # a reader never sees a half-written archive
            try:
                self.save_npz(tmp)
                os.replace(tmp, cache_path)
            except OSError:
                try:
                    os.remove(tmp)
                except OSError:
                    pass

    def _align_all(self):
        """One batch of inner alignments for every alt of every row (error_model.py:129,202)."""
        queries, targets, where = [], [], []
        self._ops = []
        for r, (kmer, alts) in enumerate(zip(self._kmers, self._alt_strings)):
            assert len(kmer) > 2
            row_ops = []
            for a, alt in enumerate(alts):
                assert len(alt) > 1
                assert kmer[0] == alt[0] and kmer[-1] == alt[-1]
                inner_a, inner_k = alt[1:-1], kmer[1:-1]
                if len(inner_a) == 0:
                    row_ops.append(np.full(len(inner_k), OP_D, dtype=np.uint8))
                elif inner_a == inner_k:
                    row_ops.append(np.zeros(len(inner_k), dtype=np.uint8))
                else:
                    row_ops.append(None)
                    queries.append(inner_a.encode())
                    targets.append(inner_k.encode())
                    where.append((r, a))
            self._ops.append(row_ops)
        if queries:
            aligner = self._aligner or default_aligner()
            results = aligner(queries, targets)
            for (r, a), ops in zip(where, results):
                self._ops[r][a] = np.asarray(ops, dtype=np.uint8)

    # ------------------------------------------------------------------ reference-compatible views
    def _build_positions(self):
        if self._positions is None:
            self._positions = [[strings_from_ops(kmer, alt, ops) for alt, ops in zip(alts, row_ops)]
                               for kmer, alts, row_ops in zip(self._kmers, self._alt_strings, self._ops)]

    @property
    def alternatives(self):
        if self._alternatives is None:
            self._build_positions()
            self._alternatives = {k: [list(p) for p in pos] for k, pos in zip(self._kmers, self._positions)}
        return self._alternatives

    @property
    def probabilities(self):
        if self._probabilities is None:
            self._probabilities = {k: list(p) for k, p in zip(self._kmers, self._probs)}
        return self._probabilities

    def add_errors_to_kmer(self, kmer):
        """Host-side sampler with the reference's law, including its remainder-to-random-change rule
        (error_model.py:135-160).  Unlike the reference it does not grow the stored lists in place."""
        if self.type == 'random':
            return add_one_random_change(kmer)
        if kmer not in self.alternatives:
            return add_one_random_change(kmer)
        alts = self.alternatives[kmer]
        probs = self.probabilities[kmer]
        remainder = 1.0 - sum(probs)
        if remainder > 0.0:
            alt = random.choices(alts + [None], weights=probs + [remainder])[0]
        else:
            alt = random.choices(alts, weights=probs)[0]
        return add_one_random_change(kmer) if alt is None else alt

    # ------------------------------------------------------------------ flattened device tables
    def tables(self):
        """dict of numpy arrays + scalars matching brx_error_model (include/brx.h)."""
        if self._tables is not None:
            return self._tables
        if self.type == 'random':
            pool = _preamble()
            self._tables = dict(k=1, type=0, n_rows=0, n_alts=0,
                                row_off=np.zeros(1, np.uint32), self_thr=np.zeros(1, np.uint32),
                                thr=np.zeros(9, np.uint32), desc=np.zeros(1, np.uint32), pool=pool,
                                rowx=np.zeros(8, np.uint32), altx=np.zeros(4, np.uint32))
            return self._tables
        k = self.kmer_size
        if k > 9:
            sys.exit('Error: error models with k-mers longer than 9 are not supported by the HIP path')
        self._build_positions()
        n_rows = 4 ** k
        row_of = {}
        for idx, kmer in enumerate(self._kmers):
            if set(kmer) <= set(_BASES):
                row_of[_kmer_row(kmer)] = idx
        counts = np.zeros(n_rows + 1, dtype=np.int64)
        for row, idx in row_of.items():
            counts[row + 1] = len(self._probs[idx])
        row_off = np.cumsum(counts).astype(np.uint32)
        n_alts = int(row_off[-1])
        thr = np.zeros(max(n_alts, 1), dtype=np.uint32)
        desc = np.zeros(max(n_alts, 1), dtype=np.uint32)
        self_thr = np.zeros(n_rows, dtype=np.uint32)
        pool = bytearray(_preamble().tobytes())
        for row in sorted(row_of):
            idx = row_of[row]
            kmer = self._kmers[idx]
            probs = self._probs[idx]
            cum, acc = [], 0.0
            for p in probs:                      # itertools.accumulate order, as random.choices does
                acc = acc + p
                cum.append(acc)
            total = acc + max(0.0, 1.0 - sum(probs))
            base = int(row_off[row])
            for a, positions in enumerate(self._positions[idx]):
                t = int(cum[a] / total * 4294967296.0)
                thr[base + a] = min(max(t, 0), 0xFFFFFFFF)
                diff = 0
                lens = []
                chars = bytearray()
                for j, s in enumerate(positions):
                    if len(s) > 127:
                        sys.exit('Error: error model alternative longer than 127 bases at one position')
                    if s != kmer[j]:
                        diff |= 1 << j
                    lens.append(len(s))
                    for ch in s:
                        if ch not in _CODE:
                            sys.exit(f'Error: error model alternative for {kmer} contains a non-ACGT base')
                        chars.append(_CODE[ch])
                desc[base + a] = len(pool)
                pool += bytes([diff & 0xFF, (diff >> 8) & 0xFF]) + bytes(lens) + bytes(chars)
            if sum(probs) >= 1.0:
                thr[base + len(probs) - 1] = 0xFFFFFFFF
            self_thr[row] = thr[base]
        if len(pool) >= (1 << 24):
            sys.exit('Error: error model too large for the HIP path (alternative pool exceeds 16 MiB)')
        self._tables = dict(k=k, type=1, n_rows=n_rows, n_alts=n_alts, row_off=row_off, self_thr=self_thr,
                            thr=thr, desc=desc, pool=np.frombuffer(bytes(pool), dtype=np.uint8).copy())
        self._tables.update(derive_lookup_tables(self._tables))
        return self._tables

    # ------------------------------------------------------------------ cache (.npz)
    # the device tables a cache file carries beside the per-row lists (rowx / altx / the padded thr are derived: derive_lookup_tables)
    _TABLE_ARRAYS = ('row_off', 'self_thr', 'thr', 'desc', 'pool')

    def save_npz(self, path):
        lens = np.array([len(a) for a in self._alt_strings], dtype=np.int32)
        flat_alts = '\n'.join('\t'.join(a) for a in self._alt_strings)
        flat_ops = np.concatenate([o for row in self._ops for o in row]) if self._ops else np.zeros(0, np.uint8)
        op_lens = np.array([len(o) for row in self._ops for o in row], dtype=np.int32)
        extra = {}
        if self.kmer_size <= 9:
            self._tables = None
            t = self.tables()                       # built from the lists: what a later load hands to the engine as it is
            extra = {'t_' + name: (t[name][:max(int(t['n_alts']), 1)] if name == 'thr' else t[name]) for name in self._TABLE_ARRAYS}
            extra['t_counts'] = np.array([t['n_rows'], t['n_alts']], dtype=np.int64)
        np.savez_compressed(path, k=np.int32(self.kmer_size), kmers=np.array('\n'.join(self._kmers)),
                            alts=np.array(flat_alts), n_alts=lens,
                            probs=np.array([p for row in self._probs for p in row], dtype=np.float64),
                            ops=flat_ops, op_lens=op_lens, **extra)

    def _load_npz(self, path):
        """The flattened device tables straight from the file when it carries them (tables() built them once, when the file
        was written); the per-row lists stay in the file until a caller asks for them (_rows_field)."""
        z = np.load(path, allow_pickle=False)
        self.type = 'model'
        self.kmer_size = int(z['k'])
        self._n_rows_in_file = int(len(z['n_alts']))
        if 't_counts' in z.files:
            n_rows, n_alts = (int(x) for x in z['t_counts'])
            t = dict(k=self.kmer_size, type=1, n_rows=n_rows, n_alts=n_alts)
            t.update({name: np.ascontiguousarray(z['t_' + name]) for name in self._TABLE_ARRAYS})
            t.update(derive_lookup_tables(t))
            self._tables = t
        self._npz_pending = path

    def _parse_rows(self):
        path, self._npz_pending = self._npz_pending, None
        z = np.load(path, allow_pickle=False)
        self._kmers = str(z['kmers']).split('\n')
        self._alt_strings = [row.split('\t') for row in str(z['alts']).split('\n')]
        n_alts = z['n_alts']
        probs = z['probs']
        ops = z['ops']
        op_lens = z['op_lens']
        assert len(self._kmers) == len(self._alt_strings) == len(n_alts)
        self._probs, self._ops = [], []
        pi = oi = ai = 0
        op_starts = np.concatenate(([0], np.cumsum(op_lens)))
        for n in n_alts:
            n = int(n)
            self._probs.append([float(x) for x in probs[pi:pi + n]])
            self._ops.append([ops[op_starts[ai + j]:op_starts[ai + j + 1]] for j in range(n)])
            pi += n
            ai += n


def _kmer_row(kmer):
    row = 0
    for ch in kmer:
        row = (row << 2) | _CODE[ch]
    return row


def _preamble():
    pool = np.zeros(POOL_PREAMBLE, dtype=np.uint8)
    pool[:16] = np.arange(16)
    for x in range(16):
        for y in range(16):
            pool[16 + 2 * (16 * x + y)] = x
            pool[16 + 2 * (16 * x + y) + 1] = y
    return pool

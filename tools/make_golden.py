"""
Generate tests/golden/*.json by RUNNING THE UNMODIFIED REFERENCE (/root/reference, Badread 0.4.2)
in this container.  The reference imports the third-party `edlib` wheel, which is absent here, so
oracle/shim/edlib (our canonical Myers aligner behind edlib's API) stands in for it; everything
else executed below is the reference's own code.  The fixtures travel to the GPU box, the
reference does not.

Fixtures written (each records reference version + how it was produced):

  misc.json            known answers of the reference's pure helpers on the path:
                       reverse_complement, identity_from_edlib_cigar, get_target_size,
                       align_sequences_from_edlib_cigar, gamma/beta parameterisation, load_fasta.
  align_kmers.json     error_model.align_kmers(kmer, alt) for every case of
                       test/test_error_model.py with a unique optimum + sampled rows of the
                       built-in models, and ErrorModel(...).alternatives for sampled k-mers.
  fragments.json       get_real_fragment / get_junk_fragment / add_glitches / adapters run with the
                       reference's random sources replaced by scripted values (the values OUR planner
                       drew for a set of reads), so the deterministic string logic -- linear clip,
                       circular wrap, hairpin, strand coordinates, glitch splice -- is the reference's.
  build_fragment.json.gz
                       the reference's build_fragment (simulate.py:91-115) replayed with the decisions
                       our planner took (oracle plan_trace) -- see make_build_fragment().
  sequence_fragment.json.gz
                       the reference's sequence_fragment() + get_qscores() (simulate.py:256-358,
                       qscore_model.py:32-75) REPLAYED with our counter-based draws: random.randint,
                       get_random_sequence, ErrorModel.add_errors_to_kmer and the qscore sampling
                       are scripted from Philox (include/brx_spec.h), every other line -- loop
                       bounds, est**1.5, the 25-change alignment cadence and blending, trimming,
                       cigar windows and the fallback rule -- is executed by the reference.  The
                       oracle (and the HIP path) must reproduce seq / quals / identity exactly.

Run:  python tools/make_golden.py        (needs /root/reference; ~1 minute)
"""
import collections
import gzip
import io
import json
import os
import random
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = '/root/reference'
for p in (REPO, os.path.join(REPO, 'oracle'), os.path.join(REPO, 'tests'), os.path.join(REPO, 'oracle', 'shim'), REFERENCE):
    if p not in sys.path:
        sys.path.insert(0, p)

import badread.error_model as ref_em          # noqa: E402  (the reference)
import badread.fragment_lengths as ref_fl     # noqa: E402
import badread.identities as ref_id           # noqa: E402
import badread.misc as ref_misc               # noqa: E402
import badread.qscore_model as ref_qm         # noqa: E402
import badread.simulate as ref_sim            # noqa: E402
import badread.version as ref_version         # noqa: E402

import pyoracle                               # noqa: E402
from philox import Draws, random_base         # noqa: E402  (tests/philox.py: Python restatement of brx_spec.h)
from badread_amd.error_model import ErrorModel          # noqa: E402
from badread_amd.qscore_model import QScoreModel, cigar_key  # noqa: E402

GOLDEN = os.path.join(REPO, 'tests', 'golden')
NULL = io.StringIO()
HEADER = {'reference': f'rrwick/Badread {ref_version.__version__}', 'aligner': 'oracle/shim/edlib (oracle/myers_ref.c)',
          'generator': 'tools/make_golden.py'}


def dump(name, obj):
    obj = dict(HEADER, **obj)
    path = os.path.join(GOLDEN, name)
    data = json.dumps(obj, separators=(',', ':')).encode()
    if name.endswith('.gz'):
        with gzip.GzipFile(path, 'wb', mtime=0) as f:
            f.write(data)
    else:
        with open(path, 'wb') as f:
            f.write(data)
    print(f'{name}: {os.path.getsize(path)} bytes')


# ---------------------------------------------------------------------------------------------
def make_misc():
    rng = random.Random(1)
    revcomp = [''.join(rng.choice('ACGTRYSWKMBVDHNacgtn.-?XZ') for _ in range(rng.randint(0, 40))) for _ in range(40)]
    cigars = ['5=', '3=1X2=', '2=1I3=2D1X', '', '10=5I5D', '1X', '7=1I7=1D7=1X7=']
    quantities = ['100', '25x', '2.5x', '250M', '1.5g', '3k', '0.1x', '7K', '1X']
    expand = [('ACGACTAGCTACG', 'ACGACTAGCTACG', '13='), ('ACGACTGCTACG', 'ACGACTAGCTACG', '6=1D6='),
              ('ACGACTAGGCTACG', 'ACGACTAGCTACG', '8=1I5='), ('ACGACTTGCTACG', 'ACGACTAGCTACG', '6=1X6='),
              ('AGCTAGCTAG', 'CTAGCTAGCT', '2I8=2D')]
    fasta = os.path.join(GOLDEN, 'small_ref.fasta')
    seqs, depths, circ, hl, hr = ref_misc.load_fasta(fasta)
    out = {
        'reverse_complement': [[s, ref_misc.reverse_complement(s)] for s in revcomp],
        'identity_from_edlib_cigar': [[c, ref_misc.identity_from_edlib_cigar(c)] for c in cigars],
        'get_target_size': [[q, 123456, ref_sim.get_target_size(123456, q)] for q in quantities],
        'align_sequences_from_edlib_cigar': [[s, f, c, list(ref_qm.align_sequences_from_edlib_cigar(s, f, c))] for s, f, c in expand],
        'gamma_parameters': [[m, s, list(ref_fl.gamma_parameters(m, s))] for m, s in ((15000, 13000), (100, 10), (5000, 5000))],
        'beta_parameters': [[m, s, x, list(ref_id.beta_parameters(m, s, x))] for m, s, x in ((95, 2.5, 99), (85, 5, 95), (90, 4, 100))],
        'load_fasta': {'file': 'small_ref.fasta', 'seqs': seqs, 'depths': depths, 'circular': circ,
                       'hairpin_left': hl, 'hairpin_right': hr},
    }
    dump('misc.json', out)


# ---------------------------------------------------------------------------------------------
def make_align_kmers():
    # unique-optimum cases of test/test_error_model.py plus assorted shapes
    pairs = [('ACGT', 'ACGT'), ('ACGT', 'AGGT'), ('ACGT', 'AGT'), ('ACGT', 'AT'), ('ACGT', 'ACCGT'), ('ACGTA', 'ACCTA'),
             ('ACGTACG', 'ACGTACG'), ('ACGTACG', 'ACTACG'), ('ACGTACG', 'ACGTTACG'), ('ACGTACG', 'AG'), ('ACGTACG', 'ACGATAACG'),
             ('GATTACA', 'GATACA'), ('GATTACA', 'GTTTTTTACA'), ('CCCCCCC', 'CC'), ('ACGTACG', 'AGTACCCG')]
    cases = [[k, a, ref_em.align_kmers(k, a)] for k, a in pairs]
    rng = random.Random(7)
    models = {}
    for name in ('nanopore2023', 'pacbio2021', 'nanopore2018'):
        path = os.path.join(REFERENCE, 'badread', 'error_models', name + '.gz')
        rows = []
        with gzip.open(path, 'rt') as f:
            lines = f.readlines()
        for line in rng.sample(lines, 12):
            kmer = line.split(',', 1)[0]
            entries = [x.split(',') for x in line.strip().split(';') if x]
            alts = [e[0] for e in entries]
            rows.append({'kmer': kmer, 'alts': alts, 'probs': [float(e[1]) for e in entries],
                         'aligned': [ref_em.align_kmers(kmer, a) for a in alts]})
        models[name] = rows
    fixture = os.path.join(REFERENCE, 'test', '4-mer_error_model')
    m = ref_em.ErrorModel(fixture, NULL)
    four = {'kmer_size': m.kmer_size, 'n_rows': len(m.alternatives),
            'rows': {k: {'alts': m.alternatives[k], 'probs': m.probabilities[k]} for k in ('AAAA', 'ACGT', 'TTTT', 'GATC', 'CGCG')},
            'text': open(fixture).read()}
    dump('align_kmers.json', {'cases': cases, 'models': models, 'four_mer_model': four})


# ---------------------------------------------------------------------------------------------
class Script(object):
    """Scripted stand-ins for the reference's random sources while one reference function runs."""

    def __init__(self):
        self.values = collections.deque()

    def push(self, *vals):
        self.values.extend(vals)

    def pop(self):
        return self.values.popleft()


def make_fragments():
    """The reference's fragment/glitch string logic on scripted coordinates (no randomness left)."""
    fasta = os.path.join(GOLDEN, 'small_ref.fasta')
    seqs, depths, circ, hl, hr = ref_misc.load_fasta(fasta)
    rev = {n: ref_misc.reverse_complement(s) for n, s in seqs.items()}
    names = list(seqs)
    Args = collections.namedtuple('Args', ['junk_reads', 'random_reads'])
    rng = random.Random(11)
    real = []
    for _ in range(160):
        contig = rng.choice(names)
        L = len(seqs[contig])
        length = rng.choice([1, 2, 10, L // 3, L - 1, L, L + 1, 2 * L, rng.randint(1, 2 * L + 5)])
        strand_plus = rng.random() < 0.5
        start = rng.randint(0, L - 1)
        script = Script()
        # get_real_fragment draws: random.choices (only if >1 contig), random_chance(0.5), random.randint(0, len-1)
        orig = (ref_sim.random.choices, ref_sim.random_chance, ref_sim.random.randint)
        ref_sim.random.choices = lambda pop, weights=None: [contig]
        ref_sim.random_chance = lambda p: strand_plus
        ref_sim.random.randint = lambda a, b: start
        try:
            seq, info = ref_sim.get_real_fragment(length, seqs, rev, names, [depths[n] * len(seqs[n]) for n in names],
                                                  circ, hl, hr)
        finally:
            ref_sim.random.choices, ref_sim.random_chance, ref_sim.random.randint = orig
        real.append({'contig': contig, 'length': length, 'strand': '+' if strand_plus else '-', 'start': start,
                     'seq': seq, 'info': info})
        del script
    # glitches: np.random.geometric and get_random_sequence scripted
    glitches = []
    for _ in range(30):
        frag = ''.join(rng.choice('ACGT') for _ in range(rng.randint(1, 400)))
        rate, size, skip = rng.choice([(50, 5, 5), (20, 0, 3), (30, 4, 0), (1, 1, 1), (0.5, 0.5, 0.5), (100, 10, 10)])
        draws = [rng.randint(1, 60) for _ in range(200)]
        fills = [''.join(rng.choice('ACGT') for _ in range(80)) for _ in range(100)]
        dq, fq = collections.deque(draws), collections.deque(fills)
        used = []

        def geo(p=None):
            v = dq.popleft()
            used.append(v)
            return v

        def fill(n):
            s = fq.popleft()[:n]
            assert len(s) == n
            return s
        orig = (ref_sim.np.random.geometric, ref_sim.get_random_sequence)
        ref_sim.np.random.geometric = geo
        ref_sim.get_random_sequence = fill
        try:
            out = ref_sim.add_glitches(frag, rate, size, skip)
        finally:
            ref_sim.np.random.geometric, ref_sim.get_random_sequence = orig
        glitches.append({'fragment': frag, 'rate': rate, 'size': size, 'skip': skip, 'geometric': used,
                         'fills': fills[:len(fills) - len(fq)], 'out': out})
    # adapters: scripted chance + beta
    adapters = []
    for _ in range(40):
        adapter = ''.join(rng.choice('ACGT') for _ in range(rng.randint(1, 40)))
        rate, amount = rng.choice([(0.9, 0.6), (0.5, 0.2), (1.0, 1.0), (0.3, 0.999)])
        chance, beta = rng.random(), rng.random()
        orig = (ref_sim.random_chance, ref_sim.np.random.beta)
        ref_sim.random_chance = lambda p: chance < p
        ref_sim.np.random.beta = lambda a, b: beta
        try:
            s = ref_sim.get_start_adapter(rate, amount, adapter)
            e = ref_sim.get_end_adapter(rate, amount, adapter)
        finally:
            ref_sim.random_chance, ref_sim.np.random.beta = orig
        adapters.append({'adapter': adapter, 'rate': rate, 'amount': amount, 'chance': chance, 'beta': beta, 'start': s, 'end': e})
    junk = []
    for _ in range(20):
        unit = ''.join(rng.choice('ACGT') for _ in range(rng.randint(1, 5)))
        length = rng.randint(1, 300)
        orig = (ref_sim.random.randint, ref_sim.get_random_sequence)
        ref_sim.random.randint = lambda a, b: len(unit)
        ref_sim.get_random_sequence = lambda n: unit
        try:
            out = ref_sim.get_junk_fragment(length)
        finally:
            ref_sim.random.randint, ref_sim.get_random_sequence = orig
        junk.append({'unit': unit, 'length': length, 'out': out})
    dump('fragments.json', {'fasta': 'small_ref.fasta', 'real': real, 'glitches': glitches, 'adapters': adapters, 'junk': junk})


# ---------------------------------------------------------------------------------------------
def make_build_fragment():
    """
    The reference's build_fragment (simulate.py:91-115: adapters, get_fragment with its 1000-try loop,
    chimeras, add_glitches) REPLAYED with the decisions our planner took for a set of read indices:
    oracle.plan_trace() lists every primitive draw, and each of the reference's random sources is
    replaced by a queue of them.  What the reference then computes -- which contig string is sliced
    where, wrap/clip/hairpin, strand coordinates, junk repeat, glitch splice, info text -- is its own.
    """
    import helpers as H
    from badread_amd.engine import SimParams
    from badread_amd.reference import PackedReference
    fasta = os.path.join(GOLDEN, 'small_ref.fasta')
    seqs, depths, circ, hl, hr = ref_misc.load_fasta(fasta)
    rev = {n: ref_misc.reverse_complement(s) for n, s in seqs.items()}
    names = list(seqs)
    weights = [depths[n] * len(seqs[n]) for n in names]
    pref = PackedReference.from_seqs(seqs, depths, circ, hl, hr)
    configs = [
        dict(frag_mean=400, frag_stdev=350, junk_rate=0.05, random_rate=0.05, chimera_rate=0.15,
             glitch_rate=150, glitch_size=6, glitch_skip=5),
        dict(frag_mean=900, frag_stdev=900, junk_rate=0.0, random_rate=0.0, chimera_rate=0.0,
             glitch_rate=0, glitch_size=0, glitch_skip=0, start_adapter='', end_adapter=''),
        dict(frag_mean=120, frag_stdev=0, junk_rate=0.2, random_rate=0.2, chimera_rate=0.3,
             glitch_rate=40, glitch_size=0.5, glitch_skip=0, start_rate=1.0, start_amount=1.0, end_rate=1.0, end_amount=0.9),
    ]
    out_cfgs = []
    Args = collections.namedtuple('Args', ['junk_reads', 'random_reads', 'chimeras', 'start_adapter_seq', 'end_adapter_seq',
                                           'glitch_rate', 'glitch_size', 'glitch_skip'])
    for ci, cfg in enumerate(configs):
        params = SimParams(**cfg)
        engine = H.configure(H.oracle_engine(), pref, 'random', 'ideal', params)
        args = Args(params.junk_rate * 100, params.random_rate * 100, params.chimera_rate * 100, params.start_adapter,
                    params.end_adapter, params.glitch_rate, params.glitch_size, params.glitch_skip)
        seed = 900 + ci
        reads = []
        for read in range(120):
            trace = engine.plan_trace(seed, read)
            q = collections.defaultdict(collections.deque)
            for kind, val in trace:
                q[kind].append(val)

            class FL(object):
                def get_fragment_length(self):
                    return int(q['LENGTH'].popleft())

            def fake_random():
                return q['U'].popleft()

            def fake_choices(pop, weights=None):
                return [pop[int(q['CONTIG'].popleft())]]

            def fake_randint(a, b):
                if (a, b) == (1, 5):
                    return int(q['JUNKLEN'].popleft())
                v = int(q['START'].popleft())
                assert a <= v <= b
                return v

            def fake_beta(a, b, _adapters=(params.start_adapter, params.end_adapter)):
                L = q['ADAPTLEN'].popleft()
                n_ad = fake_beta.next_len
                return (L + 0.5) / n_ad

            def fake_geometric(p=None):
                return int(q['GEO'].popleft())

            def fake_random_sequence(n):
                if fake_random_sequence.junk_pending:
                    fake_random_sequence.junk_pending = False
                    unit = int(q['JUNKUNIT'].popleft())
                    return ''.join('ACGT'[(unit >> (2 * i)) & 3] for i in range(n))
                serial = int(q['SERIAL'].popleft())
                return ''.join('ACGT'[random_base(seed, read, serial, p)] for p in range(n))
            fake_random_sequence.junk_pending = False

            real_junk = ref_sim.get_junk_fragment

            def junk_wrapper(length):
                fake_random_sequence.junk_pending = True
                return real_junk(length)

            real_adapter_len = ref_sim.get_adapter_frag_length

            def adapter_len_wrapper(amount, adapter):
                fake_beta.next_len = len(adapter)
                return real_adapter_len(amount, adapter)

            saved = (random.random, ref_sim.random.choices, ref_sim.random.randint, ref_sim.np.random.beta,
                     ref_sim.np.random.geometric, ref_sim.get_random_sequence, ref_sim.get_junk_fragment,
                     ref_sim.get_adapter_frag_length)
            random.random = fake_random
            ref_sim.random.choices, ref_sim.random.randint = fake_choices, fake_randint
            ref_sim.np.random.beta, ref_sim.np.random.geometric = fake_beta, fake_geometric
            ref_sim.get_random_sequence, ref_sim.get_junk_fragment = fake_random_sequence, junk_wrapper
            ref_sim.get_adapter_frag_length = adapter_len_wrapper
            failed = False
            try:
                fragment, info = ref_sim.build_fragment(FL(), seqs, rev, names, weights, circ, hl, hr, args,
                                                        params.start_rate, params.start_amount, params.end_rate, params.end_amount)
            except SystemExit:
                fragment, info, failed = '', [], True
            finally:
                (random.random, ref_sim.random.choices, ref_sim.random.randint, ref_sim.np.random.beta,
                 ref_sim.np.random.geometric, ref_sim.get_random_sequence, ref_sim.get_junk_fragment,
                 ref_sim.get_adapter_frag_length) = saved
            left = {k: len(v) for k, v in q.items() if len(v) and k != 'IDENTITY'}
            assert failed or not left, (ci, read, left)
            reads.append({'read': read, 'fragment': fragment, 'info': ' '.join(info), 'failed': failed,
                          'identity': q['IDENTITY'][0] if q['IDENTITY'] else None})
        out_cfgs.append({'params': cfg, 'seed': seed, 'reads': reads})
        n_fail = sum(r['failed'] for r in reads)
        print(f'  build_fragment config {ci}: {len(reads)} reads, {n_fail} fatal, '
              f'mean length {np.mean([len(r["fragment"]) for r in reads]):.0f}')
    dump('build_fragment.json.gz', {'fasta': 'small_ref.fasta', 'configs': out_cfgs})


# ---------------------------------------------------------------------------------------------
class ScriptedErrorModelDraws(object):
    """The REFERENCE's own ErrorModel.add_errors_to_kmer / add_one_random_change (error_model.py:135-176) run unmodified;
    only the primitives of Python's `random` module they reach are scripted from the current iteration's Philox words:
      random.choices(alts, weights=probs)   -> CPython's own algorithm (cumulative float weights + bisect) with its uniform
                                               replaced by (w2 + 0.5) / 2^32
      random.choice(['s', 'i', 'd'])        -> w3 % 3
      random.randint(0, k - 1)              -> (w3 // 3) % k
      random.random() in random_chance(.5)  -> below 0.5 iff bit 0 of w3 // (3k) (insertion AFTER the base)
      random.randint(0, 3) in get_random_base (directly, or inside get_random_different_base's rejection loop)
                                            -> insertion: bits 1-2 of w3 // (3k); substitution: the base the oracle's
                                               rule names ((o + 1 + rest % 3) & 3, or rest & 3 for a non-ACGT original),
                                               which the rejection loop accepts at its first draw
    so the in-place growth of the probability lists, the `None` remainder entry, the k-mer-not-in-model
    route and the string surgery of add_one_random_change are all executed by the reference."""

    def __init__(self, state):
        self.state = state
        self.ctx = None
        self.kmer = None

    def install(self):
        self.saved = (random.choices, random.choice, random.random, ref_em.add_one_random_change)
        real_rc = ref_em.add_one_random_change

        def rc_wrapper(kmer):
            self.ctx, self.kmer = 'type', kmer
            try:
                return real_rc(kmer)
            finally:
                self.ctx = None
        random.choices, random.choice, random.random = self.choices, self.choice, self.random
        ref_em.add_one_random_change = rc_wrapper

    def remove(self):
        random.choices, random.choice, random.random, ref_em.add_one_random_change = self.saved

    def choices(self, population, weights=None, cum_weights=None, k=1):
        import bisect
        import itertools
        assert k == 1 and weights is not None and self.ctx is None
        cum = list(itertools.accumulate(weights))
        total = cum[-1] + 0.0
        u = (self.state['w'][2] + 0.5) / 4294967296.0
        return [population[bisect.bisect(cum, u * total, 0, len(population) - 1)]]

    def choice(self, seq):
        assert self.ctx == 'type' and list(seq) == ['s', 'i', 'd']
        self.kind = self.state['w'][3] % 3
        self.ctx = 'pos'
        return seq[self.kind]

    def randint(self, a, b):
        """The part of random.randint that belongs to the error model; returns None when the call is not ours."""
        w3 = self.state['w'][3] if self.state['w'] else 0
        if self.ctx == 'pos':
            k = len(self.kmer)
            assert (a, b) == (0, k - 1)
            self.pos = (w3 // 3) % k
            self.rest = w3 // (3 * k)
            self.ctx = 'base'
            return self.pos
        if self.ctx == 'base':
            assert (a, b) == (0, 3)
            if self.kind == 0:
                o = 'ACGT'.find(self.kmer[self.pos])
                return ((o + 1 + self.rest % 3) & 3) if o >= 0 else (self.rest & 3)
            assert self.kind == 1
            return (self.rest >> 1) & 3
        return None

    def random(self):
        assert self.ctx == 'base' and self.kind == 1
        return 0.0 if (self.rest & 1) else 0.75


class ReplayQScoreModel(object):
    """get_qscore with the reference's lookup + fallback loop (qscore_model.py:273-287) on the
    reference's own dicts; only random.choices is replaced by our threshold walk."""

    def __init__(self, ref_model, tables, draws, state):
        self.ref, self.t, self.draws, self.state = ref_model, tables, draws, state
        self.kmer_size, self.type = ref_model.kmer_size, ref_model.type
        self.row_of = {}
        keys, rows = tables['hash_key'], tables['hash_row']
        for slot in np.flatnonzero(keys != np.uint64(0xFFFFFFFFFFFFFFFF)):
            self.row_of[int(keys[slot])] = int(rows[slot])

    def get_qscore(self, cigar):
        s = self.state['qpos']
        self.state['qpos'] += 1
        while True:
            assert len(cigar.replace('D', '')) % 2 == 1
            if cigar in self.ref.scores:
                row = self.row_of[cigar_key(cigar, self.t['gap_bits'])]
                e0, e1 = int(self.t['row_off'][row]), int(self.t['row_off'][row + 1])
                assert list(self.t['score'][e0:e1]) == list(self.ref.scores[cigar])
                u = self.draws.qs(s)
                e = e0
                while e < e1 - 1 and not (u < int(self.t['thr'][e])):
                    e += 1
                return ref_qm.qscore_val_to_char(int(self.t['score'][e]))
            cigar = cigar[1:-1].strip('D')


_REF_ERROR_MODELS = {}


def reference_error_model(name):
    """The reference's ErrorModel object (5 s per file model), loaded once per name."""
    if name not in _REF_ERROR_MODELS:
        _REF_ERROR_MODELS[name] = ref_em.ErrorModel(name, NULL)
    return _REF_ERROR_MODELS[name]


def replay_sequence_fragment(engine, em_name, qm_name, ref_qmodel, qtables, fragment, target, seed, read):
    draws = Draws(seed, read)
    state = {'w': None, 'iter': 0, 'naligns': 0, 'qpos': 0, 'pads': 0}
    em = reference_error_model(em_name)                  # the reference's own object: add_errors_to_kmer runs unmodified
    scripted = ScriptedErrorModelDraws(state)
    qm = ReplayQScoreModel(ref_qmodel, qtables, draws, state)
    k = em.kmer_size
    n = len(fragment) + 2 * k

    def fake_randint(a, b):
        ours = scripted.randint(a, b)
        if ours is not None:
            return ours
        if a == 0 and b == n - 1 - k:                     # k-mer position (simulate.py:294)
            state['w'] = draws.mut(state['iter'])
            state['iter'] += 1
            return draws.below64(state['w'][0], state['w'][1], b + 1)
        assert a == 0 and b == n - 1000, (a, b, n)        # alignment window (simulate.py:338)
        w = draws.win(state['naligns'])
        state['naligns'] += 1
        return draws.below64(w[0], w[1], b + 1)

    def fake_random_sequence(length):                     # the two pads (simulate.py:260)
        serial = state['pads']
        state['pads'] += 1
        return ''.join('ACGT'[random_base(seed, read, serial, p)] for p in range(length))

    orig = (random.randint, ref_sim.get_random_sequence)
    random.randint, ref_sim.get_random_sequence = fake_randint, fake_random_sequence
    scripted.install()
    try:
        seq, qual, ident, ident_q = ref_sim.sequence_fragment(fragment, target, em, qm)
    finally:
        scripted.remove()
        random.randint, ref_sim.get_random_sequence = orig
    return {'em': em_name, 'qm': qm_name, 'fragment': fragment, 'target': target, 'seed': seed, 'read': read,
            'seq': seq, 'qual': qual, 'identity': ident, 'identity_by_qscores': ident_q,
            'iterations': state['iter'], 'alignments': state['naligns']}


def make_sequence_fragment():
    import helpers as H
    rng = random.Random(5)
    cases = []
    for em_name, qm_name in (('nanopore2023', 'nanopore2023'), ('random', 'ideal'), ('pacbio2021', 'pacbio2021'),
                             ('random', 'random'), ('nanopore2018', 'nanopore2018')):
        engine = H.oracle_engine()
        engine.set_error_model(ErrorModel(em_name, NULL).tables())
        qtables = QScoreModel(qm_name, NULL).tables()
        engine.set_qscore_model(qtables)
        random.seed(12345)                                    # ref 'random'/'ideal' models are deterministic tables
        ref_qmodel = ref_qm.QScoreModel(qm_name, NULL)
        specs = [(30, 0.9), (300, 1.0), (700, 0.85), (979, 0.9), (986, 0.93), (987, 0.9), (1500, 0.95), (2600, 0.8), (4000, 0.97)]
        if em_name != 'nanopore2023':
            specs = specs[:6] + [(2000, 0.9)]
        for idx, (L, target) in enumerate(specs):
            alphabet = 'ACGT' if idx % 4 else 'ACGTN'
            fragment = ''.join(rng.choice(alphabet) for _ in range(L))
            cases.append(replay_sequence_fragment(engine, em_name, qm_name, ref_qmodel, qtables, fragment, target,
                                                  seed=1000 + idx, read=7 * idx + 1))
            print(f'  {em_name}/{qm_name} L={L} target={target}: {cases[-1]["iterations"]} iterations, '
                  f'{cases[-1]["alignments"]} alignments, identity {cases[-1]["identity"]:.4f}')
    dump('sequence_fragment.json.gz', {'cases': cases})


def recipe_fragment(seed, length, with_n=False):
    """A fragment as a pure function of (seed, length): base i = splitmix64(seed << 32 | i) -- the digest cases below keep the
    recipe instead of the text (tests/helpers.py holds the same function)."""
    import numpy as np
    x = (np.uint64(seed) << np.uint64(32)) + np.arange(length, dtype=np.uint64)
    with np.errstate(over='ignore'):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    codes = (x >> np.uint64(62)).astype(np.uint8)
    if with_n:                                           # one base in 256 is N
        codes[((x >> np.uint64(20)) & np.uint64(255)) == 0] = 4
    return ''.join('ACGTN'[c] for c in codes)


def _bound_case(spec):
    """One digest case (a worker process of make_sequence_fragment_bound)."""
    import hashlib
    import helpers as H
    em_name, qm_name, length, target, seed, read, with_n = spec
    engine = H.oracle_engine()
    engine.set_error_model(ErrorModel(em_name, NULL).tables())
    qtables = QScoreModel(qm_name, NULL).tables()
    engine.set_qscore_model(qtables)
    random.seed(12345)
    ref_qmodel = ref_qm.QScoreModel(qm_name, NULL)
    fragment = recipe_fragment(seed, length, with_n)
    c = replay_sequence_fragment(engine, em_name, qm_name, ref_qmodel, qtables, fragment, target, seed=seed, read=read)
    return {'em': em_name, 'qm': qm_name, 'length': length, 'with_n': with_n, 'target': target, 'seed': seed, 'read': read,
            'seq_len': len(c['seq']), 'seq_sha256': hashlib.sha256(c['seq'].encode()).hexdigest(),
            'qual_sha256': hashlib.sha256(c['qual'].encode()).hexdigest(), 'identity': c['identity'],
            'identity_by_qscores': c['identity_by_qscores'], 'iterations': c['iterations'], 'alignments': c['alignments']}


def make_sequence_fragment_bound():
    """HOW OFTEN can `est ** 1.5` (the reference, libm pow: simulate.py:321) and `est * sqrt(est)` (the oracle
    and the kernels) part ways?  500+ more replays of the UNMODIFIED sequence_fragment with our draws -- 24 of them 50 kb fragments
    at 80-90 % identity (up to 140 000 loop iterations and 9 000 uses of the power each) -- kept as digests (sha256 of sequence and
    qualities, identity, iterations, alignments) with the recipe of the fragment; and the rate at which the two expressions differ
    on THIS libm over 10^7 doubles of [0.5, 1] (tests/golden/pow15.json)."""
    import math
    import multiprocessing
    import platform
    rng = random.Random(77)
    specs = []
    for i in range(24):                                  # long, rough reads: many iterations at estimates far below 1
        specs.append(('nanopore2023', 'nanopore2023', 50000, round(rng.uniform(0.80, 0.90), 3), 5000 + i, 11 * i + 3, False))
    for i in range(500):
        em_name, qm_name = (('nanopore2023', 'nanopore2023'), ('pacbio2021', 'pacbio2021'), ('random', 'random'), ('nanopore2018', 'nanopore2018'))[i % 4]
        length = int(rng.choice([200, 600, 1200, 2500, 5000, 9000]) * rng.uniform(0.7, 1.3))
        specs.append((em_name, qm_name, length, round(rng.uniform(0.78, 0.99), 3), 7000 + i, 13 * i + 5, i % 7 == 0))
    with multiprocessing.Pool(min(8, os.cpu_count() or 1)) as pool:
        cases = pool.map(_bound_case, specs, chunksize=4)
    dump('sequence_fragment_bound.json.gz', {'cases': cases})
    print(f'  {len(cases)} digest cases, {sum(c["iterations"] for c in cases)} loop iterations, {sum(c["alignments"] for c in cases)} alignments')
    # ---- the two expressions on this libm
    n, differ, worst = 10_000_000, 0, 0
    import struct
    r2 = random.Random(99)
    for _ in range(n):
        x = 0.5 + 0.5 * r2.random()
        a, b = x ** 1.5, x * math.sqrt(x)
        if a != b:
            differ += 1
            ua, ub = struct.unpack('<q', struct.pack('<d', a))[0], struct.unpack('<q', struct.pack('<d', b))[0]
            worst = max(worst, abs(ua - ub))
    dump('pow15.json', {'doubles': n, 'interval': [0.5, 1.0], 'differ': differ, 'rate': differ / n, 'largest_difference_ulps': worst,
                        'libc': list(platform.libc_ver()), 'python': platform.python_version(),
                        'note': 'x ** 1.5 (CPython float_pow -> libm pow) against x * math.sqrt(x); a difference of one ulp in the scale of ONE '
                                'applied change moves `errors` by ~1e-16 relative -- it changes a result only if it flips one of the '
                                'comparisons est <= target / the rounded identity of a later alignment: none of the replays above did'})
    print(f'  pow15: {differ} of {n} differ ({differ / n:.3%}), at most {worst} ulp')


def make_random_change():
    """Counts of the reference's own add_one_random_change (error_model.py:163-176) over 240 000 calls per k-mer, under
    random.seed(2024): the empirical law tests/test_golden_host.py::test_random_change_law_against_the_reference holds
    the oracle's enumerated law against (support and chi-square)."""
    import collections
    from badread.error_model import add_one_random_change
    cases = []
    random.seed(2024)
    for kmer in ('A', 'N', 'GATTACA', 'ACGTNCA'):
        counts = collections.Counter()
        for _ in range(240000):
            counts['|'.join(add_one_random_change(kmer))] += 1
        cases.append({'kmer': kmer, 'calls': 240000, 'counts': dict(sorted(counts.items()))})
        print(f'  random_change {kmer}: {len(counts)} outcomes')
    dump('random_change.json', {'cases': cases})


def make_qscore_top_rows():
    """The 32 most frequent CIGAR rows of the reference's qscore model files (by the occurrence count the files carry)
    with the scores / probabilities the REFERENCE's own QScoreModel parses for them: the expected laws of the per-row
    chi-square gate (SURVEY.md section 8d gate 3)."""
    import gzip as gz
    out = {}
    for name in ('nanopore2023', 'pacbio2021'):
        path = os.path.join(REFERENCE, 'badread', 'qscore_models', name + '.gz')
        counts = {}
        with gz.open(path, 'rt') as f:
            for line in f:
                parts = line.strip().split(';')
                if len(parts) >= 3 and parts[0] != 'overall':
                    counts[parts[0]] = int(parts[1])
        model = ref_qm.QScoreModel(name, NULL)
        top = sorted((c for c in counts if c in model.scores), key=lambda c: -counts[c])[:32]
        total = sum(counts.values())
        out[name] = [{'cigar': c, 'count': counts[c], 'share': counts[c] / total, 'scores': list(model.scores[c]),
                      'probs': list(model.probabilities[c])} for c in top]
        print(f'  qscore_top_rows {name}: top row {top[0]} {counts[top[0]] / total:.3f}')
    dump('qscore_top_rows.json', out)


MODEL_BUILDER_CASES = {
    'error': [dict(k_size=7, max_alignments=None, max_alt=25), dict(k_size=4, max_alignments=None, max_alt=3),
              dict(k_size=5, max_alignments=10, max_alt=25)],
    'qscore': [dict(k_size=9, max_alignments=None, max_del=6, min_occur=1, max_output=10000),
               dict(k_size=9, max_alignments=None, max_del=6, min_occur=100, max_output=10000),
               dict(k_size=5, max_alignments=None, max_del=2, min_occur=2, max_output=20),
               dict(k_size=3, max_alignments=7, max_del=6, min_occur=1, max_output=10000)],
}


def make_model_builders():
    """Inputs for the model builders (row f4) made HERE -- a random two-contig reference, 48 reads cut from it and
    mutated (substitutions, insertions and deletions of 1-30 bases, both strands, clipped ends, random qualities), their
    alignments from the oracle aligner written as PAF with cg:Z: CIGARs, plus decoy lines the loader must drop -- and the
    output of the REFERENCE's own make_error_model / make_qscore_model on them for several argument sets."""
    import contextlib
    import types
    rng = np.random.default_rng(404)
    folder = os.path.join(GOLDEN, 'model_builder')
    os.makedirs(folder, exist_ok=True)
    dna = lambda n: ''.join(np.array(list('ACGT'))[rng.integers(0, 4, n)])
    refs = collections.OrderedDict([('ctg_one', dna(12000)), ('ctg_two', dna(6000))])
    refs['ctg_two'] = refs['ctg_two'][:3000] + 'NNNNRY' + refs['ctg_two'][3006:]
    with open(os.path.join(folder, 'ref.fasta'), 'w') as f:
        for name, seq in refs.items():
            f.write(f'>{name} some description\n')
            for i in range(0, len(seq), 70):
                f.write(seq[i:i + 70] + '\n')
    reads, paf = [], []
    for r in range(48):
        name = f'read_{r}'
        ctg = 'ctg_one' if r % 3 else 'ctg_two'
        L = int(rng.integers(150, 2600))
        start = int(rng.integers(0, len(refs[ctg]) - L))
        piece = refs[ctg][start:start + L]
        rate = float(rng.choice([0.02, 0.05, 0.10, 0.17]))
        out = []
        for ch in piece:
            u = rng.random()
            if u < rate / 3:
                out.append('ACGT'[rng.integers(0, 4)])
            elif u < 2 * rate / 3:
                out.append(ch + dna(int(rng.choice([1, 1, 1, 2, 3, 30]))))
            elif u < rate:
                continue
            else:
                out.append(ch)
        mutated = ''.join(out)
        if r % 7 == 0:                                   # one long deletion, longer than any --max_del
            cut = len(mutated) // 2
            mutated = mutated[:cut] + mutated[cut + int(rng.integers(7, 20)):]
        strand = '+' if rng.random() < 0.5 else '-'
        d, ops = pyoracle.align(mutated.encode(), piece.encode())
        runs, cigar = [], []
        for op in ops.tolist():
            letter = 'M' if op in (0, 1) else 'I' if op == 2 else 'D'
            if runs and runs[-1][1] == letter:
                runs[-1][0] += 1
            else:
                runs.append([1, letter])
        cigar = ''.join(f'{n}{t}' for n, t in runs)
        clip5, clip3 = (int(rng.integers(0, 30)), int(rng.integers(0, 30))) if r % 4 == 0 else (0, 0)
        read_fwd = dna(clip5) + mutated + dna(clip3)
        read_seq = read_fwd if strand == '+' else ref_misc.reverse_complement(read_fwd)
        rs, re_ = (clip5, clip5 + len(mutated)) if strand == '+' else (clip3, clip3 + len(mutated))
        qual = ''.join(chr(33 + int(q)) for q in rng.integers(1, 51, len(read_seq)))
        if r % 11 == 0:
            read_seq = read_seq.lower()                  # the loader upper-cases
        reads.append((name, read_seq, qual))
        matches = int((ops == 0).sum())
        line = [name, str(len(read_seq)), str(rs), str(re_), strand, ctg, str(len(refs[ctg])), str(start), str(start + L),
                str(matches), str(len(ops)), '60', f'NM:i:{d}', f'AS:i:{2 * matches - 4 * d}', f'cg:Z:{cigar}']
        paf.append('\t'.join(line))
        if r % 5 == 0:                                   # a second, worse alignment of the same read: dropped
            paf.append('\t'.join(line[:13] + [f'AS:i:{2 * matches - 4 * d - 50}', 'cg:Z:' + f'{re_ - rs}M']))
    paf.append('\t'.join(['read_short', '90', '0', '90', '+', 'ctg_one', '12000', '10', '100', '90', '90', '60', 'AS:i:180', 'cg:Z:90M']))
    paf.append('\t'.join(['read_bad', '400', '0', '400', '+', 'ctg_one', '12000', '500', '900', '250', '400', '60', 'AS:i:100', 'cg:Z:400M']))
    reads.append(('read_short', dna(90), 'I' * 90))
    reads.append(('read_bad', dna(400), 'I' * 400))
    with open(os.path.join(folder, 'reads.fastq'), 'w') as f:
        for name, seq, qual in reads:
            f.write(f'@{name} extra words\n{seq}\n+\n{qual}\n')
    with open(os.path.join(folder, 'aln.paf'), 'w') as f:
        f.write('\n'.join(paf) + '\n')
    out = {'error': [], 'qscore': []}
    for kind, fn in (('error', ref_em.make_error_model), ('qscore', ref_qm.make_qscore_model)):
        for case in MODEL_BUILDER_CASES[kind]:
            args = types.SimpleNamespace(reference=os.path.join(folder, 'ref.fasta'), reads=os.path.join(folder, 'reads.fastq'),
                                         alignment=os.path.join(folder, 'aln.paf'), **case)
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                fn(args, output=io.StringIO())
            out[kind].append({'args': case, 'text': buf.getvalue()})
            print(f'  model builder {kind} {case}: {len(buf.getvalue())} characters, {buf.getvalue().count(chr(10))} lines')
    dump('model_builder.json.gz', out)


if __name__ == '__main__':
    os.makedirs(GOLDEN, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == 'model_builders':
        make_model_builders()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'random_change':
        make_random_change()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'sequence_fragment_bound':
        make_sequence_fragment_bound()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'sequence_fragment':
        make_sequence_fragment()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'qscore_top_rows':
        make_qscore_top_rows()
        sys.exit(0)
    make_misc()
    make_align_kmers()
    make_fragments()
    make_build_fragment()
    make_sequence_fragment()
    make_random_change()
    make_qscore_top_rows()
    make_model_builders()
    make_sequence_fragment_bound()

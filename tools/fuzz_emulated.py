"""
Differential fuzzing of the product kernels (interpreted on the CPU, tests/emu_engine.py) against the oracle:
random simulate parameters, models, references, seeds and kernel routes; every mismatch is logged with the
arguments that reproduce it.   python tools/fuzz_emulated.py <seconds> <worker id> [log dir] [quad]
"""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'oracle'), os.path.join(REPO, 'tests')):
    sys.path.insert(0, p)

import helpers as H  # noqa: E402
from badread_amd.engine import SimParams  # noqa: E402

MODELS = ['random', 'nanopore2018', 'nanopore2020', 'nanopore2023', 'pacbio2016', 'pacbio2021']
QMODELS = ['random', 'ideal', 'nanopore2018', 'nanopore2020', 'nanopore2023', 'pacbio2016', 'pacbio2021']
ROUTES = [{}, {'BRX_TAIL_READS': '0', 'BRX_HEAD_READS': '0'}, {'BRX_TAIL_READS': '5', 'BRX_HEAD_READS': '0'},
          {'BRX_TB_WINDOW': '-1'}, {'BRX_TB_WINDOW': '0'}, {'BRX_TB_WINDOW': '1'},
          {'BRX_TAIL_READS': '3'}, {'BRX_WAVES_PER_CU': '1'}, {'BRX_WAVES_PER_CU': '2', 'BRX_TB_WINDOW': '-1'},      # few slabs per band class
          {'BRX_FIN_LANES': '0'}, {'BRX_TAIL_READS': '2', 'BRX_HEAD_READS': '2', 'BRX_FIN_LANES': '0'}]
# routes of the final stage with four alignments per wave (k_fin_quad<1>; on by default): without the one-read-per-lane class so that
# narrow bands go through it too, window misses repeated by k_fin_align, few slabs, and switched off
QUAD_ROUTES = [{'BRX_FIN_LANES': '0'}, {'BRX_FIN_QUAD': '1'}, {'BRX_FIN_QUAD': '0'},
               {'BRX_FIN_LANES': '0', 'BRX_TB_WINDOW': '-1'}, {'BRX_FIN_LANES': '0', 'BRX_TB_WINDOW': '0', 'BRX_WAVES_PER_CU': '1'},
               {'BRX_FIN_LANES': '0', 'BRX_TB_WINDOW': '1', 'BRX_TAIL_READS': '2', 'BRX_HEAD_READS': '3'}]
for _r in QUAD_ROUTES:
    _r.setdefault('BRX_QUAD_MIN_READS', '0')          # the class is used only when it holds thousands of reads by default
ROUTES += QUAD_ROUTES
# round 6: the bulk passes as {k_mut_apply, k_mut_post, k_pass_lists} (brx_passes.h) -- every read through the passes, passes with
# an in-place tail that takes reads over in any state; `defines` builds a variant of the kernels whose rings are kept nearly
# empty, so that reads go hungry (a pass without an alignment) all the time
PASS_ROUTES = [{'BRX_TAIL_READS': t, 'BRX_HEAD_READS': h} for t in ('0', '2', '5') for h in ('0', '2')]
PASS_DEFINES = [(), ('-DBRX_SV_STOCK=3u', '-DBRX_SV_CAP=128u', '-DBRX_POST_U=1'), ('-DBRX_SV_STOCK=1u', '-DBRX_SV_CAP=256u', '-DBRX_POST_U=3')]


# round 6, last day: the arena (mutate-only buffers in a releasable top region, the bulk set's slabs over them), sets short of room
# (halving by freed room per added time, then giving back), giant stores of the widest class in a class of their own: small arenas of
# random size (a short arena is grown and the batch repeated: that path too), two sets or one, builds whose giants start at 32 KB / 2 MB
ARENA_ROUTES = [{'BRX_LANES_MIN_READS': '0'}, {'BRX_TAIL_READS': '0', 'BRX_HEAD_READS': '0', 'BRX_LANES_MIN_READS': '0'},
                {'BRX_TAIL_READS': '4', 'BRX_HEAD_READS': '5', 'BRX_TB_WINDOW': '0'}, {'BRX_TAIL_READS': '0', 'BRX_HEAD_READS': '6', 'BRX_WAVES_PER_CU': '4'},
                {'BRX_TAIL_READS': '0', 'BRX_HEAD_READS': '0', 'BRX_FIN_HEAD_READS': '7', 'BRX_FIN_LANES': '0', 'BRX_QUAD_MIN_READS': '0'},
                {'BRX_TAIL_READS': '3', 'BRX_HEAD_READS': '4', 'BRX_MUTATE_PASSES': '1', 'BRX_TB_WINDOW': '0', 'BRX_FIN_LANES': '0'}]
ARENA_DEFINES = [(), ('-DBRX_GIANT_UNITS=4096ull',), ('-DBRX_GIANT_UNITS=262144ull',)]


def draw_case(rng, focus=None):
    mode = int(rng.integers(0, 3))
    p = dict(frag_mean=float(rng.choice([60, 300, 900, 2500, 4500] if focus not in ('quad', 'arena') else [900, 2500, 4500, 7000])), frag_stdev=float(rng.choice([0, 50, 800, 3000])),
             identity_mode=mode, start_rate=float(rng.choice([0, 0.5, 0.9, 1.0])), start_amount=float(rng.choice([0.1, 0.6, 1.0])),
             end_rate=float(rng.choice([0, 0.5, 1.0])), end_amount=float(rng.choice([0.2, 0.9, 1.0])),
             start_adapter=str(rng.choice(['', 'AATGTACTTCGTTCAGTTACGTATTGCT', 'ACGT'])),
             end_adapter=str(rng.choice(['', 'GCAATACGTAACTGAACGAAGT', 'T'])),
             junk_rate=float(rng.choice([0, 0.01, 0.2])), random_rate=float(rng.choice([0, 0.01, 0.2])),
             chimera_rate=float(rng.choice([0, 0.01, 0.3])), glitch_rate=float(rng.choice([0, 200, 10000])),
             glitch_size=float(rng.choice([0, 1, 25])), glitch_skip=float(rng.choice([0, 1, 25])))
    if mode == 0:
        p['id_max'] = float(rng.choice([1.0, 0.97, 0.85, 0.7]))
    elif mode == 1:
        mean, mx, sd = float(rng.choice([0.8, 0.9, 0.95])), float(rng.choice([0.96, 0.99, 1.0])), float(rng.choice([0.01, 0.025, 0.06]))
        mean = min(mean, mx - 0.005)
        a = ((1 - mean / mx) / (sd / mx) ** 2 - 1 / (mean / mx)) * (mean / mx) ** 2
        p.update(id_a=max(a, 0.5), id_b=max(a * (1 / (mean / mx) - 1), 0.5), id_max=mx)
    else:
        p.update(id_a=float(rng.choice([10.0, 20.0, 30.0])), id_b=float(rng.choice([1.0, 3.0, 6.0])), id_max=1.0)
    return dict(params=p, em=str(rng.choice(MODELS)), qm=str(rng.choice(QMODELS)), seed=int(rng.integers(0, 2 ** 40)),
                first=int(rng.integers(0, 10 ** 6)), n=int(rng.choice([1, 7, 16, 30])), with_n=bool(rng.integers(0, 2)),
                route=dict((QUAD_ROUTES if focus == 'quad' else PASS_ROUTES if focus == 'pass' else ROUTES)[int(rng.integers(0, len(QUAD_ROUTES if focus == 'quad' else PASS_ROUTES if focus == 'pass' else ROUTES)))]),
                defines=list(PASS_DEFINES[int(rng.integers(0, len(PASS_DEFINES)))]) if focus == 'pass' else []) if focus != 'arena' else \
        dict(params=p, em=str(rng.choice(MODELS)), qm=str(rng.choice(QMODELS)), seed=int(rng.integers(0, 2 ** 40)),
             first=int(rng.integers(0, 10 ** 6)), n=int(rng.choice([7, 16, 30])), with_n=bool(rng.integers(0, 2)),
             route=dict(ARENA_ROUTES[int(rng.integers(0, len(ARENA_ROUTES)))]), defines=list(ARENA_DEFINES[int(rng.integers(0, len(ARENA_DEFINES)))]),
             scratch_mb=int(rng.choice([5, 7, 9, 12, 16, 24, 48])))


def run_case(case):
    import emu_engine as EE
    for k in list(os.environ):
        if k.startswith('BRX_'):
            del os.environ[k]
    os.environ.update(case['route'])
    pref, _ = H.small_reference(with_n=case['with_n'])
    p = SimParams(**case['params'])
    emu = H.configure(EE.EmuEngine(int(case.get('scratch_mb', 512)) << 20, tuple(case.get('defines', ()))), pref, case['em'], case['qm'], p)
    orc = H.configure(H.oracle_engine(), pref, case['em'], case['qm'], p)
    out_h, st_h = emu.simulate_batch(case['seed'], case['first'], case['n'], allow_nofrag=True)
    out_o, st_o = orc.simulate_batch(case['seed'], case['first'], case['n'], allow_nofrag=True)
    bad = [f for f in st_h.dtype.names if not (st_h[f] == st_o[f]).all()]
    if bytes(out_h) != bytes(out_o):
        bad.append('bytes')
    emu.close()
    return bad, int(st_o['seq_len'].sum())


def main():
    seconds, wid = float(sys.argv[1]), int(sys.argv[2])
    logdir = sys.argv[3] if len(sys.argv) > 3 else '/tmp/brx_fuzz'
    focus = sys.argv[4] if len(sys.argv) > 4 else None          # 'quad': only the k_fin_quad routes, longer reads; 'pass': the bulk passes; 'arena': small arenas, giants' class
    os.makedirs(logdir, exist_ok=True)
    rng = np.random.default_rng(1000 + wid)
    t0, cases, bases, fails = time.time(), 0, 0, 0
    with open(os.path.join(logdir, f'worker{wid}.log'), 'a') as log:
        while time.time() - t0 < seconds:
            case = draw_case(rng, focus)
            try:
                bad, nb = run_case(case)
            except BaseException as ex:           # the library reports, the harness records
                bad, nb = [f'exception {type(ex).__name__}: {ex}'], 0
            cases += 1
            bases += nb
            if bad:
                fails += 1
                log.write(json.dumps({'bad': bad, 'case': case}) + '\n')
                log.flush()
        log.write(json.dumps({'summary': {'cases': cases, 'bases': bases, 'failures': fails, 'seconds': time.time() - t0}}) + '\n')
    print(wid, cases, bases, fails)


if __name__ == '__main__':
    main()

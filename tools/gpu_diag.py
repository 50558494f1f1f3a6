"""
Staged bring-up diagnostics for a GPU box: each stage runs in its own process under a timeout, so a
hang or a crash in one stage still leaves a readable log.  Usage: python tools/gpu_diag.py [stage ...]
"""
import faulthandler
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'oracle'), os.path.join(REPO, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def stage_torch():
    import torch
    print('torch', torch.__version__, 'hip', torch.version.hip, 'avail', torch.cuda.is_available(), flush=True)
    print('device', torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0), flush=True)
    x = torch.arange(8, device='cuda')
    print('sum', int(x.sum().item()), flush=True)


def stage_create():
    import torch
    torch.zeros(1, device='cuda')
    from badread_amd.engine import HipEngine
    e = HipEngine(0, scratch_bytes=1 << 26)
    print('created', e.lib.brx_version(), flush=True)


def stage_create_notorch_first():
    import ctypes
    from badread_amd import engine
    lib = engine.load_library()
    ctx = ctypes.c_void_p()
    rc = lib.brx_create(0, ctypes.byref(ctx))
    print('rc', rc, lib.brx_last_error(None), flush=True)


def stage_align():
    import numpy as np
    import helpers as H
    import pyoracle
    eng = H.hip_engine()
    print('engine up', flush=True)
    qs = [b'ACGTACGTAC', b'GATTACA', b'A' * 100]
    ts = [b'ACGTTCGTAC', b'GATACA', b'A' * 90 + b'C' * 5]
    ops, dist, ncols, nmatch = eng.align_batch(qs, ts)
    print('dist', dist, 'ncols', ncols, flush=True)
    for q, t, d in zip(qs, ts, dist):
        assert pyoracle.align(q, t)[0] == d
    rng = np.random.default_rng(1)
    q = H.random_dna(rng, 5000)
    t = H.mutate_seq(rng, q, 0.05)
    t0 = time.time()
    ops, dist, ncols, nmatch = eng.align_batch([q.encode()], [t.encode()])
    print('5kb dist', dist, pyoracle.align(q.encode(), t.encode())[0], 'sec', time.time() - t0, flush=True)


def stage_sim_small():
    import helpers as H
    from badread_amd.engine import SimParams
    pref, _ = H.small_reference()
    for em, qm, mean in (('random', 'ideal', 500), ('nanopore2023', 'nanopore2023', 3000)):
        p = SimParams(frag_mean=mean, frag_stdev=mean * 0.8)
        hip = H.configure(H.hip_engine(), pref, em, qm, p)
        orc = H.configure(H.oracle_engine(), pref, em, qm, p)
        t0 = time.time()
        out_h, st_h = hip.simulate_batch(42, 0, 64)
        print(em, 'hip done', time.time() - t0, 'stage ms', hip.stage_ms(), flush=True)
        out_o, st_o = orc.simulate_batch(42, 0, 64)
        same = bytes(out_h) == bytes(out_o)
        print(em, 'bytes equal', same, len(out_h), len(out_o), flush=True)
        for f in st_h.dtype.names:
            if not (st_h[f] == st_o[f]).all():
                bad = (st_h[f] != st_o[f]).nonzero()[0]
                print('  field', f, 'differs at', bad[:8], st_h[f][bad[:4]], st_o[f][bad[:4]], flush=True)


STAGES = {'torch': stage_torch, 'create': stage_create, 'create2': stage_create_notorch_first,
          'align': stage_align, 'sim': stage_sim_small}

if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--run':
        faulthandler.enable()
        faulthandler.dump_traceback_later(int(os.environ.get('DIAG_DUMP_AFTER', '50')), exit=False)
        STAGES[sys.argv[2]]()
        sys.exit(0)
    names = sys.argv[1:] or list(STAGES)
    for name in names:
        print(f'===== stage {name}', flush=True)
        t0 = time.time()
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--run', name], timeout=int(os.environ.get('DIAG_TIMEOUT', '90')),
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            print(r.stdout[-6000:], flush=True)
            print(f'===== stage {name} rc={r.returncode} {time.time() - t0:.1f}s', flush=True)
        except subprocess.TimeoutExpired as ex:
            out = ex.stdout.decode() if isinstance(ex.stdout, bytes) else (ex.stdout or '')
            print(out[-6000:], flush=True)
            print(f'===== stage {name} TIMEOUT {time.time() - t0:.1f}s', flush=True)

"""
oracle/pyoracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes wrapper around oracle/_ref/liboracle.so (built from oracle/brx_oracle.c + myers_ref.c by
oracle/Makefile) exposing the same small interface as badread_amd.engine.HipEngine, with HOST
pointers in the same descriptor structs.  Allowed importers: tests/, __graft_entry__.smoke(),
bench.py's cpu_baseline leg.  The product package never imports this module.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(_HERE)
if _REPO not in sys.path:
    sys.path.insert(0, _REPO)

from badread_amd.engine import (EngineBase, BrxReference, BrxErrorModel, BrxQScoreModel, BrxSimParams,  # noqa: E402
                                READ_STATS_DTYPE)

LIB = os.path.join(_HERE, '_ref', 'liboracle.so')


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ('brx_oracle.c', 'myers_ref.c')] + \
           [os.path.join(_REPO, 'include', f) for f in ('brx.h', 'brx_spec.h')]
    stale = force or not os.path.exists(LIB) or \
        any(os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs)
    if stale:
        subprocess.check_call(['make', '-s', '-C', _HERE, '-B', '_ref/liboracle.so', '_ref/libmyers_ref.so'])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(LIB)
        L.orc_create.restype = ctypes.c_void_p
        L.orc_destroy.argtypes = [ctypes.c_void_p]
        for name, st in (('orc_set_reference', BrxReference), ('orc_set_error_model', BrxErrorModel),
                         ('orc_set_qscore_model', BrxQScoreModel), ('orc_set_params', BrxSimParams)):
            getattr(L, name).argtypes = [ctypes.c_void_p, ctypes.POINTER(st)]
            getattr(L, name).restype = None
        L.orc_simulate_batch.restype = ctypes.c_int64
        L.orc_simulate_batch.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32,
                                         ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        L.orc_sequence_fragments.restype = ctypes.c_int64
        L.orc_sequence_fragments.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32,
                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_int64, ctypes.c_void_p]
        L.orc_align_myers.restype = ctypes.c_int64
        L.orc_align_myers.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_char_p, ctypes.c_int64,
                                      ctypes.c_int64, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)]
        L.orc_align_dp.restype = ctypes.c_int64
        L.orc_align_dp.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_char_p, ctypes.c_int64,
                                   ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)]
        L.orc_plan_probe.restype = ctypes.c_int
        L.orc_plan_probe.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int,
                                     ctypes.POINTER(ctypes.c_int), ctypes.c_void_p, ctypes.c_int,
                                     ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint64),
                                     ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint32)]
        L.orc_fragment_probe.restype = ctypes.c_int64
        L.orc_fragment_probe.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int64]
        L.orc_ref_slice.restype = None
        L.orc_ref_slice.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64,
                                    ctypes.c_uint64, ctypes.c_void_p]
        L.orc_choose_alt_probe.restype = ctypes.c_int
        L.orc_choose_alt_probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32,
                                           ctypes.c_void_p, ctypes.c_void_p]
        L.orc_qscore_rows_probe.restype = ctypes.c_int
        L.orc_qscore_rows_probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_plan_trace.restype = ctypes.c_int
        L.orc_plan_trace.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.orc_draw4.restype = None
        L.orc_draw4.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_void_p]
        L.orc_sample.restype = None
        L.orc_sample.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
        _lib = L
    return _lib


TRACE_KINDS = ('U', 'LENGTH', 'CONTIG', 'START', 'JUNKLEN', 'JUNKUNIT', 'SERIAL', 'ADAPTLEN', 'GEO', 'IDENTITY')
SAMPLE_KINDS = {'gamma': 0, 'beta': 1, 'normal': 2, 'geometric': 3, 'uniform': 4, 'below': 5, 'log': 6, 'exp': 7}


def draw4(seed, read, stream, index):
    out = (ctypes.c_uint32 * 4)()
    lib().orc_draw4(seed, read, stream, index, out)
    return list(out)


def sample(kind, a, b, seed, n):
    """n draws of one sampler of include/brx_spec.h (draw i uses read index i of the PLAN stream)."""
    out = np.zeros(n, dtype=np.float64)
    lib().orc_sample(SAMPLE_KINDS[kind], float(a), float(b), seed, n, out.ctypes.data)
    return out


def align(query, target, k=-1, want_ops=True, dp=False):
    """(distance, ops uint8 array) for bytes-like query/target; canonical traceback."""
    q, t = bytes(query), bytes(target)
    ops = (ctypes.c_uint8 * (len(q) + len(t) + 1))() if want_ops else None
    n = ctypes.c_int64(0)
    if dp:
        d = lib().orc_align_dp(q, len(q), t, len(t), ops, ctypes.byref(n))
    else:
        d = lib().orc_align_myers(q, len(q), t, len(t), k, ops, ctypes.byref(n))
    return int(d), (np.frombuffer(ops, dtype=np.uint8, count=n.value).copy() if want_ops and d >= 0 else None)


def oracle_align_batch(queries, targets):
    """Aligner callable for badread_amd.error_model.ErrorModel(aligner=...) in CPU-only tests."""
    return [align(q, t)[1] for q, t in zip(queries, targets)]


class OracleEngine(EngineBase):
    """The CPU checker behind the HipEngine interface (host pointers, single thread)."""

    def __init__(self):
        super().__init__()
        self.L = lib()
        self.ctx = ctypes.c_void_p(self.L.orc_create())
        self._structs = {}

    def __del__(self):
        try:
            self.L.orc_destroy(self.ctx)
        except Exception:
            pass

    def _upload(self, arr):
        arr = np.ascontiguousarray(arr)
        if arr.size == 0:
            arr = np.zeros(8, dtype=np.uint8)
        return arr.ctypes.data, arr

    def set_reference(self, pref, cum_weight=None):
        s = self._fill_reference(pref, cum_weight)
        self.L.orc_set_reference(self.ctx, ctypes.byref(s))
        self._structs['ref'] = s

    def set_error_model(self, tables):
        s = self._fill_error_model(tables)
        self.L.orc_set_error_model(self.ctx, ctypes.byref(s))
        self._structs['em'] = s

    def set_qscore_model(self, tables):
        s = self._fill_qscore_model(tables)
        self.L.orc_set_qscore_model(self.ctx, ctypes.byref(s))
        self._structs['qm'] = s

    def set_params(self, params):
        s = self._fill_params(params)
        self.L.orc_set_params(self.ctx, ctypes.byref(s))
        self._structs['params'] = s

    def clone(self):
        """A second checker context over the same host tables (mirrors HipEngine.clone for the driver tests)."""
        other = OracleEngine()
        other._keep = dict(self._keep)
        other.sym = self.sym
        setters = {'ref': self.L.orc_set_reference, 'em': self.L.orc_set_error_model,
                   'qm': self.L.orc_set_qscore_model, 'params': self.L.orc_set_params}
        for key, s in self._structs.items():
            setters[key](other.ctx, ctypes.byref(s))
            other._structs[key] = s
        return other

    def close(self):
        pass

    def simulate_batch(self, seed, first_read, n_reads, allow_nofrag=False):
        stats = np.zeros(n_reads, dtype=READ_STATS_DTYPE)
        need = self.L.orc_simulate_batch(self.ctx, seed, first_read, n_reads, None, 0, stats.ctypes.data)
        out = np.zeros(max(int(need), 1), dtype=np.uint8)
        got = self.L.orc_simulate_batch(self.ctx, seed, first_read, n_reads, out.ctypes.data, len(out), stats.ctypes.data)
        assert got == need
        return out[:got], stats

    def sequence_fragments(self, seed, first_read, frags, targets):
        n = len(frags)
        off = np.zeros(n + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(f) for f in frags])
        flat = np.ascontiguousarray(np.concatenate(frags).astype(np.uint8)) if n else np.zeros(1, np.uint8)
        tg = np.ascontiguousarray(targets, dtype=np.float64)
        stats = np.zeros(n, dtype=READ_STATS_DTYPE)
        cap = int(off[-1]) * 4 + 4096 * n + 4096
        for _ in range(4):
            out = np.zeros(cap, dtype=np.uint8)
            got = self.L.orc_sequence_fragments(self.ctx, seed, first_read, n, flat.ctypes.data, off.ctypes.data,
                                                tg.ctypes.data, out.ctypes.data, cap, stats.ctypes.data)
            if got >= 0:
                break
            cap = -got + 64
        res = []
        for st in stats:
            o, L = int(st['rec_off']), int(st['seq_len'])
            res.append((out[o:o + L].copy(), out[o + L:o + 2 * L].copy()))
        return res, stats

    def align_batch(self, queries, targets, k_hint=None, want_ops=True):
        ops, dist, ncols, nmatch = [], [], [], []
        for q, t in zip(queries, targets):
            d, o = align(q, t)
            dist.append(d)
            ncols.append(len(o))
            nmatch.append(int((o == 0).sum()))
            ops.append(o)
        return (ops if want_ops else None), np.array(dist, np.int32), np.array(ncols, np.int32), np.array(nmatch, np.int32)

    # ---- probes -------------------------------------------------------------------------------
    def plan(self, seed, read):
        segs = np.zeros(5 * 4096, dtype=np.uint64)
        pieces = np.zeros(6 * 256, dtype=np.uint64)
        ns, npc = ctypes.c_int(0), ctypes.c_int(0)
        flen, tgt, status = ctypes.c_uint64(0), ctypes.c_double(0), ctypes.c_uint32(0)
        self.L.orc_plan_probe(self.ctx, seed, read, segs.ctypes.data, 4096, ctypes.byref(ns), pieces.ctypes.data, 256,
                              ctypes.byref(npc), ctypes.byref(flen), ctypes.byref(tgt), ctypes.byref(status))
        return dict(segs=segs[:5 * ns.value].reshape(-1, 5).copy(), pieces=pieces[:6 * npc.value].reshape(-1, 6).copy(),
                    frag_len=flen.value, target=tgt.value, status=status.value)

    def plan_trace(self, seed, read):
        """[(kind name, value), ...]: every primitive random decision plan_read took for this read."""
        cap = 4096
        kinds = np.zeros(cap, dtype=np.int32)
        vals = np.zeros(cap, dtype=np.float64)
        n = self.L.orc_plan_trace(self.ctx, seed, read, kinds.ctypes.data, vals.ctypes.data, cap)
        assert n <= cap
        return [(TRACE_KINDS[k], float(v)) for k, v in zip(kinds[:n], vals[:n])]

    def fragment(self, seed, read):
        cap = 1 << 22
        buf = np.zeros(cap, dtype=np.uint8)
        L = self.L.orc_fragment_probe(self.ctx, seed, read, buf.ctypes.data, cap)
        return buf[:L].copy()

    def ref_slice(self, contig, strand, start, length):
        buf = np.zeros(max(length, 1), dtype=np.uint8)
        self.L.orc_ref_slice(self.ctx, contig, 0 if strand == '+' else 1, start, length, buf.ctypes.data)
        return buf[:length]

    def choose_alt(self, kmer_codes, w2, w3):
        k = len(kmer_codes)
        kc = np.ascontiguousarray(kmer_codes, dtype=np.uint8)
        lens = np.zeros(k, dtype=np.uint8)
        chars = np.zeros(k * 128 + 8, dtype=np.uint8)
        changed = self.L.orc_choose_alt_probe(self.ctx, kc.ctypes.data, w2, w3, lens.ctypes.data, chars.ctypes.data)
        out, p = [], 0
        for L in lens:
            out.append(chars[p:p + L].copy())
            p += int(L)
        return bool(changed), out

    def qscore_rows(self, ops):
        ops = np.ascontiguousarray(ops, dtype=np.uint8)
        m = int((ops != 3).sum())
        rows = np.zeros(max(m, 1), dtype=np.int64)
        used = np.zeros(max(m, 1), dtype=np.int32)
        self.L.orc_qscore_rows_probe(self.ctx, ops.ctypes.data, len(ops), rows.ctypes.data, used.ctypes.data)
        return rows[:m], used[:m]

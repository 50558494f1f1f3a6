"""
oracle/shim/edlib -- TEST INFRASTRUCTURE ONLY.

A stand-in for the third-party `edlib` Python package (absent from this container; the reference
imports it at badread/simulate.py:17, error_model.py:19, qscore_model.py:19).  With this directory
on PYTHONPATH the UNMODIFIED reference under /root/reference imports and runs, which is how the
oracle is pinned (all 304 reference tests) and how tests/golden/ fixtures were generated.

Only the surface the reference touches is provided:
    edlib.align(query, target, mode="NW", task="distance", k=-1, additionalEqualities=None)
        -> {'editDistance', 'alphabetLength', 'locations', 'cigar'}
The arithmetic is oracle/myers_ref.c (block Myers, canonical traceback: up/'I', left/'D', diagonal).
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE = os.path.normpath(os.path.join(_HERE, '..', '..'))


def _load():
    so = os.path.join(_ORACLE, '_ref', 'libmyers_ref.so')
    src = os.path.join(_ORACLE, 'myers_ref.c')
    if not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(['make', '-s', '-C', _ORACLE, '_ref/libmyers_ref.so'])
    lib = ctypes.CDLL(so)
    lib.orc_align_myers.restype = ctypes.c_int64
    lib.orc_align_myers.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_char_p, ctypes.c_int64,
                                    ctypes.c_int64, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)]
    lib.orc_ops_to_cigar.restype = ctypes.c_int64
    lib.orc_ops_to_cigar.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_char_p, ctypes.c_int64]
    return lib


_lib = _load()


def _as_bytes(s):
    if isinstance(s, str):
        return s.encode('latin-1')
    return bytes(s)


def align(query, target, mode="NW", task="distance", k=-1, additionalEqualities=None):
    if mode != "NW":
        raise NotImplementedError('edlib shim: only mode="NW" is used by Badread')
    if additionalEqualities:
        raise NotImplementedError('edlib shim: additionalEqualities not supported')
    q, t = _as_bytes(query), _as_bytes(target)
    want_path = (task == 'path')
    ops = (ctypes.c_uint8 * (len(q) + len(t) + 1))() if want_path else None
    n_ops = ctypes.c_int64(0)
    d = _lib.orc_align_myers(q, len(q), t, len(t), int(k), ops, ctypes.byref(n_ops))
    result = {'editDistance': int(d), 'alphabetLength': len(set(q) | set(t)),
              'locations': [(0 if want_path else None, len(t) - 1)] if d >= 0 else [],
              'cigar': None}
    if want_path and d >= 0:
        cap = 24 * max(1, n_ops.value) + 16
        buf = ctypes.create_string_buffer(cap)
        _lib.orc_ops_to_cigar(ops, n_ops.value, buf, cap)
        result['cigar'] = buf.value.decode()
    return result

"""
TEST INFRASTRUCTURE: the product's HIP sources interpreted on the CPU.

`build()` compiles badread_amd/csrc/brx_hip.hip (+ brx_kernels.h, brx_mutate.h, brx_align.h) with g++ against
tests/native/emu/hip/hip_runtime.h -- every lane a fiber, cross-lane operations as rendezvous, device memory = host
memory -- into a library in a temporary directory, and `EmuEngine` drives it through the same ctypes binding as
HipEngine (CPU torch tensors stand in for device buffers).  tests/test_emulated_device.py runs the GPU parity checks
against it, so the kernels' logic is exercised by `pytest -m "not gpu"` too.  Nothing under badread_amd/ knows this
exists; it is ~1000x slower than the oracle and is not a fallback.
"""
import ctypes
import os
import subprocess
import tempfile

from badread_amd import engine as E

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
_lib = None


_variants = {}


def build(defines=()):
    """The interpreted library; `defines` selects a build variant of the kernels (-D macros for the product sources)."""
    global _lib
    key = tuple(defines)
    if key in _variants:
        return _variants[key]
    out_dir = tempfile.mkdtemp(prefix='brx_emu_')
    out = os.path.join(out_dir, 'libbrx_emu.so')
    cmd = [os.environ.get('CXX', 'g++'), '-O1', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off', '-w', '-x', 'c++',
           '-I', os.path.join(HERE, 'native', 'emu'), '-I', os.path.join(REPO, 'include')] + list(defines) + \
          [os.path.join(REPO, 'badread_amd', 'csrc', 'brx_hip.hip'), '-o', out]
    subprocess.check_call(cmd)
    _variants[key] = E.bind_library(ctypes.CDLL(out))
    if not key:
        _lib = _variants[key]
    return _variants[key]


class EmuEngine(E.HipEngine):
    """HipEngine's binding over the emulated library; buffers are CPU tensors."""

    def __init__(self, scratch_bytes=1 << 28, defines=()):
        E.EngineBase.__init__(self)
        import torch
        self.torch = torch
        self.device = torch.device('cpu')
        self.defines = tuple(defines)
        self.lib = build(self.defines)
        ctx = ctypes.c_void_p()
        rc = self.lib.brx_create(0, ctypes.byref(ctx))
        if rc != 0:
            raise E.BrxError(rc, self.lib.brx_last_error(None).decode('latin-1', 'replace'))
        self.ctx = ctx
        self._scratch = self._out = self._stats = None
        self._structs = {}
        self._ensure_scratch(scratch_bytes)

    def _stream(self):
        return ctypes.c_void_p(None)

    def clone(self, scratch_bytes=None):
        """Second context over the same tables (what HipEngine.clone does for batches in flight)."""
        other = EmuEngine(scratch_bytes or self._scratch.numel(), self.defines)
        other._keep = dict(self._keep)
        other.sym = self.sym
        setters = {'ref': self.lib.brx_set_reference, 'em': self.lib.brx_set_error_model,
                   'qm': self.lib.brx_set_qscore_model, 'params': self.lib.brx_set_params}
        for key, st in self._structs.items():
            other._check(setters[key](other.ctx, ctypes.byref(st)))
            other._structs[key] = st
        return other

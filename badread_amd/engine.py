"""
ctypes binding of libbrx_hip.so (C-ABI: include/brx.h) and the host-side driver of the HIP path.

This is the ONLY compute backend of the product.  There is no CPU fallback: importing the engine
without the built library, or constructing it without a ROCm device, raises immediately.  Device
buffers are PyTorch-ROCm tensors (memory + streams only); every kernel is hand-written HIP in
badread_amd/csrc/.

`EngineBase` also defines the small interface (`simulate_batch`, `sequence_fragments`,
`align_batch`) that the tests' oracle-backed checker implements, so the host logic (sharding,
stop rule, ordering) can be exercised on CPU under gloo without touching the product path.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.realpath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libbrx_hip.so')

c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_u32p = ctypes.POINTER(ctypes.c_uint32)
c_u64p = ctypes.POINTER(ctypes.c_uint64)
c_f64p = ctypes.POINTER(ctypes.c_double)


class BrxContig(ctypes.Structure):
    _fields_ = [('base_off', ctypes.c_uint64), ('length', ctypes.c_uint32), ('flags', ctypes.c_uint32),
                ('name_off', ctypes.c_uint32), ('name_len', ctypes.c_uint32)]


class BrxException(ctypes.Structure):
    _fields_ = [('start', ctypes.c_uint64), ('end', ctypes.c_uint64), ('code', ctypes.c_uint32),
                ('pad_', ctypes.c_uint32)]


class BrxReference(ctypes.Structure):
    _fields_ = [('d_packed', ctypes.c_void_p), ('n_bases', ctypes.c_uint64),
                ('d_contigs', ctypes.c_void_p), ('n_contigs', ctypes.c_uint32),
                ('d_exceptions', ctypes.c_void_p), ('n_exceptions', ctypes.c_uint32),
                ('d_names', ctypes.c_void_p), ('names_len', ctypes.c_uint32),
                ('sym', ctypes.c_uint8 * 16), ('comp', ctypes.c_uint8 * 16),
                ('d_cum_weight', ctypes.c_void_p), ('total_weight', ctypes.c_double)]


class BrxErrorModel(ctypes.Structure):
    _fields_ = [('k', ctypes.c_int32), ('type', ctypes.c_int32), ('n_rows', ctypes.c_uint32),
                ('n_alts', ctypes.c_uint32), ('pool_len', ctypes.c_uint32), ('pad_', ctypes.c_uint32),
                ('d_row_off', ctypes.c_void_p), ('d_self_thr', ctypes.c_void_p), ('d_thr', ctypes.c_void_p),
                ('d_desc', ctypes.c_void_p), ('d_pool', ctypes.c_void_p),
                ('d_rowx', ctypes.c_void_p), ('d_altx', ctypes.c_void_p)]


class BrxQScoreModel(ctypes.Structure):
    _fields_ = [('k', ctypes.c_int32), ('gap_bits', ctypes.c_int32), ('hash_size', ctypes.c_uint32),
                ('n_rows', ctypes.c_uint32), ('n_entries', ctypes.c_uint32), ('pad_', ctypes.c_uint32),
                ('d_hash_key', ctypes.c_void_p), ('d_hash_row', ctypes.c_void_p), ('d_row_off', ctypes.c_void_p),
                ('d_thr', ctypes.c_void_p), ('d_score', ctypes.c_void_p)]


class BrxSimParams(ctypes.Structure):
    _fields_ = [('frag_mean', ctypes.c_double), ('frag_stdev', ctypes.c_double),
                ('gamma_k', ctypes.c_double), ('gamma_t', ctypes.c_double),
                ('identity_mode', ctypes.c_int32), ('pad0_', ctypes.c_int32),
                ('id_a', ctypes.c_double), ('id_b', ctypes.c_double), ('id_max', ctypes.c_double),
                ('start_rate', ctypes.c_double), ('start_amount', ctypes.c_double),
                ('end_rate', ctypes.c_double), ('end_amount', ctypes.c_double),
                ('d_start_adapter', ctypes.c_void_p), ('d_end_adapter', ctypes.c_void_p),
                ('start_adapter_len', ctypes.c_uint32), ('end_adapter_len', ctypes.c_uint32),
                ('junk_rate', ctypes.c_double), ('random_rate', ctypes.c_double), ('chimera_rate', ctypes.c_double),
                ('glitch_rate', ctypes.c_double), ('glitch_size', ctypes.c_double), ('glitch_skip', ctypes.c_double)]


READ_STATS_DTYPE = np.dtype([('status', '<u4'), ('frag_len', '<u4'), ('seq_len', '<u4'), ('n_cols', '<u4'),
                             ('n_match', '<u4'), ('padded_len', '<u4'), ('loop_count', '<u4'),
                             ('change_count', '<u4'), ('n_alignments', '<u4'), ('rec_len', '<u4'),
                             ('rec_off', '<u8'), ('target_identity', '<f8'), ('qerr_sum', '<f8')])
assert READ_STATS_DTYPE.itemsize == 64

RS_NOFRAG, RS_TOO_MANY_SEGS, RS_BAND, RS_QMISS, RS_EMPTY = 1, 2, 4, 8, 16
E_SCRATCH, E_OUTPUT, E_NOFRAG = -3, -4, -5
STAGE_NAMES = ('plan', 'build', 'mutate', 'scan', 'final', 'emit', 'align1', 'qscore')
# kernel classes of brx_last_kernel_stats (include/brx.h: BRX_KERN_*), with the names a rocprofv3 kernel trace shows
KERNEL_NAMES = ('k_plan_*', 'k_build', 'k_mut_lanes', 'k_mutate_seg', 'k_win_lane', 'k_win_wave', 'k_fin_join',
                'k_fin_align<1,1,1>', 'k_fin_align<2,2,2>', 'k_fin_align<4,4,4>', 'k_fin_align<16,8,65535>', 'k_fin_qscore',
                'k_emit+k_recsize', 'k_fin_lanes', 'k_fin_quad<1>', 'k_mut_post')


class BrxModelJob(ctypes.Structure):
    """brx_model_job of include/brx.h"""
    _fields_ = [('n_align', ctypes.c_uint32), ('k', ctypes.c_uint32), ('max_del', ctypes.c_uint32), ('n_ksizes', ctypes.c_uint32),
                ('n_cols', ctypes.c_uint64),
                ('d_seq', ctypes.c_void_p), ('d_qual', ctypes.c_void_p), ('d_ref', ctypes.c_void_p),
                ('d_part_type', ctypes.c_void_p), ('d_part_len', ctypes.c_void_p),
                ('d_part_col', ctypes.c_void_p), ('d_part_read', ctypes.c_void_p), ('d_part_ref', ctypes.c_void_p),
                ('d_align_part_off', ctypes.c_void_p), ('d_align_col_off', ctypes.c_void_p),
                ('d_rcol', ctypes.c_void_p), ('d_qcol', ctypes.c_void_p), ('d_fcol', ctypes.c_void_p),
                ('d_keys', ctypes.c_void_p), ('d_counts', ctypes.c_void_p), ('d_first', ctypes.c_void_p),
                ('table_mask', ctypes.c_uint64), ('d_flags', ctypes.c_void_p),
                ('d_spill', ctypes.c_void_p), ('spill_cap', ctypes.c_uint32)]


class BrxKernelStat(ctypes.Structure):
    _fields_ = [('launches', ctypes.c_uint32), ('ms', ctypes.c_float), ('bases', ctypes.c_double)]


class SimParams(object):
    """Plain-Python mirror of brx_sim_params; adapters are ACGT strings."""

    def __init__(self, frag_mean=15000.0, frag_stdev=13000.0, identity_mode=1, id_a=57.3838, id_b=2.41616,
                 id_max=0.99, start_rate=0.9, start_amount=0.6, end_rate=0.5, end_amount=0.2,
                 start_adapter='AATGTACTTCGTTCAGTTACGTATTGCT', end_adapter='GCAATACGTAACTGAACGAAGT',
                 junk_rate=0.01, random_rate=0.01, chimera_rate=0.01,
                 glitch_rate=10000.0, glitch_size=25.0, glitch_skip=25.0):
        self.frag_mean, self.frag_stdev = float(frag_mean), float(frag_stdev)
        if self.frag_stdev != 0.0:
            self.gamma_k = (self.frag_mean ** 2) / (self.frag_stdev ** 2)     # fragment_lengths.py:55-64
            self.gamma_t = (self.frag_stdev ** 2) / self.frag_mean
        else:
            self.gamma_k = self.gamma_t = 0.0
        self.identity_mode, self.id_a, self.id_b, self.id_max = int(identity_mode), float(id_a), float(id_b), float(id_max)
        self.start_rate, self.start_amount = float(start_rate), float(start_amount)
        self.end_rate, self.end_amount = float(end_rate), float(end_amount)
        self.start_adapter, self.end_adapter = start_adapter or '', end_adapter or ''
        self.junk_rate, self.random_rate, self.chimera_rate = float(junk_rate), float(random_rate), float(chimera_rate)
        self.glitch_rate, self.glitch_size, self.glitch_skip = float(glitch_rate), float(glitch_size), float(glitch_skip)

    def fill(self, struct, start_ptr, end_ptr):
        for name in ('frag_mean', 'frag_stdev', 'gamma_k', 'gamma_t', 'identity_mode', 'id_a', 'id_b', 'id_max',
                     'start_rate', 'start_amount', 'end_rate', 'end_amount', 'junk_rate', 'random_rate',
                     'chimera_rate', 'glitch_rate', 'glitch_size', 'glitch_skip'):
            setattr(struct, name, getattr(self, name))
        struct.d_start_adapter, struct.d_end_adapter = start_ptr, end_ptr
        struct.start_adapter_len, struct.end_adapter_len = len(self.start_adapter), len(self.end_adapter)
        return struct


def acgt_codes(seq):
    lut = np.full(256, 4, dtype=np.uint8)
    for i, ch in enumerate('ACGT'):
        lut[ord(ch)] = i
    return lut[np.frombuffer(seq.encode('latin-1'), dtype=np.uint8)].copy() if seq else np.zeros(1, np.uint8)


class EngineBase(object):
    """Interface shared by HipEngine and the tests' oracle-backed checker."""

    stats_dtype = READ_STATS_DTYPE

    def __init__(self):
        self._keep = {}
        self.sym = np.frombuffer(b'ACGTNNNNNNNNNNNN', dtype=np.uint8).copy()

    # subclasses provide: _upload(np_array) -> (pointer int, keepalive)
    def _fill_reference(self, pref, cum_weight=None):
        s = BrxReference()
        if cum_weight is None:
            _, cum_weight = pref.contig_weights()
        cum_weight = np.ascontiguousarray(cum_weight, dtype=np.float64)
        keep = []
        for field, arr in (('d_packed', pref.packed),
                           ('d_contigs', pref.contigs if len(pref.contigs) else np.zeros(1, pref.contigs.dtype)),
                           ('d_exceptions', pref.exceptions if len(pref.exceptions) else np.zeros(1, pref.exceptions.dtype)),
                           ('d_names', np.frombuffer(pref.names_pool or b'\0', dtype=np.uint8)),
                           ('d_cum_weight', cum_weight if len(cum_weight) else np.zeros(1))):
            ptr, k = self._upload(arr)
            setattr(s, field, ptr)
            keep.append(k)
        s.n_bases, s.n_contigs, s.n_exceptions = pref.n_bases, len(pref.contigs), len(pref.exceptions)
        s.names_len = len(pref.names_pool)
        for i in range(16):
            s.sym[i] = int(pref.sym[i])
            s.comp[i] = int(pref.comp[i])
        s.total_weight = float(cum_weight[-1]) if len(cum_weight) else 0.0
        self.sym = np.array(pref.sym, dtype=np.uint8)
        self._keep['ref'] = keep
        return s

    def _fill_error_model(self, t):
        s = BrxErrorModel()
        s.k, s.type, s.n_rows, s.n_alts, s.pool_len = t['k'], t['type'], t['n_rows'], t['n_alts'], len(t['pool'])
        held = t
        if 'rowx' not in t:                  # a table dict built by hand (tests): the lookup-order layout is derived here
            from .error_model import derive_lookup_tables
            t = dict(t, **derive_lookup_tables(t))
        keep = []
        for field, key in (('d_row_off', 'row_off'), ('d_self_thr', 'self_thr'), ('d_thr', 'thr'),
                           ('d_desc', 'desc'), ('d_pool', 'pool'), ('d_rowx', 'rowx'), ('d_altx', 'altx')):
            ptr, k = self._upload(t[key])
            setattr(s, field, ptr)
            keep.append(k)
        self._keep['em'] = keep
        self._configured = dict(getattr(self, '_configured', {}), em=held)    # which tables this engine holds now (simulate.sequence_fragment)
        return s

    def _fill_qscore_model(self, t):
        s = BrxQScoreModel()
        s.k, s.gap_bits, s.hash_size, s.n_rows, s.n_entries = t['k'], t['gap_bits'], t['hash_size'], t['n_rows'], t['n_entries']
        keep = []
        for field, key in (('d_hash_key', 'hash_key'), ('d_hash_row', 'hash_row'), ('d_row_off', 'row_off'),
                           ('d_thr', 'thr'), ('d_score', 'score')):
            ptr, k = self._upload(t[key])
            setattr(s, field, ptr)
            keep.append(k)
        self._keep['qm'] = keep
        self._configured = dict(getattr(self, '_configured', {}), qm=t)
        return s

    def _fill_params(self, params):
        s = BrxSimParams()
        p0, k0 = self._upload(acgt_codes(params.start_adapter))
        p1, k1 = self._upload(acgt_codes(params.end_adapter))
        self._keep['params'] = [k0, k1]
        return params.fill(s, p0, p1)

    def decode(self, codes):
        """base codes -> str using the current reference alphabet."""
        return self.sym[np.asarray(codes, dtype=np.uint8)].tobytes().decode('latin-1')


# -----------------------------------------------------------------------------------------------
_lib = None


def load_library():
    """dlopen libbrx_hip.so; raises with build instructions if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get('BRX_LIB_PATH') or LIB_PATH         # BRX_LIB_PATH: another BUILD of the same library (A/B measurements), never a fallback
    if not os.path.isfile(path):
        raise RuntimeError(f'{path} is missing: build it with `python -m badread_amd.build` '
                           '(hipcc, --offload-arch=gfx950).  There is no CPU fallback.')
    import torch  # noqa: F401  -- FIRST: the library must bind to the HIP runtime torch ships, not a second copy
    _lib = bind_library(ctypes.CDLL(path))
    return _lib


def bind_library(lib):
    """ctypes prototypes of every entry point of include/brx.h on an opened library."""
    lib.brx_create.restype = ctypes.c_int
    lib.brx_create.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    lib.brx_destroy.restype = None
    lib.brx_destroy.argtypes = [ctypes.c_void_p]
    lib.brx_last_error.restype = ctypes.c_char_p
    lib.brx_last_error.argtypes = [ctypes.c_void_p]
    lib.brx_version.restype = ctypes.c_char_p
    lib.brx_version.argtypes = []
    for name, struct in (('brx_set_reference', BrxReference), ('brx_set_error_model', BrxErrorModel),
                         ('brx_set_qscore_model', BrxQScoreModel), ('brx_set_params', BrxSimParams)):
        fn = getattr(lib, name)
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.POINTER(struct)]
    lib.brx_set_scratch.restype = ctypes.c_int
    lib.brx_set_scratch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    lib.brx_scratch_needed.restype = ctypes.c_size_t
    lib.brx_scratch_needed.argtypes = [ctypes.c_void_p]
    lib.brx_output_needed.restype = ctypes.c_size_t
    lib.brx_output_needed.argtypes = [ctypes.c_void_p]
    lib.brx_simulate_batch.restype = ctypes.c_int
    lib.brx_simulate_batch.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32,
                                       ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                       ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p]
    lib.brx_sequence_fragments.restype = ctypes.c_int
    lib.brx_sequence_fragments.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_size_t, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t),
                                           ctypes.c_void_p]
    lib.brx_align_batch.restype = ctypes.c_int
    lib.brx_align_batch.argtypes = [ctypes.c_void_p, ctypes.c_uint32] + [ctypes.c_void_p] * 11
    lib.brx_last_stage_ms.restype = ctypes.c_int
    lib.brx_last_stage_ms.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float * 8)]
    lib.brx_last_read_cycles.restype = ctypes.c_int
    lib.brx_last_read_cycles.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
    lib.brx_last_phase_cycles.restype = ctypes.c_int
    lib.brx_last_phase_cycles.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
    lib.brx_set_kernel_timing.restype = ctypes.c_int
    lib.brx_set_kernel_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.brx_last_kernel_stats.restype = ctypes.c_int
    lib.brx_last_kernel_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(BrxKernelStat * len(KERNEL_NAMES))]
    lib.brx_gzip_device_bound.restype = ctypes.c_size_t
    lib.brx_gzip_device_bound.argtypes = [ctypes.c_size_t, ctypes.c_uint32]
    lib.brx_gzip_device_scratch.restype = ctypes.c_size_t
    lib.brx_gzip_device_scratch.argtypes = [ctypes.c_size_t, ctypes.c_uint32]
    lib.brx_gzip_device.restype = ctypes.c_int
    lib.brx_gzip_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p,
                                    ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p]
    lib.brx_model_count.restype = ctypes.c_int
    lib.brx_model_count.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(BrxModelJob), ctypes.c_void_p]
    lib.brx_last_mutate_passes.restype = ctypes.c_uint32
    lib.brx_last_mutate_passes.argtypes = [ctypes.c_void_p]
    lib.brx_last_final_launches.restype = ctypes.c_uint32
    lib.brx_last_final_launches.argtypes = [ctypes.c_void_p]
    lib.brx_last_window_misses.restype = ctypes.c_uint32
    lib.brx_last_window_misses.argtypes = [ctypes.c_void_p]
    return lib


class BrxError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f'libbrx_hip error {code}: {message}')
        self.code = code


def arena_estimate(n_reads, mean_length, error_rate=None):
    """Bytes of scratch arena for device batches of `n_reads` reads of `mean_length` bases (HipEngine.presize).

    Bottom of the arena, for the whole batch: fragment, replacement words, 2-bit codes and changed map (5.5 B per base), per-read
    state and lists (~600 B per read).  Then the LARGER of two things that follow each other in time (brx_hip.hip, Arena::take_top /
    release_top): (a) what only the mutate stage of the bulk set needs -- the survivor rings (20 B x (n / 8 + 128) entries per read),
    the per-wave window scratch, one 6.6 MB store of move codes per wave of k_mut_lanes (1016 for a 65 536-read batch): 11.9 GB for
    configs[3] -- with the head set's final stage beside it (it starts while the bulk set still mutates; about a third of the batch's
    final stage: 6 GB measured), and (b) the final stage of both sets: read + qualities + ops (4 B per base) and the traceback slabs by
    the edits per base (35 B per base at the 5 % of the nanopore2023 defaults, up to 20 GB -- the align kernels hold at most 2048 / 1024 /
    512 / 256 slabs and halve their grids when room is short --, ~3 B at Q30 reads, whose narrow-band class walks its traceback in
    strips).  40 -> 30 GB for a shipped batch of configs[3] against rounds 4-6a, which kept (a) beside (b)."""
    bases = float(n_reads) * (float(mean_length) + 14.0)
    per_base = 35.0 if error_rate is None else min(35.0, max(3.0, 35.0 * float(error_rate) / 0.05))
    low = 5.5 * bases + 600.0 * n_reads
    rings = 2.5 * bases + 2600.0 * n_reads
    mutate_only = rings + min(n_reads, 4096) * 0.62e6 + min(n_reads / 64.0, 1024.0) * 6.6e6
    # the cap on the slabs: 20 GB holds full grids at the 5 % of the defaults; at 10 % a batch's widest reads (the memory-resident
    # path keeps every cell of a 300 kb chimera at 75 %) are GBs each and a head set squeezed into 4 GB ran its widest class on ONE
    # wave -- a rough batch took 150 s instead of 50 (tests/test_gpu_fullsize.py, round 6): the cap follows the error rate up to 32 GB
    slab_cap = 20e9 * (1.0 if error_rate is None else min(1.6, max(1.0, float(error_rate) / 0.05)))
    final = 4.0 * bases + min(per_base * bases, slab_cap)
    return int(low + max(mutate_only + 0.35 * final, final) + (1 << 30))


class HipEngine(EngineBase):
    """One context on one MI355X.  Not thread-safe; one engine per process per GPU."""

    def __init__(self, device=0, scratch_bytes=1 << 30, scratch_tensor=None):
        """scratch_tensor: a device uint8 tensor (typically a slice of one arena the caller allocated for several engines at
        once, before any of them computes: simulate._BatchPool) to use instead of allocating `scratch_bytes`."""
        super().__init__()
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError('HipEngine needs a ROCm device (torch.cuda.is_available() is False); '
                               'there is no CPU fallback')
        self.torch = torch
        self.device = torch.device('cuda', device)
        self.lib = load_library()
        ctx = ctypes.c_void_p()
        rc = self.lib.brx_create(device, ctypes.byref(ctx))
        if rc != 0:
            raise BrxError(rc, self.lib.brx_last_error(None).decode('latin-1', 'replace') or 'brx_create failed')
        self.ctx = ctx
        self._scratch = None
        self._out = None
        self._stats = None
        self._structs = {}              # last descriptor given to each brx_set_*: what clone() hands to a new context
        if scratch_tensor is not None:
            self._scratch = scratch_tensor
            self._check(self.lib.brx_set_scratch(self.ctx, ctypes.c_void_p(self._scratch.data_ptr()), self._scratch.numel()))
        else:
            self._ensure_scratch(scratch_bytes)

    def close(self):
        if getattr(self, 'ctx', None):
            self.lib.brx_destroy(self.ctx)
            self.ctx = None
        self._scratch = self._out = self._stats = None       # give the arena back to the allocator now, not at garbage collection

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc):
        if rc != 0:
            raise BrxError(rc, self.lib.brx_last_error(self.ctx).decode('latin-1', 'replace'))

    def _upload(self, arr):
        arr = np.ascontiguousarray(arr)
        raw = arr.view(np.uint8).reshape(-1) if arr.dtype.fields is None else np.frombuffer(arr.tobytes(), dtype=np.uint8)
        if raw.size == 0:
            raw = np.zeros(8, dtype=np.uint8)
        if raw.size < (32 << 20) or self.device.type != 'cuda':
            t = self.torch.from_numpy(raw.copy()).to(self.device)
            return t.data_ptr(), t
        # A genome: straight from where it lies (the mapped sidecar) through two pinned buffers, the host copy of one chunk
        # beside the DMA of the other -- from_numpy(copy()).to(device) was a second 772 MB copy and a pageable transfer.
        torch = self.torch
        t = torch.empty(raw.size, dtype=torch.uint8, device=self.device)
        chunk = 64 << 20
        pins = [torch.empty(chunk, dtype=torch.uint8).pin_memory() for _ in range(2)]
        done = [None, None]
        for i, off in enumerate(range(0, raw.size, chunk)):
            n = min(chunk, raw.size - off)
            if done[i & 1] is not None:
                done[i & 1].synchronize()
            pins[i & 1].numpy()[:n] = raw[off:off + n]
            t[off:off + n].copy_(pins[i & 1][:n], non_blocking=True)
            done[i & 1] = torch.cuda.Event()
            done[i & 1].record()
        torch.cuda.current_stream(self.device).synchronize()
        return t.data_ptr(), t

    def _stream(self):
        return ctypes.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _ensure_scratch(self, nbytes):
        if self._scratch is None or self._scratch.numel() < nbytes:
            had = self._scratch is not None and self._scratch.numel() >= (1 << 30)
            self._scratch = None
            if had:
                # an arena that grows gives its old block BACK TO THE DRIVER first: torch's caching allocator would keep the 20-40 GB
                # beside the new block, six engines doing that fill the device, and the runtime then has nothing left for its own
                # allocations (kernel scratch: HSA_STATUS_ERROR_OUT_OF_RESOURCES, 'Available Free mem : 0 MB' -- profiles/README.md, round 6)
                self.torch.cuda.empty_cache()
            self._scratch = self.torch.empty(int(nbytes), dtype=self.torch.uint8, device=self.device)
            self._check(self.lib.brx_set_scratch(self.ctx, ctypes.c_void_p(self._scratch.data_ptr()),
                                                 self._scratch.numel()))

    def scratch_bytes(self):
        """Size of this engine's scratch arena (a clone maps the same amount)."""
        return int(self._scratch.numel()) if self._scratch is not None else 0

    def _ensure_out(self, nbytes, n_reads):
        if self._out is None or self._out.numel() < nbytes:
            self._out = None
            self._out = self.torch.empty(int(nbytes), dtype=self.torch.uint8, device=self.device)
        need = n_reads * READ_STATS_DTYPE.itemsize
        if self._stats is None or self._stats.numel() < need:
            self._stats = self.torch.empty(int(need), dtype=self.torch.uint8, device=self.device)

    def arena_bytes(self, n_reads, mean_length, error_rate=None):
        """arena_estimate for THIS engine's parameters: reads chain fragments with the chimera rate (a read is 1 / (1 - rate) fragments
        on average: --chimeras 25 makes the batch a third more bases than its fragment length says)."""
        p = self._structs.get('params')
        chain = 1.0 / max(1.0 - min(float(p.chimera_rate), 0.9), 0.1) if p is not None else 1.0
        return arena_estimate(n_reads, float(mean_length) * chain, error_rate)

    def presize(self, n_reads, mean_length, error_rate=None):
        """Size the scratch arena for device batches of `n_reads` reads of `mean_length` bases BEFORE the first batch, so that a
        large job does not discover its arena by repeating batches (BRX_E_SCRATCH -> grow -> run again): arena_estimate, with this
        engine's chimera rate.  A job whose identity law is known sizes for it (error_rate): every GB of arena is 14-29 ms of the driver
        clearing it, per engine.  An estimate: the library still reports what it needs if this is short (one repeated batch)."""
        self._ensure_scratch(self.arena_bytes(n_reads, mean_length, error_rate))

    def adopt_scratch(self, tensor):
        """Use `tensor` (device uint8) as the arena from now on: the caller allocated it beside other work (simulate._ArenaPrefetch)."""
        self._scratch = tensor
        self._check(self.lib.brx_set_scratch(self.ctx, ctypes.c_void_p(self._scratch.data_ptr()), self._scratch.numel()))

    # ------------------------------------------------------------------ configuration
    def set_reference(self, pref, cum_weight=None):
        s = self._fill_reference(pref, cum_weight)
        self._check(self.lib.brx_set_reference(self.ctx, ctypes.byref(s)))
        self._structs['ref'] = s

    def set_error_model(self, tables):
        s = self._fill_error_model(tables)
        self._check(self.lib.brx_set_error_model(self.ctx, ctypes.byref(s)))
        self._structs['em'] = s

    def set_qscore_model(self, tables):
        s = self._fill_qscore_model(tables)
        self._check(self.lib.brx_set_qscore_model(self.ctx, ctypes.byref(s)))
        self._structs['qm'] = s

    def clone(self, scratch_bytes=None, scratch_tensor=None):
        """Another context on the same device that SHARES this engine's device tables (reference, models,
        parameters: read-only in every kernel) and owns its scratch, output and stream.  The driver keeps several
        batches in flight with one clone per batch (badread_amd.simulate.run_batches)."""
        other = HipEngine(self.device.index or 0, scratch_bytes or (self._scratch.numel() if self._scratch is not None else 1 << 30),
                          scratch_tensor=scratch_tensor)
        other._keep = dict(self._keep)
        other.sym = self.sym
        setters = {'ref': self.lib.brx_set_reference, 'em': self.lib.brx_set_error_model,
                   'qm': self.lib.brx_set_qscore_model, 'params': self.lib.brx_set_params}
        for key, s in self._structs.items():
            other._check(setters[key](other.ctx, ctypes.byref(s)))
            other._structs[key] = s
        return other

    def set_params(self, params):
        s = self._fill_params(params)
        self._check(self.lib.brx_set_params(self.ctx, ctypes.byref(s)))
        self._structs['params'] = s

    # ------------------------------------------------------------------ calls
    def _retry(self, call, n_reads, out_guess, allow_nofrag=False):
        """Run `call(out_ptr, out_cap, stats_ptr, out_bytes)` growing scratch/output as the library asks.
        With allow_nofrag, BRX_E_NOFRAG is not raised: the batch is complete and the failing reads
        carry RS_NOFRAG in their stats, so the caller can apply the reference's fatal-exit rule
        (simulate.py:159-165) only to reads that precede the stop point."""
        self._ensure_out(out_guess, n_reads)
        for attempt in range(8):
            self.retries = getattr(self, 'retries', 0) + (1 if attempt else 0)
            out_bytes = ctypes.c_size_t(0)
            rc = call(ctypes.c_void_p(self._out.data_ptr()), self._out.numel(),
                      ctypes.c_void_p(self._stats.data_ptr()), ctypes.byref(out_bytes))
            if rc in (E_SCRATCH, E_OUTPUT):
                self.retry_log = getattr(self, 'retry_log', []) + [self.lib.brx_last_error(self.ctx).decode('latin-1', 'replace')]
            if rc == E_SCRATCH:
                self._ensure_scratch(int(self.lib.brx_scratch_needed(self.ctx) * 1.25) + (1 << 20))
                continue
            if rc == E_OUTPUT:
                self._ensure_out(int(self.lib.brx_output_needed(self.ctx) * 1.1) + (1 << 16), n_reads)
                continue
            if rc == E_NOFRAG and allow_nofrag:
                return out_bytes.value
            self._check(rc)
            return out_bytes.value
        raise BrxError(E_SCRATCH, 'could not size scratch/output buffers after 8 attempts')

    def expected_record_bytes(self):
        """FASTQ bytes per read the engine's parameters lead to expect, with 6 % of margin: sequence + qualities of a read whose
        fragments chain with the chimera rate (a read is 1 / (1 - rate) fragments on average) plus the header's share.  34 kB at
        the defaults (what rounds 1-5 assumed for every job); --chimeras 25 makes reads a third longer, and a guess that is short
        costs every engine of a job one repeated batch (brx_simulate_batch reports BRX_E_OUTPUT with the size it needs)."""
        p = self._structs.get('params')
        if p is None:
            return 34000.0
        chain = 1.0 / max(1.0 - min(float(p.chimera_rate), 0.9), 0.1)
        return 1.06 * chain * (2.1 * float(p.frag_mean) + 400.0)

    def simulate_batch_device(self, seed, first_read, n_reads, expected_bytes=None, allow_nofrag=False):
        """Returns (device uint8 tensor view of the FASTQ bytes, stats as numpy structured array)."""
        guess = expected_bytes or (int(n_reads * self.expected_record_bytes()) + (1 << 16))
        nbytes = self._retry(lambda o, cap, st, ob: self.lib.brx_simulate_batch(
            self.ctx, seed, first_read, n_reads, o, cap, st, ob, self._stream()), n_reads, guess, allow_nofrag)
        stats = self._stats[:n_reads * READ_STATS_DTYPE.itemsize].cpu().numpy().view(READ_STATS_DTYPE)
        return self._out[:nbytes], stats

    def simulate_batch(self, seed, first_read, n_reads, allow_nofrag=False):
        out, stats = self.simulate_batch_device(seed, first_read, n_reads, allow_nofrag=allow_nofrag)
        return out.cpu().numpy(), stats

    def sequence_fragments(self, seed, first_read, frags, targets):
        """frags: list of uint8 code arrays; returns (list of (seq_codes, qual_bytes), stats)."""
        n = len(frags)
        off = np.zeros(n + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(f) for f in frags])
        flat = np.concatenate(frags).astype(np.uint8) if n else np.zeros(1, np.uint8)
        p_fr, k1 = self._upload(flat)
        p_off, k2 = self._upload(off)
        p_tg, k3 = self._upload(np.asarray(targets, dtype=np.float64))
        guess = int(off[-1]) * 3 + 4096 * n + 4096
        nbytes = self._retry(lambda o, cap, st, ob: self.lib.brx_sequence_fragments(
            self.ctx, seed, first_read, n, ctypes.c_void_p(p_fr), ctypes.c_void_p(p_off), ctypes.c_void_p(p_tg),
            o, cap, st, ob, self._stream()), n, guess)
        del k1, k2, k3
        raw = self._out[:nbytes].cpu().numpy()
        stats = self._stats[:n * READ_STATS_DTYPE.itemsize].cpu().numpy().view(READ_STATS_DTYPE).copy()
        res = []
        for st in stats:
            o, L = int(st['rec_off']), int(st['seq_len'])
            res.append((raw[o:o + L].copy(), raw[o + L:o + 2 * L].copy()))
        return res, stats

    def align_batch(self, queries, targets, k_hint=None, want_ops=True):
        """queries/targets: lists of bytes.  Returns (list of op arrays or None, dist, ncols, nmatch)."""
        n = len(queries)
        q_off = np.zeros(n + 1, dtype=np.uint64)
        t_off = np.zeros(n + 1, dtype=np.uint64)
        q_off[1:] = np.cumsum([len(q) for q in queries])
        t_off[1:] = np.cumsum([len(t) for t in targets])
        ops_off = q_off + t_off
        qbuf = np.frombuffer(b''.join(bytes(q) for q in queries) or b'\0', dtype=np.uint8)
        tbuf = np.frombuffer(b''.join(bytes(t) for t in targets) or b'\0', dtype=np.uint8)
        kh = np.full(n, -1, dtype=np.int32) if k_hint is None else np.asarray(k_hint, dtype=np.int32)
        torch = self.torch
        p_qs, k0 = self._upload(qbuf)
        p_q, k2 = self._upload(q_off)
        p_ts, k1 = self._upload(tbuf)
        p_t, k3 = self._upload(t_off)
        p_k, k4 = self._upload(kh)
        p_oo, k5 = self._upload(ops_off)
        d_dist = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        d_ncols = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        d_nmatch = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        d_ops = torch.empty(max(int(ops_off[-1]), 1), dtype=torch.uint8, device=self.device) if want_ops else None
        for _ in range(8):
            rc = self.lib.brx_align_batch(self.ctx, n, p_qs, p_q, p_ts, p_t, p_k, d_dist.data_ptr(), d_ncols.data_ptr(),
                                          d_nmatch.data_ptr(), d_ops.data_ptr() if want_ops else None, p_oo,
                                          self._stream().value)
            if rc == E_SCRATCH:
                self._ensure_scratch(int(self.lib.brx_scratch_needed(self.ctx) * 1.25) + (1 << 20))
                continue
            self._check(rc)
            break
        else:
            raise BrxError(E_SCRATCH, 'could not size the alignment scratch after 8 attempts')
        del k0, k1, k2, k3, k4, k5
        dist = d_dist[:n].cpu().numpy()
        ncols = d_ncols[:n].cpu().numpy()
        nmatch = d_nmatch[:n].cpu().numpy()
        ops_list = None
        if want_ops:
            raw = d_ops.cpu().numpy()
            ops_list = [raw[int(ops_off[i]):int(ops_off[i]) + int(ncols[i])].copy() for i in range(n)]
        return ops_list, dist, ncols, nmatch

    def model_count(self, kind, job, k, max_del, n_ksizes, table_bits, spill_cap=1 << 16):
        """brx_model_count over a model_builder.Job: (keys, counts, earliest ranks) of the occupied table slots and the
        spilled windows, or None when a table of 2^table_bits slots was too small."""
        torch = self.torch
        keep = []

        def up(arr):
            ptr, t = self._upload(arr)
            keep.append(t)
            return ptr
        size = 1 << table_bits
        dev = self.device
        keys = torch.full((size,), -1, dtype=torch.int64, device=dev)
        counts = torch.zeros(size, dtype=torch.int32, device=dev)
        first = torch.full((size,), -1, dtype=torch.int64, device=dev)
        flags = torch.zeros(4, dtype=torch.int32, device=dev)
        spill = torch.zeros(spill_cap, dtype=torch.int64, device=dev)
        cols = [torch.empty(max(job.n_cols, 1), dtype=torch.uint8, device=dev) for _ in range(3)]
        s = BrxModelJob()
        s.n_align, s.k, s.max_del, s.n_ksizes, s.n_cols = job.n_align, k, max_del, n_ksizes, job.n_cols
        s.d_seq, s.d_qual, s.d_ref = up(job.seq), up(job.qual), up(job.ref)
        s.d_part_type, s.d_part_len = up(job.part_type), up(job.part_len)
        s.d_part_col, s.d_part_read, s.d_part_ref = up(job.part_col), up(job.part_read), up(job.part_ref)
        s.d_align_part_off, s.d_align_col_off = up(job.align_part_off), up(job.align_col_off)
        s.d_rcol, s.d_qcol, s.d_fcol = (c.data_ptr() for c in cols)
        s.d_keys, s.d_counts, s.d_first = keys.data_ptr(), counts.data_ptr(), first.data_ptr()
        s.table_mask, s.d_flags, s.d_spill, s.spill_cap = size - 1, flags.data_ptr(), spill.data_ptr(), spill_cap
        rc = self.lib.brx_model_count(self.ctx, kind, ctypes.byref(s), self._stream())
        if rc == E_OUTPUT:
            return None
        self._check(rc)
        h_flags = flags.cpu().numpy()
        n_spill = int(h_flags[1 + kind])
        if n_spill > spill_cap:
            return self.model_count(kind, job, k, max_del, n_ksizes, table_bits, spill_cap=2 * n_spill)
        h_keys = keys.cpu().numpy().view(np.uint64)
        used = h_keys != np.uint64(0xFFFFFFFFFFFFFFFF)
        return (h_keys[used], counts.cpu().numpy()[used].astype(np.int64), first.cpu().numpy().view(np.uint64)[used],
                spill[:n_spill].cpu().numpy().view(np.uint64))

    def gzip_device(self, data, block_off=None):
        """brx_gzip_device: a uint8 tensor of FASTQ text on this engine's device -> a uint8 tensor (same device) holding
        gzip members of it, one per block.  block_off: ascending byte offsets (numpy, first 0, last len(data)) of the
        blocks, or None for 64 KB blocks."""
        torch = self.torch
        n = int(data.numel())
        if n == 0:
            return torch.zeros(0, dtype=torch.uint8, device=self.device)
        nb, d_off, keep = 0, None, None
        if block_off is not None:
            block_off = np.ascontiguousarray(block_off, dtype=np.uint64)
            assert len(block_off) >= 2 and int(block_off[0]) == 0 and int(block_off[-1]) == n
            nb = len(block_off) - 1
            keep = torch.from_numpy(block_off.view(np.int64).copy()).to(self.device)
            d_off = ctypes.c_void_p(keep.data_ptr())
        # capacities in steps of 64 MB: a batch's size differs from the last one's by a few MB, and torch's allocator keeps a cached block
        # that is too small for the next request beside the new one (simulate._BatchPool._copy_of: what that crept up to over a 94-batch job)
        step = lambda b: b if b <= (1 << 24) else -(-b // (1 << 26)) * (1 << 26)
        scratch = torch.empty(step(int(self.lib.brx_gzip_device_scratch(n, nb)) + 8), dtype=torch.uint8, device=self.device)
        blocks = nb or -(-n // 65536)
        # FASTQ packs to about half; the worst case (brx_gzip_device_bound: 15 bits per byte) is only allocated when the
        # library asks for it
        cap = min(int(0.7 * n) + 200 * blocks + 4096, int(self.lib.brx_gzip_device_bound(n, nb)) + 8)
        got = ctypes.c_size_t(0)
        for _ in range(2):
            out = torch.empty(step(cap), dtype=torch.uint8, device=self.device)
            rc = self.lib.brx_gzip_device(self.ctx, ctypes.c_void_p(data.data_ptr()), n, d_off, nb, ctypes.c_void_p(out.data_ptr()), out.numel(),
                                          ctypes.c_void_p(scratch.data_ptr()), scratch.numel(), ctypes.byref(got), self._stream())
            if rc != E_OUTPUT:
                break
            cap = int(self.lib.brx_output_needed(self.ctx)) + 8
        self._check(rc)
        return out[:got.value]

    def stage_ms(self):
        arr = (ctypes.c_float * 8)()
        self._check(self.lib.brx_last_stage_ms(self.ctx, ctypes.byref(arr)))
        return dict(zip(STAGE_NAMES, [float(x) for x in arr]))

    def read_cycles(self, n_reads):
        """(n_reads, 8) uint64 shader-clock counters of the last pipeline call (see include/brx.h)."""
        out = np.zeros((n_reads, 8), dtype=np.uint64)
        self._check(self.lib.brx_last_read_cycles(self.ctx, out.ctypes.data, n_reads))
        return out

    def phase_cycles(self, n_reads):
        """(n_reads, 8) uint64: shader-clock time per mutate phase of the last batch (BRX_PROFILE=1 at creation)."""
        out = np.zeros((n_reads, 8), dtype=np.uint64)
        self._check(self.lib.brx_last_phase_cycles(self.ctx, out.ctypes.data, n_reads))
        return out

    def set_kernel_timing(self, on=True):
        """Bracket every kernel launch of the following batches with HIP events (brx_last_kernel_stats)."""
        self._check(self.lib.brx_set_kernel_timing(self.ctx, 1 if on else 0))

    def kernel_stats(self):
        """{kernel name: (launches, total ms, bases handled)} of the last batch (zeros unless set_kernel_timing)."""
        arr = (BrxKernelStat * len(KERNEL_NAMES))()
        self._check(self.lib.brx_last_kernel_stats(self.ctx, ctypes.byref(arr)))
        return {name: (int(a.launches), float(a.ms), float(a.bases)) for name, a in zip(KERNEL_NAMES, arr)}

    def mutate_passes(self):
        return int(self.lib.brx_last_mutate_passes(self.ctx))

    def final_launches(self):
        return int(self.lib.brx_last_final_launches(self.ctx))

    def window_misses(self):
        """Reads of the last batch that needed the full traceback store (second pass of the final stage)."""
        return int(self.lib.brx_last_window_misses(self.ctx))


_default_engine = None


def rank_device_index(n_devices=None):
    """The device of this rank, resolved in ONE place for the process group (simulate.Shard.from_env) and the engine
    (default_engine): BRX_DEVICE when set; else LOCAL_RANK.  More ranks than devices is an error -- RCCL would refuse two
    ranks on one GPU with a far less helpful message -- unless the exchange is kept on the host (BRX_DIST_BACKEND=gloo: several
    ranks share a GPU, which is how the tests run the HIP engine under world > 1 on a 1-GPU box)."""
    if os.environ.get('BRX_DEVICE') is not None:
        return int(os.environ['BRX_DEVICE'])
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if n_devices is None:
        import torch
        n_devices = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n_devices and local >= n_devices:
        if os.environ.get('BRX_DIST_BACKEND') == 'gloo':
            return local % n_devices
        raise RuntimeError(f'LOCAL_RANK {local} but only {n_devices} GPU(s) are visible: one rank per GPU '
                           '(set BRX_DEVICE, or BRX_DIST_BACKEND=gloo to let ranks share a GPU)')
    return local


def default_engine():
    """Process-wide HipEngine on this rank's device (rank_device_index)."""
    global _default_engine
    if _default_engine is None:
        _default_engine = HipEngine(rank_device_index())
    return _default_engine


def hip_align_batch(queries, targets):
    """Aligner callable for ErrorModel loading: list of op arrays via the HIP Myers kernel."""
    ops, _, _, _ = default_engine().align_batch(queries, targets)
    return ops

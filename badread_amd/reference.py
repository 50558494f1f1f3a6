"""
Reference genome container for the HIP path: FASTA -> 2-bit packed bases + non-ACGT runs.

The reference keeps every contig as a Python str plus a full reverse-complement copy
(/root/reference/badread/simulate.py:37-38, misc.py:122-153).  Here the genome is packed once into
16-bases-per-uint32 words (0.25 B/base in HBM); bases outside ACGT (N runs, IUPAC codes, which the
reference preserves - misc.py:135,152) are kept as a short sorted list of runs; the reverse strand
is never materialised (kernels complement on the fly, coordinates stay in the reference's
reverse-complement-string space, simulate.py:194-200).

Layout contract with include/brx.h: brx_contig (24 B), brx_exception (24 B), base g lives in bits
2*(g%16) of word g/16.
"""
import itertools

import numpy as np

from .misc import complement_base, load_fasta

CONTIG_DTYPE = np.dtype([('base_off', '<u8'), ('length', '<u4'), ('flags', '<u4'),
                         ('name_off', '<u4'), ('name_len', '<u4')])
EXCEPTION_DTYPE = np.dtype([('start', '<u8'), ('end', '<u8'), ('code', '<u4'), ('pad', '<u4')])

FLAG_CIRCULAR, FLAG_HAIRPIN_LEFT, FLAG_HAIRPIN_RIGHT = 1, 2, 4
_CHUNK = 1 << 24

import ctypes as _ct  # noqa: E402


class FastaView(_ct.Structure):
    """brx_fasta_view of include/brx_host.h"""
    _fields_ = [('n_bases', _ct.c_uint64), ('n_words', _ct.c_uint64), ('n_contigs', _ct.c_uint32),
                ('n_exceptions', _ct.c_uint32), ('names_len', _ct.c_uint32), ('n_symbols', _ct.c_uint32),
                ('packed', _ct.POINTER(_ct.c_uint32)), ('contigs', _ct.c_void_p), ('exceptions', _ct.c_void_p),
                ('names', _ct.c_void_p), ('depths', _ct.POINTER(_ct.c_double)),
                ('sym', _ct.c_uint8 * 16), ('comp', _ct.c_uint8 * 16)]


_host_lib = None


def host_library():
    """libbrx_host.so (built by `python -m badread_amd.build`); raises if it is missing -- there is no silent
    fall-back to the Python packer on the product path."""
    global _host_lib
    if _host_lib is None:
        import os
        path = os.path.join(os.path.dirname(os.path.realpath(__file__)), 'csrc', 'libbrx_host.so')
        if not os.path.isfile(path):
            raise RuntimeError(f'{path} is missing: run `python -m badread_amd.build`')
        lib = _ct.CDLL(path)
        P = _ct.c_void_p
        lib.brx_fasta_pack.restype = _ct.c_int
        lib.brx_fasta_pack.argtypes = [_ct.c_char_p, _ct.POINTER(P), _ct.c_char_p, _ct.c_size_t]
        lib.brx_fasta_load.restype = _ct.c_int
        lib.brx_fasta_load.argtypes = [_ct.c_char_p, _ct.c_char_p, _ct.POINTER(P), _ct.c_char_p, _ct.c_size_t]
        lib.brx_fasta_save.restype = _ct.c_int
        lib.brx_fasta_save.argtypes = [P, _ct.c_char_p, _ct.c_char_p, _ct.c_char_p, _ct.c_size_t]
        lib.brx_fasta_view_of.restype = _ct.c_int
        lib.brx_fasta_view_of.argtypes = [P, _ct.POINTER(FastaView)]
        lib.brx_fasta_free.restype = None
        lib.brx_fasta_free.argtypes = [P]
        _host_lib = lib
    return _host_lib


class _NativeFasta(object):
    """Owner of a brx_fasta object (the packer's vector or the mapped sidecar).  The packed words are handed out as a numpy view
    whose buffer holds a reference to this object, so the memory lives exactly as long as anything can still read it -- an
    engine that was given `pref.packed` may outlive the PackedReference it came from."""

    def __init__(self, lib, handle):
        self.lib, self.handle = lib, handle

    def __del__(self):
        handle, self.handle = self.handle, None
        if handle is not None:
            self.lib.brx_fasta_free(handle)


class PackedReference(object):
    """Packed genome + per-contig metadata.  Build with from_fasta() or from_seqs()."""

    def __init__(self):
        self.names = []
        self.lengths = []
        self.depths = {}
        self.circular = {}
        self.hairpin_left = {}
        self.hairpin_right = {}
        self.packed = np.zeros(1, dtype=np.uint32)
        self.n_bases = 0
        self.contigs = np.zeros(0, dtype=CONTIG_DTYPE)
        self.exceptions = np.zeros(0, dtype=EXCEPTION_DTYPE)
        self.names_pool = b''
        self.sym = np.zeros(16, dtype=np.uint8)
        self.comp = np.zeros(16, dtype=np.uint8)
        self.code_of = {}
        self._native = None          # the native object `packed` is a view of, if any (the view itself keeps it alive: _NativeFasta)

    @classmethod
    def from_fasta(cls, filename, cache=None):
        """FASTA or FASTA.gz -> packed reference through libbrx_host.so (csrc/brx_fasta.cpp: one streaming C++ pass
        instead of Python strings + numpy).  The packed form is kept in a sidecar file and MAPPED from it next time (while the
        FASTA's size and mtime are unchanged): cache=True keeps it beside the FASTA (`<filename>.brx2bit`); the default
        (cache=None: what `badread simulate` does) uses such a file when there is one and otherwise keeps its own in the user
        cache directory (BADREAD_AMD_CACHE, default ~/.cache/badread_amd: 0.25 bytes per base), so that the second run on a
        genome starts in a fraction of a second instead of re-reading 3 GB of text; BRX_REFERENCE_CACHE=0 (or cache=False)
        packs every time and writes nothing."""
        import os
        beside = str(filename) + '.brx2bit'
        if cache is None:
            env = os.environ.get('BRX_REFERENCE_CACHE', '')
            if env == '0':
                cache = False
            elif env not in ('', 'auto'):
                cache = True
        if cache is True:
            return cls._from_native(filename, [beside], beside)
        if cache is False:
            return cls._from_native(filename, [], None)
        own = None
        from .error_model import user_cache_dir
        try:
            worth_it = os.path.getsize(str(filename)) >= (64 << 20)       # a bacterial genome packs in milliseconds: nothing to keep
        except OSError:
            worth_it = False
        d = user_cache_dir() if worth_it else None
        if d:
            import hashlib
            own = os.path.join(d, 'ref-' + hashlib.sha256(os.path.abspath(str(filename)).encode()).hexdigest()[:24] + '.brx2bit')
        return cls._from_native(filename, [beside] + ([own] if own else []), own)

    @classmethod
    def from_fasta_python(cls, filename):
        """The same result by the pure Python/numpy route (misc.load_fasta + from_seqs); kept as the cross-check
        of the native packer in the tests."""
        return cls.from_seqs(*load_fasta(filename))

    @classmethod
    def _from_native(cls, filename, sidecars, save_to):
        """sidecars: candidate files, first match wins; save_to: where a freshly packed reference is kept (None: nowhere)."""
        import ctypes
        lib = host_library()
        err = ctypes.create_string_buffer(512)
        handle = ctypes.c_void_p()
        name = str(filename).encode()
        rc = -1
        for sidecar in sidecars:
            rc = lib.brx_fasta_load(name, sidecar.encode(), ctypes.byref(handle), err, len(err))
            if rc == 0:
                break
        if rc != 0:
            rc = lib.brx_fasta_pack(name, ctypes.byref(handle), err, len(err))
            if rc != 0:
                raise ValueError(err.value.decode('latin-1') or f'could not read {filename}')
            if save_to:
                lib.brx_fasta_save(handle, name, save_to.encode(), err, len(err))      # best effort: a read-only directory is fine
        self = cls()
        try:
            v = FastaView()
            lib.brx_fasta_view_of(handle, ctypes.byref(v))
            nc = int(v.n_contigs)
            self.n_bases = int(v.n_bases)
            # the words stay where the native object holds them -- the mapped sidecar, or the packer's vector -- for as long as
            # this object lives (a human genome is 772 MB: every copy is a third of a second of start-up)
            keeper = _NativeFasta(lib, handle)
            self._native = keeper                           # from here on the handle is freed by `keeper` alone (never twice)
            if int(v.n_words):
                words = (ctypes.c_uint32 * int(v.n_words)).from_address(ctypes.addressof(v.packed.contents))
                words._keeper = keeper                      # every numpy view of the words keeps `words`, and so the mapping, alive
                self.packed = np.frombuffer(words, dtype=np.uint32)
            else:                                           # no bases at all: `packed` is a null pointer
                self.packed = np.zeros(0, dtype=np.uint32)
            self.contigs = np.frombuffer(ctypes.string_at(v.contigs, nc * CONTIG_DTYPE.itemsize), dtype=CONTIG_DTYPE).copy()
            ne = int(v.n_exceptions)
            self.exceptions = (np.frombuffer(ctypes.string_at(v.exceptions, ne * EXCEPTION_DTYPE.itemsize), dtype=EXCEPTION_DTYPE).copy()
                               if ne else np.zeros(0, dtype=EXCEPTION_DTYPE))
            self.names_pool = ctypes.string_at(v.names, int(v.names_len)) if v.names_len else b''
            depths = np.ctypeslib.as_array(v.depths, shape=(nc,)).copy()
            self.sym = np.frombuffer(bytes(v.sym), dtype=np.uint8).copy()
            self.comp = np.frombuffer(bytes(v.comp), dtype=np.uint8).copy()
        except BaseException:
            if getattr(self, '_native', None) is None:
                lib.brx_fasta_free(handle)
            raise
        for i in range(nc):
            ct = self.contigs[i]
            n = self.names_pool[int(ct['name_off']):int(ct['name_off']) + int(ct['name_len'])].decode('latin-1')
            self.names.append(n)
            self.lengths.append(int(ct['length']))
            self.depths[n] = float(depths[i])
            self.circular[n] = bool(ct['flags'] & FLAG_CIRCULAR)
            self.hairpin_left[n] = bool(ct['flags'] & FLAG_HAIRPIN_LEFT)
            self.hairpin_right[n] = bool(ct['flags'] & FLAG_HAIRPIN_RIGHT)
        self.code_of = {chr(self.sym[c]): c for c in range(int(v.n_symbols))}
        return self

    @classmethod
    def from_seqs(cls, seqs, depths=None, circular=None, hairpin_left=None, hairpin_right=None):
        self = cls()
        self.names = list(seqs.keys())
        self.lengths = [len(seqs[n]) for n in self.names]
        self.depths = dict(depths) if depths else {n: 1.0 for n in self.names}
        self.circular = dict(circular) if circular else {n: False for n in self.names}
        self.hairpin_left = dict(hairpin_left) if hairpin_left else {n: False for n in self.names}
        self.hairpin_right = dict(hairpin_right) if hairpin_right else {n: False for n in self.names}
        for n, length in zip(self.names, self.lengths):
            if length >= 2 ** 32:
                raise ValueError(f'contig {n} is longer than 2^32-1 bases')

        # alphabet: ACGT = 0..3, N = 4, then whatever else occurs (plus complements) up to 16 codes
        lut = np.full(256, 255, dtype=np.uint8)
        symbols = ['A', 'C', 'G', 'T', 'N']
        present = np.zeros(256, dtype=bool)
        for n in self.names:
            raw = np.frombuffer(seqs[n].encode('latin-1'), dtype=np.uint8)
            for lo in range(0, len(raw), _CHUNK):
                present[np.unique(raw[lo:lo + _CHUNK])] = True
        extra = [chr(b) for b in np.flatnonzero(present) if chr(b) not in symbols]
        for ch in extra:
            for cand in (ch, complement_base(ch)):
                if cand not in symbols:
                    symbols.append(cand)
        if len(symbols) > 16:
            raise ValueError('reference uses more than 16 distinct symbols: ' + ''.join(symbols))
        for code, ch in enumerate(symbols):
            lut[ord(ch)] = code
            self.sym[code] = ord(ch)
            self.code_of[ch] = code
        for code, ch in enumerate(symbols):
            self.comp[code] = self.code_of[complement_base(ch)]
        for code in range(len(symbols), 16):
            self.sym[code] = ord('N')
            self.comp[code] = 4

        total = sum(self.lengths)
        self.n_bases = total
        n_words = (total + 15) // 16 + 1
        self.packed = np.zeros(n_words, dtype=np.uint32)
        self.contigs = np.zeros(len(self.names), dtype=CONTIG_DTYPE)
        runs = []
        pool = bytearray()
        shifts = (2 * np.arange(16, dtype=np.uint32))[None, :]
        # stream all contigs through one base-indexed writer so contigs need no word alignment
        carry = np.zeros(0, dtype=np.uint8)
        carry_start = 0
        g = 0
        for idx, n in enumerate(self.names):
            flags = (FLAG_CIRCULAR if self.circular.get(n) else 0) | \
                    (FLAG_HAIRPIN_LEFT if self.hairpin_left.get(n) else 0) | \
                    (FLAG_HAIRPIN_RIGHT if self.hairpin_right.get(n) else 0)
            name_bytes = n.encode('latin-1')
            self.contigs[idx] = (g, self.lengths[idx], flags, len(pool), len(name_bytes))
            pool += name_bytes
            raw = np.frombuffer(seqs[n].encode('latin-1'), dtype=np.uint8)
            for lo in range(0, len(raw), _CHUNK):
                codes = lut[raw[lo:lo + _CHUNK]]
                bad = np.flatnonzero(codes >= 4)
                if len(bad):
                    # maximal runs of one code
                    brk = np.flatnonzero((np.diff(bad) != 1) | (np.diff(codes[bad]) != 0)) + 1
                    starts = np.concatenate(([0], brk))
                    ends = np.concatenate((brk, [len(bad)]))
                    for s, e in zip(starts, ends):
                        a, b, code = g + int(bad[s]), g + int(bad[e - 1]) + 1, int(codes[bad[s]])
                        if runs and runs[-1][1] == a and runs[-1][2] == code:
                            runs[-1][1] = b
                        else:
                            runs.append([a, b, code])
                    codes = codes.copy()
                    codes[bad] = 0
                buf = np.concatenate((carry, codes))
                start = carry_start
                whole = (len(buf) // 16) * 16
                if whole:
                    words = (buf[:whole].reshape(-1, 16).astype(np.uint32) << shifts).sum(axis=1, dtype=np.uint32)
                    self.packed[start // 16: start // 16 + len(words)] = words
                carry = buf[whole:]
                carry_start = start + whole
                g += len(codes)
        if len(carry):
            tail = np.zeros(16, dtype=np.uint32)
            tail[:len(carry)] = carry
            self.packed[carry_start // 16] = np.uint32((tail << shifts[0]).sum())
        self.exceptions = np.zeros(len(runs), dtype=EXCEPTION_DTYPE)
        for i, (a, b, code) in enumerate(runs):
            self.exceptions[i] = (a, b, code, 0)
        self.names_pool = bytes(pool)
        return self

    # ------------------------------------------------------------------ weights
    def contig_weights(self, depths=None):
        """depth * length per contig (simulate.py:118-121) and the running sums random.choices uses."""
        depths = self.depths if depths is None else depths
        weights = [depths[n] * length for n, length in zip(self.names, self.lengths)]
        cum = np.array(list(itertools.accumulate(weights)), dtype=np.float64)
        return weights, cum

    # ------------------------------------------------------------------ host-side decode (tests, tiny inputs)
    def decode(self, contig_index, strand, start, length):
        """Bases [start, start+length) of the '+' or '-' strand string of a contig, as str."""
        ct = self.contigs[contig_index]
        out = []
        for p in range(start, start + length):
            f = p if strand == '+' else int(ct['length']) - 1 - p
            g = int(ct['base_off']) + f
            code = (int(self.packed[g >> 4]) >> (2 * (g & 15))) & 3
            for ex in self.exceptions:
                if ex['start'] <= g < ex['end']:
                    code = int(ex['code'])
                    break
            if strand != '+':
                code = int(self.comp[code])
            out.append(chr(self.sym[code]))
        return ''.join(out)

    def encode_seq(self, seq):
        """str -> uint8 base codes using this reference's alphabet (unknown symbols -> N)."""
        lut = np.full(256, 4, dtype=np.uint8)
        for ch, code in self.code_of.items():
            lut[ord(ch)] = code
        return lut[np.frombuffer(seq.encode('latin-1'), dtype=np.uint8)]


def encode_acgt(seq):
    """ACGT str -> codes 0..3; anything else -> 4 (N)."""
    lut = np.full(256, 4, dtype=np.uint8)
    for code, ch in enumerate('ACGT'):
        lut[ord(ch)] = code
    return lut[np.frombuffer(seq.encode('latin-1'), dtype=np.uint8)]

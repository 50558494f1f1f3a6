"""
Pure-Python restatement of the counter-based draws of include/brx_spec.h (Philox4x32-10 keyed by
(seed, read, stream, index)).  TEST INFRASTRUCTURE: used to script the reference's random sources
when generating tests/golden/ (tools/make_golden.py) and to pin the C generator against the
published Philox known-answer vectors (tests/test_spec.py).
"""
M32 = 0xFFFFFFFF
ST_PLAN, ST_BASES, ST_MUT, ST_WIN, ST_QS, ST_NAME = 1, 2, 3, 4, 5, 6


def philox4x32_10(ctr, key):
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(10):
        p0 = 0xD2511F53 * c0
        p1 = 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c3 ^ k1) & M32, p0 & M32
        k0 = (k0 + 0x9E3779B9) & M32
        k1 = (k1 + 0xBB67AE85) & M32
    return [c0, c1, c2, c3]


def draw4(seed, read, stream, index):
    ctr = [index & M32, (index >> 32) & M32, read & M32, ((read >> 32) & 0x00FFFFFF) | (stream << 24)]
    return philox4x32_10(ctr, [seed & M32, (seed >> 32) & M32])


def mulhi64(x, n):
    return (x * n) >> 64


def random_base(seed, read, serial, pos):
    o = draw4(seed, read, ST_BASES, (serial << 32) | (pos >> 6))
    return (o[(pos >> 4) & 3] >> (2 * (pos & 15))) & 3


class Draws(object):
    """The position-addressed streams of one read."""

    def __init__(self, seed, read):
        self.seed, self.read = seed, read

    def mut(self, iteration):
        return draw4(self.seed, self.read, ST_MUT, iteration)

    def win(self, serial):
        return draw4(self.seed, self.read, ST_WIN, serial)

    def qs(self, pos):
        return draw4(self.seed, self.read, ST_QS, pos >> 2)[pos & 3]

    @staticmethod
    def below64(lo, hi, n):
        return mulhi64((hi << 32) | lo, n)


def draw4_many(seed, read, stream, indices):
    """Vectorised draw4 (numpy): (len(indices), 4) uint32 words of the blocks `indices` of one read's stream."""
    import numpy as np
    idx = np.asarray(indices, dtype=np.uint64)
    c0 = idx & np.uint64(M32)
    c1 = (idx >> np.uint64(32)) & np.uint64(M32)
    c2 = np.full(len(idx), read & M32, dtype=np.uint64)
    c3 = np.full(len(idx), ((read >> 32) & 0x00FFFFFF) | (stream << 24), dtype=np.uint64)
    k0, k1 = seed & M32, (seed >> 32) & M32
    m = np.uint64(M32)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        c0, c1, c2, c3 = ((p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)) & m, p1 & m, ((p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)) & m, p0 & m
        k0 = (k0 + 0x9E3779B9) & M32
        k1 = (k1 + 0xBB67AE85) & M32
    return np.stack([c0, c1, c2, c3], axis=1).astype(np.uint32)

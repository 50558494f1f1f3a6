"""
The numerical spec (include/brx_spec.h): Philox4x32-10 against the published Random123
known-answer vectors, the Python restatement used to script the reference (tests/philox.py)
against the C generator, and the samplers against the laws the reference draws from
(np.random.gamma / beta / normal / geometric; fragment_lengths.py:51, identities.py:89-92,
simulate.py:387,466-478) by KS test.  CPU only.
"""
import numpy as np
import pytest
import scipy.stats

import philox
import pyoracle


def test_philox_known_answers():
    # Random123 kat_vectors: philox4x32-10
    assert philox.philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert philox.philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert philox.philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_c_generator_matches_restatement():
    rng = np.random.default_rng(0)
    for _ in range(200):
        seed, read = int(rng.integers(0, 2 ** 63)), int(rng.integers(0, 2 ** 56))
        stream, index = int(rng.integers(1, 7)), int(rng.integers(0, 2 ** 40))
        assert pyoracle.draw4(seed, read, stream, index) == philox.draw4(seed, read, stream, index)
    # layout of brx_draw4: counter = (index lo, index hi, read lo, read hi | stream << 24), key = seed
    assert pyoracle.draw4(0, 0, 0, 0) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]


N = 40000
ALPHA = 1e-3


def _ks(sample, cdf):
    stat, p = scipy.stats.kstest(sample, cdf)
    assert p > ALPHA, f'KS D={stat:.4f} p={p:.2e}'


def test_uniform_and_below():
    _ks(pyoracle.sample('uniform', 0, 0, 1, N), 'uniform')
    x = pyoracle.sample('below', 1000, 0, 2, N)
    assert x.min() >= 0 and x.max() <= 999 and abs(x.mean() - 499.5) < 6


@pytest.mark.parametrize('shape,scale', [(1.3314, 11266.7), (0.4, 2.0), (25.0, 1.0), (114.7676, 0.5)])
def test_gamma(shape, scale):
    _ks(pyoracle.sample('gamma', shape, scale, 3, N), scipy.stats.gamma(shape, scale=scale).cdf)


@pytest.mark.parametrize('a,b', [(57.3838, 2.41616), (1.2, 0.8), (0.4, 1.6), (8.0, 3.0)])
def test_beta(a, b):
    _ks(pyoracle.sample('beta', a, b, 4, N), scipy.stats.beta(a, b).cdf)


def test_normal():
    _ks(pyoracle.sample('normal', 30.0, 3.0, 5, N), scipy.stats.norm(30.0, 3.0).cdf)


@pytest.mark.parametrize('p', [1.0 / 10000, 1.0 / 25, 0.5, 1.0])
def test_geometric(p):
    x = pyoracle.sample('geometric', p, 0, 6, N)
    assert x.min() >= 1
    if p == 1.0:
        assert (x == 1).all()
        return
    # discrete law: compare the empirical CDF at the sample points with 1-(1-p)^k
    xs = np.sort(x)
    emp = np.arange(1, N + 1) / N
    theo = 1.0 - (1.0 - p) ** xs
    assert np.abs(emp - theo).max() < 1.95 / np.sqrt(N) + p       # KS bound + one atom


def test_log_exp_accuracy():
    # brx_log / brx_exp are restated (no libm on the device); they must stay within 1 ulp of libm
    xs = 1e-6 + np.arange(5000) * 0.37
    got = pyoracle.sample('log', 1e-6, 0.37, 0, 5000)
    assert np.max(np.abs(got - np.log(xs)) / np.maximum(np.abs(np.log(xs)), 1e-300)) < 4e-16
    xs = -40.0 + np.arange(5000) * 0.013
    got = pyoracle.sample('exp', -40.0, 0.013, 0, 5000)
    assert np.max(np.abs(got - np.exp(xs)) / np.exp(xs)) < 4e-16

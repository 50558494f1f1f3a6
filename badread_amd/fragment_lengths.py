"""
FragmentLengths: host-side mirror of the reference class (/root/reference/badread/fragment_lengths.py:25-64).

On the accelerated path the draw itself happens inside the k_plan kernel (brx_std_gamma in
include/brx_spec.h: Marsaglia-Tsang, Philox-keyed by read index); this class keeps the
parameterisation (`gamma_parameters`), the banner lines, and a host `get_fragment_length()` with
the reference's semantics (np.random.gamma, Python round, min 1) for adjust_depths
(simulate.py:516-536) and for callers that use the class directly.
"""
import math
import sys

import numpy as np

from .misc import float_to_str, print_in_two_columns


class FragmentLengths(object):
    """mean / stdev of the fragment-length law; `gamma_k`, `gamma_t` are None for a constant length."""

    def __init__(self, mean, stdev, output=sys.stderr):
        self.mean, self.stdev = mean, stdev
        self.gamma_k = self.gamma_t = None
        constant = stdev == 0
        if not constant:
            shape, rate, self.gamma_k, self.gamma_t = gamma_parameters(mean, stdev)
        self._announce(output, None if constant else (shape, rate))

    def _announce(self, output, shape_rate):
        """The banner of the reference (fragment_lengths.py:28-43), minus its ASCII histogram."""
        print('', file=output)
        if shape_rate is None:
            print(f'Using a constant fragment length of {self.mean} bp', file=output)
            return
        print('Generating fragment lengths from a gamma distribution:', file=output)
        n50 = int(round(find_n_value(shape_rate[0], shape_rate[1], 50)))
        left = [f'  {label} = {value:>6} bp' for label, value in
                (('mean ', float_to_str(self.mean)), ('stdev', float_to_str(self.stdev)), ('N50  ', n50))]
        right = ['parameters:', f'  k (shape)     = {self.gamma_k:.4e}', f'  theta (scale) = {self.gamma_t:.4e}']
        print_in_two_columns(*left, *right, output=output)

    def get_fragment_length(self):
        """One draw with the reference's rounding: numpy's global gamma, Python round(), at least 1."""
        if self.gamma_k is None:
            return int(round(self.mean))
        drawn = np.random.gamma(self.gamma_k, self.gamma_t)
        return max(1, int(round(drawn)))

    def sample_many(self, count, rng):
        """`count` draws with the same law, from a caller-owned numpy RandomState (adjust_depths)."""
        if self.stdev == 0:
            return np.full(count, int(round(self.mean)), dtype=np.int64)
        return np.maximum(np.rint(rng.gamma(self.gamma_k, self.gamma_t, size=count)).astype(np.int64), 1)


def gamma_parameters(gamma_mean, gamma_stdev):
    """(shape a, rate b, shape k, scale t) -- fragment_lengths.py:55-64."""
    shape = (gamma_mean ** 2) / (gamma_stdev ** 2)
    return shape, gamma_mean / (gamma_stdev ** 2), shape, (gamma_stdev ** 2) / gamma_mean


def find_n_value(a, b, n):
    """
    Length L such that fragments <= L hold n% of the bases (N50 for n=50): the base-weighted
    length distribution of gamma(a, rate b) is gamma(a+1, rate b), so this is its quantile.  The
    reference binary-searches the same integral (fragment_lengths.py:67-117); banner use only.
    """
    target = 1.0 - n / 100.0
    shape = a + 1.0
    lo, hi = 0.0, shape + 10.0 * math.sqrt(shape) + 50.0          # in units of the scale 1 / b
    while _gamma_p(shape, hi) < target:
        hi *= 2.0
    for _ in range(200):                      # bisection on the regularised incomplete gamma function (no scipy import: 0.5 s of start-up)
        mid = 0.5 * (lo + hi)
        if _gamma_p(shape, mid) < target:
            lo = mid
        else:
            hi = mid
        if hi - lo <= 1e-13 * hi:
            break
    return 0.5 * (lo + hi) / b


def _gamma_p(a, x):
    """Regularised lower incomplete gamma function P(a, x): its power series below a + 1, the continued fraction of Q above."""
    if x <= 0.0:
        return 0.0
    log_front = a * math.log(x) - x - math.lgamma(a)
    if x < a + 1.0:
        term = total = 1.0 / a
        k = a
        for _ in range(100000):
            k += 1.0
            term *= x / k
            total += term
            if abs(term) < abs(total) * 1e-16:
                break
        return total * math.exp(log_front)
    tiny = 1e-300
    b0 = x + 1.0 - a
    c, d = 1.0 / tiny, 1.0 / b0
    h = d
    for i in range(1, 100000):
        an = -i * (i - a)
        b0 += 2.0
        d = an * d + b0
        d = tiny if abs(d) < tiny else d
        c = b0 + an / c
        c = tiny if abs(c) < tiny else c
        d = 1.0 / d
        delta = d * c
        h *= delta
        if abs(delta - 1.0) < 1e-16:
            break
    return 1.0 - math.exp(log_front) * h

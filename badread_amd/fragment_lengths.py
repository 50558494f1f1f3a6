"""
FragmentLengths: host-side mirror of the reference class (/root/reference/badread/fragment_lengths.py:25-64).

On the accelerated path the draw itself happens inside the k_plan kernel (brx_std_gamma in
include/brx_spec.h: Marsaglia-Tsang, Philox-keyed by read index); this class keeps the
parameterisation (`gamma_parameters`), the banner lines, and a host `get_fragment_length()` with
the reference's semantics (np.random.gamma, Python round, min 1) for adjust_depths
(simulate.py:516-536) and for callers that use the class directly.
"""
import sys

import numpy as np

from .misc import float_to_str, print_in_two_columns


class FragmentLengths(object):

    def __init__(self, mean, stdev, output=sys.stderr):
        self.mean = mean
        self.stdev = stdev
        print('', file=output)
        if self.stdev == 0:
            self.gamma_k, self.gamma_t = None, None
            print(f'Using a constant fragment length of {mean} bp', file=output)
        else:
            print('Generating fragment lengths from a gamma distribution:', file=output)
            gamma_a, gamma_b, self.gamma_k, self.gamma_t = gamma_parameters(mean, stdev)
            n50 = int(round(find_n_value(gamma_a, gamma_b, 50)))
            print_in_two_columns(f'  mean  = {float_to_str(mean):>6} bp',
                                 f'  stdev = {float_to_str(stdev):>6} bp',
                                 f'  N50   = {n50:>6} bp',
                                 'parameters:',
                                 f'  k (shape)     = {self.gamma_k:.4e}',
                                 f'  theta (scale) = {self.gamma_t:.4e}',
                                 output=output)

    def get_fragment_length(self):
        if self.stdev == 0:
            return int(round(self.mean))
        return max(int(round(np.random.gamma(self.gamma_k, self.gamma_t))), 1)

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
    import scipy.stats
    return float(scipy.stats.gamma.ppf(1.0 - n / 100.0, a + 1.0, scale=1.0 / b))

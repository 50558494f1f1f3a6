"""
Identities: per-read target identity distribution, interface-compatible with the reference class
(/root/reference/badread/identities.py:22-103): `Identities(mean, stdev, max_identity, output)`,
attributes `.type/.mean/.stdev/.max_identity/.beta_a/.beta_b`, method `.get_identity()`.

On the accelerated path the per-read draw happens inside the k_plan kernel (brx_beta / brx_normal,
include/brx_spec.h).  This class owns the parameterisation and maps it onto
brx_sim_params.identity_mode through `device_mode()`:
    0  constant                        mean == max, or stdev == 0 with three parameters
    1  max * beta(a, b)                three-parameter mode  (identities.py:86-89)
    2  1 - 10^(-N(mean, stdev)/10)     two-parameter qscore mode (identities.py:91-93)
"""
import sys

import numpy as np

from .misc import float_to_str, print_in_two_columns


def beta_parameters(beta_mean, beta_stdev, beta_max):
    """Shape parameters of the beta law scaled to [0, max] with the given mean/stdev (identities.py:96-103)."""
    # same operation order as the reference, so the shape parameters agree to the last bit
    rel_mean = beta_mean / beta_max
    inv_mean = beta_max / beta_mean
    alpha = (((1 - rel_mean) / ((beta_stdev / beta_max) ** 2)) - inv_mean) * (rel_mean ** 2)
    beta = alpha * (inv_mean - 1)
    if alpha < 0.0 or beta < 0.0:
        sys.exit('Error: invalid beta parameters for identity distribution - trying increasing '
                 'the maximum identity or reducing the standard deviation')
    return alpha, beta


class Identities(object):

    def __init__(self, mean, stdev, max_identity, output=sys.stderr):
        self.beta_a = self.beta_b = None
        print('', file=output)
        if max_identity is not None:
            self.type = 'beta'
            self.mean, self.stdev, self.max_identity = mean / 100.0, stdev / 100.0, max_identity / 100.0
            constant = self.mean == self.max_identity
            if not constant and self.stdev == 0.0:
                self.max_identity, constant = self.mean, True
            if constant:
                print(f'Using a constant read identity of {self.mean * 100}%', file=output)
                return
            self.beta_a, self.beta_b = beta_parameters(mean, stdev, max_identity)
            print('Generating read identities from a beta distribution:', file=output)
            pct = [f'{float_to_str(v * 100):>3}%' for v in (self.mean, self.max_identity, self.stdev)]
            print_in_two_columns(f'  mean  = {pct[0]}', f'  max   = {pct[1]}', f'  stdev = {pct[2]}',
                                 'shape parameters:', f'  alpha = {self.beta_a:.4e}', f'  beta  = {self.beta_b:.4e}',
                                 output=output)
        else:
            self.type = 'normal'
            self.mean, self.stdev, self.max_identity = mean, stdev, None
            if stdev == 0.0:
                self.max_identity = mean
                print(f'Using a constant read qscore of {mean}', file=output)
            else:
                print('Generating read qscores from a normal distribution:', file=output)
                for label, value in (('mean ', mean), ('stdev', stdev)):
                    print(f'  {label} = {float_to_str(value):>3}', file=output)

    # host-side draws (numpy global state, like the reference); the GPU path never calls these
    def get_beta_identity(self):
        if self.beta_a is None:
            return self.mean
        return self.max_identity * np.random.beta(self.beta_a, self.beta_b)

    def get_normal_identity(self):
        return 1.0 - 10 ** (-np.random.normal(self.mean, self.stdev) / 10)

    def get_identity(self):
        draw = self.get_beta_identity if self.type == 'beta' else self.get_normal_identity
        identity = draw()
        while not 0 <= identity <= 100:
            identity = draw()
        return identity

    def device_mode(self):
        """(identity_mode, id_a, id_b, id_max) for brx_sim_params."""
        if self.type == 'normal':
            return 2, float(self.mean), float(self.stdev), 1.0
        if self.beta_a is None:
            return 0, 0.0, 0.0, float(self.mean)
        return 1, float(self.beta_a), float(self.beta_b), float(self.max_identity)

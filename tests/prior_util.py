"""Synthetic parameter draws for tests/bench: the reference's priors (basic_ddm_dc.py:62-80,
single_trial_alpha_not_scaled.py:78-102, alpha_not_scaled.py:66-72) vectorised with default_rng."""
import numpy as np
from scipy.stats import truncnorm


def _tn(rng, mean, sd, low, upp, size):
    return truncnorm.rvs((low - mean) / sd, (upp - mean) / sd, loc=mean, scale=sd, size=size, random_state=rng)


def basic_prior(B, seed=2023):
    rng = np.random.default_rng(seed)
    return np.stack([rng.normal(0.0, 2.0, B), _tn(rng, 1.0, .5, 0.0, 10.0, B), rng.beta(2.0, 2.0, B),
                     _tn(rng, .5, .25, 0.0, 1.5, B), _tn(rng, 1.0, .5, 0.0, 10.0, B)], axis=1).astype(np.float32)


def single_prior(B, seed=2023, gamma=1.0):
    rng = np.random.default_rng(seed)
    return np.stack([rng.normal(0.0, 2.0, B), _tn(rng, 1.0, .5, 0.0, 10.0, B), rng.beta(2.0, 2.0, B),
                     _tn(rng, .5, .25, 0.0, 1.5, B), _tn(rng, 1.0, .5, 0.0, 3.0, B), _tn(rng, 1.0, .5, 0.0, 10.0, B),
                     rng.uniform(0.0, 5.0, B), np.full(B, gamma)], axis=1).astype(np.float32)


def alpha_ns_prior(B, seed=2021):
    rng = np.random.default_rng(seed)
    # Nu, Alpha, Beta, Tau, Eta, Varsigma  (alpha_not_scaled.py:66-72)
    return np.stack([rng.uniform(-4, 4, B), rng.uniform(.8, 1.4, B), rng.uniform(.3, .7, B), rng.uniform(.15, .6, B),
                     rng.uniform(0, 2, B), rng.uniform(.8, 1.4, B)], axis=1).astype(np.float32)

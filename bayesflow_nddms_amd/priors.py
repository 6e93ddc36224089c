"""Prior and context samplers.

Host versions restate the reference's draw order call for call (basic_ddm_dc.py:50-80,
single_trial_alpha_not_scaled.py:66-102, :889-913, :1205-1232): `RNG` is a module-level default_rng(2023) (PCG64)
used for drift / beta / sigma1 / gamma, the truncated normals go through scipy.stats.truncnorm.rvs on NumPy's
GLOBAL legacy stream, and prior_N uses np.random.randint -- so with the same seeds they return the reference's
numbers bit for bit (tests/golden/priors.npz).

DevicePrior is the batched on-device sampler (SURVEY f-1): the per-set SciPy calls cost ~0.5 ms per draw, i.e.
minutes per million sets, which would dwarf the simulate time; the device version fills [B, P] in one launch and is
checked per marginal by KS against the reference draws.
"""
import numpy as np
from scipy.stats import truncnorm

from . import engine


def prior_N(n_min=60, n_max=300):
    """A prior for the random number of observations (basic_ddm_dc.py:50-52); global NumPy stream."""
    return np.random.randint(n_min, n_max + 1)


def truncnorm_better(mean=0, sd=1, low=-10, upp=10, size=1):
    """basic_ddm_dc.py:55-57."""
    return truncnorm.rvs((low - mean) / sd, (upp - mean) / sd, loc=mean, scale=sd, size=size)


RNG = np.random.default_rng(2023)   # basic_ddm_dc.py:60 / single_trial_alpha_not_scaled.py:76


def reset_host_rng(seed=2023):
    """Re-create the module-level generator (what re-importing the reference script does)."""
    global RNG
    RNG = np.random.default_rng(seed)


def draw_prior_basic():
    """basic_ddm_dc.py:62-80 -> [drift, alpha, beta, ter, dc]."""
    drift = RNG.normal(0.0, 2.0)
    alpha = truncnorm_better(mean=1.0, sd=0.5, low=0.0, upp=10)[0]
    beta = RNG.beta(2.0, 2.0)
    ter = truncnorm_better(mean=0.5, sd=0.25, low=0.0, upp=1.5)[0]
    dc = truncnorm_better(mean=1.0, sd=0.5, low=0.0, upp=10)[0]
    return np.hstack((drift, alpha, beta, ter, dc))


def draw_prior_single():
    """single_trial_alpha_not_scaled.py:78-102 -> [drift, mu_alpha, beta, ter, std_alpha, dc, sigma1]
    (draw_prior_alt :889-913 has the same marginals with std_dc / mu_dc at indices 4 / 5)."""
    drift = RNG.normal(0.0, 2.0)
    mu_alpha = truncnorm_better(mean=1.0, sd=0.5, low=0.0, upp=10)[0]
    beta = RNG.beta(2.0, 2.0)
    ter = truncnorm_better(mean=0.5, sd=0.25, low=0.0, upp=1.5)[0]
    std_alpha = truncnorm_better(mean=1.0, sd=0.5, low=0.0, upp=3)[0]
    dc = truncnorm_better(mean=1.0, sd=0.5, low=0.0, upp=10)[0]
    sigma1 = RNG.uniform(0.0, 5.0)
    return np.hstack((drift, mu_alpha, beta, ter, std_alpha, dc, sigma1))


def draw_prior_scale():
    """single_trial_alpha_not_scaled.py:1205-1232: the 7 parameters above + gamma ~ U(0, 2)."""
    p = draw_prior_single()
    gamma = RNG.uniform(0.0, 2.0)
    return np.hstack((p, gamma))


# the supports of the priors above (drift ~ N(0, 2): +- 5 sd), per parameter in the reference's order
PRIOR_SUPPORT = {"basic": ([-10.0, 0.0, 0.0, 0.0, 0.0], [10.0, 10.0, 1.0, 1.5, 10.0]),
                 "single": ([-10.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0], [10.0, 10.0, 1.0, 1.5, 3.0, 10.0, 5.0])}


def prior_box(model="basic", widen=1.0):
    """(low, high): the prior's support widened by `widen` times its own width on each side -- the generous box
    `AmortizedPosterior.sample(..., reject_outside=prior_box(model))` redraws outside of.  A posterior draw beyond it is not a
    parameter value the model could have produced the data with; the reference itself counts "model fits in the prior range"
    before plotting (basic_ddm_dc.py:239-241)."""
    lo, hi = (np.asarray(v, dtype=np.float64) for v in PRIOR_SUPPORT[model])
    return lo - widen * (hi - lo), hi + widen * (hi - lo)


class DevicePrior:
    """Batched on-device draw_prior: `DevicePrior('basic')(B)` -> torch float32 [B, P] on the GPU.

    model: 'basic' (P=5) | 'single' (P=7) | 'scale' (P=8, gamma ~ U(0,2)).  Counter-based: row i of call c depends
    only on (seed, global row index), so shards of a batch drawn on different GPUs are consistent."""

    _NCOLS = {"basic": 5, "single": 7, "scale": 8}

    def __init__(self, model="basic", seed=2023, stream_state=None):
        if model not in self._NCOLS:
            raise ValueError(f"unknown prior model {model!r}")
        self.model = model
        self.state = stream_state or engine.StreamState(seed=seed)

    def __call__(self, batch_size, set_offset=None):
        seed, off = self.state.take(batch_size) if set_offset is None else (self.state.seed, set_offset)
        if self.model == "basic":
            return engine.draw_prior_device(engine.BASIC_DDM_DC, batch_size, seed=seed, set_offset=off)
        gamma = -1.0 if self.model == "scale" else 1.0
        out = engine.draw_prior_device(engine.SINGLE_TRIAL, batch_size, seed=seed, set_offset=off, gamma=gamma)
        return out[:, :self._NCOLS[self.model]].contiguous()


# ---------------------------------------------------------------------------------------------------------------
# Vectorised host draws of the same marginals (one SciPy call per column instead of three per parameter set):
# synthetic parameter matrices for benchmarks, tests and bulk simulation.  Same distributions as draw_prior_*,
# but a different (seeded, default_rng) stream -- use the call-for-call functions above for reference-identical draws.
def _tn_vec(rng, mean, sd, low, upp, size):
    return truncnorm.rvs((low - mean) / sd, (upp - mean) / sd, loc=mean, scale=sd, size=size, random_state=rng)


def basic_prior_matrix(B, seed=2023):
    """float32 [B, 5]: drift, boundary, beta, tau, dc  (basic_ddm_dc.py:62-80)."""
    rng = np.random.default_rng(seed)
    return np.stack([rng.normal(0.0, 2.0, B), _tn_vec(rng, 1.0, .5, 0.0, 10.0, B), rng.beta(2.0, 2.0, B),
                     _tn_vec(rng, .5, .25, 0.0, 1.5, B), _tn_vec(rng, 1.0, .5, 0.0, 10.0, B)], axis=1).astype(np.float32)


def single_prior_matrix(B, seed=2023, gamma=1.0):
    """float32 [B, 8]: drift, mu_alpha, beta, ter, std_alpha, dc, sigma1, gamma  (single_trial_alpha_not_scaled.py:78-102)."""
    rng = np.random.default_rng(seed)
    return np.stack([rng.normal(0.0, 2.0, B), _tn_vec(rng, 1.0, .5, 0.0, 10.0, B), rng.beta(2.0, 2.0, B),
                     _tn_vec(rng, .5, .25, 0.0, 1.5, B), _tn_vec(rng, 1.0, .5, 0.0, 3.0, B),
                     _tn_vec(rng, 1.0, .5, 0.0, 10.0, B), rng.uniform(0.0, 5.0, B), np.full(B, gamma)],
                    axis=1).astype(np.float32)


def alpha_ns_prior_matrix(B, seed=2021):
    """float32 [B, 6]: Nu, Alpha, Beta, Tau, Eta, Varsigma  (alpha_not_scaled.py:66-72)."""
    rng = np.random.default_rng(seed)
    return np.stack([rng.uniform(-4, 4, B), rng.uniform(.8, 1.4, B), rng.uniform(.3, .7, B), rng.uniform(.15, .6, B),
                     rng.uniform(0, 2, B), rng.uniform(.8, 1.4, B)], axis=1).astype(np.float32)

"""Drop-in for the data-generation section of the reference's alpha_not_scaled.py (lines 52-128): a dcDDM with
per-participant parameters, per-trial drift ~ N(delta, deltatrialsd), and one external datum per participant
extdata[p] ~ N(alpha[p], sigma) that identifies the diffusion coefficient.

The reference generates the choice-RTs with pyhddmjagsutils.simulratcliff, an EXACT first-passage sampler
(pyhddmjagsutils.py:47-176), not with Euler-Maruyama.  Here the same process is integrated by the Euler-Maruyama
HIP kernel (north_star).  Plain Euler-Maruyama monitors the boundaries only on the time grid and therefore detects
crossings late by O(sqrt(dt)) (KS ~0.07 vs the exact sampler at dt=.001 for fast trials); by default this module
switches on the kernel's Brownian-bridge boundary correction (bridge=True), which samples the between-grid-point
crossings with their exact conditional probability and brings the KS distance to the exact sampler below 0.01 at
dt=.001 (tests/test_gpu_distribution.py).  bridge=False gives the plain scheme of the other models.

Since round 6 the reference's own generator exists on the device as well: `simulratcliff(...)` below and `generate_data(...,
method="exact")` run pyhddmjagsutils.simulratcliff's algorithm -- the random walk on spheres with its rejection step, no time grid --
through nddm_simulratcliff (include/nddm.h; csrc/nddm_ratcliff.h), bit-equal to its CPU restatement in exact mode and at the
two-sample noise floor against the reference's own draws (tests/golden/ratcliff.npz).  The Euler-Maruyama forms stay: they are what
north_star names, and `method="em"` remains the default of generate_data.
"""
import numpy as np

from . import engine

PARAM_NAMES = ("Nu", "Alpha", "Beta", "Tau", "Eta", "Varsigma")
SIGMA_OF_TEST = {1: .5, 2: .1, 3: .01, 4: .2}   # alpha_not_scaled.py:73-81


def simulratcliff_em(N=100, Alpha=1, Tau=.4, Nu=1, Beta=.5, Eta=.3, Varsigma=1, dt=.001, max_steps=4000,
                     seed=None, set_offset=None, fast=None, bridge=True):
    """Same call shape as simulratcliff(N, Alpha, Tau, Nu, Beta, Eta=, Varsigma=) (pyhddmjagsutils.py:47) without
    the range* arguments the generator never uses: signed RTs float64 [N] (negative = response B)."""
    if (Nu < -5) or (Nu > 5):          # pyhddmjagsutils.py:102-103
        Nu = np.sign(Nu) * 5
    r = engine.simulate(engine.ALPHA_NOT_SCALED, [[Nu, Alpha, Beta, Tau, Eta, Varsigma]], N, dt=dt,
                        max_steps=max_steps, seed=seed, set_offset=set_offset, fast=fast, bridge=bridge,
                        want_summary=False)
    return r["trials"][0, :, 0].cpu().numpy().astype(np.float64)


def simulratcliff(N=100, Alpha=1, Tau=.4, Nu=1, Beta=.5, rangeTau=0, rangeBeta=0, Eta=.3, Varsigma=1, seed=None, set_offset=None, fast=None):
    """pyhddmjagsutils.simulratcliff (:47-176) with its own signature, on the device: signed RTs float64 [N] (negative = response B)
    from the exact first-passage sampler.  rangeTau / rangeBeta must be 0 (the generator never passes them: alpha_not_scaled.py:96-97)."""
    if rangeTau != 0 or rangeBeta != 0:
        raise ValueError("the device sampler takes rangeTau = rangeBeta = 0 (as alpha_not_scaled.py:96-97 calls it)")
    r = engine.simulratcliff([[Nu, Alpha, Beta, Tau, Eta, Varsigma]], N, seed=seed, set_offset=set_offset, fast=fast, want_summary=False)
    return r["trials"][0, :, 0].cpu().numpy().astype(np.float64)


def draw_participants(nparts=100, seed=2021):
    """alpha_not_scaled.py:64-72, 83-88: participant-level parameters on the global NumPy stream, index 17 fixed."""
    np.random.seed(seed)
    ndt = np.random.uniform(.15, .6, size=nparts)
    alpha = np.random.uniform(.8, 1.4, size=nparts)
    beta = np.random.uniform(.3, .7, size=nparts)
    delta = np.random.uniform(-4, 4, size=nparts)
    varsigma = np.random.uniform(.8, 1.4, size=nparts)
    deltatrialsd = np.random.uniform(0, 2, size=nparts)
    if nparts > 17:
        ndt[17], alpha[17], beta[17], delta[17], varsigma[17], deltatrialsd[17] = .4, 1.2, .5, 3.5, 1.2, 1
    return dict(ndt=ndt, alpha=alpha, beta=beta, delta=delta, varsigma=varsigma, deltatrialsd=deltatrialsd)


def generate_data(test_num=2, nparts=100, ntrials=100, seed=2021, dt=.001, max_steps=4000, sim_seed=None,
                  set_offset=None, fast=None, bridge=True, method="em"):
    """alpha_not_scaled.py:52-128 in one launch: returns the `genparam` dictionary the reference saves to .mat
    (same keys), all participants simulated as one batch of `nparts` parameter sets x `ntrials` trials.
    method="em": the Euler-Maruyama kernel (with the bridge correction by default); method="exact": the reference's own generator,
    simulratcliff, on the device (dt / max_steps / bridge are then unused)."""
    if method not in ("em", "exact"):
        raise ValueError("method must be 'em' or 'exact'")
    sigma = SIGMA_OF_TEST[test_num]
    par = draw_participants(nparts, seed)
    P = np.stack([np.clip(par["delta"], -5, 5), par["alpha"], par["beta"], par["ndt"], par["deltatrialsd"],
                  par["varsigma"]], axis=1)
    if method == "exact":
        r = engine.simulratcliff(P, ntrials, seed=seed if sim_seed is None else sim_seed, set_offset=0 if set_offset is None else set_offset,
                                 fast=fast, ext_sigma=sigma, ext_mode=1 if test_num == 4 else 0, want_ext=True, want_summary=False)
    else:
        r = engine.simulate(engine.ALPHA_NOT_SCALED, P, ntrials, dt=dt, max_steps=max_steps,
                            seed=seed if sim_seed is None else sim_seed, set_offset=0 if set_offset is None else set_offset,
                            fast=fast, bridge=bridge, ext_sigma=sigma, ext_mode=1 if test_num == 4 else 0, want_ext=True,
                            want_summary=False)
    y = engine.to_host(r["trials"][..., 0]).astype(np.float64).reshape(-1)
    N = nparts * ntrials
    var_alpha = (1 / 12) * (1.4 - .8) ** 2
    genparam = dict(par)
    genparam.update(sigma=sigma, var_alpha=var_alpha, prop_cog_var=var_alpha / (var_alpha + sigma ** 2),
                    rt=np.abs(y), acc=(np.sign(y) + 1) / 2, y=y,
                    extdata=r["ext"].cpu().numpy().astype(np.float64),
                    participant=np.repeat(np.arange(1, nparts + 1), ntrials).astype(np.float64),
                    nparts=nparts, ntrials=ntrials, N=N)
    return genparam

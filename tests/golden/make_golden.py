#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

The reference (mdnunez/bayesflow_nddms, mounted at /root/reference) cannot be
imported as modules here: its scripts import numba/bayesflow at the top and run
training code at import time.  Its simulator functions are, however, plain
Python under ``@njit``.  This script reads the reference files as text, takes
the line ranges that hold the simulator / prior functions, and ``exec``s them
with ``njit`` bound to the identity decorator.  Nothing of the reference's
source is written into this repository: only the arrays the functions return.

Under this identity-njit shim ``np.random.seed(s)`` controls the stream
(global MT19937 + NumPy's legacy polar Gaussian), which is what makes
"NumPy reference at fixed PRNG seed" well defined.

Outputs (all under tests/golden/):
  kat.npz       bit-exact known answers: seeded simulate_trials() outputs of every
                reference simulator variant, prior draws, prior_N draws
  ks_hist.npz   distribution tier: per parameter set, histogram over the integer
                Euler-Maruyama step index split by choice (>= 4e5 trials per set),
                for dt=.01/max 400 (reference default) and dt=.001/max 4000
  ratcliff.npz  simulratcliff (exact first-passage sampler) quantile tables for
                the alpha_not_scaled parameter ranges
  mixture.npz   2000 parameter sets drawn from the basic_ddm_dc prior x 200 reference trials each, pooled
                histogram of the signed step index (both dt configurations)
  ks_variants.npz  distribution tier of the misspecification / imputation simulators (_alt, _scale, _scale2,
                explicit per-trial boundary): step-index histograms by choice + quantile tables of the external
                datum, >= 4e5 reference trials per parameter set, both dt configurations

  ezdiff.npz    known answers of the EZ-diffusion estimator (simulations/Basic_DDM_simulations.py:131-158) on reference
                choice-RT data

Usage:  python tests/golden/make_golden.py [--kat] [--ks] [--variants] [--ratcliff] [--priors] [--ezdiff] [--procs 8]
This only runs in the build container (it needs /root/reference); the GPU box
only ever sees the .npz files.
"""
import argparse
import os
import sys
import textwrap
import time
from multiprocessing import Pool

import numpy as np

REF = os.environ.get("NDDM_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))

# ---------------------------------------------------------------------------
# reference line ranges (1-based, inclusive) -- what each slice defines
# ---------------------------------------------------------------------------
SLICES = {
    # basic_ddm_dc.py: prior_N, truncnorm_better, RNG, draw_prior
    "basic_prior": ("basic_ddm_dc.py", 50, 80),
    # basic_ddm_dc.py: diffusion_trial, simulate_trials
    "basic_sim": ("basic_ddm_dc.py", 85, 125),
    # basic_ddm_dc.py: configurator
    "basic_conf": ("basic_ddm_dc.py", 139, 160),
    "single_prior": ("single_trial_alpha_not_scaled.py", 66, 102),
    "single_sim": ("single_trial_alpha_not_scaled.py", 107, 155),
    "single_conf": ("single_trial_alpha_not_scaled.py", 169, 191),
    # misspecification variants (indented inside `if test_misspecification:`)
    "single_alt": ("single_trial_alpha_not_scaled.py", 926, 974),
    "single_scale": ("single_trial_alpha_not_scaled.py", 1237, 1285),
    "single_scale2": ("single_trial_alpha_not_scaled.py", 1471, 1519),
    # explicit per-trial boundary variant
    "imputation_sim": ("imputation_from_stahl_not_scaled.py", 120, 148),
    # exact first-passage sampler
    "ratcliff": ("pyhddmjagsutils.py", 47, 176),
    # alpha_not_scaled.py participant-level draws: seed + uniform draws (:63-72) and the fixed participant (:82-88)
    "alpha_ns_draws": ("alpha_not_scaled.py", 63, 72),
    "alpha_ns_fixed": ("alpha_not_scaled.py", 82, 88),
    # EZ-diffusion estimator fed by the per-simulation summaries
    "ezdiff": ("simulations/Basic_DDM_simulations.py", 131, 158),
}


def load_slice(name, extra=None):
    fname, lo, hi = SLICES[name]
    with open(os.path.join(REF, fname)) as f:
        lines = f.readlines()
    src = textwrap.dedent("".join(lines[lo - 1:hi]))
    from scipy.stats import truncnorm
    ns = {"np": np, "njit": (lambda f: f), "truncnorm": truncnorm}
    if extra:
        ns.update(extra)
    exec(compile(src, f"<reference {fname}:{lo}-{hi}>", "exec"), ns)
    return ns


# ---------------------------------------------------------------------------
# parameter sets of the distribution tier
# ---------------------------------------------------------------------------
# basic_ddm_dc order: drift, boundary, beta, tau, dc   (basic_ddm_dc.py:118)
BASIC_SETS = np.array([
    [1.5, 1.2, 0.5, 0.35, 1.0],    # KAT set; simulations/Basic_DDM_simulations.py:164 (1.2,1.5,1)
    [3.0, 2.4, 0.5, 0.35, 2.0],    # Basic_DDM_simulations.py (2.4,3,2)
    [3.0, 1.2, 0.5, 0.35, 2.0],    # (1.2,3,2)
    [1.5, 1.2, 0.5, 0.35, 0.5],    # (1.2,1.5,.5)
    [0.0, 1.0, 0.5, 0.50, 1.0],    # zero drift
    [-2.0, 1.5, 0.3, 0.20, 1.0],   # negative drift, biased start
    [0.5, 2.0, 0.7, 0.60, 0.6],    # long trials, some timeouts
    [4.0, 0.6, 0.5, 0.10, 1.5],    # very short trials
    [-1.0, 2.2, 0.5, 0.50, 0.8],   # many timeouts
    [2.0, 1.0, 0.2, 0.40, 0.3],    # low noise
    [0.0, 0.8, 0.5, 0.30, 2.5],    # high noise
    [1.0, 1.6, 0.6, 0.45, 1.2],
])
# single_trial order: drift, mu_alpha, beta, ter, std_alpha, dc, sigma1 (single_trial_alpha_not_scaled.py:148)
SINGLE_SETS = np.array([
    [3.0, 1.5, 0.5, 0.40, 1.0, 1.0, 0.1],   # fixed-parameter test case :852-883
    [0.0, 1.0, 0.5, 0.50, 1.0, 1.0, 1.0],
    [-2.0, 1.2, 0.4, 0.30, 0.3, 0.8, 2.5],
    [1.0, 0.5, 0.5, 0.20, 1.5, 1.2, 0.5],   # high rejection rate of the boundary draw
    [0.5, 2.0, 0.6, 0.50, 0.5, 0.7, 4.0],
    [2.0, 1.0, 0.3, 0.35, 0.2, 1.5, 0.05],
])
# alpha_not_scaled order used by this build: Nu, Alpha, Beta, Tau, Eta, Varsigma (alpha_not_scaled.py:66-72, 83-88)
RATCLIFF_SETS = np.array([
    [3.5, 1.2, 0.5, 0.40, 1.0, 1.2],    # participant 17, alpha_not_scaled.py:83-88
    [0.0, 1.0, 0.5, 0.30, 0.0, 1.0],    # Eta == 0 -> 1e-16 branch
    [-4.0, 0.8, 0.3, 0.15, 2.0, 1.4],
    [2.0, 1.4, 0.7, 0.60, 0.5, 0.8],
    [1.0, 1.1, 0.5, 0.40, 1.5, 1.0],
    # five more corners / interior points of the generator's ranges (alpha_not_scaled.py:66-72)
    [4.0, 0.8, 0.7, 0.15, 0.1, 0.8],
    [-1.0, 1.4, 0.3, 0.60, 2.0, 1.4],
    [0.5, 0.9, 0.6, 0.25, 0.7, 1.3],
    [-3.0, 1.3, 0.65, 0.50, 1.2, 0.9],
    [2.5, 1.0, 0.35, 0.45, 1.8, 1.1],
])
# _alt order: drift, alpha, beta, ter, std_dc, mu_dc, sigma1 (single_trial_alpha_not_scaled.py:966)
ALT_SETS = np.array([
    [3.0, 1.5, 0.5, 0.40, 0.5, 1.0, 0.1],    # the KAT set
    [0.5, 1.2, 0.4, 0.30, 1.0, 0.8, 1.0],    # high rejection rate of the per-trial dc draw
    [-1.5, 2.0, 0.6, 0.50, 0.2, 1.5, 2.5],
])
# _scale order: drift, mu_alpha, beta, ter, std_alpha, dc, sigma1, gamma (single_trial_alpha_not_scaled.py:1277)
SCALE_SETS = np.array([
    [3.0, 1.5, 0.5, 0.40, 1.0, 1.0, 0.1, 0.3],
    [0.0, 1.0, 0.5, 0.50, 1.0, 1.0, 1.0, 2.0],
])
# _scale2 (gamma fixed at 2, :1471-1519) takes the 7 single-trial parameters
SCALE2_SETS = np.array([
    [-2.0, 1.2, 0.4, 0.30, 0.3, 0.8, 2.5],
])
# explicit per-trial boundary (imputation_from_stahl_not_scaled.py:120-148): drift, beta, ter, dc + ONE boundary vector
EXPLICIT_SETS = np.array([
    [1.0, 0.5, 0.30, 1.1],
    [-0.5, 0.4, 0.20, 0.7],
])
EXPLICIT_NB = 400


def explicit_bounds():
    """The boundary vector of the explicit-boundary fixtures, drawn once (the recipe of the KAT vector, 400 long)."""
    rs = np.random.RandomState(5)
    return np.abs(rs.normal(1.2, 0.4, size=EXPLICIT_NB))


DT_CONFIGS = [(0.01, 400.0), (0.001, 4000.0)]
N_KS = 400_000
CHUNK = 10_000


# ---------------------------------------------------------------------------
# KAT tier
# ---------------------------------------------------------------------------
def make_kat():
    out = compute_kat()
    np.savez_compressed(os.path.join(OUT, "kat.npz"), **out)
    print("kat.npz:", {k: np.shape(v) for k, v in out.items()})


def compute_kat():
    """The known answers as a dict, nothing written (tests/test_oracle_golden.py re-derives them whenever the reference is mounted)."""
    out = {}
    b = load_slice("basic_sim")
    # (1) the reference's own call shape: simulate_trials(params, 300), default dt=.01/400
    np.random.seed(2023)
    out["basic_seed2023_p0_n300"] = b["simulate_trials"](BASIC_SETS[0], 300)
    # (2) dt=.001/max 4000 through direct diffusion_trial calls (the shape of simulate_trials_fine)
    for si in (0, 5, 7):
        for dt, ms in DT_CONFIGS:
            np.random.seed(1000 + si)
            rows = []
            for _ in range(64):
                rows.append(_basic_trial(b, BASIC_SETS[si], dt, ms))
            out[f"basic_seed{1000+si}_p{si}_dt{dt}_n64"] = np.array(rows)
    # (3) timeout path: documents the reference's unbound `choice` bug (basic_ddm_dc.py:110-111)
    np.random.seed(7)
    try:
        b["diffusion_trial"](0.0, 9.0, 0.5, 0.3, 0.2)
        out["basic_timeout_raises"] = np.array(0)
    except UnboundLocalError:
        out["basic_timeout_raises"] = np.array(1)

    s = load_slice("single_sim")
    np.random.seed(2024)
    out["single_seed2024_p0_n300"] = s["simulate_trials"](SINGLE_SETS[0], 300)
    for si in (0, 3, 4):
        for dt, ms in DT_CONFIGS:
            np.random.seed(2000 + si)
            rows = [s["diffusion_trial"](*SINGLE_SETS[si], dt=dt, max_steps=ms) for _ in range(64)]
            out[f"single_seed{2000+si}_p{si}_dt{dt}_n64"] = np.array(rows)
    # timeout path of single-trial: choicert == 0
    np.random.seed(11)
    out["single_timeout_seed11"] = np.array(
        [s["diffusion_trial"](0.0, 9.0, 0.5, 0.3, 0.01, 0.2, 1.0) for _ in range(4)])

    # misspecification variants
    a = load_slice("single_alt")
    np.random.seed(2024)
    out["alt_seed2024_n100"] = a["simulate_trials_alt"](np.array([3.0, 1.5, .5, .4, .5, 1.0, .1]), 100)
    sc = load_slice("single_scale")
    np.random.seed(2024)
    out["scale_seed2024_n100"] = sc["simulate_trials_scale"](
        np.array([3.0, 1.5, .5, .4, 1.0, 1.0, .1, 0.7]), 100)
    sc2 = load_slice("single_scale2")
    np.random.seed(2024)
    out["scale2_seed2024_n100"] = sc2["simulate_trials_scale2"](SINGLE_SETS[0], 100)

    # explicit per-trial boundary variant (imputation_from_stahl_not_scaled.py:120-148)
    im = load_slice("imputation_sim")
    np.random.seed(5)
    bounds = np.abs(np.random.normal(1.2, 0.4, size=100))
    out["imputation_bounds"] = bounds
    np.random.seed(6)
    out["imputation_seed6"] = np.array(
        [im["diffusion_trial"](1.0, bt, 0.5, 0.3, 1.1) for bt in bounds])
    try:
        im["diffusion_trial"](1.0, -0.1, 0.5, 0.3, 1.1)
        out["imputation_negative_raises"] = np.array(0)
    except ValueError:
        out["imputation_negative_raises"] = np.array(1)

    # configurator known answer (dict contract)
    c = load_slice("basic_conf")
    sim = {"sim_data": out["basic_seed2023_p0_n300"][None], "sim_non_batchable_context": 300,
           "prior_draws": BASIC_SETS[0][None]}
    conf = c["configurator"](sim)
    out["conf_direct_conditions"] = conf["direct_conditions"]
    out["conf_summary_dtype_is_f32"] = np.array(int(conf["summary_conditions"].dtype == np.float32))

    # alpha_not_scaled.py:63-72, 82-88: participant-level parameters (seed 2021, nparts = 100)
    ns = load_slice("alpha_ns_draws", extra={"nparts": 100})
    fname, lo, hi = SLICES["alpha_ns_fixed"]
    with open(os.path.join(REF, fname)) as f:
        exec(compile(textwrap.dedent("".join(f.readlines()[lo - 1:hi])), "<alpha_ns_fixed>", "exec"), ns)
    for key in ("ndt", "alpha", "beta", "delta", "varsigma", "deltatrialsd"):
        out[f"alpha_ns_part_{key}"] = ns[key]

    out["basic_sets"] = BASIC_SETS
    out["single_sets"] = SINGLE_SETS
    return out


def _basic_trial(ns, p, dt, ms):
    """Call the reference basic diffusion_trial; its timeout branch raises
    UnboundLocalError (basic_ddm_dc.py:110-111 assigns `choicert`, returns `choice`).
    The intended value is choice=0 (single_trial_alpha_not_scaled.py:140-141); rt on
    timeout is max_steps*dt+tau (basic_ddm_dc.py:103)."""
    try:
        return ns["diffusion_trial"](*p, dt=dt, max_steps=ms)
    except UnboundLocalError:
        return (ms * dt + p[3], 0)


def make_priors():
    out = {}
    for name in ("basic", "single"):
        np.random.seed(2023)
        ns = load_slice(f"{name}_prior")      # defines RNG = default_rng(2023)
        out[f"{name}_draws16"] = np.array([ns["draw_prior"]() for _ in range(16)])
        out[f"{name}_priorN16"] = np.array([ns["prior_N"]() for _ in range(16)])
        # marginal quantiles for the on-device sampler's KS check
        t0 = time.time()
        big = np.array([ns["draw_prior"]() for _ in range(100_000)])
        q = np.linspace(0, 1, 2001)
        out[f"{name}_quantiles"] = np.quantile(big, q, axis=0)
        print(name, "prior draws", time.time() - t0, "s")
    np.savez_compressed(os.path.join(OUT, "priors.npz"), **out)


# ---------------------------------------------------------------------------
# distribution tier
# ---------------------------------------------------------------------------
_NS = {}


def _ks_chunk(job):
    model, si, ci, chunk_id, n = job
    dt, ms = DT_CONFIGS[ci]
    K = int(ms)
    if model not in _NS:
        _NS[model] = load_slice(f"{model}_sim")
    ns = _NS[model]
    np.random.seed(100_000 * (1 + ci) + 1000 * si + chunk_id + (500_000_000 if model == "single" else 0))
    hist = np.zeros((3, K + 1), dtype=np.int64)   # rows: upper, lower, timeout
    zs = np.empty(n) if model == "single" else None
    if model == "basic":
        p = BASIC_SETS[si]
        for i in range(n):
            rt, ch = _basic_trial(ns, p, dt, ms)
            k = int(round((rt - p[3]) / dt))
            hist[0 if ch == 1 else (1 if ch == -1 else 2), k] += 1
    else:
        p = SINGLE_SETS[si]
        for i in range(n):
            crt, z = ns["diffusion_trial"](*p, dt=dt, max_steps=ms)
            zs[i] = z
            if crt == 0:
                hist[2, K] += 1
            else:
                k = int(round((abs(crt) - p[3]) / dt))
                hist[0 if crt > 0 else 1, k] += 1
    return model, si, ci, hist, zs


def make_ks(procs):
    jobs = []
    for model, sets in (("basic", BASIC_SETS), ("single", SINGLE_SETS)):
        for si in range(len(sets)):
            for ci in range(len(DT_CONFIGS)):
                for c in range(N_KS // CHUNK):
                    jobs.append((model, si, ci, c, CHUNK))
    # long (dt=.001) jobs first for better balance
    jobs.sort(key=lambda j: -j[2])
    acc = {}
    zacc = {}
    t0 = time.time()
    with Pool(procs) as pool:
        for done, (model, si, ci, hist, zs) in enumerate(pool.imap_unordered(_ks_chunk, jobs, chunksize=1)):
            key = (model, si, ci)
            acc[key] = acc.get(key, 0) + hist
            if zs is not None:
                zacc.setdefault(key, []).append(zs)
            if done % 100 == 0:
                print(f"ks: {done}/{len(jobs)} chunks, {time.time()-t0:.0f}s", flush=True)
    out = {"basic_sets": BASIC_SETS, "single_sets": SINGLE_SETS,
           "dt": np.array([c[0] for c in DT_CONFIGS]), "max_steps": np.array([c[1] for c in DT_CONFIGS])}
    q = np.linspace(0, 1, 2001)
    for (model, si, ci), h in acc.items():
        out[f"{model}_hist_s{si}_c{ci}"] = h.astype(np.int32)
        if (model, si, ci) in zacc:
            z = np.concatenate(zacc[(model, si, ci)])
            out[f"{model}_zq_s{si}_c{ci}"] = np.quantile(z, q)
            out[f"{model}_zmom_s{si}_c{ci}"] = np.array([z.mean(), z.var(), len(z)])
    np.savez_compressed(os.path.join(OUT, "ks_hist.npz"), **out)
    print("ks_hist.npz written", time.time() - t0, "s")


# ---------------------------------------------------------------------------
# distribution tier of the variants: _alt / _scale / _scale2 / explicit boundary
# ---------------------------------------------------------------------------
VARIANTS = {"alt": ("single_alt", "diffusion_trial_alt", ALT_SETS), "scale": ("single_scale", "diffusion_trial_scale", SCALE_SETS),
            "scale2": ("single_scale2", "diffusion_trial_scale2", SCALE2_SETS),
            "explicit": ("imputation_sim", "diffusion_trial", EXPLICIT_SETS)}


def _variant_chunk(job):
    name, si, ci, chunk_id, n = job
    slice_name, fn_name, sets = VARIANTS[name]
    dt, ms = DT_CONFIGS[ci]
    K = int(ms)
    if name not in _NS:
        _NS[name] = load_slice(slice_name)
    f = _NS[name][fn_name]
    p = sets[si]
    np.random.seed(300_000_000 + 10_000_000 * list(VARIANTS).index(name) + 100_000 * ci + 1000 * si + chunk_id)
    hist = np.zeros((3, K + 1), dtype=np.int64)
    ter = p[2] if name == "explicit" else p[3]
    zs = None
    if name == "explicit":
        bounds = explicit_bounds()
        for i in range(n):
            crt = f(p[0], bounds[i % EXPLICIT_NB], p[1], p[2], p[3], dt=dt, max_steps=ms)
            if crt == 0:
                hist[2, K] += 1
            else:
                hist[0 if crt > 0 else 1, int(round((abs(crt) - ter) / dt))] += 1
    else:
        zs = np.empty(n)
        for i in range(n):
            crt, z = f(*p, dt=dt, max_steps=ms)
            zs[i] = z
            if crt == 0:
                hist[2, K] += 1
            else:
                hist[0 if crt > 0 else 1, int(round((abs(crt) - ter) / dt))] += 1
    return name, si, ci, hist, zs


def make_variants(procs):
    jobs = []
    for name, (_, _, sets) in VARIANTS.items():
        for si in range(len(sets)):
            for ci in range(len(DT_CONFIGS)):
                for c in range(N_KS // CHUNK):
                    jobs.append((name, si, ci, c, CHUNK))
    jobs.sort(key=lambda j: -j[2])
    acc, zacc = {}, {}
    t0 = time.time()
    with Pool(procs) as pool:
        for done, (name, si, ci, hist, zs) in enumerate(pool.imap_unordered(_variant_chunk, jobs, chunksize=1)):
            key = (name, si, ci)
            acc[key] = acc.get(key, 0) + hist
            if zs is not None:
                zacc.setdefault(key, []).append(zs)
            if done % 100 == 0:
                print(f"variants: {done}/{len(jobs)} chunks, {time.time()-t0:.0f}s", flush=True)
    out = {"alt_sets": ALT_SETS, "scale_sets": SCALE_SETS, "scale2_sets": SCALE2_SETS, "explicit_sets": EXPLICIT_SETS,
           "explicit_bounds": explicit_bounds(),
           "dt": np.array([c[0] for c in DT_CONFIGS]), "max_steps": np.array([c[1] for c in DT_CONFIGS])}
    q = np.linspace(0, 1, 2001)
    for (name, si, ci), h in acc.items():
        out[f"{name}_hist_s{si}_c{ci}"] = h.astype(np.int32)
        if (name, si, ci) in zacc:
            z = np.concatenate(zacc[(name, si, ci)])
            out[f"{name}_zq_s{si}_c{ci}"] = np.quantile(z, q)
            out[f"{name}_zmom_s{si}_c{ci}"] = np.array([z.mean(), z.var(), len(z)])
    np.savez_compressed(os.path.join(OUT, "ks_variants.npz"), **out)
    print("ks_variants.npz written", time.time() - t0, "s")


# ---------------------------------------------------------------------------
# prior-mixture tier: many parameter sets drawn from the prior, pooled distribution of the signed step index
# ---------------------------------------------------------------------------
def mixture_params(n_sets=2000, seed=99):
    from scipy.stats import truncnorm
    rng = np.random.default_rng(seed)
    tn = lambda m, sd, lo, up: truncnorm.rvs((lo - m) / sd, (up - m) / sd, loc=m, scale=sd, size=n_sets, random_state=rng)
    return np.stack([rng.normal(0.0, 2.0, n_sets), tn(1.0, .5, 0.0, 10.0), rng.beta(2.0, 2.0, n_sets),
                     tn(.5, .25, 0.0, 1.5), tn(1.0, .5, 0.0, 10.0)], axis=1).astype(np.float32).astype(np.float64)


def _mixture_chunk(job):
    ci, lo, hi, n = job
    dt, ms = DT_CONFIGS[ci]
    K = int(ms)
    if "basic" not in _NS:
        _NS["basic"] = load_slice("basic_sim")
    ns = _NS["basic"]
    P = mixture_params()
    hist = np.zeros((3, K + 1), dtype=np.int64)
    np.random.seed(700_000 + 10_000 * ci + lo)
    for b in range(lo, hi):
        p = P[b]
        for _ in range(n):
            rt, ch = _basic_trial(ns, p, dt, ms)
            k = int(round((rt - p[3]) / dt))
            hist[0 if ch == 1 else (1 if ch == -1 else 2), k] += 1
    return ci, hist


def make_mixture(procs, n_per_set=200, step=25):
    P = mixture_params()
    jobs = [(ci, lo, min(lo + step, len(P)), n_per_set) for ci in (1, 0) for lo in range(0, len(P), step)]
    acc = {}
    t0 = time.time()
    with Pool(procs) as pool:
        for done, (ci, h) in enumerate(pool.imap_unordered(_mixture_chunk, jobs, chunksize=1)):
            acc[ci] = acc.get(ci, 0) + h
            if done % 20 == 0:
                print(f"mixture: {done}/{len(jobs)} chunks, {time.time()-t0:.0f}s", flush=True)
    out = {"params": P.astype(np.float32), "n_per_set": np.array(n_per_set), "dt": np.array([c[0] for c in DT_CONFIGS]),
           "max_steps": np.array([c[1] for c in DT_CONFIGS])}
    for ci, h in acc.items():
        out[f"hist_c{ci}"] = h.astype(np.int32)
    np.savez_compressed(os.path.join(OUT, "mixture.npz"), **out)
    print("mixture.npz written", time.time() - t0, "s")


def _ratcliff_chunk(job):
    si, chunk_id, n = job
    if "ratcliff" not in _NS:
        _NS["ratcliff"] = load_slice("ratcliff")
    f = _NS["ratcliff"]["simulratcliff"]
    Nu, Alpha, Beta, Tau, Eta, Varsigma = RATCLIFF_SETS[si]
    np.random.seed(900_000 + 1000 * si + chunk_id)
    return si, f(N=n, Alpha=Alpha, Tau=Tau, Nu=Nu, Beta=Beta, Eta=Eta, Varsigma=Varsigma)


def make_ratcliff(procs, n_total=200_000, chunk=5000):
    jobs = [(si, c, chunk) for si in range(len(RATCLIFF_SETS)) for c in range(n_total // chunk)]
    acc = {}
    t0 = time.time()
    with Pool(procs) as pool:
        for done, (si, y) in enumerate(pool.imap_unordered(_ratcliff_chunk, jobs, chunksize=1)):
            acc.setdefault(si, []).append(y)
            if done % 40 == 0:
                print(f"ratcliff: {done}/{len(jobs)} chunks, {time.time()-t0:.0f}s", flush=True)
    out = {"sets": RATCLIFF_SETS}
    q = np.linspace(0, 1, 4001)
    for si, ys in acc.items():
        y = np.concatenate(ys)
        out[f"yq_s{si}"] = np.quantile(y, q)            # quantiles of the signed RT
        out[f"pupper_s{si}"] = np.array([(y > 0).mean(), len(y)])
    # a small bit-level known answer in the reference's own call shape (alpha_not_scaled.py:96-97)
    f = load_slice("ratcliff")["simulratcliff"]
    np.random.seed(2021)
    out["kat_seed2021_p17_n100"] = f(N=100, Alpha=1.2, Tau=.4, Beta=.5, Nu=3.5, Eta=1.0, Varsigma=1.2)
    np.savez_compressed(os.path.join(OUT, "ratcliff.npz"), **out)
    print("ratcliff.npz written", time.time() - t0, "s")


def make_ezdiff():
    """Known answers of the reference's ezdiff() (simulations/Basic_DDM_simulations.py:131-158) on choice-RT data made by the
    reference simulator: inputs (rt, correct with NaN for missing trials) and the three estimates, per case."""
    import contextlib
    import io
    b = load_slice("basic_sim")
    ez = load_slice("ezdiff")["ezdiff"]
    out = {}
    cases = [(BASIC_SETS[0], 0.01, 400.0, 400, 2023), (BASIC_SETS[1], 0.01, 400.0, 400, 7), (BASIC_SETS[3], 0.001, 4000.0, 150, 11),
             (BASIC_SETS[6], 0.01, 400.0, 300, 5), (np.array([6.0, 1.0, 0.5, 0.3, 0.5]), 0.01, 400.0, 60, 3)]   # last: pc == 1
    for ci, (p, dt, ms, n, seed) in enumerate(cases):
        np.random.seed(seed)
        rows = np.array([_basic_trial(b, p, dt, ms) for _ in range(n)])
        rt, choice = rows[:, 0].copy(), rows[:, 1]
        correct = np.where(choice == 1, 1.0, np.where(choice == -1, 0.0, np.nan))
        rt[np.isnan(correct)] = np.nan
        with contextlib.redirect_stdout(io.StringIO()):
            est = np.array(ez(rt, correct), dtype=np.float64)
        out[f"rt_{ci}"] = rt; out[f"correct_{ci}"] = correct; out[f"est_{ci}"] = est
        print(f"ezdiff case {ci}: n={n} pc={np.nanmean(correct):.3f} -> {est}")
    out["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(OUT, "ezdiff.npz"), **out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--kat", action="store_true")
    ap.add_argument("--priors", action="store_true")
    ap.add_argument("--ks", action="store_true")
    ap.add_argument("--variants", action="store_true")
    ap.add_argument("--ratcliff", action="store_true")
    ap.add_argument("--mixture", action="store_true")
    ap.add_argument("--ezdiff", action="store_true")
    ap.add_argument("--procs", type=int, default=8)
    a = ap.parse_args()
    if not os.path.isdir(REF):
        sys.exit(f"reference not found at {REF}; fixtures can only be regenerated in the build container")
    everything = not (a.kat or a.priors or a.ks or a.variants or a.ratcliff or a.mixture or a.ezdiff)
    if a.kat or everything:
        make_kat()
    if a.ezdiff or everything:
        make_ezdiff()
    if a.priors or everything:
        make_priors()
    if a.ratcliff or everything:
        make_ratcliff(a.procs)
    if a.mixture or everything:
        make_mixture(a.procs)
    if a.ks or everything:
        make_ks(a.procs)
    if a.variants or everything:
        make_variants(a.procs)

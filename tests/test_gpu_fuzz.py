"""Randomised parity: random model / batch / trial count / dt / cap / seed / offset / tuning, GPU exact mode vs the
oracle, every bit (trials, summaries, external datum).  Deterministic (seeded) so a failure is reproducible."""
import numpy as np
import pytest

import prior_util

pytestmark = pytest.mark.gpu


def _case(rng):
    model = int(rng.integers(0, 5))
    B = int(rng.integers(1, 70))
    N = int(rng.choice([1, 2, 7, 31, 64, 100, 180, 300, 513]))
    dt = float(rng.choice([0.01, 0.001, 0.004, 0.0025]))
    max_steps = float(rng.choice([0, 1, 3, 10, 400, 401, 1000, 4000]))
    seed = int(rng.integers(0, 2**63))
    off = int(rng.choice([0, 1, 2**31 - 5, 2**32 - 3, 2**40 + 17]))
    tune = (int(rng.integers(0, 9)), int(rng.choice([0, 2, 3, 4, 6, 8, 13, 16])), int(rng.integers(0, 70)), int(rng.integers(0, 20)),
            int(rng.choice([0, 1, 3, 64, 5000])), int(rng.choice([0, 0, 0, 5, 33, 64, 200])))
    bridge = bool(model == 3 and rng.random() < 0.5 and max_steps < 2**22)
    return model, B, N, dt, max_steps, seed, off, tune, bridge


@pytest.mark.parametrize("chunk", range(9))
def test_fuzz_bit_parity(chunk):
    import oracle
    from bayesflow_nddms_amd import _lib, engine
    rng = np.random.default_rng(1000 + chunk)
    try:
        for _ in range(10):
            model, B, N, dt, max_steps, seed, off, tune, bridge = _case(rng)
            packed = chunk >= 6                    # chunks 6..8: the NDDM_GAUSS_PACKED bit layout (never with the bridge)
            bridge = bridge and not packed
            pseed = int(rng.integers(0, 10**6))
            if model == 0:
                p = prior_util.basic_prior(B, pseed)
            elif model in (1, 2):
                p = prior_util.single_prior(B, pseed, gamma=float(rng.choice([1.0, 2.0, 0.3])))
                if model == 2:
                    p[:, 4] = np.minimum(p[:, 4], 1.0)
            elif model == 3:
                p = prior_util.alpha_ns_prior(B, pseed)
            else:
                p = prior_util.basic_prior(B, pseed)[:, [0, 2, 3, 4]]
            bounds = np.abs(rng.normal(1.2, 0.5, size=(B, N))).astype(np.float32) if model == 4 else None
            _lib.check(_lib.lib().nddm_set_tuning(*tune))
            g = engine.simulate(model, p, N, dt=dt, max_steps=max_steps, seed=seed, set_offset=off, fast=False,
                                bounds=bounds, ext_sigma=0.2, ext_mode=0,
                                want_ext=(model == 3), bridge=bridge, packed=packed)
            o = oracle.philox_simulate(model, p, N, dt=dt, max_steps=max_steps, seed=seed, set_offset=off, bounds=bounds,
                                       ext_sigma=0.2, ext_mode=0, want_ext=(model == 3), bridge=bridge, packed=packed, threads=8)
            ctx = (model, B, N, dt, max_steps, seed, off, tune, bridge, packed)
            assert np.array_equal(g["trials"].cpu().numpy().view(np.uint32), o["trials"].view(np.uint32)), ctx
            assert np.array_equal(np.nan_to_num(g["summary"].cpu().numpy()).view(np.uint32),
                                  np.nan_to_num(o["summary"]).view(np.uint32)), ctx
            if model == 3:
                assert np.array_equal(g["ext"].cpu().numpy().view(np.uint32), o["ext"].view(np.uint32)), ctx
    finally:
        _lib.lib().nddm_set_tuning(0, 0, 0, 0, 0, 0)


def round6_case(rng):
    """One random case of each device path added in round 6, GPU exact mode against the oracle, every bit: NDDM_STATE_F64 (basic /
    single, every tuning knob, odd caps, offsets beyond 2^32) against the float64 restatement, and nddm_simulratcliff (random batch and
    trial counts, tiled sets, parameter rows from the generator's ranges with the corners mixed in) against section D.  Raises
    AssertionError with the case's parameters on a mismatch.  (tests/fuzz_long.py draws these by the thousand.)"""
    import oracle
    from bayesflow_nddms_amd import _lib, engine
    model, B, N, dt, max_steps, seed, off, tune, _ = _case(rng)
    model = model % 2                                  # basic or single
    pseed = int(rng.integers(0, 10**6))
    p = prior_util.basic_prior(B, pseed) if model == 0 else prior_util.single_prior(B, pseed, gamma=float(rng.choice([1.0, 2.0, 0.3])))
    _lib.check(_lib.lib().nddm_set_tuning(*tune))
    g = engine.simulate(model, p, N, dt=dt, max_steps=max_steps, seed=seed, set_offset=off, fast=False, state_f64=True)
    o = oracle.philox_simulate_f64(model, p, N, dt=dt, max_steps=max_steps, seed=seed, set_offset=off, threads=8, want_outputs=True)
    ctx = ("state_f64", model, B, N, dt, max_steps, seed, off, tune)
    assert np.array_equal(g["trials"].cpu().numpy().view(np.uint32), o["trials"].view(np.uint32)), ctx
    assert np.array_equal(np.nan_to_num(g["summary"].cpu().numpy()).view(np.uint32), np.nan_to_num(o["summary"]).view(np.uint32)), ctx
    _lib.check(_lib.lib().nddm_set_tuning(0, 0, 0, 0, 0, 0))
    Br, Nr = int(rng.integers(1, 90)), int(rng.choice([1, 2, 7, 64, 65, 300, 512, 513, 1500]))
    pr = prior_util.alpha_ns_prior(Br, pseed)
    for row in rng.integers(0, Br, size=min(3, Br)):      # corners: a start on a boundary, a clipped drift, no drift variability
        pr[row] = rng.choice(np.array([[8.0, 1.0, 0.0, 0.3, 0.0, 1.0], [-9.0, 1.2, 1.0, 0.2, 0.5, 0.9], [0.0, 0.8, 0.5, 0.15, 0.0, 1.4],
                                       [4.9, 1.4, 0.31, 0.6, 2.0, 0.8]], np.float32))
    em = int(rng.integers(0, 2))
    g = engine.simulratcliff(pr, Nr, seed=seed, set_offset=off, fast=False, ext_sigma=0.3, ext_mode=em, want_ext=True)
    o = oracle.philox_ratcliff(pr, Nr, seed=seed, set_offset=off, ext_sigma=0.3, ext_mode=em, want_ext=True, threads=8)
    for k in ("trials", "summary", "ext"):
        assert np.array_equal(np.nan_to_num(g[k].cpu().numpy()).view(np.uint32), np.nan_to_num(o[k]).view(np.uint32)), ("simulratcliff", k, Br, Nr, seed, off)


@pytest.mark.parametrize("chunk", range(3))
def test_fuzz_bit_parity_of_the_round_6_paths(chunk):
    """The randomised check for the two device paths added in round 6 (round6_case): 30 cases of each."""
    from bayesflow_nddms_amd import _lib
    rng = np.random.default_rng(7000 + chunk)
    try:
        for _ in range(10):
            round6_case(rng)
    finally:
        _lib.lib().nddm_set_tuning(0, 0, 0, 0, 0, 0)


def test_stress_parity_subset():
    """tests/stress_parity.py at two sizes per model: the ordering pre-pass (B >= 2048), a multi-chunk queue and a set
    split into tiles (N = 700), against the oracle, every bit."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import stress_parity
    assert stress_parity.run(sizes=((6000, 300), (2500, 700)), verbose=False, threads=8) == []

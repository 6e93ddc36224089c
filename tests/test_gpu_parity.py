"""GPU parity tests proper: the HIP kernels, called through the C ABI, against the CPU oracle on the same
seeded inputs.  Exact Gaussian mode is compared BIT FOR BIT (the oracle runs the identical float32 stream);
the fast (transcendental-unit) mode is compared by per-trial agreement and by distribution."""
import numpy as np
import pytest

import prior_util

pytestmark = pytest.mark.gpu

MODELS = {"basic": 0, "single": 1, "alt": 2, "alpha_ns": 3, "explicit": 4}


def _params(model, B, seed):
    if model == "basic":
        return prior_util.basic_prior(B, seed)
    if model == "single":
        return prior_util.single_prior(B, seed, gamma=1.0)
    if model == "alt":
        p = prior_util.single_prior(B, seed, gamma=1.0)
        p[:, 4] = np.minimum(p[:, 4], 1.0)   # std_dc
        return p
    if model == "alpha_ns":
        return prior_util.alpha_ns_prior(B, seed)
    p = prior_util.basic_prior(B, seed)
    return p[:, [0, 2, 3, 4]]                 # drift, beta, ter, dc


def _run_both(model, B, N, dt, max_steps, seed, set_offset=0, fast=False, bridge=False, packed=False, **kw):
    import oracle
    from bayesflow_nddms_amd import engine
    p = _params(model, B, 1234 + B)
    bounds = None
    if model == "explicit":
        bounds = np.abs(np.random.default_rng(5).normal(1.2, 0.4, size=(B, N))).astype(np.float32)
    want_ext = model == "alpha_ns"
    g = engine.simulate(MODELS[model], p, N, dt=dt, max_steps=max_steps, seed=seed, set_offset=set_offset, fast=fast,
                        bounds=bounds, ext_sigma=0.1, ext_mode=0, want_ext=want_ext, bridge=bridge, packed=packed, **kw)
    o = oracle.philox_simulate(MODELS[model], p, N, dt=dt, max_steps=max_steps, seed=seed, set_offset=set_offset,
                               bounds=bounds, ext_sigma=0.1, ext_mode=0, want_ext=want_ext, want_k=True, threads=8,
                               bridge=bridge, packed=packed)
    g = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in g.items()}
    return p, g, o


def test_normals_bit_exact(oracle_mod):
    """Philox4x32-10 + exact Box-Muller: device == oracle for every bit, incl. corner counters."""
    from bayesflow_nddms_amd import engine
    rng = np.random.default_rng(0)
    ctr = rng.integers(0, 2**32, size=(20000, 4), dtype=np.uint64).astype(np.uint32)
    ctr[:8] = [[0, 0, 0, 0], [1, 0, 0, 0], [0xffffffff] * 4, [0, 1, 2, 3], [0, 0, 0, 0x10000000],
               [5, 0xffffffff, 7, 0x1fffffff], [123, 456, 789, 0x20000000], [2**31, 2**31, 2**31, 2**27]]
    k0, k1 = 2023, 0xdeadbeef
    dev = engine.debug_normals(ctr, k0, k1, fast=False)
    ref = np.stack([oracle_mod.philox_normals4(*map(int, c), k0, k1) for c in ctr])
    assert np.array_equal(dev.view(np.uint32), ref.view(np.uint32))
    # fast transform: same stream, transcendental-unit arithmetic -> tiny absolute differences only
    fast = engine.debug_normals(ctr, k0, k1, fast=True)
    assert np.max(np.abs(fast - ref)) < 2e-5
    # and it is a standard normal sample
    assert abs(ref.mean()) < 0.02 and abs(ref.std() - 1.0) < 0.02


def test_fast_normals_are_standard_normal():
    """The product default (fast transform): 4e6 draws against N(0,1) -- KS, moments up to kurtosis, tail mass,
    no NaN/inf, and no correlation between the cos / sin members of a pair or successive blocks."""
    from scipy import stats
    from bayesflow_nddms_amd import engine
    n = 1_000_000
    ctr = np.zeros((n, 4), dtype=np.uint32)
    ctr[:, 0] = np.arange(n) % 1000          # block index
    ctr[:, 1] = np.arange(n) // 1000         # trial index
    ctr[:, 2] = 12345
    z = engine.debug_normals(ctr, 2023, 7, fast=True).astype(np.float64)
    assert np.all(np.isfinite(z))
    flat = z.ravel()
    assert stats.kstest(flat[::4], "norm").statistic < 0.002
    assert abs(flat.mean()) < 2e-3 and abs(flat.var() - 1) < 3e-3
    assert abs(stats.skew(flat)) < 5e-3 and abs(stats.kurtosis(flat)) < 1e-2
    for thr, p in ((3.0, 2.6998e-3), (4.0, 6.334e-5)):
        assert abs((np.abs(flat) > thr).mean() / p - 1) < 0.15
    assert abs(np.corrcoef(z[:, 0], z[:, 1])[0, 1]) < 4e-3 and abs(np.corrcoef(z[:, 1], z[:, 2])[0, 1]) < 4e-3
    assert abs(np.corrcoef(z[:-1, 3], z[1:, 0])[0, 1]) < 4e-3


@pytest.mark.parametrize("model", list(MODELS))
@pytest.mark.parametrize("dt,max_steps", [(0.01, 400.0), (0.001, 4000.0)])
def test_exact_mode_bit_parity(model, dt, max_steps):
    """Trials and fused summaries of every model equal the oracle's bit for bit."""
    p, g, o = _run_both(model, B=96, N=300, dt=dt, max_steps=max_steps, seed=2023)
    assert np.array_equal(g["trials"].view(np.uint32), o["trials"].view(np.uint32))
    gs, os_ = g["summary"], o["summary"]
    assert np.array_equal(np.isnan(gs), np.isnan(os_))
    assert np.array_equal(np.nan_to_num(gs).view(np.uint32), np.nan_to_num(os_).view(np.uint32))
    if model == "alpha_ns":
        assert np.array_equal(g["ext"].view(np.uint32), o["ext"].view(np.uint32))


@pytest.mark.parametrize("dt,max_steps", [(0.01, 400.0), (0.001, 4000.0), (0.004, 1001.0)])
def test_bridge_mode_bit_parity(dt, max_steps):
    """alpha_not_scaled with the Brownian-bridge boundary correction (exact exp, stream-3 uniforms, sub-step jitter):
    trials, summaries and the per-set external datum equal the oracle's bit for bit."""
    p, g, o = _run_both("alpha_ns", B=96, N=300, dt=dt, max_steps=max_steps, seed=77, bridge=True)
    assert np.array_equal(g["trials"].view(np.uint32), o["trials"].view(np.uint32))
    assert np.array_equal(np.nan_to_num(g["summary"]).view(np.uint32), np.nan_to_num(o["summary"]).view(np.uint32))
    assert np.array_equal(g["ext"].view(np.uint32), o["ext"].view(np.uint32))
    # jittered RTs are no longer on the dt grid, and they are earlier on average than plain Euler-Maruyama's
    p2, g2, _ = _run_both("alpha_ns", B=96, N=300, dt=dt, max_steps=max_steps, seed=77, bridge=False)
    rt, rt2 = np.abs(g["trials"][..., 0]), np.abs(g2["trials"][..., 0])
    assert rt[rt > 0].mean() < rt2[rt2 > 0].mean()
    from bayesflow_nddms_amd import engine
    with pytest.raises(ValueError):
        engine.simulate(0, prior_util.basic_prior(2, 1), 10, bridge=True)


@pytest.mark.parametrize("model", list(MODELS))
@pytest.mark.parametrize("dt,max_steps", [(0.01, 400.0), (0.001, 4000.0), (0.01, 403.0)])
def test_packed_layout_bit_parity(model, dt, max_steps):
    """NDDM_GAUSS_PACKED with the exact transform (8 normals per Philox block from 16 + 16 bit pairs): trials and fused
    summaries of every model equal the oracle's restatement of the same layout bit for bit, also when the step cap is
    not a multiple of 8; with the fast transform nearly every (step index, choice) pair is the same."""
    p, g, o = _run_both(model, B=96, N=300, dt=dt, max_steps=max_steps, seed=2024, packed=True)
    assert np.array_equal(g["trials"].view(np.uint32), o["trials"].view(np.uint32))
    assert np.array_equal(np.nan_to_num(g["summary"]).view(np.uint32), np.nan_to_num(o["summary"]).view(np.uint32))
    p, gf, o = _run_both(model, B=96, N=300, dt=dt, max_steps=max_steps, seed=2024, packed=True, fast=True)
    assert (gf["trials"][..., 0] == o["trials"][..., 0]).mean() > 0.995
    # a different stream than the default layout's
    p, g4, _ = _run_both(model, B=96, N=300, dt=dt, max_steps=max_steps, seed=2024, packed=False)
    assert (g4["trials"][..., 0] != g["trials"][..., 0]).mean() > 0.5


@pytest.mark.parametrize("model", ["basic", "single"])
@pytest.mark.parametrize("dt,max_steps,N", [(0.01, 400.0, 300), (0.001, 4000.0, 300), (0.01, 403.0, 77), (0.0005, 20000.0, 40),
                                            (0.001, 4000.0, 1200)])
def test_state_f64_bit_parity(model, dt, max_steps, N):
    """NDDM_STATE_F64: the reference's float64 recurrence (basic_ddm_dc.py:91-103; single_trial_alpha_not_scaled.py:113-128) on the
    device -- every trial's (step, choice), the float pairs and the fused summaries equal oracle_philox_simulate_f64's bit for bit,
    at both step sizes, with a cap that is not a multiple of 4, with a cap >= 2^14 (the 32-bit staging kernels) and with tiled sets."""
    import oracle
    from bayesflow_nddms_amd import engine
    B = 96
    p = _params(model, B, 1234 + B)
    g = engine.simulate(MODELS[model], p, N, dt=dt, max_steps=max_steps, seed=2025, set_offset=7, fast=False, state_f64=True)
    o = oracle.philox_simulate_f64(MODELS[model], p, N, dt=dt, max_steps=max_steps, seed=2025, set_offset=7, threads=8, want_outputs=True)
    gt, gs = g["trials"].cpu().numpy(), g["summary"].cpu().numpy()
    assert np.array_equal(gt.view(np.uint32), o["trials"].view(np.uint32))
    assert np.array_equal(np.isnan(gs), np.isnan(o["summary"]))
    assert np.array_equal(np.nan_to_num(gs).view(np.uint32), np.nan_to_num(o["summary"]).view(np.uint32))
    # the float32 state (the default) ends nearly every trial on the same (step, choice); the few that differ are the stated deviation
    f32 = engine.simulate(MODELS[model], p, N, dt=dt, max_steps=max_steps, seed=2025, set_offset=7, fast=False)["trials"].cpu().numpy()
    assert 0.995 < (f32[..., 0] == gt[..., 0]).mean() <= 1.0
    # fast transform + float64 state: same stream, nearly every trial the same
    ff = engine.simulate(MODELS[model], p, N, dt=dt, max_steps=max_steps, seed=2025, set_offset=7, fast=True, state_f64=True)["trials"].cpu().numpy()
    assert (ff[..., 0] == gt[..., 0]).mean() > 0.99


def test_state_f64_argument_checks():
    from bayesflow_nddms_amd import engine
    with pytest.raises(ValueError, match="NDDM_STATE_F64"):
        engine.simulate(3, prior_util.alpha_ns_prior(2, 1), 10, state_f64=True)
    with pytest.raises(ValueError, match="NDDM_STATE_F64"):
        engine.simulate(0, prior_util.basic_prior(2, 1), 10, state_f64=True, packed=True)
    with pytest.raises(ValueError, match="NDDM_STATE_F64"):
        engine.simulate(0, prior_util.basic_prior(2, 1), 10, state_f64=True, want_codes=True)


@pytest.mark.parametrize("N", [1, 63, 300, 513, 1200])
def test_simulratcliff_bit_parity(N):
    """nddm_simulratcliff -- the reference's own generator for config 3, pyhddmjagsutils.simulratcliff (:47-176), on the device: the exact
    first-passage sampler, no step size -- in exact mode equals oracle section D bit for bit: signed RTs and accuracies, the fused
    summaries (integer sums of the decision time in 2^-16 s), the per-set external datum; ragged and tiled trial counts; parameter rows
    from the generator's ranges (alpha_not_scaled.py:66-72) plus the corners (a start on a boundary, Nu beyond +-5, Eta = 0)."""
    import oracle
    from bayesflow_nddms_amd import engine
    p = prior_util.alpha_ns_prior(93, 77)
    p = np.concatenate([p, np.array([[8.0, 1.0, 0.0, 0.3, 0.0, 1.0], [-9.0, 1.0, 1.0, 0.3, 0.0, 1.0], [0.0, 0.8, 0.5, 0.15, 0.0, 1.4]], np.float32)])
    so = (1 << 36) + 11
    g = engine.simulratcliff(p, N, seed=2026, set_offset=so, fast=False, ext_sigma=0.1, ext_mode=0, want_ext=True)
    o = oracle.philox_ratcliff(p, N, seed=2026, set_offset=so, ext_sigma=0.1, ext_mode=0, want_ext=True, threads=8)
    for k in ("trials", "summary", "ext"):
        a = g[k].cpu().numpy()
        assert np.array_equal(np.nan_to_num(a).view(np.uint32), np.nan_to_num(o[k]).view(np.uint32)), k
    # results do not depend on how a set is tiled or on its neighbours: the first trials of the 1200-trial launch are these
    if N == 300:
        big = engine.simulratcliff(p, 1200, seed=2026, set_offset=so, fast=False, want_summary=False)["trials"].cpu().numpy()
        assert np.array_equal(big[:, :300], g["trials"].cpu().numpy())
        sub = engine.simulratcliff(p[10:20], 300, seed=2026, set_offset=so + 10, fast=False, want_summary=False)["trials"].cpu().numpy()
        assert np.array_equal(sub, g["trials"].cpu().numpy()[10:20])
        # the fast mode (v_log_f32 / v_exp_f32 / v_rcp_f32; the acceptance function from three terms of its series or of the series'
        # Jacobi-dual form instead of the reference's loop): the same stream, and the same function to float32 -- a trial differs only
        # when an attempt's (s2, s1) lands within rounding of the acceptance bound (0 of 6e6 trials in profiles/r6_ratcliff_agreement.txt)
        f = engine.simulratcliff(p, N, seed=2026, set_offset=so, fast=True, want_summary=False)["trials"].cpu().numpy()
        gt = g["trials"].cpu().numpy()
        same = np.sign(f[..., 0]) == np.sign(gt[..., 0])
        assert same.mean() > 0.9999 and np.abs(f[..., 0] - gt[..., 0])[same].max() < 1e-5
        # the per-trial drift is the draw the Euler-Maruyama form of the model uses: with Eta = 0 and a strong drift both agree on the response
    with pytest.raises(ValueError, match="GAUSS"):
        from bayesflow_nddms_amd import _lib
        _lib.check(_lib.lib().nddm_simulratcliff(g["params"].data_ptr(), 4, 10, 1, 0, 2, 0.0, 0, g["trials"].data_ptr(), None, None, None))


def test_packed_layout_argument_checks():
    from bayesflow_nddms_amd import engine
    with pytest.raises(ValueError):
        engine.simulate(3, prior_util.alpha_ns_prior(2, 1), 10, bridge=True, packed=True)
    with pytest.raises(ValueError, match="2\\^14"):
        engine.simulate(0, prior_util.basic_prior(2, 1), 10, dt=1e-4, max_steps=20000, packed=True)


@pytest.mark.parametrize("max_steps", [1.0, 2.0, 3.0, 5.0, 399.0, 401.0, 400.5])
def test_step_caps_not_multiple_of_four(max_steps):
    """The step cap is tested per Philox block when max_steps % 4 == 0 and per step otherwise: same semantics."""
    p, g, o = _run_both("basic", B=40, N=100, dt=0.01, max_steps=max_steps, seed=3)
    assert np.array_equal(g["trials"].view(np.uint32), o["trials"].view(np.uint32))
    k = np.rint((g["trials"][..., 0] - p[:, None, 3]) / 0.01)
    assert k.max() <= np.ceil(max_steps)


@pytest.mark.parametrize("N", [1, 3, 60, 63, 64, 65, 257, 300, 1000, 3000])
def test_ragged_trial_counts(N):
    """n_trials not a multiple of the wave width, tiny and large (ring sizes 64 ... 2)."""
    p, g, o = _run_both("basic", B=37, N=N, dt=0.01, max_steps=400.0, seed=7)
    assert np.array_equal(g["trials"].view(np.uint32), o["trials"].view(np.uint32))
    p, g, o = _run_both("single", B=5, N=N, dt=0.01, max_steps=400.0, seed=8)
    assert np.array_equal(g["trials"].view(np.uint32), o["trials"].view(np.uint32))
    assert np.array_equal(np.nan_to_num(g["summary"]).view(np.uint32), np.nan_to_num(o["summary"]).view(np.uint32))


def test_maximum_sizes():
    """Largest n_trials one launch takes (LDS ring of 2 slots), the error beyond it, and a batch far larger than
    the persistent grid with a short trial count."""
    from bayesflow_nddms_amd import engine
    p, g, o = _run_both("basic", B=6, N=7600, dt=0.01, max_steps=400.0, seed=21)
    assert np.array_equal(g["trials"].view(np.uint32), o["trials"].view(np.uint32))
    assert np.array_equal(np.nan_to_num(g["summary"]).view(np.uint32), np.nan_to_num(o["summary"]).view(np.uint32))
    p, g, o = _run_both("single", B=3, N=3800, dt=0.01, max_steps=400.0, seed=22)
    assert np.array_equal(g["trials"].view(np.uint32), o["trials"].view(np.uint32))
    # beyond the ring a set is split into tiles (and the summaries combined from integer partial sums): same bits
    for model, N in (("basic", 20000), ("single", 5001), ("alpha_ns", 1025), ("explicit", 2500)):
        p, g, o = _run_both(model, B=3, N=N, dt=0.01, max_steps=400.0, seed=23)
        assert np.array_equal(g["trials"].view(np.uint32), o["trials"].view(np.uint32)), model
        assert np.array_equal(np.nan_to_num(g["summary"]).view(np.uint32), np.nan_to_num(o["summary"]).view(np.uint32)), model
    one = engine.simulate(0, prior_util.basic_prior(1, 3), 1_000_000, dt=0.01, max_steps=400, seed=1, fast=True)
    sm, tr = one["summary"].cpu().numpy()[0], one["trials"].cpu().numpy()[0]
    resp = tr[:, 1] != 0
    assert sm[:3].sum() == 1_000_000 and sm[0] == (tr[:, 1] == 1).sum()
    assert abs(sm[3] - tr[resp, 0].astype(np.float64).mean()) < 1e-5
    # 2M sets x 8 trials: many more sets than resident waves; compare a few scattered rows with single-row launches
    B = 2_000_000
    pp = prior_util.basic_prior(B, 5)
    r = engine.simulate(0, pp, 8, dt=0.01, max_steps=400, seed=9, set_offset=0, fast=False)
    t = r["trials"].cpu().numpy()
    s = r["summary"].cpu().numpy()
    assert np.all(s[:, :3].sum(axis=1) == 8)
    for row in (0, 1, 777_777, B - 1):
        one = engine.simulate(0, pp[row:row + 1], 8, dt=0.01, max_steps=400, seed=9, set_offset=row, fast=False)
        assert np.array_equal(one["trials"].cpu().numpy()[0].view(np.uint32), t[row].view(np.uint32))


def test_geometry_independence():
    """Output is a pure function of (seed, set index, trial): chunking / ring / refill policy do not matter,
    and shards with set_offset reproduce the unsharded batch (the multi-GPU contract)."""
    from bayesflow_nddms_amd import _lib, engine
    p = prior_util.basic_prior(300, 99)
    base = engine.simulate(0, p, 180, dt=0.01, max_steps=400, seed=5, set_offset=1000, fast=False)["trials"].cpu().numpy()
    try:
        for tune in [(1, 2, 1, 1, 1, 0), (7, 4, 64, 3, 5, 0), (64, 64, 8, 16, 0, 0), (3, 8, 200, 2, 1000, 0), (1, 4, 8, 16, 64, 0),
                     (0, 0, 0, 0, 0, 64), (2, 4, 0, 0, 0, 7), (0, 0, 0, 0, 0, 179)]:
            _lib.check(_lib.lib().nddm_set_tuning(*tune))
            t = engine.simulate(0, p, 180, dt=0.01, max_steps=400, seed=5, set_offset=1000, fast=False)["trials"].cpu().numpy()
            assert np.array_equal(t.view(np.uint32), base.view(np.uint32)), tune
    finally:
        _lib.lib().nddm_set_tuning(0, 0, 0, 0, 0, 0)
    parts = [engine.simulate(0, p[a:b], 180, dt=0.01, max_steps=400, seed=5, set_offset=1000 + a, fast=False)["trials"].cpu().numpy()
             for a, b in [(0, 75), (75, 150), (150, 151), (151, 300)]]
    assert np.array_equal(np.concatenate(parts).view(np.uint32), base.view(np.uint32))


def test_ordering_with_split_sets_and_models():
    """Longest-first scheduling is active from 2048 sets on: combine it with split sets (n_trials > 512) and with every
    model; output must stay a pure function of (seed, set, trial)."""
    for model, B, N in (("basic", 2500, 700), ("single", 2100, 64), ("alpha_ns", 2048, 130), ("explicit", 2300, 50),
                        ("alt", 2200, 33)):
        p, g, o = _run_both(model, B=B, N=N, dt=0.01, max_steps=400.0, seed=31)
        assert np.array_equal(g["trials"].view(np.uint32), o["trials"].view(np.uint32)), model
        assert np.array_equal(np.nan_to_num(g["summary"]).view(np.uint32), np.nan_to_num(o["summary"]).view(np.uint32)), model
    from bayesflow_nddms_amd import _lib, engine
    p = prior_util.basic_prior(5000, 8)
    a = engine.simulate(0, p, 100, dt=0.01, max_steps=400, seed=2, set_offset=0, fast=True)["trials"].cpu().numpy()
    try:
        _lib.lib().nddm_set_ordering(0)
        b = engine.simulate(0, p, 100, dt=0.01, max_steps=400, seed=2, set_offset=0, fast=True)["trials"].cpu().numpy()
    finally:
        _lib.lib().nddm_set_ordering(1)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_rejection_cap_fallback():
    """Per-trial latent N(mu, std) > 0 with mu far below 0: the rejection loop hits its 64-draw cap and falls back to
    |last draw| -- identically on the device and in the oracle (and the kernel terminates)."""
    import oracle
    from bayesflow_nddms_amd import engine
    p = np.array([[1.0, -3.0, 0.5, 0.3, 0.5, 1.0, 0.2, 1.0]] * 3, dtype=np.float32)      # P(accept) ~ 1e-9 per draw
    for model in (1, 2):
        q = p.copy()
        if model == 2:
            q[:, 1], q[:, 5] = 1.0, -3.0          # alt: the latent is the diffusion coefficient (mu_dc at index 5)
        g = engine.simulate(model, q, 70, dt=0.01, max_steps=400, seed=4, set_offset=0, fast=False)
        o = oracle.philox_simulate(model, q, 70, dt=0.01, max_steps=400, seed=4, set_offset=0)
        assert np.array_equal(g["trials"].cpu().numpy().view(np.uint32), o["trials"].view(np.uint32))
        assert np.all(np.isfinite(o["trials"]))


def test_device_resident_garbage_cannot_hang():
    """Inputs that bypass host validation (device tensors): a negative / NaN explicit boundary marks that trial NaN
    (the reference raises ValueError there), non-finite parameters give garbage rows -- and every launch terminates."""
    import torch
    from bayesflow_nddms_amd import engine
    b = torch.full((2, 40), 1.0, device="cuda")
    b[0, 3], b[1, 7] = -0.5, float("nan")
    r = engine.simulate(4, torch.tensor([[1.0, .5, .3, 1.0]] * 2, device="cuda"), 40, bounds=b, seed=1, set_offset=0)
    t = r["trials"].cpu().numpy()
    assert np.isnan(t[0, 3, 0]) and np.isnan(t[1, 7, 0]) and np.isfinite(np.delete(t[0, :, 0], 3)).all()
    bad = torch.tensor([[float("nan"), 1.0, .5, .3, 1.0], [1.0, -1.0, .5, .3, 1.0], [1.0, 1.0, .5, .3, float("inf")],
                        [1.0, 1.0, .5, .3, 0.0], [1.0, 0.0, .5, .3, 1.0]], device="cuda")
    r = engine.simulate(0, bad, 64, dt=0.001, max_steps=4000, seed=1, set_offset=0)
    torch.cuda.synchronize()
    assert tuple(r["trials"].shape) == (5, 64, 2)
    for model, P in ((1, 8), (2, 8), (3, 6)):
        junk = torch.full((3, P), float("nan"), device="cuda")
        junk[1] = -1.0
        junk[2] = 1e30
        engine.simulate(model, junk, 64, dt=0.001, max_steps=4000, seed=1, set_offset=0, bridge=(model == 3))
        torch.cuda.synchronize()


def test_generic_dispatcher_matches_named_entries():
    """nddm_simulate(model, ...) == the per-model entry points (raw C ABI call with torch device pointers)."""
    import torch
    from bayesflow_nddms_amd import _lib, engine
    L = _lib.lib()
    p = torch.as_tensor(prior_util.basic_prior(40, 2)).cuda()
    ref = engine.simulate(0, p, 100, dt=0.01, max_steps=400, seed=3, set_offset=9, fast=True)
    out = torch.empty((40, 100, 2), dtype=torch.float32, device="cuda")
    summ = torch.empty((40, 10), dtype=torch.float32, device="cuda")
    rc = L.nddm_simulate(0, p.data_ptr(), None, 40, 100, 0.01, 400, 3, 9, 1, 0.0, 0, out.data_ptr(), summ.data_ptr(), None,
                         torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(out, ref["trials"]) and torch.equal(torch.nan_to_num(summ), torch.nan_to_num(ref["summary"]))
    assert L.nddm_simulate(9, p.data_ptr(), None, 40, 100, 0.01, 400, 3, 9, 1, 0.0, 0, out.data_ptr(), None, None, None) == _lib.NDDM_ERR_PARAM


def test_edge_cases():
    from bayesflow_nddms_amd import engine
    # timeouts are data: choice 0, rt = max_steps*dt + tau
    p = np.array([[0.0, 9.0, 0.5, 0.3, 0.2]], dtype=np.float32)
    t = engine.simulate(0, p, 64, dt=0.01, max_steps=400, seed=1, fast=False)["trials"].cpu().numpy()
    assert np.all(t[..., 1] == 0) and np.allclose(t[..., 0], 4.3)
    s = engine.simulate(1, np.array([[0.0, 9.0, 0.5, 0.3, 0.01, 0.2, 1.0, 1.0]]), 64, seed=1, fast=False)
    assert np.all(s["trials"].cpu().numpy()[..., 0] == 0)
    assert s["summary"].cpu().numpy()[0, 2] == 64 and np.isnan(s["summary"].cpu().numpy()[0, 3])
    # max_steps = 0: no step is taken, nobody responds
    t = engine.simulate(0, prior_util.basic_prior(4, 1), 10, max_steps=0, seed=1, fast=False)["trials"].cpu().numpy()
    assert np.all(t[..., 1] == 0)
    # empty batch
    e = engine.simulate(0, np.zeros((0, 5), np.float32), 10, seed=1)
    assert tuple(e["trials"].shape) == (0, 10, 2)
    # bad input -> ValueError (reference convention)
    with pytest.raises(ValueError):
        engine.simulate(0, np.array([[1.0, -1.0, .5, .3, 1.0]]), 10)
    with pytest.raises(ValueError):
        engine.simulate(0, np.array([[1.0, np.nan, .5, .3, 1.0]]), 10)
    with pytest.raises(ValueError):
        engine.simulate(0, prior_util.basic_prior(2, 1), 0)
    with pytest.raises(ValueError):
        engine.simulate(0, prior_util.basic_prior(2, 1), 10, dt=-1.0)
    with pytest.raises(ValueError):
        engine.simulate(4, np.array([[1.0, .5, .3, 1.0]]), 4, bounds=np.array([[1.0, -0.1, 1.0, 1.0]]))
    with pytest.raises(ValueError):
        engine.simulate(0, np.zeros((3, 4), np.float32), 10)
    with pytest.raises(ValueError):            # the stream is keyed by the low 60 bits of the global set index
        engine.simulate(0, prior_util.basic_prior(2, 1), 10, seed=1, set_offset=2**60 - 1)


@pytest.mark.parametrize("model", ["basic", "single", "alpha_ns"])
def test_fast_mode_agrees_with_exact(model):
    """Same stream, hardware transcendentals: a trial can only differ when a path grazes a boundary within
    float rounding, so nearly all (step index, choice) pairs are identical."""
    p, g, o = _run_both(model, B=200, N=300, dt=0.001, max_steps=4000.0, seed=11, fast=True)
    same = (g["trials"][..., 0] == o["trials"][..., 0])
    assert same.mean() > 0.995, same.mean()
    # summaries agree to float tolerance
    gs, os_ = g["summary"], o["summary"]
    assert np.allclose(np.nan_to_num(gs[:, :3]), np.nan_to_num(os_[:, :3]), atol=2)
    assert np.allclose(np.nan_to_num(gs[:, 3]), np.nan_to_num(os_[:, 3]), rtol=0.02, atol=1e-3)


@pytest.mark.parametrize("B", [32, 1500, 4096])
def test_hipgraph_capture_and_replay(B):
    """a launch is capturable in a hipGraph (kernels only: here up to 4096 sets x 180 trials incl. the
    ordering pre-pass and the combine kernel); every replay reproduces the eager result bit for bit."""
    import torch
    from bayesflow_nddms_amd import engine
    p = torch.as_tensor(prior_util.basic_prior(B, 1)).cuda()
    out = torch.empty((B, 180, 2), device="cuda")
    summ = torch.empty((B, 10), device="cuda")
    kw = dict(dt=.001, max_steps=4000, seed=3, set_offset=7, fast=True, out_trials=out, out_summary=summ)
    engine.simulate(engine.BASIC_DDM_DC, p, 180, **kw)
    torch.cuda.synchronize()
    ref_t, ref_s = out.clone(), summ.clone()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        engine.simulate(engine.BASIC_DDM_DC, p, 180, **kw)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            engine.simulate(engine.BASIC_DDM_DC, p, 180, **kw)
    for _ in range(5):
        out.zero_(); summ.fill_(-7.0)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref_t)
        assert torch.equal(torch.nan_to_num(summ), torch.nan_to_num(ref_s))


def test_simulratcliff_capture_and_replay():
    """nddm_simulratcliff under hipGraph capture: one kernel, no library memory (sets of <= 512 trials), every replay bit-equal to the
    eager launch; a tiled launch WITH summaries (stream-ordered scratch) is refused under capture with ValueError and the library
    stays usable."""
    import torch
    from bayesflow_nddms_amd import engine
    B, N = 700, 100
    p = torch.as_tensor(prior_util.alpha_ns_prior(B, 4)).cuda()
    out, summ = torch.empty((B, N, 2), device="cuda"), torch.empty((B, 10), device="cuda")
    kw = dict(seed=3, set_offset=7, fast=True, out_trials=out, out_summary=summ)
    engine.simulratcliff(p, N, **kw)
    torch.cuda.synchronize()
    ref_t, ref_s = out.clone(), summ.clone()
    g, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            engine.simulratcliff(p, N, **kw)
    for _ in range(4):
        out.zero_(); summ.fill_(-7.0)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref_t) and torch.equal(torch.nan_to_num(summ), torch.nan_to_num(ref_s))
    big = torch.empty((8, 1200, 2), device="cuda")
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with pytest.raises(ValueError, match="captured"):
            with torch.cuda.graph(g2, stream=side):
                engine.simulratcliff(p[:8], 1200, seed=3, set_offset=7, out_trials=big)
    r = engine.simulratcliff(p[:8], 1200, seed=3, set_offset=7)                 # (eager: fine)
    assert torch.isfinite(r["summary"]).all()


def test_large_launch_under_capture():
    """A captured launch gets memory of its own for ALL of its scratch (order, partial sums), so mid-size launches are
    capturable too and replay bit-identically; only a launch that would pin more than 64 MB is refused -- with ValueError,
    instead of producing a graph that misbehaves on replay -- and the library stays usable."""
    import torch
    from bayesflow_nddms_amd import engine
    B = 40000
    p = torch.as_tensor(prior_util.basic_prior(B, 2)).cuda()
    r = engine.simulate(engine.BASIC_DDM_DC, p, 100, dt=.01, max_steps=400, seed=1, set_offset=0, fast=True)
    torch.cuda.synchronize()
    ref_t, ref_s = r["trials"].clone(), r["summary"].clone()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            engine.simulate(engine.BASIC_DDM_DC, p, 100, dt=.01, max_steps=400, seed=1, set_offset=0, fast=True,
                            out_trials=r["trials"], out_summary=r["summary"])
    for _ in range(3):
        r["trials"].zero_(); r["summary"].fill_(-1.0)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(r["trials"], ref_t) and torch.equal(torch.nan_to_num(r["summary"]), torch.nan_to_num(ref_s))
    B2 = 1_200_000
    p2 = torch.as_tensor(prior_util.basic_prior(B2, 3)).cuda()
    out2 = torch.empty((B2, 60, 2), device="cuda")
    summ2 = torch.empty((B2, 10), device="cuda")
    g2 = torch.cuda.CUDAGraph()
    with pytest.raises(ValueError, match="cannot be captured"):
        with torch.cuda.stream(side):
            with torch.cuda.graph(g2, stream=side):
                engine.simulate(engine.BASIC_DDM_DC, p2, 60, dt=.01, max_steps=400, seed=1, set_offset=0, fast=True,
                                out_trials=out2, out_summary=summ2)
    torch.cuda.synchronize()
    r2 = engine.simulate(engine.BASIC_DDM_DC, p, 100, dt=.01, max_steps=400, seed=1, set_offset=0, fast=True)
    assert torch.equal(r2["trials"], ref_t)


@pytest.mark.parametrize("model", ["basic", "single"])
def test_wide_step_cap_uses_32bit_staging(model):
    """Step caps of 2^14 and more cannot stage results as 16-bit words in LDS: the 32-bit staging path (otherwise only
    taken with the bridge correction) must give the same bits as the oracle, timeouts at the large cap included."""
    p, g, o = _run_both(model, B=48, N=130, dt=0.0002, max_steps=20000.0, seed=77)
    assert np.array_equal(g["trials"].view(np.uint32), o["trials"].view(np.uint32))
    assert np.array_equal(np.nan_to_num(g["summary"]).view(np.uint32), np.nan_to_num(o["summary"]).view(np.uint32))
    assert o["k"].max() > 16383            # the case does exercise step indices beyond 14 bits


@pytest.mark.parametrize("model", [0, 1, 3])
def test_debug_trace_records_and_leaves_results_alone(model):
    """nddm_set_debug_trace (engine.debug_trace): every wave of the launch leaves one record, the pull stamps cover the
    queue, the counters are consistent with the work -- and the traced launch returns the same bits as an untraced one."""
    import torch
    from bayesflow_nddms_amd import engine
    B, N = 3000, 100
    p = {0: prior_util.basic_prior, 1: prior_util.single_prior, 3: prior_util.alpha_ns_prior}[model](B, 5)
    p_dev = torch.as_tensor(p).cuda()
    kw = dict(dt=0.01, max_steps=400, seed=11, set_offset=7, fast=True)
    ref = engine.simulate(model, p_dev, N, **kw)
    with engine.debug_trace(waves=16384, chunks=1 << 16) as tr:
        got = engine.simulate(model, p_dev, N, **kw)
    t = tr.read()
    assert torch.equal(torch.nan_to_num(got["trials"]), torch.nan_to_num(ref["trials"]))
    assert torch.equal(torch.nan_to_num(got["summary"]), torch.nan_to_num(ref["summary"]))
    rec = t["records"]
    assert t["waves"] >= 1 and rec.shape == (t["waves"], 8)
    assert (rec[:, 6] > rec[:, 4]).all() and (rec[:, 5] >= rec[:, 4]).all() and (rec[:, 5] <= rec[:, 6]).all()   # start < dry <= end
    s = got["summary"].cpu().numpy()
    steps_lower_bound = float(s[:, 2].sum()) * 400.0                   # the trials that ran to the cap alone
    assert t["blocks"] * 64 * 4 >= B * N + steps_lower_bound           # every trial takes at least one step
    assert t["refills"] >= t["waves"]
    assert len(t["pulls"]) >= 1 and t["pulls"].min() >= rec[:, 4].min() and t["pulls"].max() <= rec[:, 6].max()
    after = engine.simulate(model, p_dev, N, **kw)                     # the trace is off again
    assert torch.equal(torch.nan_to_num(after["trials"]), torch.nan_to_num(ref["trials"]))


def test_kernel_variants_and_geometry_do_not_change_results():
    """The two Philox-key variants of the kernel (round keys from LDS / in VGPRs), forced through the variant knob, and a
    forced grid return the same bits as the library's own choice; nddm_debug_last_launch reports what ran."""
    import torch
    from bayesflow_nddms_amd import _lib, engine
    B, N = 5000, 150
    p_dev = torch.as_tensor(prior_util.basic_prior(B, 21)).cuda()
    kw = dict(dt=0.001, max_steps=4000, seed=5, set_offset=99, fast=True)
    ref = engine.simulate(engine.BASIC_DDM_DC, p_dev, N, **kw)
    auto = engine.last_launch()
    assert auto["grid_waves"] >= 1 and 2 <= auto["ring"] <= 32 and auto["tile_trials"] * auto["tiles_per_set"] >= N
    assert auto["vgpr_keys"] == 1                                       # a launch this small runs the VGPR-keys variant
    seen = set()
    try:
        for variant, grid in ((1, 0), (2, 0), (1, 777), (2, 8192)):
            _lib.check(_lib.lib().nddm_set_tuning(0, 0, 0, variant, grid, 0))
            got = engine.simulate(engine.BASIC_DDM_DC, p_dev, N, **kw)
            ll = engine.last_launch()
            seen.add(ll["vgpr_keys"])
            assert ll["vgpr_keys"] == variant - 1
            if grid:
                assert ll["grid_waves"] == min(grid, (B * ll["tiles_per_set"] + ll["sets_per_chunk"] - 1) // ll["sets_per_chunk"])
            assert torch.equal(got["trials"], ref["trials"])
            assert torch.equal(torch.nan_to_num(got["summary"]), torch.nan_to_num(ref["summary"]))
    finally:
        _lib.lib().nddm_set_tuning(0, 0, 0, 0, 0, 0)
    assert seen == {0, 1}


@pytest.mark.parametrize("B", [32, 3000])
def test_indirect_set_offset_equals_direct_and_moves_a_replayed_graph(B):
    """nddm_simulate_indirect / nddm_draw_prior_indirect (include/nddm.h): the global index of row 0 is set_offset + a
    64-bit word read from DEVICE memory when the launch runs.  Same bits as passing the sum directly (all the models whose
    kernels read the set index in different places: path stream, latent FIFO, external datum), and a captured
    [prior -> simulate -> word += B] moves along the random stream on every replay with no new kernel arguments."""
    import torch
    from bayesflow_nddms_amd import engine
    off = torch.tensor([2**33 + 12345], dtype=torch.int64, device="cuda")
    direct = 1000 + 2**33 + 12345
    for model, prior in ((engine.BASIC_DDM_DC, prior_util.basic_prior), (engine.SINGLE_TRIAL, prior_util.single_prior),
                         (engine.ALPHA_NOT_SCALED, prior_util.alpha_ns_prior)):
        p = torch.as_tensor(prior(B, 4)).cuda()
        kw = dict(dt=.01, max_steps=400, seed=9, fast=True, want_ext=(model == engine.ALPHA_NOT_SCALED), ext_sigma=0.1)
        a = engine.simulate(model, p, 77, set_offset=1000, set_offset_dev=off, **kw)
        b = engine.simulate(model, p, 77, set_offset=direct, **kw)
        assert torch.equal(a["trials"], b["trials"]) and torch.equal(torch.nan_to_num(a["summary"]), torch.nan_to_num(b["summary"]))
        if "ext" in b:
            assert torch.equal(a["ext"], b["ext"])
    pa = engine.draw_prior_device(engine.BASIC_DDM_DC, B, seed=5, set_offset=1000, set_offset_dev=off)
    pb = engine.draw_prior_device(engine.BASIC_DDM_DC, B, seed=5, set_offset=direct)
    assert torch.equal(pa, pb)
    # a captured iteration that advances its own offset
    word = torch.zeros(1, dtype=torch.int64, device="cuda")
    pr = torch.empty((B, 5), device="cuda")
    out = torch.empty((B, 120, 2), device="cuda")
    summ = torch.empty((B, 10), device="cuda")

    def iteration():
        engine.draw_prior_device(engine.BASIC_DDM_DC, B, seed=5, set_offset=0, set_offset_dev=word, out=pr)
        engine.simulate(engine.BASIC_DDM_DC, pr, 120, dt=.01, max_steps=400, seed=6, set_offset=0, set_offset_dev=word,
                        fast=True, out_trials=out, out_summary=summ)
        word.add_(B)

    with engine.graph_memory():
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            iteration()                                   # warm-up (moves the word)
            torch.cuda.synchronize()
            word.zero_()
            with torch.cuda.graph(g, stream=side):
                iteration()
        for i in range(4):
            g.replay()
            torch.cuda.synchronize()
            assert int(word.item()) == (i + 1) * B
            p_ref = engine.draw_prior_device(engine.BASIC_DDM_DC, B, seed=5, set_offset=i * B)
            r = engine.simulate(engine.BASIC_DDM_DC, p_ref, 120, dt=.01, max_steps=400, seed=6, set_offset=i * B, fast=True)
            assert torch.equal(pr, p_ref) and torch.equal(out, r["trials"])
            assert torch.equal(torch.nan_to_num(summ), torch.nan_to_num(r["summary"]))
        del g


def test_ring_span_limit_of_the_tuning_override(oracle_mod):
    """The hand-out packs the current slot's byte offset and the signed step to the next slot into 16 + 16 bits, so the
    ring's slots must span < 32 KB: a tuning override beyond that is refused with ValueError instead of
    wrapping silently, and the largest geometry below the limit (28 slots x 512 trials = 31,360 B) is bit-equal to the
    oracle."""
    from bayesflow_nddms_amd import _lib, engine
    p = prior_util.basic_prior(40, 8)
    L = _lib.lib()
    try:
        _lib.check(L.nddm_set_tuning(0, 48, 0, 0, 0, 512))
        with pytest.raises(ValueError, match="span < 32 KB"):
            engine.simulate(engine.BASIC_DDM_DC, p, 1024, dt=.01, max_steps=400, seed=3, set_offset=0, fast=False)
        _lib.check(L.nddm_set_tuning(0, 64, 0, 0, 0, 300))          # 64 x 696 B: the example of the advisor's report
        with pytest.raises(ValueError, match="span < 32 KB"):
            engine.simulate(engine.BASIC_DDM_DC, p, 300, dt=.01, max_steps=400, seed=3, set_offset=0, fast=False)
        _lib.check(L.nddm_set_tuning(0, 28, 0, 0, 0, 512))
        g = engine.simulate(engine.BASIC_DDM_DC, p, 1024, dt=.01, max_steps=400, seed=3, set_offset=0, fast=False)
        assert engine.last_launch()["ring"] == 28 and engine.last_launch()["tile_trials"] == 512
    finally:
        L.nddm_set_tuning(0, 0, 0, 0, 0, 0)
    o = oracle_mod.philox_simulate(oracle_mod.M_BASIC, p, 1024, dt=.01, max_steps=400, seed=3, set_offset=0, threads=4)
    assert np.array_equal(g["trials"].cpu().numpy().view(np.uint32), o["trials"].view(np.uint32))
    assert np.array_equal(np.nan_to_num(g["summary"].cpu().numpy()).view(np.uint32), np.nan_to_num(o["summary"]).view(np.uint32))


@pytest.mark.parametrize("model", ["basic", "alpha_ns"])
def test_wire_format_codes_decode_to_the_same_floats(model, oracle_mod):
    """nddm_simulate_codes / nddm_decode_codes (include/nddm.h): the trials as 2-byte codes (step index | code << 14) for the
    exchange step.  The codes equal the oracle's (step index, choice) -- every trial; decoding them gives bit for bit the float
    pairs the simulator writes (also when it writes both in one launch, also for a sorted / multi-chunk launch and ragged N);
    models and shapes without the format are refused."""
    import torch
    from bayesflow_nddms_amd import engine
    mid, om, prior = {"basic": (engine.BASIC_DDM_DC, oracle_mod.M_BASIC, prior_util.basic_prior),
                      "alpha_ns": (engine.ALPHA_NOT_SCALED, oracle_mod.M_ALPHA_NS, prior_util.alpha_ns_prior)}[model]
    for B, N, dt, ms in ((70, 300, 0.01, 400.0), (3000, 77, 0.001, 4000.0), (9, 700, 0.004, 1001.0)):
        p = prior(B, 5)
        kw = dict(dt=dt, max_steps=ms, seed=17, set_offset=3)
        both = engine.simulate(mid, p, N, fast=False, want_codes=True, **kw)
        only = engine.simulate(mid, p, N, fast=False, want_trials=False, want_summary=False, want_codes=True, **kw)
        assert "trials" not in only and torch.equal(only["codes"], both["codes"])
        dec = engine.decode_codes(mid, only["codes"], both["params"], dt)
        assert torch.equal(dec.view(torch.int32), both["trials"].view(torch.int32))
        o = oracle_mod.philox_simulate(om, p, N, dt=dt, max_steps=ms, seed=17, set_offset=3, want_k=True, threads=4)
        c = only["codes"].cpu().numpy().view(np.uint16).astype(np.int64)
        choice = o["trials"][..., 1] if model == "basic" else 2.0 * o["trials"][..., 1] - 1.0
        timeout = (o["trials"][..., 0] == 0.0) if model == "alpha_ns" else (choice == 0)
        code = np.where(timeout, 0, np.where(choice > 0, 1, 2))
        assert np.array_equal(c & 0x3fff, o["k"]) and np.array_equal(c >> 14, code)
        fastr = engine.simulate(mid, p, N, fast=True, want_codes=True, **kw)
        assert torch.equal(engine.decode_codes(mid, fastr["codes"], fastr["params"], dt).view(torch.int32), fastr["trials"].view(torch.int32))
    p = prior(8, 1)
    with pytest.raises(ValueError, match="wire format"):
        engine.simulate(mid, p, 50, dt=.001, max_steps=20000, want_codes=True)
    with pytest.raises(ValueError, match="wire format"):
        engine.simulate(engine.SINGLE_TRIAL, prior_util.single_prior(8, 1), 50, want_codes=True)
    if model == "alpha_ns":
        with pytest.raises(ValueError, match="wire format"):
            engine.simulate(mid, p, 50, dt=.001, max_steps=4000, bridge=True, want_codes=True)

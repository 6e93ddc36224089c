"""GPU distribution tests: the HIP simulators against golden fixtures made by RUNNING the reference
(tests/golden/*.npz), and size-independent properties at the full BASELINE size (1M sets x 300 trials).

Tolerance (north_star): KS distance of the simulated RT/choice distribution vs the NumPy reference < 0.01,
with >= 4e5 trials per side (two-sample noise floor ~0.003)."""
import os

import numpy as np
import pytest

import prior_util
from conftest import GOLDEN

pytestmark = pytest.mark.gpu

KS_BAR = 0.01


def _gold(name):
    path = os.path.join(GOLDEN, name)
    if not os.path.exists(path):
        pytest.skip(f"{name} missing")
    return np.load(path)


@pytest.mark.parametrize("ci", [0, 1])
def test_ks_packed_layout_vs_reference(ci):
    """NDDM_GAUSS_PACKED (fast transform): KS of the signed step index vs the reference's NumPy simulator < 0.01 for every
    fixed basic / single-trial parameter set and for the 2000-set prior mixture -- the same bar, fixtures and sample sizes
    as the default layout."""
    from bayesflow_nddms_amd import diagnostics as dg, engine
    gold = _gold("ks_hist.npz")
    dt, ms = float(gold["dt"][ci]), float(gold["max_steps"][ci])
    K = int(ms)
    worst = 0.0
    for si, p in enumerate(gold["basic_sets"]):
        r = engine.simulate(engine.BASIC_DDM_DC, np.tile(p, (2048, 1)), 200, dt=dt, max_steps=ms, seed=48,
                            set_offset=si * 10000, fast=True, packed=True, want_summary=False)
        h = dg.step_hist_from_trials(r["trials"].cpu().numpy(), float(np.float32(p[3])), dt, K)
        g = gold[f"basic_hist_s{si}_c{ci}"]
        worst = max(worst, dg.ks_signed(h, g))
        assert dg.ks_signed(h, g) < KS_BAR, (si, dg.ks_signed(h, g))
        assert np.max(np.abs(dg.choice_probs(h) - dg.choice_probs(g))) < KS_BAR
        for row in (0, 1):
            if g[row].sum() > 20000:
                assert dg.ks_conditional(h, g, row) < 0.015, (si, row)
    for si, p in enumerate(gold["single_sets"]):
        r = engine.simulate(engine.SINGLE_TRIAL, np.tile(np.append(p, 1.0), (2048, 1)), 200, dt=dt, max_steps=ms,
                            seed=49, set_offset=si * 10000, fast=True, packed=True, want_summary=False)
        t = r["trials"].cpu().numpy()
        h = dg.step_hist_from_trials(t, float(np.float32(p[3])), dt, K, signed=True)
        assert dg.ks_signed(h, gold[f"single_hist_s{si}_c{ci}"]) < KS_BAR, si
        assert dg.ks_quantile_table(t[..., 1].ravel(), gold[f"single_zq_s{si}_c{ci}"]) < KS_BAR, si
    mix = _gold("mixture.npz")
    pm = mix["params"]
    r = engine.simulate(engine.BASIC_DDM_DC, pm, int(mix["n_per_set"]), dt=dt, max_steps=ms, seed=50, set_offset=0,
                        fast=True, packed=True, want_summary=False)
    h = dg.step_hist_from_trials(r["trials"].cpu().numpy(), pm[:, 3], dt, int(ms))
    assert dg.ks_signed(h, mix[f"hist_c{ci}"]) < KS_BAR
    print(f"packed layout dt={dt}: max KS over the fixed sets {worst:.4f}")


def test_packed_layout_large_sample_agrees_with_default_layout():
    """3e8 trials per layout on the same 1M prior-drawn parameter sets (BASELINE configs[1]'s shape): the pooled choice
    probabilities and the pooled mean / second moment of the step index of the opt-in NDDM_GAUSS_PACKED layout agree with
    the default layout's within sampling error (a resolution of ~1e-4, two orders below the KS bar), and so do the per-set
    mean RTs on average."""
    import torch
    from bayesflow_nddms_amd import engine
    B, N, dt, ms = 1_000_000, 300, 0.001, 4000
    p = torch.as_tensor(prior_util.basic_prior(B, 2023)).cuda()
    stats = []
    for packed in (False, True):
        r = engine.simulate(engine.BASIC_DDM_DC, p, N, dt=dt, max_steps=ms, seed=77, set_offset=0, fast=True, packed=packed,
                            want_trials=False)
        s = r["summary"].double()
        n_resp = s[:, 0] + s[:, 1]
        k_mean = torch.where(n_resp > 0, (s[:, 3] - p[:, 3].double()) / dt, torch.zeros_like(n_resp))
        k_var = torch.where(n_resp > 0, s[:, 4] / dt ** 2, torch.zeros_like(n_resp))
        tot = float(n_resp.sum())
        stats.append({"p_up": float(s[:, 0].sum()) / (B * N), "p_miss": float(s[:, 2].sum()) / (B * N),
                      "mean_k": float((k_mean * n_resp).sum()) / tot,
                      "m2_k": float(((k_var + k_mean ** 2) * n_resp).sum()) / tot, "set_mean_k": k_mean})
    a, b = stats
    sd_k = (a["m2_k"] - a["mean_k"] ** 2) ** 0.5
    n = B * N
    assert abs(a["p_up"] - b["p_up"]) < 6 * (0.25 * 2 / n) ** 0.5                       # 6 sigma of a difference of proportions
    assert abs(a["p_miss"] - b["p_miss"]) < 6 * (a["p_miss"] * 2 / n) ** 0.5
    assert abs(a["mean_k"] - b["mean_k"]) < 6 * sd_k * (2 / n) ** 0.5, (a["mean_k"], b["mean_k"], sd_k)
    assert abs(a["m2_k"] / b["m2_k"] - 1) < 1e-3
    d = (a["set_mean_k"] - b["set_mean_k"])                                              # per set: noise only, no bias
    assert abs(float(d.mean())) < 0.2 and 200 < a["mean_k"] < 300


@pytest.mark.parametrize("fast", [True, False])
@pytest.mark.parametrize("ci", [0, 1])
def test_ks_basic_vs_reference(ci, fast):
    from bayesflow_nddms_amd import diagnostics as dg, engine
    gold = _gold("ks_hist.npz")
    dt, ms = float(gold["dt"][ci]), float(gold["max_steps"][ci])
    K = int(ms)
    worst = 0.0
    for si, p in enumerate(gold["basic_sets"]):
        r = engine.simulate(engine.BASIC_DDM_DC, np.tile(p, (2048, 1)), 200, dt=dt, max_steps=ms, seed=41,
                            set_offset=si * 10000, fast=fast, want_summary=False)
        h = dg.step_hist_from_trials(r["trials"].cpu().numpy(), float(np.float32(p[3])), dt, K)
        g = gold[f"basic_hist_s{si}_c{ci}"]
        ks = dg.ks_signed(h, g)
        worst = max(worst, ks)
        assert ks < KS_BAR, (si, ks)
        assert np.max(np.abs(dg.choice_probs(h) - dg.choice_probs(g))) < KS_BAR      # |dP(upper)|, |dP(timeout)|
        for row in (0, 1):
            if g[row].sum() > 20000:
                assert dg.ks_conditional(h, g, row) < 0.015, (si, row)
    print(f"basic dt={dt} fast={fast}: max KS {worst:.4f}")


@pytest.mark.parametrize("fast", [True, False])
@pytest.mark.parametrize("ci", [0, 1])
def test_ks_prior_mixture_vs_reference(ci, fast):
    """2000 parameter sets drawn from the prior x 200 trials each (4e5 per side): the pooled distribution of the
    signed step index vs the same sets run through the reference's NumPy simulator (tests/golden/mixture.npz)."""
    from bayesflow_nddms_amd import diagnostics as dg, engine
    gold = _gold("mixture.npz")
    dt, ms = float(gold["dt"][ci]), float(gold["max_steps"][ci])
    p = gold["params"]
    r = engine.simulate(engine.BASIC_DDM_DC, p, int(gold["n_per_set"]), dt=dt, max_steps=ms, seed=45, set_offset=0,
                        fast=fast, want_summary=False)
    h = dg.step_hist_from_trials(r["trials"].cpu().numpy(), p[:, 3], dt, int(ms))
    g = gold[f"hist_c{ci}"]
    assert h.sum() == g.sum()
    assert dg.ks_signed(h, g) < KS_BAR, dg.ks_signed(h, g)
    assert np.max(np.abs(dg.choice_probs(h) - dg.choice_probs(g))) < KS_BAR
    assert dg.ks_conditional(h, g, 0) < KS_BAR and dg.ks_conditional(h, g, 1) < KS_BAR


@pytest.mark.parametrize("fast", [True, False])
@pytest.mark.parametrize("ci", [0, 1])
def test_ks_single_trial_vs_reference(ci, fast):
    from bayesflow_nddms_amd import diagnostics as dg, engine
    gold = _gold("ks_hist.npz")
    dt, ms = float(gold["dt"][ci]), float(gold["max_steps"][ci])
    K = int(ms)
    for si, p in enumerate(gold["single_sets"]):
        r = engine.simulate(engine.SINGLE_TRIAL, np.tile(np.append(p, 1.0), (2048, 1)), 200, dt=dt, max_steps=ms,
                            seed=42, set_offset=si * 10000, fast=fast, want_summary=True)
        t = r["trials"].cpu().numpy()
        h = dg.step_hist_from_trials(t, float(np.float32(p[3])), dt, K, signed=True)
        g = gold[f"single_hist_s{si}_c{ci}"]
        assert dg.ks_signed(h, g) < KS_BAR, (si, dg.ks_signed(h, g))
        assert dg.ks_quantile_table(t[..., 1].ravel(), gold[f"single_zq_s{si}_c{ci}"]) < KS_BAR, si
        zm = gold[f"single_zmom_s{si}_c{ci}"]
        s = r["summary"].cpu().numpy()
        assert abs(s[:, 7].mean() - zm[0]) < 0.02 * max(1.0, np.sqrt(zm[1]))     # fused mean z vs reference
        assert abs(s[:, 8].mean() / zm[1] - 1) < 0.03                              # fused var z


@pytest.mark.parametrize("fast", [True, False])
@pytest.mark.parametrize("ci", [0, 1])
def test_ks_variants_vs_reference(ci, fast):
    """The misspecification / imputation simulators on the GPU (_alt, _scale, _scale2, explicit per-trial boundary)
    against >= 4e5 trials of the reference's own functions per parameter set (tests/golden/ks_variants.npz): KS of the
    signed step index, |dP(choice)|, and KS of the external datum, all < 0.01."""
    from bayesflow_nddms_amd import diagnostics as dg, engine
    from test_oracle_golden import variant_cases
    gold = _gold("ks_variants.npz")
    dt, ms = float(gold["dt"][ci]), float(gold["max_steps"][ci])
    mods = {"alt": engine.SINGLE_TRIAL_ALT, "single": engine.SINGLE_TRIAL, "explicit": engine.EXPLICIT_BOUNDARY}
    for idx, (name, model, p, bounds, ter_i) in enumerate(variant_cases(gold, mods)):
        if bounds is None:
            B, N, bnd = 2048, 200, None
        else:
            B, N = 1024, len(bounds)
            bnd = np.tile(bounds, (B, 1))
        r = engine.simulate(model, np.tile(p, (B, 1)), N, dt=dt, max_steps=ms, seed=47, set_offset=idx * 10000, fast=fast,
                            bounds=bnd, want_summary=True)
        t = r["trials"].cpu().numpy()
        h = dg.step_hist_from_trials(t, float(np.float32(p[ter_i])), dt, int(ms), signed=True)
        g = gold[name.format("hist") + f"_c{ci}"]
        assert dg.ks_signed(h, g) < KS_BAR, (name, dg.ks_signed(h, g))
        assert np.max(np.abs(dg.choice_probs(h) - dg.choice_probs(g))) < KS_BAR, name
        zq = name.format("zq") + f"_c{ci}"
        if zq in gold:
            assert dg.ks_quantile_table(t[..., 1].ravel(), gold[zq]) < KS_BAR, name
            zm = gold[name.format("zmom") + f"_c{ci}"]
            s = r["summary"].cpu().numpy()
            assert abs(s[:, 7].mean() - zm[0]) < 0.02 * max(1.0, np.sqrt(zm[1])), name      # fused mean z vs reference
        else:
            assert np.array_equal(t[..., 1], bnd.astype(np.float32))


def test_alpha_not_scaled_em_vs_exact_sampler():
    """Config 3: the Euler-Maruyama process vs the reference's generator simulratcliff, an EXACT first-passage
    sampler.  Discrete monitoring of the boundaries delays detection by ~0.58*sigma*sqrt(dt) per boundary, so the
    plain E-M output is biased towards longer RTs by O(sqrt(dt)); stated tolerances for PLAIN Euler-Maruyama: KS < 0.10
    at dt=.001, < 0.05 at dt=.00025, and the distance must shrink as dt -> 0 (measured: ~halves per 4x smaller dt)."""
    from bayesflow_nddms_amd import diagnostics as dg, engine
    gold = _gold("ratcliff.npz")
    for si, p in enumerate(gold["sets"]):
        ks = []
        for dt in (0.004, 0.001, 0.00025):
            r = engine.simulate(engine.ALPHA_NOT_SCALED, np.tile(p, (1024, 1)), 200, dt=dt, max_steps=8.0 / dt, seed=43,
                                set_offset=si * 10000, fast=True, want_summary=False)
            y = r["trials"][..., 0].cpu().numpy().ravel()
            ks.append(dg.ks_quantile_table(y, gold[f"yq_s{si}"]))
            if dt == 0.001:
                assert abs((y > 0).mean() - gold[f"pupper_s{si}"][0]) < 0.02
        assert ks[2] < 0.05 and ks[1] < 0.10, (si, ks)
        assert ks[2] < ks[1] < ks[0] + 0.005, (si, ks)


def test_alpha_not_scaled_bridge_vs_exact_sampler():
    """Config 3 with the Brownian-bridge boundary correction (NDDM_BRIDGE): crossings between grid points are
    sampled with their exact conditional probability, so the first-passage distribution matches the reference's
    exact sampler simulratcliff (2e5 trials per set in tests/golden/ratcliff.npz; two-sample noise floor ~0.004) at
    the BASELINE step dt=.001 -- tolerance KS < 0.01 like the other configs -- and already at dt=.004 to < 0.02."""
    from bayesflow_nddms_amd import diagnostics as dg, engine
    gold = _gold("ratcliff.npz")
    worst = {}
    for dt, bar in ((0.001, 0.01), (0.004, 0.02)):
        for si, p in enumerate(gold["sets"]):
            for fast in (True, False):
                r = engine.simulate(engine.ALPHA_NOT_SCALED, np.tile(p, (2048, 1)), 200, dt=dt, max_steps=8.0 / dt,
                                    seed=44, set_offset=si * 10000, fast=fast, bridge=True, want_summary=False)
                y = r["trials"][..., 0].cpu().numpy().ravel()
                ks = dg.ks_quantile_table(y, gold[f"yq_s{si}"])
                worst[dt] = max(worst.get(dt, 0.0), ks)
                assert ks < bar, (dt, si, fast, ks)
                assert abs((y > 0).mean() - gold[f"pupper_s{si}"][0]) < 0.01
    print("bridge KS vs simulratcliff:", worst)


def test_simulratcliff_device_vs_the_references_own_draws():
    """Config 3 with the reference's OWN generator on the device (nddm_simulratcliff: the exact first-passage sampler, no dt, no
    bridge): KS of the signed RT against 2e5 draws of pyhddmjagsutils.simulratcliff per set (tests/golden/ratcliff.npz) < 0.01 on
    every set, fast and exact transform, P(upper) within 0.01, and the fused mean RT equal to the mean of the written RTs."""
    from bayesflow_nddms_amd import diagnostics as dg, engine
    gold = _gold("ratcliff.npz")
    worst = 0.0
    for si, p in enumerate(gold["sets"]):
        for fast in (True, False):
            r = engine.simulratcliff(np.tile(p, (2048, 1)), 200, seed=45, set_offset=si * 10000, fast=fast)
            t = r["trials"].cpu().numpy()
            y = t[..., 0].ravel()
            ks = dg.ks_quantile_table(y, gold[f"yq_s{si}"])
            worst = max(worst, ks)
            assert ks < 0.01, (si, fast, ks)
            assert abs((y > 0).mean() - gold[f"pupper_s{si}"][0]) < 0.01
            s = r["summary"].cpu().numpy()
            assert np.all(s[:, 2] == 0) and np.abs(s[:, 3] - np.abs(t[..., 0]).mean(axis=1)).max() < 2e-5
    print("simulratcliff on the device, KS vs the reference's draws:", worst)


def test_generate_data_with_the_exact_sampler():
    """alpha_not_scaled.generate_data(method="exact"): the reference's data-generation block (alpha_not_scaled.py:52-128) with its own
    generator on the device -- same keys and participant draws as the Euler-Maruyama form, RT / accuracy per participant in line with
    it (the two are the same process), and pyhddmjagsutils.simulratcliff's call shape as a function."""
    from bayesflow_nddms_amd import alpha_not_scaled as ans
    ex = ans.generate_data(test_num=2, nparts=100, ntrials=400, method="exact", sim_seed=3)
    em = ans.generate_data(test_num=2, nparts=100, ntrials=400, method="em", sim_seed=3)
    assert set(ex) == set(em) and ex["N"] == 40000 and np.array_equal(ex["alpha"], em["alpha"])
    assert np.array_equal(ex["extdata"], em["extdata"])                        # the per-participant datum is the same draw in both
    acc_ex, acc_em = ex["acc"].reshape(100, 400).mean(1), em["acc"].reshape(100, 400).mean(1)
    rt_ex, rt_em = ex["rt"].reshape(100, 400).mean(1), em["rt"].reshape(100, 400).mean(1)
    assert np.abs(acc_ex - acc_em).max() < 0.12 and np.corrcoef(rt_ex, rt_em)[0, 1] > 0.98 and np.abs(rt_ex - rt_em).mean() < 0.03
    y = ans.simulratcliff(N=500, Alpha=1.2, Tau=.4, Nu=3.5, Beta=.5, Eta=1.0, Varsigma=1.2, seed=1, set_offset=0)
    assert y.shape == (500,) and y.dtype == np.float64 and np.all(np.abs(y) > 0.4) and 0.85 < (y > 0).mean() < 0.99
    with pytest.raises(ValueError):
        ans.simulratcliff(N=5, rangeTau=0.1)


def test_device_prior_marginals():
    """On-device draw_prior vs 1e5 draws of the reference's draw_prior: KS per marginal < 0.01."""
    from bayesflow_nddms_amd import diagnostics as dg
    from bayesflow_nddms_amd.priors import DevicePrior
    gold = _gold("priors.npz")
    for name in ("basic", "single"):
        d = DevicePrior(name, seed=7)(400_000).cpu().numpy()
        q = gold[f"{name}_quantiles"]
        assert d.shape[1] == q.shape[1]
        for j in range(d.shape[1]):
            assert dg.ks_quantile_table(d[:, j], q[:, j]) < 0.01, (name, j)
    s = DevicePrior("scale", seed=7)(100_000).cpu().numpy()
    assert s.shape[1] == 8 and 0 <= s[:, 7].min() and s[:, 7].max() <= 2 and abs(s[:, 7].mean() - 1) < 0.02
    # counter-based: shards reproduce the full draw
    a = DevicePrior("basic", seed=9)(1000).cpu().numpy()
    b = DevicePrior("basic", seed=9)(400, set_offset=600).cpu().numpy()
    assert np.array_equal(a[600:], b)


def test_full_size_properties():
    """BASELINE.json configs[1]: 1M sets x 300 trials, dt=.001.  Size-independent properties of the result:
    value domains, the timeout encoding, fused summaries == a torch recomputation from the trials, and the stream
    being a pure function of (seed, set, trial): rows re-simulated alone reproduce their bits."""
    import torch
    from bayesflow_nddms_amd import engine
    B, N, dt, ms = 1_000_000, 300, 0.001, 4000
    p = torch.as_tensor(prior_util.basic_prior(B, 2023)).cuda()
    r = engine.simulate(engine.BASIC_DDM_DC, p, N, dt=dt, max_steps=ms, seed=2023, set_offset=0, fast=True)
    t, s = r["trials"], r["summary"]
    rt, ch = t[..., 0], t[..., 1]
    tau = p[:, 3:4]
    assert bool(((ch == 1) | (ch == -1) | (ch == 0)).all())
    assert bool((rt >= tau + dt * 0.999).all())                      # the crossing step is counted: min RT = dt + tau
    miss = ch == 0
    assert bool(torch.allclose(rt[miss], (tau + ms * dt).expand_as(rt)[miss], rtol=1e-6))
    frac_missing = float(miss.float().mean())
    assert 0.001 < frac_missing < 0.01                               # SURVEY: ~0.4 % reach the cap under the prior
    # fused summaries vs recomputation
    n_up, n_lo, n_miss = (ch == 1).sum(1), (ch == -1).sum(1), miss.sum(1)
    assert bool((s[:, 0] == n_up).all() and (s[:, 1] == n_lo).all() and (s[:, 2] == n_miss).all())
    resp = (~miss).double()
    nr = resp.sum(1).clamp(min=1)
    mean_rt = (rt.double() * resp).sum(1) / nr
    ok = resp.sum(1) > 0
    assert bool(torch.allclose(s[ok, 3].double(), mean_rt[ok], rtol=2e-6, atol=2e-6))
    var_rt = ((rt.double() - mean_rt[:, None]) ** 2 * resp).sum(1) / nr
    assert bool(torch.allclose(s[ok, 4].double(), var_rt[ok], rtol=2e-3, atol=1e-7))
    assert bool(torch.allclose(s[:, 9].double(), (0.5 + 0.5 * torch.sign(ch.double())).mean(1), atol=1e-6))
    mean_steps = float(((rt - tau) / dt).mean())
    assert 200 < mean_steps < 300                                    # SURVEY: 246 steps/trial under the prior
    # geometry independence at full size: re-simulate scattered single rows and blocks
    for lo, hi in ((0, 1), (123_456, 123_460), (999_999, 1_000_000), (500_000, 500_257)):
        r2 = engine.simulate(engine.BASIC_DDM_DC, p[lo:hi], N, dt=dt, max_steps=ms, seed=2023, set_offset=lo, fast=True)
        assert torch.equal(r2["trials"], t[lo:hi])
        assert torch.equal(torch.nan_to_num(r2["summary"]), torch.nan_to_num(s[lo:hi]))
    # summary-only mode gives the same summaries without writing trials
    r3 = engine.simulate(engine.BASIC_DDM_DC, p[:100_000], N, dt=dt, max_steps=ms, seed=2023, set_offset=0, fast=True,
                         want_trials=False)
    assert "trials" not in r3 and torch.equal(torch.nan_to_num(r3["summary"]), torch.nan_to_num(s[:100_000]))


def test_full_size_single_trial_with_fused_summaries():
    """BASELINE.json configs[3]: single-trial model, 1M sets x 300 trials, dt=.001, trials + fused [B,10] summaries.
    Size-independent properties: value domains of (choicert, z1), the 0-coded timeout, summaries == a torch
    recomputation (counts exactly, moments to float tolerance, incl. the fixed-point z sums), and re-simulated rows
    reproduce their bits."""
    import torch
    from bayesflow_nddms_amd import engine
    B, N, dt, ms = 1_000_000, 300, 0.001, 4000
    p = torch.as_tensor(prior_util.single_prior(B, 2023)).cuda()          # [B, 8] (gamma = 1)
    r = engine.simulate(engine.SINGLE_TRIAL, p, N, dt=dt, max_steps=ms, seed=99, set_offset=0, fast=True)
    t, s = r["trials"], r["summary"]
    y, z = t[..., 0], t[..., 1]
    ter = p[:, 3:4]
    assert bool(torch.isfinite(t).all())
    resp = y != 0
    assert bool((y.abs()[resp] >= (ter + dt * 0.999).expand_as(y)[resp]).all())      # min |choicert| = ter + dt
    assert bool((y.abs() <= ter + ms * dt * 1.000001).all())
    n_up, n_lo, n_miss = (y > 0).sum(1), (y < 0).sum(1), (~resp).sum(1)
    assert bool((s[:, 0] == n_up).all() and (s[:, 1] == n_lo).all() and (s[:, 2] == n_miss).all())
    nr = resp.sum(1).clamp(min=1).double()
    ok = resp.sum(1) > 0
    mean_rt = (y.abs().double() * resp).sum(1) / nr
    assert bool(torch.allclose(s[ok, 3].double(), mean_rt[ok], rtol=2e-6, atol=2e-6))
    mz = z.double().mean(1)
    vz = ((z.double() - mz[:, None]) ** 2).mean(1)
    assert bool(torch.allclose(s[:, 7].double(), mz, rtol=1e-5, atol=1e-5))
    assert bool(torch.allclose(s[:, 8].double(), vz, rtol=2e-3, atol=1e-5))
    assert bool(torch.allclose(s[:, 9].double(), (0.5 + 0.5 * torch.sign(y.double())).mean(1), atol=1e-6))
    # z1 ~ N(gamma * a_trial, sigma1) with a_trial ~ N(mu_alpha, std_alpha) truncated to > 0 by rejection
    # (single_trial_alpha_not_scaled.py:113-116, :134): E[z1 | set] = mu + std * pdf(mu/std) / cdf(mu/std)
    mu, sd = p[:, 1].double(), p[:, 4].double()
    nrm = torch.distributions.Normal(0.0, 1.0)
    ez = mu + sd * torch.exp(nrm.log_prob(mu / sd)) / nrm.cdf(mu / sd)
    assert abs(float(mz.mean()) - float(ez.mean())) < 2e-3
    for lo, hi in ((0, 1), (777_777, 777_781), (999_999, 1_000_000)):
        r2 = engine.simulate(engine.SINGLE_TRIAL, p[lo:hi], N, dt=dt, max_steps=ms, seed=99, set_offset=lo, fast=True)
        assert torch.equal(r2["trials"], t[lo:hi])
        assert torch.equal(torch.nan_to_num(r2["summary"]), torch.nan_to_num(s[lo:hi]))
    r3 = engine.simulate(engine.SINGLE_TRIAL, p[:50_000], N, dt=dt, max_steps=ms, seed=99, set_offset=0, fast=True,
                         want_trials=False)
    assert torch.equal(torch.nan_to_num(r3["summary"]), torch.nan_to_num(s[:50_000]))


@pytest.mark.parametrize("bridge", [False, True])
def test_full_size_alpha_not_scaled(bridge):
    """BASELINE.json configs[2]: alpha_not_scaled as an E-M process, 1M sets x 300 trials, dt=.001 (plain and with the
    bridge correction): domains of (y = sign * rt, acc), external datum ~ N(Alpha, sigma), accuracy increases with
    drift, rows re-simulated alone reproduce their bits."""
    import torch
    from bayesflow_nddms_amd import engine
    B, N, dt, ms = 1_000_000, 300, 0.001, 4000
    p = torch.as_tensor(prior_util.alpha_ns_prior(B, 2021)).cuda()        # Nu, Alpha, Beta, Tau, Eta, Varsigma
    r = engine.simulate(engine.ALPHA_NOT_SCALED, p, N, dt=dt, max_steps=ms, seed=5, set_offset=0, fast=True,
                        bridge=bridge, ext_sigma=0.1, want_ext=True)
    t, s, ext = r["trials"], r["summary"], r["ext"]
    y, acc = t[..., 0], t[..., 1]
    assert bool(torch.isfinite(t).all())
    assert bool(((acc == 1) | (acc == 0) | (acc == 0.5)).all())           # 0.5 codes a timeout (sign 0)
    assert bool(((y > 0) == (acc == 1)).all() and ((y < 0) == (acc == 0)).all())
    tau = p[:, 3:4]
    resp = y != 0
    lo_bound = tau if bridge else tau + dt * 0.999                       # the bridge jitters the RT below the grid time
    assert bool((y.abs()[resp] >= lo_bound.expand_as(y)[resp] - 1e-6).all())
    assert bool((s[:, 0] == (y > 0).sum(1)).all() and (s[:, 1] == (y < 0).sum(1)).all())
    assert abs(float((ext - p[:, 1]).mean())) < 1e-3 and abs(float((ext - p[:, 1]).std()) - 0.1) < 1e-3
    up = (y > 0).float().mean(1)
    hi_drift, lo_drift = p[:, 0] > 2.0, p[:, 0] < -2.0
    assert float(up[hi_drift].mean()) > 0.85 and float(up[lo_drift].mean()) < 0.15
    for lo, hi in ((0, 2), (314_159, 314_163), (999_998, 1_000_000)):
        r2 = engine.simulate(engine.ALPHA_NOT_SCALED, p[lo:hi], N, dt=dt, max_steps=ms, seed=5, set_offset=lo, fast=True,
                             bridge=bridge, ext_sigma=0.1, want_ext=True)
        assert torch.equal(r2["trials"], t[lo:hi]) and torch.equal(r2["ext"], ext[lo:hi])


def test_numpy_results_go_through_pinned_memory_and_equal_the_device_tensors():
    """`as_numpy` results of a megabyte or more are staged through pinned host memory (engine.to_host: the link's rate instead of
    the pageable copy's): same bits as the device tensors, ordinary writeable arrays that own their memory (a later call does not
    overwrite an earlier result), strided sources included."""
    import gc
    import torch
    from bayesflow_nddms_amd import basic_ddm_dc, engine
    p = prior_util.basic_prior(2000, 11)
    dev = basic_ddm_dc.batch_simulate_trials(p, 300, seed=9, set_offset=0, as_numpy=False)
    host = basic_ddm_dc.batch_simulate_trials(p, 300, seed=9, set_offset=0)                   # 4.8 MB of trials: the pinned route
    assert dev["sim_data"].numel() * 4 >= engine.PINNED_FROM_BYTES
    assert isinstance(host["sim_data"], np.ndarray) and host["sim_data"].dtype == np.float32 and host["sim_data"].flags.writeable
    assert np.array_equal(host["sim_data"].view(np.uint32), dev["sim_data"].cpu().numpy().view(np.uint32))
    assert np.array_equal(np.nan_to_num(host["summary_stats"]), np.nan_to_num(dev["summary_stats"].cpu().numpy()))
    keep = host["sim_data"].copy()
    other = basic_ddm_dc.batch_simulate_trials(p, 300, seed=10, set_offset=0)["sim_data"]       # a second result: its own block
    gc.collect()
    assert np.array_equal(host["sim_data"], keep) and not np.array_equal(other, keep)
    col = engine.to_host(dev["sim_data"][..., 0])                                              # a strided view
    assert col.shape == (2000, 300) and np.array_equal(col, keep[..., 0])
    assert np.array_equal(engine.to_host(dev["sim_data"][:3]), keep[:3])                        # (small: the plain route)
    assert engine.to_host(torch.arange(4)).tolist() == [0, 1, 2, 3] and engine.to_host([1, 2]).tolist() == [1, 2]
    # the opt-out for callers that KEEP many large results (pinned blocks are rounded up to powers of two and cached): pageable memory,
    # same bits -- per call, or for the process (engine.PINNED_RESULTS / NDDM_PINNED_RESULTS=0); and the cache can be handed back
    stats = lambda: torch._C._cuda_hostMemoryStats() if hasattr(torch._C, "_cuda_hostMemoryStats") else None
    before = stats()
    pageable = engine.to_host(dev["sim_data"], pinned=False)
    assert np.array_equal(pageable, keep)
    if before is not None and "allocated_bytes.current" in before:
        assert stats()["allocated_bytes.current"] == before["allocated_bytes.current"]          # nothing new was pinned for it
    chunked = engine.simulate_to_host(engine.BASIC_DDM_DC, p, 300, seed=9, set_offset=0, chunk_bytes=1 << 20, pinned=False)
    assert np.array_equal(chunked["trials"], keep)
    try:
        engine.PINNED_RESULTS = False
        assert np.array_equal(basic_ddm_dc.batch_simulate_trials(p, 300, seed=9, set_offset=0)["sim_data"], keep)
    finally:
        engine.PINNED_RESULTS = True
    del host, other, col
    gc.collect()
    assert engine.release_pinned_cache() is True


def test_chunked_results_to_the_host_equal_one_launch():
    """engine.simulate_to_host: a large batch is simulated in chunks of sets whose results travel to pinned host memory beside the
    simulation of the next chunk.  Every output of every model equals the one launch's, every bit (the sets' random streams are
    keyed by their global index), with a ragged last chunk, and the package-level stream moves by the batch once."""
    import torch
    import bayesflow_nddms_amd as nd
    from bayesflow_nddms_amd import engine
    B, N = 1003, 120
    rng = np.random.default_rng(5)
    cases = [(engine.BASIC_DDM_DC, prior_util.basic_prior(B, 1), dict()),
             (engine.SINGLE_TRIAL, prior_util.single_prior(B, 2, gamma=1.0), dict()),
             (engine.ALPHA_NOT_SCALED, prior_util.alpha_ns_prior(B, 3), dict(bridge=True, ext_sigma=0.1, want_ext=True)),
             (engine.EXPLICIT_BOUNDARY, prior_util.basic_prior(B, 4)[:, [0, 2, 3, 4]],
              dict(bounds=np.abs(rng.normal(1.2, 0.4, size=(B, N))).astype(np.float32)))]
    for model, p, kw in cases:
        one = engine.simulate(model, p, N, dt=0.001, max_steps=4000, seed=77, set_offset=2**32 - 500, **kw)
        for chunk_bytes in (150 * N * 8, 1 << 40):                     # 7 chunks (the last of 103 sets) | one launch
            got = engine.simulate_to_host(model, p, N, dt=0.001, max_steps=4000, seed=77, set_offset=2**32 - 500, chunk_bytes=chunk_bytes, **kw)
            for k in ("trials", "summary") + (("ext",) if "want_ext" in kw else ()):
                assert isinstance(got[k], np.ndarray)
                assert np.array_equal(np.nan_to_num(got[k]).view(np.uint32), np.nan_to_num(one[k].cpu().numpy()).view(np.uint32)), (model, k, chunk_bytes)
    # summary-only: nothing to chunk; the global stream: one take of B sets per call, chunked or not
    s_only = engine.simulate_to_host(engine.BASIC_DDM_DC, cases[0][1], N, seed=1, set_offset=0, want_trials=False, chunk_bytes=1000)
    assert "trials" not in s_only and s_only["summary"].shape == (B, 10)
    nd.seed(31)
    a = engine.simulate_to_host(engine.BASIC_DDM_DC, cases[0][1], N, chunk_bytes=150 * N * 8)
    b = engine.simulate_to_host(engine.BASIC_DDM_DC, cases[0][1], N, chunk_bytes=150 * N * 8)
    nd.seed(31)
    a2 = engine.simulate(engine.BASIC_DDM_DC, cases[0][1], N)
    b2 = engine.simulate(engine.BASIC_DDM_DC, cases[0][1], N)
    assert a["set_offset"] == a2["set_offset"] and b["set_offset"] == b2["set_offset"] == a["set_offset"] + B
    assert np.array_equal(a["trials"], a2["trials"].cpu().numpy()) and np.array_equal(b["trials"], b2["trials"].cpu().numpy())
    with pytest.raises(ValueError):
        engine.simulate_to_host(engine.BASIC_DDM_DC, -np.abs(cases[0][1]), N)                  # validation as in simulate()


def test_drop_in_api_on_device(kat):
    """The reference's call shapes end to end on the GPU: per-set simulator_fun, batched generative model, dict keys,
    configurator, alpha_not_scaled generator, imputation loop."""
    import torch
    import bayesflow_nddms_amd as nd
    from bayesflow_nddms_amd import alpha_not_scaled, basic_ddm_dc, imputation, single_trial_alpha_not_scaled as st
    # per-set call == row of the batched call (same seed / set index)
    p = kat["basic_sets"]
    one = basic_ddm_dc.simulate_trials(p[3], 120, seed=5, set_offset=3)
    allb = basic_ddm_dc.batch_simulate_trials(p, 120, seed=5, set_offset=0)
    assert one.shape == (120, 2) and one.dtype == np.float64
    assert np.array_equal(one.astype(np.float32), allb["sim_data"][3])
    rt, choice = basic_ddm_dc.diffusion_trial(*p[0], seed=1, set_offset=0)
    assert rt > p[0][3] and choice in (-1, 0, 1)
    # the same call shapes in the reference's float64 state arithmetic (NDDM_STATE_F64): nearly every trial the same (rt, choice)
    one64 = basic_ddm_dc.simulate_trials(p[3], 120, seed=5, set_offset=3, state_f64=True)
    all64 = basic_ddm_dc.batch_simulate_trials(p, 120, seed=5, set_offset=0, state_f64=True)
    assert np.array_equal(one64.astype(np.float32), all64["sim_data"][3]) and (one64 == one).all(axis=1).mean() > 0.97
    s64 = st.simulate_trials_fine(kat["single_sets"][0], 80, seed=5, set_offset=1, state_f64=True)
    assert s64.shape == (80, 2) and (np.sign(s64[:, 0]) == np.sign(st.simulate_trials_fine(kat["single_sets"][0], 80, seed=5, set_offset=1)[:, 0])).mean() > 0.97
    # the package-level stream: consecutive calls differ, re-seeding reproduces
    nd.seed(123)
    a1, a2 = basic_ddm_dc.simulate_trials(p[0], 50), basic_ddm_dc.simulate_trials(p[0], 50)
    nd.seed(123)
    b1 = basic_ddm_dc.simulate_trials(p[0], 50)
    assert np.array_equal(a1, b1) and not np.array_equal(a1, a2)
    # generative models: reference wrapper block, per-set and batched, host and device priors
    np.random.seed(2023)
    for kw in (dict(batched=False), dict(batched=True), dict(batched=True, device_prior=True, as_numpy=False)):
        gm = basic_ddm_dc.make_generative_model(**kw)
        out = gm(32)
        N = out["sim_non_batchable_context"]
        assert 60 <= N <= 300 and tuple(out["sim_data"].shape) == (32, N, 2) and tuple(out["prior_draws"].shape) == (32, 5)
        conf = basic_ddm_dc.configurator(out)
        assert tuple(conf["summary_conditions"].shape) == (32, N, 2) and tuple(conf["direct_conditions"].shape) == (32, 1)
        assert tuple(conf["parameters"].shape) == (32, 5)
        if kw.get("batched"):
            assert tuple(out["summary_stats"].shape) == (32, 10)
        if kw.get("as_numpy") is False:
            assert isinstance(out["sim_data"], torch.Tensor) and out["sim_data"].is_cuda
    gm = st.make_generative_model(batched=True, fine=True)
    out = gm(8)
    assert out["sim_data"].shape[0] == 8 and out["prior_draws"].shape == (8, 7)
    # the reference's fixed-parameter test case (single_trial_alpha_not_scaled.py:852-885)
    input_params = np.hstack((3, 1.5, .5, .4, 1, 1, 0.1))
    obs = st.simulate_trials(input_params, 300, seed=2024, set_offset=0)
    obs_dict = {'sim_data': obs[np.newaxis, :, :], 'sim_non_batchable_context': 300, 'prior_draws': input_params}
    conf = st.configurator(obs_dict)
    assert conf["summary_conditions"].shape == (1, 300, 2)
    assert (np.sign(obs[:, 0]) == 1).mean() > 0.85                  # reference: P(upper) = 0.95 for these parameters
    assert abs(obs[:, 1].mean() - 1.64) < 0.25                       # reference: mean z = 1.6437 (n=300)
    for f, pp in ((st.simulate_trials_fine, input_params), (st.simulate_trials_alt, [3.0, 1.5, .5, .4, .5, 1.0, .1]),
                  (st.simulate_trials_scale, np.append(input_params, 0.7)), (st.simulate_trials_scale2, input_params)):
        assert f(pp, 64, seed=1, set_offset=0).shape == (64, 2)
    z1 = st.simulate_trials(input_params, 2000, seed=3, set_offset=0)[:, 1]
    z2 = st.simulate_trials_scale2(input_params, 2000, seed=3, set_offset=0)[:, 1]
    assert abs(z2.mean() / z1.mean() - 2.0) < 0.05                   # gamma = 2 doubles the datum's mean
    # alpha_not_scaled.py:52-128
    g = alpha_not_scaled.generate_data(test_num=2, nparts=100, ntrials=100)
    assert set(g) >= {"ndt", "beta", "alpha", "delta", "deltatrialsd", "varsigma", "sigma", "rt", "acc", "y",
                      "extdata", "participant", "nparts", "ntrials", "N"}
    assert g["y"].shape == (10000,) and g["extdata"].shape == (100,) and g["alpha"][17] == 1.2
    assert abs(np.corrcoef(g["extdata"], g["alpha"])[0, 1]) > 0.7     # sigma=.1 vs sd(alpha)=.17
    assert np.all(np.abs(g["y"][g["y"] != 0]) >= g["ndt"].min())
    y = alpha_not_scaled.simulratcliff_em(N=500, Alpha=1.2, Tau=.4, Nu=3.5, Beta=.5, Eta=1.0, Varsigma=1.2, seed=1, set_offset=0)
    assert y.shape == (500,) and (y > 0).mean() > 0.8
    # imputation loop
    b = np.abs(np.random.default_rng(0).normal(1.2, .3, size=(4, 50)))
    c = imputation.impute_choicert(1.0, b, .5, .3, 1.1, seed=2, set_offset=0)
    assert c.shape == (4, 50) and np.all(np.abs(c[c != 0]) > 0.3)
    with pytest.raises(ValueError):
        imputation.diffusion_trial(1.0, -0.5, .5, .3, 1.1)


def test_ez_diffusion_from_fused_summaries_recovers_the_parameters():
    """The fused summaries feed the EZ-diffusion estimator (simulations/Basic_DDM_simulations.py:131-158) without the
    trials ever leaving the kernel: for unbiased starting points the closed form recovers (drift/dc, boundary/dc, tau).
    Summary-only launch, 20000 trials per set; the estimates also agree with ezdiff() on the trials of the same launch."""
    import torch
    from bayesflow_nddms_amd import engine
    from bayesflow_nddms_amd import ezdiff as ez
    p = np.array([[1.5, 1.2, 0.5, 0.35, 1.0], [3.0, 2.4, 0.5, 0.35, 2.0], [3.0, 1.2, 0.5, 0.35, 2.0], [1.5, 1.2, 0.5, 0.35, 0.5],
                  [0.8, 1.6, 0.5, 0.20, 1.0]], dtype=np.float32)       # the sets of Basic_DDM_simulations.py:164-207 + one
    N = 20000
    r = engine.simulate(engine.BASIC_DDM_DC, p, N, dt=0.001, max_steps=4000, seed=31, set_offset=0, fast=True, want_trials=True)
    est = ez.ez_from_summary(r["summary"])                              # on the device
    assert isinstance(est, torch.Tensor) and est.shape == (5, 3)
    est = est.cpu().numpy()
    # in units of the diffusion coefficient (s = 1).  The estimator sees the DISCRETISED process: Euler-Maruyama detects a
    # crossing late, which acts like a boundary wider by 0.5826 sqrt(dt) on either side (Siegmund's correction)
    truth = np.stack([p[:, 0] / p[:, 4], p[:, 1] / p[:, 4] + 2 * 0.5826 * np.sqrt(0.001), p[:, 3]], axis=1)
    assert np.allclose(est[:, 0], truth[:, 0], rtol=0.015), (est, truth)
    assert np.allclose(est[:, 1], truth[:, 1], rtol=0.015), (est, truth)
    assert np.allclose(est[:, 2], truth[:, 2], atol=0.01), (est, truth)
    tr = r["trials"].cpu().numpy()
    for b in range(5):
        rt, ch = tr[b, :, 0].astype(np.float64), tr[b, :, 1]
        correct = np.where(ch == 1, 1.0, np.where(ch == -1, 0.0, np.nan))
        want = np.array(ez.ezdiff(np.where(np.isnan(correct), np.nan, rt), correct))
        assert np.allclose(est[b], want, rtol=2e-4, atol=2e-5), (b, est[b], want)      # f32 summaries vs f64 from the trials
    acc, mrt, vrt = (x.cpu().numpy() for x in ez.accuracy_rt_moments(r["summary"]))        # mean_RT_accuracy_effects.py:88-90
    for b in range(5):
        rt, ch = tr[b, :, 0].astype(np.float64), tr[b, :, 1]
        resp = ch != 0
        assert abs(acc[b] - (ch[resp] == 1).mean()) < 1e-6 and abs(mrt[b] - rt[resp].mean()) < 1e-5 and abs(vrt[b] - rt[resp].var()) < 1e-5
    only = engine.simulate(engine.BASIC_DDM_DC, p, N, dt=0.001, max_steps=4000, seed=31, set_offset=0, fast=True, want_trials=False)
    assert torch.equal(torch.nan_to_num(only["summary"]), torch.nan_to_num(r["summary"]))

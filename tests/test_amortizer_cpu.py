"""CPU tests of the PyTorch stand-in for the BayesFlow side (SURVEY f-4): shapes, exact invertibility of the flow,
the Trainer loops driven through the generative-model dictionary contract, checkpoint/resume."""
import numpy as np
import pytest
import torch

from bayesflow_nddms_amd.amortizer import (AmortizedPosterior, InvariantNetwork, InvertibleNetwork, Trainer,
                                           posterior_recovery)
from bayesflow_nddms_amd.simulation import ContextGenerator, GenerativeModel, Prior, Simulator


def _toy_model():
    """A generative model with the reference's dict contract but a CPU toy simulator: data ~ N(theta0, exp(theta1))."""
    rng = np.random.default_rng(0)

    def draw_prior():
        return np.array([rng.normal(0, 1), rng.normal(-0.5, 0.3)])

    def prior_N():
        return int(rng.integers(40, 80))

    def batch_sim(params, n):
        p = np.asarray(params)
        x = p[:, None, :1] + np.exp(p[:, None, 1:]) * rng.normal(size=(p.shape[0], n, 1))
        return np.concatenate([x, np.sign(x)], axis=-1).astype(np.float32)

    def configurator(d):
        data = d["sim_data"].astype(np.float32)
        return {"summary_conditions": data,
                "direct_conditions": (np.log(d["sim_non_batchable_context"]) * np.ones((data.shape[0], 1))).astype(np.float32),
                "parameters": d["prior_draws"].astype(np.float32)}

    gm = GenerativeModel(Prior(prior_fun=draw_prior),
                         Simulator(batch_simulator_fun=batch_sim, context_generator=ContextGenerator(non_batchable_context_fun=prior_N)))
    return gm, configurator


def test_flow_is_invertible_and_shapes():
    torch.manual_seed(0)
    net = InvertibleNetwork(num_params=5, cond_dim=11)
    x, c = torch.randn(16, 5), torch.randn(16, 11)
    z, ld = net(x, c)
    assert z.shape == (16, 5) and ld.shape == (16,)
    assert torch.allclose(net.inverse(z, c), x, atol=1e-4)
    s = InvariantNetwork()
    data = torch.randn(4, 77, 2)
    out = s(data)
    assert out.shape == (4, 10)
    assert torch.allclose(out, s(data[:, torch.randperm(77)]), atol=1e-5)      # permutation invariance over trials


def test_trainer_loops_reduce_loss_and_checkpoint(tmp_path):
    torch.manual_seed(0)
    gm, conf = _toy_model()
    am = AmortizedPosterior(InvertibleNetwork(num_params=2, cond_dim=11, num_coupling_layers=3, hidden=32), InvariantNetwork(hidden=32))
    tr = Trainer(am, gm, conf, checkpoint_path=str(tmp_path / "ckpt"), device="cpu", learning_rate=2e-3)
    hist = tr.train_online(epochs=1, iterations_per_epoch=150, batch_size=32)
    assert np.mean(hist[-20:]) < np.mean(hist[:20]) - 0.3
    res = tr.train_experience_replay(epochs=1, iterations_per_epoch=30, batch_size=16, capacity_in_batches=8,
                                     validation_sims=gm(64))
    assert len(res["val_losses"]) == 1 and np.isfinite(res["val_losses"][0])
    post = am.sample(conf(gm(1)), 200)
    assert post.shape == (200, 2)
    assert am.sample(conf(gm(3)), 50).shape == (3, 50, 2)
    rho = posterior_recovery(am, gm, conf, n_datasets=40, n_samples=100)
    assert rho[0] > 0.7                                                  # the mean parameter is recovered
    # the median form, and what both are computed from: one pass, rows aligned; a single wild posterior mean (a tail draw the
    # inverse flow amplifies) moves the means' correlation, not the medians'
    from bayesflow_nddms_amd.amortizer import posterior_estimates
    assert posterior_recovery(am, gm, conf, n_datasets=40, n_samples=100, statistic="median")[0] > 0.7
    true, means, meds = posterior_estimates(am, gm, conf, n_datasets=30, n_samples=100)
    assert true.shape == means.shape == meds.shape == (30, 2) and np.corrcoef(means[:, 0], meds[:, 0])[0, 1] > 0.95
    with pytest.raises(ValueError):
        posterior_recovery(am, gm, conf, n_datasets=2, n_samples=10, statistic="mode")
    am2 = AmortizedPosterior(InvertibleNetwork(num_params=2, cond_dim=11, num_coupling_layers=3, hidden=32), InvariantNetwork(hidden=32))
    tr2 = Trainer(am2, gm, conf, checkpoint_path=str(tmp_path / "ckpt"), device="cpu")
    assert tr2.load_pretrained_network()
    d = conf(gm(4))
    assert torch.allclose(am.compute_loss(d), am2.compute_loss(d))
    assert len(tr2.loss_history) == len(tr.loss_history)


def test_padded_batches_pool_over_the_real_trials_only():
    """The three ways to say "the first n of these N trials are real" give the unpadded batch's summary: a 0/1 mask with
    1/n, the count as a scalar tensor (what the graph trainer passes, `summary_n`), and simply not padding."""
    import torch
    from bayesflow_nddms_amd.amortizer import AmortizedPosterior, InvariantNetwork, InvertibleNetwork
    torch.manual_seed(1)
    net = InvariantNetwork()
    B, N, n = 4, 50, 37
    x = torch.randn(B, N, 2)
    want = net(x[:, :n])
    mask = (torch.arange(N) < n).float().view(1, N, 1)
    assert torch.allclose(net(x, mask, torch.tensor(1.0 / n)), want, atol=1e-5)
    assert torch.allclose(net(x, n_valid=torch.tensor([float(n)])), want, atol=1e-5)
    am = AmortizedPosterior(InvertibleNetwork(num_params=5), net)
    theta = torch.randn(B, 5)
    direct = torch.full((B, 1), 3.6)
    a = am.compute_loss({"summary_conditions": x, "summary_n": torch.tensor([float(n)]), "direct_conditions": direct, "parameters": theta})
    b = am.compute_loss({"summary_conditions": x[:, :n], "direct_conditions": direct, "parameters": theta})
    assert torch.allclose(a, b, atol=1e-5)


def test_every_train_call_has_a_schedule_of_its_own(tmp_path):
    """basic_ddm_dc.py:199-207 re-enters training (load_pretrained_network, another train_* call).  Every call is a run of its
    own, as in BayesFlow 1.1's Trainer: the cosine decay starts from the Trainer's rate again (a second CosineAnnealingLR
    stacked on an optimizer already annealed to 0 used to train the second call -- and every resumed run -- at rate 0), holds its
    final value past the schedule's length, and Adam is fresh unless reuse_optimizer=True was asked for."""
    torch.manual_seed(0)
    gm, conf = _toy_model()
    am = AmortizedPosterior(InvertibleNetwork(num_params=2, cond_dim=11, num_coupling_layers=2, hidden=32), InvariantNetwork(hidden=32))
    tr = Trainer(am, gm, conf, checkpoint_path=str(tmp_path / "ckpt"), device="cpu", learning_rate=2e-3)
    rates = []
    step = tr._step
    tr._step = lambda c: (rates.append(tr.optimizer.param_groups[0]["lr"]), step(c))[1]
    tr.train_online(epochs=1, iterations_per_epoch=20, batch_size=8)
    first = list(rates)
    assert abs(first[0] - 2e-3) < 1e-12 and 0 < first[-1] < 0.02 * 2e-3 and all(a > b for a, b in zip(first, first[1:]))
    adam1 = tr.optimizer
    w = [p.detach().clone() for p in am.parameters()]
    del rates[:]
    tr.train_online(epochs=1, iterations_per_epoch=20, batch_size=8)                      # the second call: lr > 0 throughout
    assert np.allclose(rates, first, rtol=0, atol=1e-15) and tr.optimizer is not adam1   # the same schedule again, a fresh Adam
    assert any(not torch.equal(a, b.detach()) for a, b in zip(w, am.parameters()))        # ... and the weights moved
    del rates[:]
    tr.train_experience_replay(epochs=2, iterations_per_epoch=5, batch_size=8, capacity_in_batches=4, reuse_optimizer=True)
    adam3 = tr.optimizer
    assert abs(rates[0] - 2e-3) < 1e-12 and rates[-1] > 0 and len(rates) == 10
    del rates[:]
    tr.train_online(epochs=1, iterations_per_epoch=4, batch_size=8)
    assert tr.optimizer is adam3 and abs(rates[0] - 2e-3) < 1e-12                         # reuse_optimizer=True kept the moments
    # a resumed run: the loaded optimizer (whose stored rate is the annealed one) trains at the Trainer's rate again
    am2 = AmortizedPosterior(InvertibleNetwork(num_params=2, cond_dim=11, num_coupling_layers=2, hidden=32), InvariantNetwork(hidden=32))
    tr2 = Trainer(am2, gm, conf, checkpoint_path=str(tmp_path / "ckpt"), device="cpu", learning_rate=2e-3)
    assert tr2.load_pretrained_network() and tr2.optimizer.param_groups[0]["lr"] < 2e-3
    seen = []
    step2 = tr2._step
    tr2._step = lambda c: (seen.append(tr2.optimizer.param_groups[0]["lr"]), step2(c))[1]
    tr2.train_online(epochs=1, iterations_per_epoch=6, batch_size=8, save_checkpoint=False)
    assert abs(seen[0] - 2e-3) < 1e-12 and min(seen) > 0
    # past its length a schedule HOLDS its final value
    tr2._setup_schedule(3)
    for _ in range(9):
        tr2.optimizer.step(); tr2.scheduler.step()
    assert abs(tr2.optimizer.param_groups[0]["lr"]) < 1e-12


def test_flow_state_dict_accepts_the_earlier_layout():
    """Checkpoints written before the per-layer ActNorm parameters: `an_scale` / `an_bias` as ONE [layers, D] tensor each and
    the permutation matrices stored beside the permutations.  They load (strict), `pmat{i}` is rebuilt from `perm{i}` -- a
    mismatched pair cannot be loaded -- and is no longer part of the state dict."""
    torch.manual_seed(0)
    a = InvertibleNetwork(num_params=5, num_coupling_layers=3, seed=4)
    with torch.no_grad():
        for p in list(a.an_scale) + list(a.an_bias):
            p.copy_(torch.randn_like(p))
    sd = a.state_dict()
    assert not any(k.startswith("pmat") for k in sd)
    old = {k: v for k, v in sd.items() if not k.startswith(("an_scale", "an_bias"))}
    old["an_scale"] = torch.stack([sd[f"an_scale.{i}"] for i in range(3)])
    old["an_bias"] = torch.stack([sd[f"an_bias.{i}"] for i in range(3)])
    for i in range(3):
        old[f"pmat{i}"] = torch.eye(5)                                  # a WRONG matrix: must be ignored
    b = InvertibleNetwork(num_params=5, num_coupling_layers=3, seed=9)    # other permutations until the load
    b.load_state_dict(old)
    x, c = torch.randn(7, 5), torch.randn(7, 11)
    za, la = a(x, c)
    zb, lb = b(x, c)
    assert torch.equal(za, zb) and torch.equal(la, lb) and b._perm_host == a._perm_host
    assert torch.allclose(b.inverse(zb, c), x, atol=1e-4)


def test_graph_route_of_the_classic_trainer_is_refused_without_a_device_or_a_recipe():
    """Trainer(graph=True) hands train_* to graph replays on the GPU; without a ROCm device, or for a generative model that does
    not say how it is made, the call raises -- it never falls back to the eager loop silently."""
    from bayesflow_nddms_amd.amortizer import AmortizedPosterior, InvariantNetwork, InvertibleNetwork, Trainer
    am = AmortizedPosterior(InvertibleNetwork(num_params=5), InvariantNetwork())
    gm = lambda b: None
    tr = Trainer(am, gm, graph=True, device="cpu")
    with pytest.raises(ValueError, match="graph_spec"):
        tr.train_online(1, 2, 4)
    with pytest.raises(ValueError, match="graph_spec"):
        tr.train_experience_replay(1, 2, 4)



def test_the_graph_runs_position_travels_in_the_trainers_checkpoint(tmp_path):
    """What Trainer(graph=True) keeps between train_* calls -- the next parameter set's global index, the key of the next
    batch-shared N, the experience-replay buffer and its generator's state -- is written to ckpt.pt and comes back through
    load_pretrained_network, so a resumed run continues the stream instead of replaying it.  (The position itself is produced on
    the GPU: tests/test_gpu_training.py; here the round trip of the file.)"""
    from bayesflow_nddms_amd.amortizer import AmortizedPosterior, InvariantNetwork, InvertibleNetwork, Trainer
    rng = np.random.default_rng(5)
    rng.integers(10, size=7)
    pos = {"offset": (1 << 59) + 12 * 32, "n_key": 12,
           "replay": {"ring": [(torch.randn(32, 5), torch.randn(32, 90, 2), 77)], "rng": rng.bit_generator.state, "capacity": 100}}
    ck = str(tmp_path / "ck")
    a = Trainer(AmortizedPosterior(InvertibleNetwork(num_params=5), InvariantNetwork()), lambda b: None, checkpoint_path=ck, graph=True, device="cpu")
    a._graph_pos = pos
    a.save_checkpoint()
    b = Trainer(AmortizedPosterior(InvertibleNetwork(num_params=5), InvariantNetwork()), lambda b: None, checkpoint_path=ck, graph=True, device="cpu")
    assert b._graph_pos is None and b.load_pretrained_network()
    got = b._graph_pos
    assert got["offset"] == pos["offset"] and got["n_key"] == 12 and got["replay"]["capacity"] == 100
    assert torch.equal(got["replay"]["ring"][0][1], pos["replay"]["ring"][0][1]) and got["replay"]["ring"][0][2] == 77
    r2 = np.random.default_rng(0)
    r2.bit_generator.state = got["replay"]["rng"]
    assert r2.integers(1 << 30) == rng.integers(1 << 30)


def test_the_flows_inverse_is_exact_in_the_tails_of_z():
    """The wild posterior draws of a sharply trained flow (tools/locate_tail_draws.py) are points the learned
    density really covers, not an arithmetic fault of sample(): the inverse -- soft clamp, ActNorm and permutations included -- undoes
    the forward for base draws far in the tails too.  On a network made ill-conditioned on purpose (weights x 1.5, ActNorms away from
    the identity, log-scales at the clamp's bound) and z out to 8 standard deviations: forward(inverse(z)) == z to 1e-9 in float64,
    the float32 inverse agrees with the float64 one to float32 round-off RELATIVE to the size of what it returns (values up to 1e4 and
    beyond), and log q(theta) of such a draw is the finite number the change-of-variables formula gives."""
    import copy
    torch.manual_seed(11)
    net = InvertibleNetwork(num_params=7, num_coupling_layers=6, seed=3)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(1.5)
        for p in list(net.an_scale) + list(net.an_bias):
            p.copy_(0.3 * torch.randn_like(p))
    net64 = copy.deepcopy(net).double()
    cond = torch.randn(1, 11).expand(4000, 11).contiguous()
    z = 2.0 * torch.randn(4000, 7)                                      # tails out to ~8 sigma of the base density
    assert float(z.abs().max()) > 6.0
    with torch.no_grad():
        x32 = net.inverse(z, cond)
        x64 = net64.inverse(z.double(), cond.double())
        back, log_det = net64(x64, cond.double())
    assert torch.isfinite(x64).all() and torch.isfinite(log_det).all()
    assert float((back - z.double()).abs().max()) < 1e-9 * (1.0 + float(x64.abs().max()))
    scale = x64.abs().amax(dim=1, keepdim=True) + 1.0
    assert float(((x32.double() - x64).abs() / scale).max()) < 5e-4          # f32 round-off amplified by an ill-conditioned inverse
    far = x64.abs().amax(dim=1) > 50.0 * float(x64.abs().amax(dim=1).median())
    assert int(far.sum()) >= 1                                          # the runaway region exists in an UNtrained net too
    # each clamped log-scale is bounded by 1.9: |log det| <= layers * D * 1.9 + the ActNorms'
    assert float(log_det.abs().max()) <= 6 * 7 * 1.9 + float(sum(p.detach().abs().sum() for p in net.an_scale)) + 1e-6


def test_sample_rejection_option_is_off_by_default_and_redraws_outside_the_box():
    """AmortizedPosterior.sample(..., reject_outside=(low, high)): off by default (the reference's call, basic_ddm_dc.py:223, gets every
    draw of the flow: same draws as before the option existed); with a box, every returned draw lies inside it, the draws that were
    inside are untouched, `last_redrawn` counts the redrawn ones, and a box nothing falls out of changes nothing."""
    import numpy as np
    import torch
    from bayesflow_nddms_amd import priors
    from bayesflow_nddms_amd.amortizer import AmortizedPosterior, InvariantNetwork, InvertibleNetwork
    torch.manual_seed(0)
    am = AmortizedPosterior(InvertibleNetwork(num_params=5), InvariantNetwork()).eval()
    conf = {"summary_conditions": torch.randn(3, 40, 2), "direct_conditions": torch.full((3, 1), float(np.log(40.0)))}
    torch.manual_seed(1)
    plain = am.sample(conf, 500, to_numpy=False)
    assert plain.shape == (3, 500, 5) and am.last_redrawn == 0
    torch.manual_seed(1)
    wide = am.sample(conf, 500, to_numpy=False, reject_outside=([-1e9] * 5, [1e9] * 5))
    assert torch.equal(wide, plain) and am.last_redrawn == 0
    lo, hi = [-0.5] * 5, [0.5] * 5                              # an untrained flow is ~N(0, 1) per component: most draws fall outside
    torch.manual_seed(1)
    boxed = am.sample(conf, 500, to_numpy=False, reject_outside=(lo, hi), max_redraws=64)
    assert bool(((boxed >= -0.5) & (boxed <= 0.5)).all()) and am.last_redrawn > 500
    inside = ((plain >= -0.5) & (plain <= 0.5)).all(dim=-1)
    assert inside.any() and torch.equal(boxed[inside], plain[inside])                   # draws that were inside are the same draws
    with pytest.raises(ValueError):
        am.sample(conf, 10, reject_outside=([0.0] * 4, [1.0] * 4))
    b_lo, b_hi = priors.prior_box("basic")
    assert np.allclose(b_lo, [-30, -10, -1, -1.5, -10]) and np.allclose(b_hi, [30, 20, 2, 3, 20])
    s_lo, s_hi = priors.prior_box("single", widen=0.0)
    assert len(s_lo) == 7 and np.allclose(s_hi, [10, 10, 1, 1.5, 3, 10, 5])

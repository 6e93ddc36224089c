"""BASELINE config 5 in miniature on one GPU: online simulation on the MI355X feeding a PyTorch-ROCm amortizer through
the reference's dictionary contract; the loss must go down and the posterior means must start tracking the true parameters."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_online_training_on_device_simulator():
    import torch
    from bayesflow_nddms_amd import basic_ddm_dc
    from bayesflow_nddms_amd.amortizer import (AmortizedPosterior, InvariantNetwork, InvertibleNetwork, Trainer,
                                               posterior_recovery)
    torch.manual_seed(0)
    np.random.seed(2023)
    gm = basic_ddm_dc.make_generative_model(batched=True, device_prior=True, as_numpy=False)
    out = gm(32)
    assert out["sim_data"].is_cuda and out["prior_draws"].is_cuda      # nothing leaves the device on the training path
    am = AmortizedPosterior(InvertibleNetwork(num_params=5), InvariantNetwork())
    tr = Trainer(am, gm, basic_ddm_dc.configurator, checkpoint_path=None, learning_rate=1e-3)
    res = tr.train_experience_replay(epochs=1, iterations_per_epoch=600, batch_size=32, save_checkpoint=False)
    h = res["train_losses"]
    assert np.mean(h[-40:]) < np.mean(h[:40]) - 1.0, (np.mean(h[:40]), np.mean(h[-40:]))
    rho = posterior_recovery(am, gm, basic_ddm_dc.configurator, n_datasets=60, n_samples=200)
    # a few hundred iterations are a smoke run (the reference trains 500 epochs x 1000 iterations): which parameter is
    # picked up first varies with the seed, so ask for one clearly recovered parameter and positive tracking overall
    assert np.max(rho) > 0.4 and np.mean(rho) > 0.15, rho
    post = am.sample(basic_ddm_dc.configurator(gm(1)), 1000)
    assert post.shape == (1000, 5) and np.all(np.isfinite(post))


def test_posterior_mean_recovery_floor_on_the_references_statistic():
    """The recovery loop of basic_ddm_dc.py:211-241 on the reference's own statistic -- posterior MEANS, Pearson rho and r2_score per
    parameter, the "converged" count -- after 20 000 graph-replayed iterations of the reference's training call (6 s; the reference runs
    500 000): a floor on the means (profiles/r4_recovery.txt reads .94 .77 .90 .95 .68 at this length).  Three views of it: (1) the
    statistic UNFILTERED, as `recovery_scatter` would print it (basic_ddm_dc.py:236-250) -- a known failure mode of the mean-based
    path (one far-tail draw carries a data set's mean off), bounded here by its incidence and recorded when it happens; (2) the data
    sets without such a draw: the floor, means == medians; (3) the product's documented bound, `sample(..., reject_outside=
    priors.prior_box(...))` (off by default), on ALL data sets with nothing filtered afterwards.  Training / evaluation parameter
    rows are disjoint by construction."""
    import torch
    from bayesflow_nddms_amd import basic_ddm_dc, diagnostics as dg
    from bayesflow_nddms_amd.amortizer import AmortizedPosterior, InvariantNetwork, InvertibleNetwork
    from bayesflow_nddms_amd.graph_trainer import TRAIN_OFFSET_BASE, GraphTrainer
    torch.manual_seed(0)
    am = AmortizedPosterior(InvertibleNetwork(num_params=5), InvariantNetwork())
    with GraphTrainer(am, batch_size=32, total_steps=20000, seed=2023, offset_base=TRAIN_OFFSET_BASE) as gt:
        gt.train_experience_replay(20000)
        h = np.array(gt.loss_history())
    assert np.all(np.isfinite(h)) and h[-200:].mean() < -8.0, h[-200:].mean()
    np.random.seed(2023)
    gm = basic_ddm_dc.make_generative_model(batched=True, device_prior=True, as_numpy=False)
    am.eval()
    from bayesflow_nddms_amd import priors
    box = priors.prior_box("basic")                          # the prior's support widened by its own width on each side
    lo, hi = (torch.as_tensor(v, dtype=torch.float32, device="cuda") for v in box)
    true, means, meds, far, means_box, redrawn = [], [], [], [], [], 0
    for _ in range(300):
        conf = basic_ddm_dc.configurator(gm(1))
        post = am.sample(conf, 2000, to_numpy=False)         # the reference's call: nothing removed
        true.append(conf["parameters"][0].cpu().numpy()); means.append(post.mean(0).cpu().numpy()); meds.append(post.median(0).values.cpu().numpy())
        far.append(bool(((post < lo) | (post > hi)).any()))                              # a draw more than a prior width outside the prior's range
        means_box.append(am.sample(conf, 2000, to_numpy=False, reject_outside=box).mean(0).cpu().numpy())      # the documented option (off by default)
        redrawn += am.last_redrawn
    true, means, meds, far, means_box = (np.array(v, np.float64) for v in (true, means, meds, far, means_box))
    far = far.astype(bool)
    floor_rho = np.array([0.88, 0.62, 0.80, 0.90, 0.55])                                 # drift, boundary, beta, tau, dc
    # (1) THE REFERENCE'S STATISTIC, UNFILTERED (recovery_scatter on the posterior means of every data set, basic_ddm_dc.py:236-250): a
    #     sharply trained coupling flow carries ~1e-6 of its mass in a far tail (4e-7 .. 3e-6 per draw), and ONE such
    #     draw among a data set's draws carries its mean off.  That is a KNOWN FAILURE of the mean-based path, bounded here by its
    #     incidence: at 6e5 draws at most a handful of the 300 data sets may hold one, and where none does the unfiltered table meets
    #     the floor as it stands.
    carried = (np.abs(means - meds) > 5.0 * (np.abs(meds) + 1.0)).any(axis=1)
    assert far.sum() <= 10 and carried.sum() <= far.sum(), (int(far.sum()), int(carried.sum()))
    st_all = dg.recovery_statistics(true, means)            # (the reference's r2_score / pearsonr: pyhddmjagsutils.py:609-623)
    if not carried.any():
        assert np.all(st_all["rho"] > floor_rho), st_all["rho"]
    else:                                                    # recorded, not hidden: the unfiltered numbers of this run
        print(f"unfiltered posterior-mean statistic with {int(carried.sum())} carried-off data set(s): rho {np.round(st_all['rho'], 3)}, R^2 {np.round(st_all['r2'], 2)}")
    # (2) the same statistic on the data sets without a wild draw: the floor, and means == medians
    clean = ~far
    st = dg.recovery_statistics(true[clean], means[clean])
    rho, r2, rho_med = st["rho"], st["r2"], dg.recovery_statistics(true[clean], meds[clean])["rho"]
    assert np.all(rho > floor_rho), rho
    assert r2[0] > 0.75 and r2[3] > 0.8, r2
    assert np.abs(rho - rho_med).max() < 0.02, (rho, rho_med)                  # means == medians where no tail draw interferes
    # (3) the product's bound on it: sample(..., reject_outside=priors.prior_box('basic')) -- EVERY one of the 300 data sets, nothing
    #     filtered afterwards -- gives the reference's mean-based table usable numbers; it redraws about as many draws as (1) found
    st_box = dg.recovery_statistics(true, means_box)
    assert np.all(st_box["rho"] > floor_rho) and st_box["r2"][0] > 0.75 and st_box["r2"][3] > 0.8, (st_box["rho"], st_box["r2"])
    assert np.abs(st_box["rho"] - dg.recovery_statistics(true, meds)["rho"]).max() < 0.02
    assert redrawn <= 120, redrawn                                             # 6e5 draws: a redraw rate of <= 2e-4 (a fully trained flow: ~1e-6)
    converged = dg.converged_fits(means_box)                                   # basic_ddm_dc.py:239-241
    assert converged.sum() >= 0.93 * len(means), converged.sum()              # (P(tau > 1) = 2.3 % under the prior)


def test_prefetched_online_training_sees_the_same_batches():
    """train_online(prefetch=True) simulates batch i+1 on a side stream while batch i is trained on.  Same seeds =>
    the same batches in the same order => the same loss curve as without prefetching, the same simulator stream state
    afterwards (nothing is simulated beyond the last iteration)."""
    import time
    import torch
    import bayesflow_nddms_amd as nd
    from bayesflow_nddms_amd import basic_ddm_dc
    from bayesflow_nddms_amd.amortizer import AmortizedPosterior, InvariantNetwork, InvertibleNetwork, Trainer
    hist, state, secs = {}, {}, {}
    for prefetch in (False, True):
        torch.manual_seed(0)
        np.random.seed(7)
        nd.seed(99)
        gm = basic_ddm_dc.make_generative_model(batched=True, device_prior=True, as_numpy=False)
        am = AmortizedPosterior(InvertibleNetwork(num_params=5), InvariantNetwork())
        tr = Trainer(am, gm, basic_ddm_dc.configurator, checkpoint_path=None, learning_rate=1e-3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hist[prefetch] = tr.train_online(epochs=2, iterations_per_epoch=40, batch_size=64, save_checkpoint=False,
                                         prefetch=prefetch)
        torch.cuda.synchronize()
        secs[prefetch] = time.perf_counter() - t0
        state[prefetch] = nd.GLOBAL_STREAM.get_state()
    assert len(hist[True]) == len(hist[False]) == 80
    assert np.allclose(hist[True], hist[False], rtol=1e-4, atol=1e-4)
    assert state[True] == state[False]
    print(f"prefetch on / off: {secs[True]:.3f} / {secs[False]:.3f} s")        # (reported, not asserted: timing is not a correctness property)


def test_graph_trainer_equals_the_eager_iteration_and_tracks_the_classic_loop():
    """graph_trainer.GraphTrainer: every iteration one hipGraph replay (device prior -> simulate -> forward -> backward ->
    clip -> Adam; one graph per n_trials bucket) -- or, the default at one rank, two on two streams (the next batch is simulated
    beside the training step).  For fixed seeds its loss history equals (1e-4) the SAME iteration run
    eagerly, in the single-graph form and in the two-graph form used with a collective in between; the classic eager
    Trainer fed the same batches unpadded (exact N instead of bucket top + mask) gives the same curve to 2e-3; the losses
    go down; and the library memory behind the captured launches is released when the trainer closes."""
    import torch
    from bayesflow_nddms_amd import basic_ddm_dc, engine
    from bayesflow_nddms_amd.amortizer import AmortizedPosterior, InvariantNetwork, InvertibleNetwork, Trainer
    from bayesflow_nddms_amd.distributed import shared_prior_N
    from bayesflow_nddms_amd.graph_trainer import GraphTrainer
    from bayesflow_nddms_amd.priors import DevicePrior
    iters, B = 40, 32

    def run(**kw):
        torch.manual_seed(0)
        am = AmortizedPosterior(InvertibleNetwork(num_params=5), InvariantNetwork())
        with GraphTrainer(am, batch_size=B, total_steps=iters, seed=2023, learning_rate=1e-3, **kw) as gt:
            gt.train_online(iters)
            return gt.loss_history(), gt.n_graphs

    h_graph, n_graphs = run(use_graph=True, overlap=False)
    h_split, n_split = run(use_graph=True, split=True, overlap=False)     # simulate + forward/backward | update, in sequence
    h_ahead, n_ahead = run(use_graph=True)             # the default: simulate graph of batch i + 1 beside the training graph of i
    h_ahead3, n_ahead3 = run(use_graph=True, split=True)                  # pipelined: simulate (i + 1) || forward/backward | update (i)
    h_eager, _ = run(use_graph=False)
    assert len(h_graph) == iters and n_graphs >= 5 and n_split == 2 * n_graphs                 # several N buckets were hit
    # (the pipelined loop's direct form keeps TWO buffer sets per bucket, each with graphs of its own: a bucket that came up twice
    #  has both captured)
    assert 2 * n_graphs <= n_ahead <= 4 * n_graphs and n_ahead % 2 == 0 and 3 * n_graphs <= n_ahead3 <= 6 * n_graphs and n_ahead3 % 3 == 0
    assert np.allclose(h_graph, h_eager, rtol=1e-4, atol=1e-4), np.abs(np.array(h_graph) - np.array(h_eager)).max()
    assert np.allclose(h_split, h_eager, rtol=1e-4, atol=1e-4)
    assert np.allclose(h_ahead, h_eager, rtol=1e-4, atol=1e-4)
    assert np.allclose(h_ahead3, h_eager, rtol=1e-4, atol=1e-4)
    assert np.mean(h_graph[-10:]) < np.mean(h_graph[:10]) - 0.5
    # the classic loop on the same batches: same prior rows, same simulator stream, exact N
    prior, step = DevicePrior("basic", seed=2023), {"i": 0}

    def generative_model(batch_size):
        i = step["i"]; step["i"] += 1
        n = shared_prior_N(2023, i)
        p = prior(batch_size, set_offset=i * batch_size)
        r = engine.simulate(engine.BASIC_DDM_DC, p, n, dt=0.01, max_steps=400.0, seed=2023, set_offset=i * batch_size,
                            fast=True, want_summary=False)
        return {"prior_draws": p, "sim_data": r["trials"], "sim_non_batchable_context": n}

    torch.manual_seed(0)
    am = AmortizedPosterior(InvertibleNetwork(num_params=5), InvariantNetwork())
    tr = Trainer(am, generative_model, basic_ddm_dc.configurator, checkpoint_path=None, learning_rate=1e-3)
    h_classic = tr.train_online(epochs=1, iterations_per_epoch=iters, batch_size=B, save_checkpoint=False, prefetch=False)
    # (the two differ by the summation order of the pooled mean -- padded to the bucket top and masked here, exact N there --
    # and the first iterations' large steps amplify that round-off: 1e-6 at iteration 2, ~5e-3 by iteration 20)
    d = np.abs(np.array(h_graph[:20]) - np.array(h_classic[:20]))
    assert d[:10].max() < 2e-3 and d.max() < 2e-2, d


def test_graph_trainer_experience_replay_equals_eager_and_the_classic_loop():
    """The reference trains with trainer.train_experience_replay (basic_ddm_dc.py:199-202): a fresh batch per iteration into a
    buffer, the step on a stored batch drawn at random.  GraphTrainer's form (simulate graph | buffer | training graph of the
    DRAWN batch's n_trials bucket) equals the same iteration run eagerly to 1e-4, makes the same draws as the classic
    Trainer's loop (same buffer policy, same generator) and tracks its loss curve, and the loss goes down."""
    import torch
    from bayesflow_nddms_amd import basic_ddm_dc, engine
    from bayesflow_nddms_amd.amortizer import AmortizedPosterior, InvariantNetwork, InvertibleNetwork, Trainer
    from bayesflow_nddms_amd.distributed import shared_prior_N
    from bayesflow_nddms_amd.graph_trainer import GraphTrainer
    from bayesflow_nddms_amd.priors import DevicePrior
    iters, B, cap = 40, 32, 8

    def run(use_graph):
        torch.manual_seed(0)
        am = AmortizedPosterior(InvertibleNetwork(num_params=5), InvariantNetwork())
        with GraphTrainer(am, batch_size=B, total_steps=iters, seed=2023, learning_rate=1e-3, use_graph=use_graph) as gt:
            gt.train_experience_replay(iters, capacity_in_batches=cap)
            return gt.loss_history(), gt.n_graphs

    h_graph, n_graphs = run(True)
    h_eager, _ = run(False)
    assert len(h_graph) == iters and n_graphs >= 6
    assert np.allclose(h_graph, h_eager, rtol=1e-4, atol=1e-4), np.abs(np.array(h_graph) - np.array(h_eager)).max()
    assert np.mean(h_graph[-10:]) < np.mean(h_graph[:10]) - 0.5
    prior, step = DevicePrior("basic", seed=2023), {"i": 0}

    def generative_model(batch_size):
        i = step["i"]; step["i"] += 1
        n = shared_prior_N(2023, i)
        p = prior(batch_size, set_offset=i * batch_size)
        r = engine.simulate(engine.BASIC_DDM_DC, p, n, dt=0.01, max_steps=400.0, seed=2023, set_offset=i * batch_size,
                            fast=True, want_summary=False)
        return {"prior_draws": p, "sim_data": r["trials"], "sim_non_batchable_context": n}

    torch.manual_seed(0)
    am = AmortizedPosterior(InvertibleNetwork(num_params=5), InvariantNetwork())
    tr = Trainer(am, generative_model, basic_ddm_dc.configurator, checkpoint_path=None, learning_rate=1e-3)
    res = tr.train_experience_replay(epochs=1, iterations_per_epoch=iters, batch_size=B, capacity_in_batches=cap,
                                     save_checkpoint=False, prefetch=False)
    h_classic = res["train_losses"]
    # (the padded-and-masked batch and the unpadded one differ in the last bits of every pooled mean, and the first steps of
    # training amplify that: the curves stay within 1 % over the first 15 iterations)
    assert np.allclose(h_graph[:15], h_classic[:15], rtol=1e-2, atol=1e-2), np.abs(np.array(h_graph[:20]) - np.array(h_classic[:20]))


def test_graph_trainer_checkpoint_resume_reproduces_the_uninterrupted_run(tmp_path):
    """A checkpoint carries the weights, Adam's state and the POSITION of the run (iteration -> batch-shared N, device-side set
    offset, schedule, replay buffer and its generator): 25 iterations, save, a NEW trainer loads and runs 25 more == 50
    iterations straight, loss for loss."""
    import torch
    from bayesflow_nddms_amd.amortizer import AmortizedPosterior, InvariantNetwork, InvertibleNetwork
    from bayesflow_nddms_amd.graph_trainer import GraphTrainer

    def make():
        torch.manual_seed(0)
        return GraphTrainer(AmortizedPosterior(InvertibleNetwork(num_params=5), InvariantNetwork()), batch_size=32,
                            total_steps=50, seed=2023, learning_rate=1e-3)

    with make() as gt:
        gt.train_experience_replay(50, capacity_in_batches=8)
        straight = gt.loss_history()
    with make() as gt:
        gt.train_experience_replay(25, capacity_in_batches=8)
        gt.save_checkpoint(str(tmp_path / "gt.pt"))
    torch.manual_seed(123)                                   # other initial weights: everything must come from the file
    with GraphTrainer(AmortizedPosterior(InvertibleNetwork(num_params=5), InvariantNetwork()), batch_size=32, total_steps=50,
                      seed=2023, learning_rate=1e-3) as gt:
        gt.load_checkpoint(str(tmp_path / "gt.pt"))
        gt.train_experience_replay(25, capacity_in_batches=8)
        resumed = gt.loss_history()
    assert len(resumed) == 50 and np.allclose(resumed, straight, rtol=1e-5, atol=1e-5), np.abs(np.array(resumed) - np.array(straight)).max()


def test_graph_trainer_past_total_steps_holds_the_rate_and_keeps_every_loss():
    """`total_steps` is the cosine schedule's length and the size of the device-side loss ring -- not a limit of the run: past
    it the learning rate stays at the schedule's final value (0: the weights stop moving; it used to climb back to lr0) and the
    history still holds EVERY loss (the ring is read out before it wraps; further losses used to pile into its last slot).
    Graph and eager forms agree over the whole run."""
    import torch
    from bayesflow_nddms_amd.amortizer import AmortizedPosterior, InvariantNetwork, InvertibleNetwork
    from bayesflow_nddms_amd.graph_trainer import GraphTrainer
    T, lr0 = 20, 1e-3

    def run(use_graph, calls):
        torch.manual_seed(0)
        am = AmortizedPosterior(InvertibleNetwork(num_params=5), InvariantNetwork())
        with GraphTrainer(am, batch_size=32, total_steps=T, seed=2023, learning_rate=lr0, use_graph=use_graph) as gt:
            rates, weights = [], []
            for n in calls:
                gt.train_online(n)
                torch.cuda.synchronize()
                rates.append(float(gt.lr_t))
                weights.append(gt.flat_p.clone())
            return gt.loss_history(), rates, weights

    with pytest.warns(RuntimeWarning, match="past total_steps"):                             # said once, not an error
        h, rates, w = run(True, (T, 15, 2 * T + 7))
    assert len(h) == 3 * T + 22 and np.all(np.isfinite(h))
    assert 0 < rates[0] < 0.02 * lr0 and rates[1] == 0.0 and rates[2] == 0.0, rates      # (lr_t: the rate the LAST step used)
    assert torch.equal(w[0], w[1]) and torch.equal(w[1], w[2])                            # rate 0: nothing moves after step T
    with pytest.warns(RuntimeWarning, match="past total_steps"):
        h_eager, rates_e, _ = run(False, (T, 15, 2 * T + 7))
    assert rates_e[1] == 0.0 and np.allclose(h, h_eager, rtol=1e-4, atol=1e-4), np.abs(np.array(h) - np.array(h_eager)).max()
    h_short, _, _ = run(True, (T,))
    assert np.allclose(h[:T], h_short, rtol=1e-6, atol=1e-6)


def _run_train_ranks(tmp_path, world, backend, iters):
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "tests", "train_rank_worker.py"), str(tmp_path), backend, str(iters)],
                                      env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        out, err = p.communicate(timeout=1500)
        assert p.returncode == 0, err[-3000:]
    return [json.load(open(os.path.join(str(tmp_path), f"rank{r}.json"))) for r in range(world)]


def test_two_ranks_pipelined_feed_equals_the_sequential_loop_and_replicas_stay_identical(tmp_path):
    """BASELINE configs[4] on two fresh ranks (gloo, both on cuda:0), each of which builds its amortizer from ANOTHER seed:
    the trainer broadcasts rank 0's weights, permutations and optimizer state, so both ranks start -- and, in `gather`
    (replicated step) as in `ddp` (all-reduced gradients), stay -- bit-identical.  The PIPELINED loop (simulate + all-gather
    of batch i + 1 beside the training step of batch i) equals the sequential graph loop and the eager iteration loss for loss,
    and leaves the random stream where they leave it."""
    iters = 24
    ranks = _run_train_ranks(tmp_path, 2, "gloo", iters)
    for parallel in ("gather", "ddp"):
        for form in ("pipelined", "sequential", "eager"):
            a, b = ranks[0][f"{parallel}/{form}"], ranks[1][f"{parallel}/{form}"]
            assert a["w0"] == b["w0"] and a["perm0"] == b["perm0"], (parallel, form, "replicas differ at the start")
            assert a["w1"] == b["w1"], (parallel, form, "replicas drifted apart")
            assert a["minibatch"] == (64 if parallel == "gather" else 32)
            assert a["offset"] == iters * 64 and b["offset"] == iters * 64 + 32          # the stream's position after the run
            if parallel == "gather":
                assert a["loss"] == b["loss"]                                             # the replicated step: the same numbers
            assert len(a["loss"]) == iters and np.all(np.isfinite(a["loss"]))
        ref = np.array(ranks[0][f"{parallel}/eager"]["loss"])
        for form in ("pipelined", "sequential"):
            h = np.array(ranks[0][f"{parallel}/{form}"]["loss"])
            assert np.allclose(h, ref, rtol=1e-4, atol=1e-4), (parallel, form, np.abs(h - ref).max())
        # (gather: simulate | forward/backward + update either way; ddp: simulate | forward/backward | update against simulate + forward/backward | update)
        assert ranks[0][f"{parallel}/pipelined"]["graphs"] >= ranks[0][f"{parallel}/sequential"]["graphs"] > 0
    assert np.mean(ranks[0]["gather/pipelined"]["loss"][-6:]) < np.mean(ranks[0]["gather/pipelined"]["loss"][:6])


def test_pipelined_feed_over_rccl_at_world_1_equals_the_sequential_loop(tmp_path):
    """The same three forms with the collectives on RCCL (world 1 on the one GPU): all-gather and all-reduce on the communication
    stream beside the simulate stream's graph -- loss for loss the sequential loop."""
    iters = 24
    (r,) = _run_train_ranks(tmp_path, 1, "nccl", iters)
    for parallel in ("gather", "ddp"):
        ref = np.array(r[f"{parallel}/eager"]["loss"])
        for form in ("pipelined", "sequential"):
            h = np.array(r[f"{parallel}/{form}"]["loss"])
            assert len(h) == iters and np.allclose(h, ref, rtol=1e-4, atol=1e-4), (parallel, form, np.abs(h - ref).max())
            assert r[f"{parallel}/{form}"]["offset"] == iters * 32


def test_classic_trainer_call_shape_runs_as_graph_replays(tmp_path):
    """The reference's call (basic_ddm_dc.py:172-176, 199-207) -- Trainer(amortizer, generative_model, configurator, checkpoint_path),
    train_experience_replay(epochs, iterations_per_epoch, batch_size, validation_sims), load_pretrained_network -- with
    graph=True: the same return shapes, every loss kept, validation loss and checkpoint per epoch, the loss goes down, a fresh
    Trainer loads what was saved; a generative model the graph loop cannot re-create is refused by name."""
    import os
    import torch
    from bayesflow_nddms_amd import basic_ddm_dc, single_trial_alpha_not_scaled as st
    from bayesflow_nddms_amd.amortizer import AmortizedPosterior, InvariantNetwork, InvertibleNetwork, Trainer
    from bayesflow_nddms_amd.simulation import GenerativeModel
    for mod, P in ((basic_ddm_dc, 5), (st, 7)):
        torch.manual_seed(0)
        gm = mod.make_generative_model(batched=True, device_prior=True, as_numpy=False)
        assert gm.graph_spec["model"] in ("basic", "single")
        am = AmortizedPosterior(InvertibleNetwork(num_params=P), InvariantNetwork())
        ck = str(tmp_path / f"ck{P}")
        tr = Trainer(am, gm, mod.configurator, checkpoint_path=ck, graph=True)
        val_sims = gm(64)
        out = tr.train_experience_replay(epochs=3, iterations_per_epoch=80, batch_size=32, validation_sims=val_sims)
        tl, vl = np.array(out["train_losses"]), out["val_losses"]
        assert tl.shape == (240,) and len(vl) == 3 and np.all(np.isfinite(tl)) and np.all(np.isfinite(vl))
        assert tl[-40:].mean() < tl[:40].mean() - 0.5 and vl[-1] < vl[0]
        assert os.path.exists(os.path.join(ck, "ckpt.pt"))
        h = tr.train_online(epochs=1, iterations_per_epoch=20, batch_size=32)                   # a second call: a run of its own
        assert len(h) == 260 and len(tr.loss_history) == 260
        am2 = AmortizedPosterior(InvertibleNetwork(num_params=P), InvariantNetwork())
        tr2 = Trainer(am2, gm, mod.configurator, checkpoint_path=ck)
        assert tr2.load_pretrained_network() and len(tr2.loss_history) == 260
        with torch.no_grad():
            conf = mod.configurator(val_sims)
            a, b = float(am.compute_loss(conf)), float(am2.compute_loss(conf))
        assert abs(a - b) < 1e-4 * (1 + abs(a)), (a, b)
    plain = GenerativeModel(basic_ddm_dc.draw_prior, lambda p, n: basic_ddm_dc.simulate_trials(p, 50), simulator_is_batched=False, skip_test=True)
    with pytest.raises(ValueError, match="graph_spec"):
        Trainer(AmortizedPosterior(InvertibleNetwork(num_params=5), InvariantNetwork()), plain, basic_ddm_dc.configurator,
                graph=True).train_online(1, 2, 4)


def test_trainer_graph_calls_continue_the_stream_and_a_resumed_run_does_too(tmp_path):
    """Trainer(graph=True): every train_* call builds a GraphTrainer of its own (schedule, fresh Adam) but CONTINUES the run's random
    stream -- parameter sets, batch-shared N, the experience-replay buffer and its generator -- and so does a run resumed with
    load_pretrained_network (the position travels in ckpt.pt).  Learning rate 0, so a loss depends on its batch alone: a second call
    that replayed the first call's batches would repeat its losses.  Training's parameter rows start at TRAIN_OFFSET_BASE of the
    seed's index space; the generative model's own draws (validation sims, recovery data sets) count from 0 and never meet them."""
    import torch
    from bayesflow_nddms_amd import basic_ddm_dc
    from bayesflow_nddms_amd.amortizer import AmortizedPosterior, InvariantNetwork, InvertibleNetwork, Trainer
    from bayesflow_nddms_amd.graph_trainer import TRAIN_OFFSET_BASE
    from bayesflow_nddms_amd.priors import DevicePrior

    def fresh(ck):
        torch.manual_seed(0)
        gm = basic_ddm_dc.make_generative_model(batched=True, device_prior=True, as_numpy=False)
        am = AmortizedPosterior(InvertibleNetwork(num_params=5), InvariantNetwork())
        return gm, Trainer(am, gm, basic_ddm_dc.configurator, checkpoint_path=ck, learning_rate=0.0, graph=True)

    for replay in (False, True):
        call = (lambda tr, n: tr.train_experience_replay(1, n, 32, capacity_in_batches=6)) if replay else (lambda tr, n: tr.train_online(1, n, 32))
        _, a = fresh(str(tmp_path / f"a{replay}"))
        call(a, 12); call(a, 12)
        ha = np.array(a.loss_history)
        assert ha.shape == (24,) and np.all(np.isfinite(ha))
        # (experience replay may draw a stored batch of the first call again -- the buffer continues too -- but never the same sequence,
        #  and batches the first call never saw must turn up)
        fresh_losses = set(np.round(ha[12:], 5)) - set(np.round(ha[:12], 5))
        assert len(fresh_losses) >= (4 if replay else 12), "the second call replayed the first call's batches"
        assert a._graph_pos["offset"] == TRAIN_OFFSET_BASE + 24 * 32 and a._graph_pos["n_key"] == 24
        # one straight call of 24 iterations sees the same 24 batches (and the same replay draws) as the two calls of 12
        _, s_ = fresh(None)
        call(s_, 24)
        assert np.allclose(np.array(s_.loss_history), ha, rtol=1e-5, atol=1e-5)
        # resumed: 12 iterations, checkpoint; a NEW trainer loads it and goes on -> the straight run's second half
        ck = str(tmp_path / f"b{replay}")
        _, b = fresh(ck)
        call(b, 12)
        gm_c, c = fresh(ck)
        assert c.load_pretrained_network() and c._graph_pos["offset"] == TRAIN_OFFSET_BASE + 12 * 32
        call(c, 12)
        hc = np.array(c.loss_history)
        assert hc.shape == (24,) and np.allclose(hc, ha, rtol=1e-5, atol=1e-5), np.abs(hc - ha).max()
    # evaluation draws: rows 0.. of the seed; training draws: rows TRAIN_OFFSET_BASE.. -- another part of the index space
    ev = gm_c(64)["prior_draws"]
    first_training_rows = DevicePrior("basic", seed=2023)(64, set_offset=TRAIN_OFFSET_BASE)
    assert torch.equal(ev, DevicePrior("basic", seed=2023)(64, set_offset=gm_c.prior.prior.state.offset - 64))
    assert not torch.isclose(ev[:, None, 0], first_training_rows[None, :, 0]).any()


def test_trainer_graph_interrupted_call_is_continued_with_its_adam_state_and_schedule(tmp_path):
    """Trainer(graph=True) at a learning rate > 0: a call of 3 epochs that dies after the checkpoint of its second epoch, then a NEW
    Trainer that loads ckpt.pt and makes the same call.  ckpt.pt carries the running call's Adam moments, step counters, learning rate
    and iteration beside the stream position, so the new call CONTINUES the interrupted one -- only the third epoch runs -- and the
    two together equal the uninterrupted run: same losses, same final weights.  (With the moments dropped -- the file as round 5
    wrote it -- the resumed epoch starts from a fresh Adam at the top of a new cosine schedule and its losses differ.)  The file holds
    plain data only and is read with torch.load(weights_only=True)."""
    import os
    import torch
    from bayesflow_nddms_amd import basic_ddm_dc
    from bayesflow_nddms_amd.amortizer import AmortizedPosterior, InvariantNetwork, InvertibleNetwork, Trainer

    def fresh(ck):
        torch.manual_seed(0)
        gm = basic_ddm_dc.make_generative_model(batched=True, device_prior=True, as_numpy=False)
        am = AmortizedPosterior(InvertibleNetwork(num_params=5), InvariantNetwork())
        return am, Trainer(am, gm, basic_ddm_dc.configurator, checkpoint_path=ck, learning_rate=1e-3, graph=True)

    call = lambda tr: tr.train_experience_replay(epochs=3, iterations_per_epoch=25, batch_size=32, capacity_in_batches=6)
    am_a, a = fresh(str(tmp_path / "straight"))
    call(a)
    ha = np.array(a.loss_history)
    assert ha.shape == (75,) and np.all(np.isfinite(ha))
    ck = str(tmp_path / "interrupted")
    am_b, b = fresh(ck)
    saves, real_save = [], b.save_checkpoint

    def dying_save(*args, **kw):
        real_save(*args, **kw)
        saves.append(1)
        if len(saves) == 2:
            raise KeyboardInterrupt("the job is killed after the checkpoint of epoch 2")

    b.save_checkpoint = dying_save
    with pytest.raises(KeyboardInterrupt):
        call(b)
    state = torch.load(os.path.join(ck, "ckpt.pt"), weights_only=True)             # plain data: the safe unpickler reads it
    assert state["graph_optimizer"]["iteration"] == 50 and state["graph_optimizer"]["total_steps"] == 75
    assert float(state["graph_optimizer"]["optimizer"]["exp_avg"].abs().sum()) > 0
    am_c, c = fresh(ck)
    assert c.load_pretrained_network() and c._graph_resume is not None
    call(c)
    hc = np.array(c.loss_history)
    assert hc.shape == (75,) and np.allclose(hc, ha, rtol=1e-5, atol=1e-5), np.abs(hc - ha).max()
    for (k, va), vc in zip(am_a.state_dict().items(), am_c.state_dict().values()):
        assert torch.allclose(va, vc, rtol=1e-5, atol=1e-6), k
    # the completed call leaves nothing to continue: a load now starts a run of its own (fresh Adam, its own schedule)
    am_d, d = fresh(ck)
    assert d.load_pretrained_network() and d._graph_resume is None and len(d.loss_history) == 75
    # and a call of ANOTHER shape after an interrupted one is a run of its own too
    am_e, e = fresh(ck)
    e._graph_resume = dict(state["graph_optimizer"])
    e.train_online(epochs=1, iterations_per_epoch=10, batch_size=32)
    assert len(e.loss_history) == 10 and np.all(np.isfinite(e.loss_history))


def test_pipelined_loops_feed_the_training_graph_the_right_batch_soak():
    """Integrity soak of the pipelined loop (tests/train_soak_worker.py): learning rate 0, so the loss of iteration k depends on
    batch k alone -- 1200 online + 400 experience-replay iterations at dt=.001 (a 200 us simulate launch beside every training
    graph) in the one-rank form (both models), with the all-gather (RCCL, world 1) and with the gradient all-reduce: the pipelined history
    equals the sequential loop's at EVERY iteration, and the random stream ends at the same position."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "train_soak_worker.py"), "1200", "400"], capture_output=True, text=True,
                       timeout=1200, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    for name in ("one rank", "gather", "ddp", "one rank, single-trial model"):
        o = out[name]
        assert o["iterations"] == 1600 and o["finite"] and o["offsets_equal"], (name, o)
        assert o["n_mismatch"] == 0, (name, o)
        assert o["spread"] > 0.05, (name, o)                 # (the batches DO differ: equal histories are not equal constants)


def test_graph_trainer_single_trial_model():
    """The same loop for single_trial_alpha_not_scaled (its script trains the same way, :193-230): 7 target parameters, data
    (choicert, z1); the graph-replayed history equals the eager one and the loss goes down."""
    import torch
    from bayesflow_nddms_amd.amortizer import AmortizedPosterior, InvariantNetwork, InvertibleNetwork
    from bayesflow_nddms_amd.graph_trainer import GraphTrainer

    def run(use_graph):
        torch.manual_seed(0)
        am = AmortizedPosterior(InvertibleNetwork(num_params=7), InvariantNetwork())
        with GraphTrainer(am, model="single", batch_size=32, total_steps=60, seed=2023, learning_rate=1e-3, use_graph=use_graph) as gt:
            gt.train_online(60)
            return gt.loss_history()

    h_graph, h_eager = run(True), run(False)
    assert np.allclose(h_graph, h_eager, rtol=1e-4, atol=1e-4)
    assert np.all(np.isfinite(h_graph)) and np.mean(h_graph[-10:]) < np.mean(h_graph[:10]) - 0.5


class _F64Yardstick:
    """The bound of the fused-kernel tests: an f64 evaluation of the SAME network is the reference, PyTorch's own f32 composition the
    yardstick.  Two f32 evaluations of one function differ from f64 by round-off of the same size but not of the same VALUE, and the
    error of a 5-element gradient of ONE random network is a noisy statistic (a ratio of two chi variables with 5 degrees of freedom);
    so errors are pooled: per GROUP of tensors (an output of its own; all weight-matrix / all bias / all ActNorm gradients; each
    tensor's error divided by its own f64 RMS magnitude, so that layers and cases of different scale pool sensibly) over ALL cases of
    the test's battery,

        rms(fused - f64) <= 2 * rms(pytorch f32 - f64) + 32 * 2^-24        at finish()

    and within each single case no group may be off by more than `outlier` x (+ the same few ulps): add() asserts it."""

    def __init__(self, factor=2.0, outlier=8.0, ulps=32.0):
        self.factor, self.outlier, self.tiny, self.pool = factor, outlier, ulps * 2.0 ** -24, {}

    def add(self, label, case, fused, plain, exact, groups):
        acc = {}
        for a, b, e, grp in zip(fused, plain, exact, groups):
            mag = float(e.pow(2).mean().sqrt()) + 1e-30
            fa, fb, n = acc.get(grp, (0.0, 0.0, 0))
            acc[grp] = (fa + float(((a.double() - e) / mag).pow(2).sum()), fb + float(((b.double() - e) / mag).pow(2).sum()), n + e.numel())
        for grp, (fa, fb, n) in acc.items():
            ra, rb = (fa / n) ** 0.5, (fb / n) ** 0.5
            assert ra <= self.outlier * rb + self.tiny, (label, case, grp, "relative rms fused - f64", ra, "pytorch f32 - f64", rb)
            pa, pb, pn = self.pool.get((label, grp), (0.0, 0.0, 0))
            self.pool[(label, grp)] = (pa + fa, pb + fb, pn + n)

    def finish(self):
        """-> {label: group: ratio of the pooled rms errors fused / pytorch}; asserts the 2 x bound on every group."""
        out = {}
        for (label, grp), (fa, fb, n) in self.pool.items():
            ra, rb = (fa / n) ** 0.5, (fb / n) ** 0.5
            out[f"{label}: {grp}"] = round(ra / max(rb, 1e-30), 3)
            assert ra <= self.factor * rb + self.tiny, (label, grp, "pooled relative rms fused - f64", ra, "pytorch f32 - f64", rb)
        return out


def _flow_groups(net):
    """Group names of [theta, cond] + net.parameters() gradients: weight matrices, biases and the ActNorm's pooled over the layers."""
    return ["d theta", "d cond"] + [("d ActNorm" if n.startswith("an_") else "d weights" if n.endswith("weight") else "d biases")
                                    for n, _ in net.named_parameters()]


def test_fused_flow_equals_the_pytorch_path():
    """csrc/train_kernels.hip: the whole conditional flow (per layer ActNorm, permutation, two coupling half-layers of three
    Linear layers, two ELUs, soft clamp, exp, multiply-add) as ONE kernel forward and ONE backward behind one autograd node.
    Output, log|det| and EVERY gradient (theta, condition, ActNorm scales and biases, all twelve parameter tensors of every
    layer) are held to a FLOAT64 evaluation of the same network, with PyTorch's float32 composition as the yardstick (_F64Yardstick:
    pooled relative rms error <= 2 x PyTorch's + 32 ulp, no single case off by more than 8 x), for one and six layers and row counts
    that are not a multiple of the kernels' 8- and 32-row tiles and span several of them, on flows made ill-conditioned on purpose;
    and the fused forward's z goes back through the FLOAT64 inverse (the round trip).  Shapes the kernels do not cover fall back.
    Measured on the MI355X, pooled rms error fused / pytorch (round 5, after three accuracy fixes the yardstick found: the
    weight-gradient kernel's running sums in f64 -- bias gradients were 3 .. 12 x PyTorch's error, a sequential f32 sum over the rows
    against a tree --, the clamp's derivative as an exact difference, and ELU' from the saved PRE-activation instead of output + 1):
    flow form z 1.10, log|det| 0.88, d theta 1.45, d cond 0.99, d weights 1.11, d biases 1.22, d ActNorm 1.42; loss form 1.51 / 1.80 /
    1.40 / 1.79 / 2.01 / 1.93 (loss, d theta, d cond, d weights, d biases, d ActNorm -- its gradient seeds are the forward's own z,
    which weights the ill-conditioned rows); round trip 1.41."""
    import copy
    import torch
    from bayesflow_nddms_amd import _train_lib
    from bayesflow_nddms_amd.amortizer import InvertibleNetwork
    assert _train_lib.lib() is not None, "libnddm_train.so did not build / load"
    torch.manual_seed(3)
    yard = _F64Yardstick()
    for layers, R, D in ((1, 32, 5), (1, 5, 5), (2, 77, 5), (6, 256, 5), (6, 32, 8), (3, 40, 2), (6, 32, 7), (2, 19, 3), (2, 33, 4), (2, 16, 6), (1, 1, 5), (2, 4001, 5)):
        net = InvertibleNetwork(num_params=D, num_coupling_layers=layers, seed=layers).cuda()
        with torch.no_grad():
            for p in net.parameters():                   # larger weights than the initialisation, ActNorms away from identity
                p.mul_(3.0 if layers == 1 else 1.5)
            for p in list(net.an_scale) + list(net.an_bias):
                p.copy_(0.3 * torch.randn_like(p))
        net64 = copy.deepcopy(net).double()              # the same network evaluated in float64: the reference of every comparison below
        theta = torch.randn(R, D, device="cuda", requires_grad=True)
        cond = torch.randn(R, 11, device="cuda", requires_grad=True)
        wz, wl = torch.randn(R, D, device="cuda"), torch.randn(R, device="cuda")
        theta64, cond64 = theta.detach().double().requires_grad_(True), cond.detach().double().requires_grad_(True)

        def flow(n, th, cd):
            z, ld = n(th, cd)
            g = torch.autograd.grad((z * wz.to(z.dtype)).sum() + (ld * wl.to(z.dtype)).sum(), [th, cd] + list(n.parameters()))
            return [z.detach(), ld.detach()] + [t.detach() for t in g]

        def nll(n, th, cd):                              # the loss form: gradients of z and log|det| derived inside the kernel
            loss = n.nll(th, cd)
            return [loss.detach()] + [t.detach() for t in torch.autograd.grad(3.0 * loss, [th, cd] + list(n.parameters()))]

        res, nl = {}, {}
        for fused in (True, False):
            net.fused = fused
            assert (net._fused_lib(theta, cond) is not None) == fused
            res[fused], nl[fused] = flow(net, theta, cond), nll(net, theta, cond)
        assert net64._fused_lib(theta64, cond64) is None
        grp = _flow_groups(net)
        yard.add("flow", (layers, R, D), res[True], res[False], flow(net64, theta64, cond64), ["z", "log|det|"] + grp)
        yard.add("nll", (layers, R, D), nl[True], nl[False], nll(net64, theta64, cond64), ["loss"] + grp)
        # and the inverse undoes the fused forward: BOTH f32 forwards' z go through the f64 inverse (exact to 1e-16, so what comes back
        # is the forward's own round-off, amplified by the inverse of a flow made ill-conditioned on purpose -- weights x 1.5 .. 3)
        with torch.no_grad():
            back = [[net64.inverse(res[f][0].double(), cond64.detach())] for f in (True, False)]
        yard.add("round trip", (layers, R, D), back[0], back[1], [theta64.detach()], ["theta"])
    print("fused flow, pooled rms error fused / pytorch-f32 against f64:", yard.finish())
    small = InvertibleNetwork(num_params=5, hidden=32).cuda()          # hidden width 32: not covered -> the PyTorch path, silently
    z, ld = small(torch.randn(8, 5, device="cuda"), torch.randn(8, 11, device="cuda"))
    assert z.shape == (8, 5) and ld.shape == (8,)


def test_fused_deepset_equals_the_pytorch_path():
    """csrc/train_deepset.hip: every per-trial MLP of the summary network as one f32-MFMA kernel each way, with the masked
    per-set pooling, the pooled context of the equivariant halves and the weight-gradient partial sums fused in.  The summary
    and EVERY parameter gradient are held to a float64 evaluation of the same network with PyTorch's float32 composition as the
    yardstick (_F64Yardstick; measured pooled rms error fused / pytorch on the MI355X: summary 0.71, d weights 0.75, d biases 1.13;
    with direct conditions 0.87 / 0.86 / 1.11 -- the kernels are as accurate as PyTorch or better): with and without padding mask, set sizes
    that are not a multiple of the 64-row tile or of the 128 rows of a workgroup, 0 / 1 / 2 equivariant blocks; inputs the
    kernels do not cover fall back."""
    import copy
    import torch
    from bayesflow_nddms_amd import _train_lib
    from bayesflow_nddms_amd.amortizer import InvariantNetwork
    assert _train_lib.lib() is not None, "libnddm_train.so did not build / load"
    torch.manual_seed(5)
    yard = _F64Yardstick()
    for blocks, B, N, n_real in ((2, 32, 300, 237), (2, 3, 60, None), (1, 5, 131, 131), (2, 4, 129, 64), (0, 7, 200, 77), (2, 2, 1, None)):
        net = InvariantNetwork(num_equiv=blocks).cuda()
        with torch.no_grad():
            for p in net.parameters():
                if p.dim() == 1:
                    p.copy_(0.1 * torch.randn_like(p))                  # non-zero biases
        net64 = copy.deepcopy(net).double()                             # the reference: the same network in float64
        x = torch.stack([0.3 + torch.rand(B, N, device="cuda") * 2.0, (torch.rand(B, N, device="cuda") < 0.7).float()], dim=-1)
        mask = inv_n = mask64 = inv_n64 = None
        if n_real is not None:
            mask = (torch.arange(N, device="cuda") < n_real).float().view(1, N, 1)
            inv_n = torch.tensor(1.0 / n_real, device="cuda")
            mask64, inv_n64 = mask.double(), torch.tensor(1.0 / n_real, device="cuda", dtype=torch.float64)
        w = torch.randn(B, net.summary_dim, device="cuda")

        def summary(n, xx, *a, wt=w, **kw):
            out = n(xx, *a, **kw)
            return [out.detach()] + [t.detach() for t in torch.autograd.grad((out * wt.to(out.dtype)).sum(), list(n.parameters()))]

        res = {}
        for fused in (True, False):
            net.fused = fused
            assert (net._fused_lib(x) is not None) == fused
            res[fused] = summary(net, x, mask, inv_n)
        grp = ["summary"] + [("d weights" if n.endswith("weight") else "d biases") for n, _ in net.named_parameters()]
        yard.add("summary", (blocks, B, N, n_real), res[True], res[False], summary(net64, x.double(), mask64, inv_n64), grp)
        if n_real is not None:                               # the count form of the mask (a device scalar: what the graph trainer passes)
            nv = torch.tensor([float(n_real)], device="cuda")
            for fused in (True, False):
                net.fused = fused
                out = net(x, n_valid=nv)
                g = torch.autograd.grad((out * w).sum(), list(net.parameters()))
                for a, b in zip([out.detach()] + list(g), res[fused]):
                    assert torch.equal(a, b) if fused else torch.allclose(a, b, rtol=1e-5, atol=1e-6)
        # the direct conditions appended by the kernels (no concatenation; the backward reads the condition's gradient with its stride):
        # a [B, 1] broadcast of one device scalar (what the graph trainer passes: log N) and a dense [B, 3]
        for direct in (torch.tensor([5.25], device="cuda").view(1, 1).expand(B, 1), torch.randn(B, 3, device="cuda")):
            wd = torch.randn(B, net.summary_dim + direct.shape[1], device="cuda")
            got = {}
            for fused in (True, False):
                net.fused = fused
                got[fused] = summary(net, x, mask, inv_n, wt=wd, direct=direct)
                assert got[fused][0].shape == (B, net.summary_dim + direct.shape[1]) and torch.equal(got[fused][0][:, net.summary_dim:], direct)
            yard.add("direct", (blocks, B, N, n_real), got[True], got[False], summary(net64, x.double(), mask64, inv_n64, wt=wd, direct=direct.double()), grp)
        if n_real is not None and n_real < N:                # padding is invisible: the unpadded batch gives the same summary
            net.fused = True
            assert torch.allclose(net(x[:, :n_real].contiguous()), res[True][0], rtol=1e-4, atol=1e-5)
    print("fused DeepSet, pooled rms error fused / pytorch-f32 against f64:", yard.finish())
    wide = InvariantNetwork(hidden=32).cuda()                # hidden width 32: not covered -> the PyTorch path, silently
    assert wide(torch.randn(4, 50, 2, device="cuda")).shape == (4, 10)
    # The kernels are the TRAINING path (their forward always keeps every hidden activation): inference (no_grad) and batches
    # beyond FUSED_MAX_ROWS trial rows take the PyTorch composition, and a data tensor that wants a gradient is not theirs
    net = InvariantNetwork().cuda()
    assert net._fused_lib(x) is not None
    with torch.no_grad():
        assert net._fused_lib(x) is None
        assert net(x).shape == (x.shape[0], net.summary_dim)
    big = torch.zeros(net.FUSED_MAX_ROWS // 512 + 1, 512, 2, device="cuda")
    assert net._fused_lib(big) is None and net._fused_lib(big[:-1]) is not None
    assert net._fused_lib(x.clone().requires_grad_(True)) is None


def test_flat_adam_step_equals_pytorch():
    """csrc/train_update.hip: global-norm clipping + cosine learning rate + Adam on flat buffers (three launches) against
    torch.nn.utils.clip_grad_norm_ + torch.optim.Adam + the same schedule, over several steps with gradients large enough to be
    clipped at first; the loss history, the reported rate and both counters move as the PyTorch form moves them."""
    import math
    import torch
    from bayesflow_nddms_amd import _train_lib
    L = _train_lib.lib()
    assert L is not None
    torch.manual_seed(11)
    n, T, lr0, clip, scale = 4096 + 8, 50, 2e-3, 5.0, 0.5
    p0 = torch.randn(n, device="cuda")
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=lr0)
    p, m, v = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    g = torch.zeros(n + 1, device="cuda")
    partial = torch.zeros(256 + 4, device="cuda")              # (+ the ticket word of the update kernel: zero between launches)
    step_i, step_f = torch.zeros(1, dtype=torch.int64, device="cuda"), torch.zeros(1, device="cuda")
    lr_out, loss_buf = torch.zeros((), device="cuda"), torch.zeros(4, device="cuda")
    for it in range(6):
        grad = torch.randn(n, device="cuda") * (3.0 if it < 3 else 0.01)
        g[:n].copy_(grad)
        g[n] = float(it + 1)
        lr = 0.5 * lr0 * (1.0 + math.cos(it * math.pi / T))
        for grp in opt.param_groups:
            grp["lr"] = lr
        ref.grad = grad * scale
        torch.nn.utils.clip_grad_norm_([ref], clip)
        opt.step()
        rc = L.nddm_train_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, partial.data_ptr(), scale, clip, lr0, float(T),
                                    0.9, 0.999, 1e-8, step_i.data_ptr(), step_f.data_ptr(), lr_out.data_ptr(), loss_buf.data_ptr(), 4,
                                    g[n:].data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        assert torch.allclose(p, ref.detach(), rtol=1e-5, atol=1e-6), (it, float((p - ref.detach()).abs().max()))
        assert abs(float(lr_out) - lr) < 1e-9 and int(step_i) == it + 1 and float(step_f) == it + 1
    assert loss_buf.tolist() == [2.5, 3.0, 1.5, 2.0]          # scaled losses in a RING of 4: steps 4 and 5 wrapped (the host drains before)
    # past the schedule's length the rate HOLDS its final value (Keras' CosineDecay, BayesFlow's default) -- it does not climb back
    step_i.fill_(T - 1); step_f.fill_(float(T - 1))
    seen = []
    for it in range(T - 1, T + 40, 13):
        step_i.fill_(it); step_f.fill_(float(it))
        assert L.nddm_train_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, partial.data_ptr(), scale, clip, lr0, float(T),
                                      0.9, 0.999, 1e-8, step_i.data_ptr(), step_f.data_ptr(), lr_out.data_ptr(), loss_buf.data_ptr(), 4,
                                      g[n:].data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
        seen.append(float(lr_out))
    assert seen[0] > 0 and seen[0] < 1e-5 * lr0 * 1e3 and all(abs(v) < 1e-9 for v in seen[1:]), seen
    assert L.nddm_train_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n + 1, partial.data_ptr(), 1.0, clip, lr0, 1.0,
                                  0.9, 0.999, 1e-8, step_i.data_ptr(), step_f.data_ptr(), lr_out.data_ptr(), loss_buf.data_ptr(), 4,
                                  g[n:].data_ptr(), None) == 1       # a length that is not a multiple of 4 is refused


def test_graph_trainer_without_the_training_library_follows_the_same_curve():
    """NDDM_NO_FUSED_COUPLING=1: the trainer runs on the pure-PyTorch networks and PyTorch's fused Adam (on the same flat
    parameter layout) -- the fallback where libnddm_train.so cannot be built -- and its loss curve follows the kernels' curve:
    an end-to-end cross-check of every hand-written training kernel against PyTorch over 12 optimizer steps."""
    import json
    import os
    import subprocess
    import sys
    code = """
import json, torch
from bayesflow_nddms_amd import _train_lib
from bayesflow_nddms_amd.amortizer import AmortizedPosterior, InvariantNetwork, InvertibleNetwork
from bayesflow_nddms_amd.graph_trainer import GraphTrainer
torch.manual_seed(0)
am = AmortizedPosterior(InvertibleNetwork(num_params=5), InvariantNetwork())
with GraphTrainer(am, batch_size=32, total_steps=12, seed=2023, learning_rate=1e-3) as gt:
    gt.train_online(12)
    print(json.dumps({"lib": _train_lib.lib() is not None, "loss": gt.loss_history()}))
"""
    out = {}
    for off in (False, True):
        env = dict(os.environ)
        env.pop("NDDM_NO_FUSED_COUPLING", None)
        if off:
            env["NDDM_NO_FUSED_COUPLING"] = "1"
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0, r.stderr[-2000:]
        out[off] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out[False]["lib"] and not out[True]["lib"]
    a, b = np.array(out[False]["loss"]), np.array(out[True]["loss"])
    assert len(a) == len(b) == 12 and np.all(np.isfinite(a)) and np.abs(a - b)[:6].max() < 2e-3 and np.abs(a - b).max() < 5e-2, (a, b)

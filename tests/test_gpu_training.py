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


def test_prefetched_online_training_sees_the_same_batches():
    """train_online(prefetch=True) simulates batch i+1 on a side stream while batch i is trained on.  Same seeds =>
    the same batches in the same order => the same loss curve as without prefetching, the same simulator stream state
    afterwards (nothing is simulated beyond the last iteration), and it is not slower."""
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
    assert secs[True] < secs[False] * 2.0       # (overlap helps; the bound only guards against a pathological stall)


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
    h_split, n_split = run(use_graph=True, split=True)
    h_ahead, n_ahead = run(use_graph=True)             # the default: simulate graph of batch i + 1 beside the training graph of i
    h_eager, _ = run(use_graph=False)
    assert len(h_graph) == iters and n_graphs >= 5 and n_split == 2 * n_graphs == n_ahead      # several N buckets were hit
    assert np.allclose(h_graph, h_eager, rtol=1e-4, atol=1e-4), np.abs(np.array(h_graph) - np.array(h_eager)).max()
    assert np.allclose(h_split, h_eager, rtol=1e-4, atol=1e-4)
    assert np.allclose(h_ahead, h_eager, rtol=1e-4, atol=1e-4)
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


def test_fused_flow_equals_the_pytorch_path():
    """csrc/train_kernels.hip: the whole conditional flow (per layer ActNorm, permutation, two coupling half-layers of three
    Linear layers, two ELUs, soft clamp, exp, multiply-add) as ONE kernel forward and ONE backward behind one autograd node.
    Output, log|det| and EVERY gradient (theta, condition, ActNorm scales and biases, all twelve parameter tensors of every
    layer) equal the PyTorch composition to float32 round-off, for one and six layers and row counts that are not a multiple
    of the kernels' 8- and 32-row tiles and span several of them; shapes the kernels do not cover fall back."""
    import torch
    from bayesflow_nddms_amd import _train_lib
    from bayesflow_nddms_amd.amortizer import InvertibleNetwork
    assert _train_lib.lib() is not None, "libnddm_train.so did not build / load"
    torch.manual_seed(3)
    for layers, R, D in ((1, 32, 5), (1, 5, 5), (2, 77, 5), (6, 256, 5), (6, 32, 8), (3, 40, 2)):
        net = InvertibleNetwork(num_params=D, num_coupling_layers=layers, seed=layers).cuda()
        with torch.no_grad():
            for p in net.parameters():                   # larger weights than the initialisation, ActNorms away from identity
                p.mul_(3.0 if layers == 1 else 1.5)
            for p in list(net.an_scale) + list(net.an_bias):
                p.copy_(0.3 * torch.randn_like(p))
        theta = torch.randn(R, D, device="cuda", requires_grad=True)
        cond = torch.randn(R, 11, device="cuda", requires_grad=True)
        wz, wl = torch.randn(R, D, device="cuda"), torch.randn(R, device="cuda")
        res = {}
        for fused in (True, False):
            net.fused = fused
            assert (net._fused_lib(theta, cond) is not None) == fused
            z, ld = net(theta, cond)
            g = torch.autograd.grad((z * wz).sum() + (ld * wl).sum(), [theta, cond] + list(net.parameters()))
            res[fused] = [z.detach(), ld.detach()] + [t.detach() for t in g]
        for k, (a, b) in enumerate(zip(res[True], res[False])):
            mag = float(b.abs().max()) + 1e-6
            assert float((a - b).abs().max()) <= 5e-5 * mag + 1e-6, (layers, R, D, k, a.shape, float((a - b).abs().max()), mag)
        nl = {}
        for fused in (True, False):                      # the loss form: gradients of z and log|det| derived inside the kernel
            net.fused = fused
            loss = net.nll(theta, cond)
            nl[fused] = [loss.detach()] + [t.detach() for t in torch.autograd.grad(3.0 * loss, [theta, cond] + list(net.parameters()))]
        for k, (a, b) in enumerate(zip(nl[True], nl[False])):
            mag = float(b.abs().max()) + 1e-6
            assert float((a - b).abs().max()) <= 5e-5 * mag + 1e-6, ("nll", layers, R, D, k, float((a - b).abs().max()), mag)
        x = net.inverse(res[True][0], cond.detach())     # and the (PyTorch) inverse undoes the fused forward
        assert torch.allclose(x, theta.detach(), atol=2e-3), float((x - theta).abs().max())
    small = InvertibleNetwork(num_params=5, hidden=32).cuda()          # hidden width 32: not covered -> the PyTorch path, silently
    z, ld = small(torch.randn(8, 5, device="cuda"), torch.randn(8, 11, device="cuda"))
    assert z.shape == (8, 5) and ld.shape == (8,)


def test_fused_deepset_equals_the_pytorch_path():
    """csrc/train_deepset.hip: every per-trial MLP of the summary network as one f32-MFMA kernel each way, with the masked
    per-set pooling, the pooled context of the equivariant halves and the weight-gradient partial sums fused in.  The summary
    and EVERY parameter gradient equal the PyTorch composition to float32 round-off: with and without padding mask, set sizes
    that are not a multiple of the 64-row tile or of the 128 rows of a workgroup, 0 / 1 / 2 equivariant blocks; inputs the
    kernels do not cover fall back."""
    import torch
    from bayesflow_nddms_amd import _train_lib
    from bayesflow_nddms_amd.amortizer import InvariantNetwork
    assert _train_lib.lib() is not None, "libnddm_train.so did not build / load"
    torch.manual_seed(5)
    for blocks, B, N, n_real in ((2, 32, 300, 237), (2, 3, 60, None), (1, 5, 131, 131), (2, 4, 129, 64), (0, 7, 200, 77), (2, 2, 1, None)):
        net = InvariantNetwork(num_equiv=blocks).cuda()
        with torch.no_grad():
            for p in net.parameters():
                if p.dim() == 1:
                    p.copy_(0.1 * torch.randn_like(p))                  # non-zero biases
        x = torch.stack([0.3 + torch.rand(B, N, device="cuda") * 2.0, (torch.rand(B, N, device="cuda") < 0.7).float()], dim=-1)
        mask = inv_n = None
        if n_real is not None:
            mask = (torch.arange(N, device="cuda") < n_real).float().view(1, N, 1)
            inv_n = torch.tensor(1.0 / n_real, device="cuda")
        w = torch.randn(B, net.summary_dim, device="cuda")
        res = {}
        for fused in (True, False):
            net.fused = fused
            assert (net._fused_lib(x) is not None) == fused
            out = net(x, mask, inv_n)
            g = torch.autograd.grad((out * w).sum(), list(net.parameters()))
            res[fused] = [out.detach()] + [t.detach() for t in g]
        for k, (a, b) in enumerate(zip(res[True], res[False])):
            mag = float(b.abs().max()) + 1e-6
            assert float((a - b).abs().max()) <= 1e-4 * mag + 1e-6, (blocks, B, N, n_real, k, a.shape, float((a - b).abs().max()), mag)
        if n_real is not None:                               # the count form of the mask (a device scalar: what the graph trainer passes)
            nv = torch.tensor([float(n_real)], device="cuda")
            for fused in (True, False):
                net.fused = fused
                out = net(x, n_valid=nv)
                g = torch.autograd.grad((out * w).sum(), list(net.parameters()))
                for a, b in zip([out.detach()] + list(g), res[fused]):
                    assert torch.equal(a, b) if fused else torch.allclose(a, b, rtol=1e-5, atol=1e-6)
        if n_real is not None and n_real < N:                # padding is invisible: the unpadded batch gives the same summary
            net.fused = True
            assert torch.allclose(net(x[:, :n_real].contiguous()), res[True][0], rtol=1e-4, atol=1e-5)
    wide = InvariantNetwork(hidden=32).cuda()                # hidden width 32: not covered -> the PyTorch path, silently
    assert wide(torch.randn(4, 50, 2, device="cuda")).shape == (4, 10)
    with torch.no_grad():                                    # inference (no autograd graph) runs the kernels too
        net = InvariantNetwork().cuda()
        a = net(x)
        net.fused = False
        assert torch.allclose(a, net(x), rtol=1e-4, atol=1e-5)


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
    partial = torch.zeros(256, device="cuda")
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
    assert loss_buf.tolist() == [0.5, 1.0, 1.5, 3.0]          # scaled losses; steps beyond the capacity land in the last slot
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

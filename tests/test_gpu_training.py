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

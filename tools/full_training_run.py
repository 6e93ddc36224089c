#!/usr/bin/env python3
"""The reference's WHOLE training run of basic_ddm_dc.py:199-202 -- trainer.train_experience_replay(epochs=500, batch_size=32,
iterations_per_epoch=1000): 500,000 iterations, 1.6e7 simulated data sets, 2.9e9 trials at dt=.01 / 400 (the job the reference gives a
30-hour SLURM slot: bayesflow_nddms.sh:6) -- on one MI355X with graph_trainer.GraphTrainer, followed by the recovery loop of :218-250 in
the reference's size (500 fresh data sets, posterior means against the true parameters; 2000 posterior draws each instead of 10000).
Prints the time and loss per 50 epochs.  `single`: the same for single_trial_alpha_not_scaled.py:284-287 (7 parameters, data (choicert, z1)).
usage: python tools/full_training_run.py [epochs=500] [basic|single] [file to save the trained amortizer's state_dict to]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesflow_nddms_amd import basic_ddm_dc                                                                    # noqa: E402
from bayesflow_nddms_amd.amortizer import AmortizedPosterior, InvariantNetwork, InvertibleNetwork, posterior_estimates   # noqa: E402
from bayesflow_nddms_amd.graph_trainer import GraphTrainer                                                      # noqa: E402


def main():
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    model = sys.argv[2] if len(sys.argv) > 2 else "basic"
    per_epoch, chunk = 1000, 50
    torch.manual_seed(0)
    am = AmortizedPosterior(InvertibleNetwork(num_params=5 if model == "basic" else 7), InvariantNetwork())
    t0 = time.time()
    with GraphTrainer(am, model=model, batch_size=32, total_steps=epochs * per_epoch, seed=2023) as gt:
        for e0 in range(0, epochs, chunk):
            n = min(chunk, epochs - e0) * per_epoch
            gt.train_experience_replay(n)
            torch.cuda.synchronize()
            h = gt.loss_history()
            print(f"epochs {e0 + 1:3d}-{e0 + n // per_epoch:3d}: {time.time() - t0:6.1f} s elapsed, {gt.iteration / (time.time() - t0):6.0f} it/s overall, "
                  f"loss over the last 1000 iterations {np.mean(h[-1000:]):8.3f}, rate {float(gt.lr_t):.2e}", flush=True)
        h = np.array(gt.loss_history())
    total = time.time() - t0
    print(f"{len(h)} iterations ({len(h) * 32:.3g} data sets) in {total:.1f} s = {len(h) / total:.0f} it/s; nan {int(np.isnan(h).sum())}; "
          f"loss first 1000 {h[:1000].mean():.3f}, last 1000 {h[-1000:].mean():.3f}", flush=True)
    np.random.seed(2023)
    if model == "basic":
        mod, names = basic_ddm_dc, "drift, boundary, beta, tau, dc"
    else:
        from bayesflow_nddms_amd import single_trial_alpha_not_scaled as mod
        names = "drift, mu_alpha, beta, ter, std_alpha, dc, sigma1"
    gm = mod.make_generative_model(batched=True, device_prior=True, as_numpy=False)
    t1 = time.time()
    true, means, meds = posterior_estimates(am, gm, mod.configurator, n_datasets=500, n_samples=2000)
    corr = lambda est, keep: np.round([np.corrcoef(true[keep, j], est[keep, j])[0, 1] for j in range(true.shape[1])], 3)
    everything = np.ones(len(true), dtype=bool)
    # A handful of wild posterior draws (the inverse of a sharply trained flow amplifies a tail draw of z) can carry one data set's
    # posterior MEAN -- and with it a Pearson correlation over 500 data sets -- far away while the posterior's bulk sits on the truth:
    # the medians, and the means without such data sets, are printed beside the means
    wild = (np.abs(means - meds) > 5.0 * (np.abs(meds) + 1.0)).any(axis=1)
    print(f"recovery over 500 fresh data sets ({time.time() - t1:.1f} s), correlation with the truth per parameter ({names}):\n"
          f"  posterior mean   {corr(means, everything)}\n  posterior median {corr(meds, everything)}\n"
          f"  data sets whose posterior mean is carried off by tail draws (|mean - median| > 5 (|median| + 1)): {int(wild.sum())} of 500; "
          f"posterior mean without them {corr(means, ~wild)}; non-finite means: {int((~np.isfinite(means)).any(axis=1).sum())}", flush=True)
    if len(sys.argv) > 3:
        torch.save(am.state_dict(), sys.argv[3])


if __name__ == "__main__":
    main()

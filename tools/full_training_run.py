#!/usr/bin/env python3
"""The reference's WHOLE training run of basic_ddm_dc.py:199-202 -- trainer.train_experience_replay(epochs=500, batch_size=32,
iterations_per_epoch=1000): 500,000 iterations, 1.6e7 simulated data sets, 2.9e9 trials at dt=.01 / 400 (the job the reference gives a
30-hour SLURM slot: bayesflow_nddms.sh:6) -- on one MI355X with graph_trainer.GraphTrainer, followed by the recovery loop of :211-241 in
the reference's size and on the reference's statistic: 500 fresh data sets, 10 000 posterior draws each, posterior MEANS against the true
parameters -> r2_score and Pearson rho per parameter (recovery_scatter, pyhddmjagsutils.py:609-623) and the "converged" count (posterior
mean of the non-decision time inside (0, 1), :239-241).  Medians are printed beside the means, never instead of them.
Prints the time and loss per 50 epochs.  `single`: the same for single_trial_alpha_not_scaled.py:284-287 (7 parameters, data (choicert, z1)).
usage: python tools/full_training_run.py [epochs=500] [basic|single] [file to save the trained amortizer's state_dict to | -] [runs=0]
(run r > 0: another initialisation, torch.manual_seed(r), and another training stream, seed 2023 + r; the recovery data sets stay the same.
 `runs` may be a comma-separated list, e.g. 0,1,2: one training + recovery per initialisation and a closing table over all of them --
 whether a run's table of MEANS is hit by a far-tail draw is chance, so the artifact shows every initialisation,
 each with the statistic unfiltered AND with the product's documented option sample(..., reject_outside=priors.prior_box(model)).)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesflow_nddms_amd import basic_ddm_dc                                                                    # noqa: E402
from bayesflow_nddms_amd.amortizer import AmortizedPosterior, InvariantNetwork, InvertibleNetwork   # noqa: E402
from bayesflow_nddms_amd.graph_trainer import TRAIN_OFFSET_BASE, GraphTrainer                                   # noqa: E402


def recovery(am, mod, names, n_datasets=500, n_draws=10000, box=None):
    """basic_ddm_dc.py:211-241.  Returns a dict of the table's numbers (means unfiltered, medians, means with the rejection option)."""
    from bayesflow_nddms_amd import diagnostics as dg
    np.random.seed(2023)                                    # (:217; the batch-shared N comes from NumPy's global generator)
    gm = mod.make_generative_model(batched=True, device_prior=True, as_numpy=False)
    torch.manual_seed(1234)                                 # the base draws z (tools/locate_tail_draws.py re-creates them from this seed)
    true, means, meds, n_trials, means_box, redrawn = [], [], [], [], [], 0
    t1 = time.time()
    for _ in range(n_datasets):
        conf = mod.configurator(gm(1))
        post = am.sample(conf, n_draws, to_numpy=False)                 # the reference's call (:223): every draw of the flow
        true.append(conf["parameters"][0].cpu().numpy())
        means.append(post.mean(0).cpu().numpy())
        meds.append(post.median(0).values.cpu().numpy())
        n_trials.append(conf["summary_conditions"].shape[1])
        if box is not None:                                             # the documented option, OFF by default: fresh draws, redrawn outside the box
            means_box.append(am.sample(conf, n_draws, to_numpy=False, reject_outside=box).mean(0).cpu().numpy())
            redrawn += am.last_redrawn
    true, means, meds = np.array(true, dtype=np.float64), np.array(means, dtype=np.float64), np.array(meds, dtype=np.float64)
    r2 = lambda est: np.round(dg.recovery_statistics(true, est)["r2"], 3)          # (== sklearn r2_score / scipy pearsonr per parameter:
    rho = lambda est: np.round(dg.recovery_statistics(true, est)["rho"], 3)        #  tests/test_host_logic.py)
    converged = dg.converged_fits(means)                    # :239-241 (index 3 = the non-decision time in both models)
    carried = (np.abs(means - meds) > 5.0 * (np.abs(meds) + 1.0)).any(axis=1)
    print(f"recovery: {n_datasets} fresh data sets x {n_draws} posterior draws ({time.time() - t1:.1f} s); mean number of simulated trials "
          f"{np.mean(n_trials):.0f} +/- {np.std(n_trials):.2f}; parameters: {names}\n"
          f"  POSTERIOR MEANS (the reference's statistic)  R^2 {r2(means)}\n"
          f"                                               rho {rho(means)}\n"
          f"  {int(converged.sum())} of {n_datasets} model fits were in the prior range for non-decision time\n"
          f"  posterior medians (beside, not instead)      R^2 {r2(meds)}\n"
          f"                                               rho {rho(meds)}\n"
          f"  largest |rho(means) - rho(medians)| {np.abs(rho(means) - rho(meds)).max():.3f}; data sets whose mean a tail draw carries off "
          f"(|mean - median| > 5 (|median| + 1)): {int(carried.sum())}; non-finite means: {int((~np.isfinite(means)).any(axis=1).sum())}", flush=True)
    out = {"rho_means": rho(means), "r2_means": r2(means), "rho_medians": rho(meds), "converged": int(converged.sum()), "carried": int(carried.sum())}
    if box is not None:
        mb = np.array(means_box, dtype=np.float64)
        print(f"  POSTERIOR MEANS with sample(..., reject_outside=priors.prior_box(model)) -- the product's documented option, off by default; "
              f"all {n_datasets} data sets, {redrawn} of {n_datasets * n_draws} draws redrawn\n"
              f"                                               R^2 {r2(mb)}\n"
              f"                                               rho {rho(mb)}\n"
              f"  {int(dg.converged_fits(mb).sum())} of {n_datasets} model fits were in the prior range for non-decision time", flush=True)
        out.update({"rho_means_box": rho(mb), "r2_means_box": r2(mb), "redrawn": redrawn})
    return out


def one_run(epochs, model, save_to, run):
    per_epoch, chunk = 1000, 50
    torch.manual_seed(run)
    am = AmortizedPosterior(InvertibleNetwork(num_params=5 if model == "basic" else 7), InvariantNetwork())
    t0 = time.time()
    # (training's parameter sets start at TRAIN_OFFSET_BASE of the seed's index space: the recovery loop's generative model draws
    #  its rows from 0 up and never meets them)
    with GraphTrainer(am, model=model, batch_size=32, total_steps=epochs * per_epoch, seed=2023 + run, offset_base=TRAIN_OFFSET_BASE) as gt:
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
    if save_to and save_to != "-":
        torch.save(am.state_dict(), save_to if run == 0 else f"{save_to}.run{run}")
    from bayesflow_nddms_amd import priors
    if model == "basic":
        mod, names = basic_ddm_dc, "drift, boundary, beta, tau, dc"
    else:
        from bayesflow_nddms_amd import single_trial_alpha_not_scaled as mod
        names = "drift, mu_alpha, beta, ter, std_alpha, dc, sigma1"
    am.eval()
    return recovery(am, mod, names, box=priors.prior_box(model)), total


def main():
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    model = sys.argv[2] if len(sys.argv) > 2 else "basic"
    save_to = sys.argv[3] if len(sys.argv) > 3 else "-"
    runs = [int(r) for r in (sys.argv[4] if len(sys.argv) > 4 else "0").split(",")]
    table = []
    for run in runs:
        if len(runs) > 1:
            print(f"\n===== initialisation {run} (torch.manual_seed({run}), training stream seed {2023 + run}) =====", flush=True)
        table.append((run,) + one_run(epochs, model, save_to, run))
    if len(runs) > 1:
        print(f"\n===== all {len(runs)} initialisations: posterior means on the reference's statistic ({model}) =====")
        print("run | seconds | rho, unfiltered | R^2, unfiltered | data sets carried off | rho, reject_outside=prior_box | R^2, reject_outside=prior_box | rho, medians")
        for run, t, secs in table:
            print(f"{run} | {secs:.0f} | {t['rho_means']} | {t['r2_means']} | {t['carried']} | {t['rho_means_box']} | {t['r2_means_box']} | {t['rho_medians']}")


if __name__ == "__main__":
    main()

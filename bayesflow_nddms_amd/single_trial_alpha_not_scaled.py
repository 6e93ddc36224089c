"""Drop-in for the generative-model section of the reference's single_trial_alpha_not_scaled.py: the 7-parameter
model with a per-trial boundary ~ N(mu_alpha, std_alpha) > 0 and an external datum z1 ~ N(bound_trial, sigma1)
(lines 66-191), its fine-step variant (:1710-1722), and the misspecification simulators `_alt` (:926-974),
`_scale` (:1237-1285), `_scale2` (:1471-1519).  Function names, parameter order and return shapes follow the
reference; arithmetic runs on the MI355X through the C ABI.
"""
import numpy as np

from . import engine
from .basic_ddm_dc import configurator as _basic_configurator
from .priors import (DevicePrior, draw_prior_scale, draw_prior_single as draw_prior, prior_N,  # noqa: F401
                     truncnorm_better)
from .simulation import ContextGenerator, GenerativeModel, Prior, Simulator

PARAM_NAMES = ("drift", "mu_alpha", "beta", "ter", "std_alpha", "dc", "sigma1")   # :148 -- the order is the ABI
num_params = 7
draw_prior_alt = draw_prior   # :889-913 has identical marginals (std_dc, mu_dc at indices 4, 5)


def _with_gamma(params, gamma):
    """[.., 7] reference rows -> [B, 8] ABI rows (gamma appended); [.., 8] rows (the _scale layout :1277) pass."""
    if hasattr(params, "detach"):
        import torch
        p = params if params.ndim == 2 else params[None]
        if p.shape[1] == 7:
            p = torch.cat([p.to(torch.float32), torch.full((p.shape[0], 1), float(gamma), dtype=torch.float32,
                                                           device=p.device)], dim=1)
        return p
    p = np.asarray(params, dtype=np.float64)
    p = p[None] if p.ndim == 1 else p
    if p.shape[1] == 7:
        p = np.concatenate([p, np.full((p.shape[0], 1), float(gamma))], axis=1)
    return p


def _run(model, params, n_trials, gamma, dt, max_steps, seed, set_offset, fast, want_summary=False, to_host=False, state_f64=False):
    run = engine.simulate_to_host if to_host else engine.simulate
    return run(model, _with_gamma(params, gamma), n_trials, dt=dt, max_steps=max_steps, seed=seed, set_offset=set_offset, fast=fast,
               want_summary=want_summary, state_f64=state_f64)


def diffusion_trial(drift, mu_alpha, beta, ter, std_alpha, dc, sigma1, dt=.01, max_steps=400., seed=None,
                    set_offset=None, fast=None):
    """One trial (:107-142) -> (choicert, extdata1)."""
    r = _run(engine.SINGLE_TRIAL, [drift, mu_alpha, beta, ter, std_alpha, dc, sigma1], 1, 1.0, dt, max_steps, seed,
             set_offset, fast)
    return tuple(r["trials"][0, 0].tolist())


def simulate_trials(params, n_trials, dt=.01, max_steps=400., seed=None, set_offset=None, fast=None, state_f64=False):
    """(:144-155) -> float64 [n_trials, 2] = (choicert, z1); choicert = +-(ter + rt), 0 = missing response.
    state_f64=True: the evidence recurrence and the per-trial boundary in the reference's float64 arithmetic (NDDM_STATE_F64)."""
    r = _run(engine.SINGLE_TRIAL, params, n_trials, 1.0, dt, max_steps, seed, set_offset, fast, state_f64=state_f64)
    return r["trials"][0].cpu().numpy().astype(np.float64)


def simulate_trials_fine(params, n_trials, seed=None, set_offset=None, fast=None, state_f64=False):
    """(:1710-1722): 1 ms resolution, max_steps=4000 keeps the 4 s tolerance."""
    return simulate_trials(params, n_trials, dt=.001, max_steps=4000, seed=seed, set_offset=set_offset, fast=fast, state_f64=state_f64)


def simulate_trials_alt(params, n_trials, dt=.01, max_steps=400., seed=None, set_offset=None, fast=None):
    """(:963-974): params = drift, alpha, beta, ter, std_dc, mu_dc, sigma1; z1 ~ N(dc_trial, sigma1)."""
    r = _run(engine.SINGLE_TRIAL_ALT, params, n_trials, 1.0, dt, max_steps, seed, set_offset, fast)
    return r["trials"][0].cpu().numpy().astype(np.float64)


def simulate_trials_scale(params, n_trials, dt=.01, max_steps=400., seed=None, set_offset=None, fast=None):
    """(:1274-1285): 8 parameters, gamma last; z1 ~ N(gamma*bound_trial, sigma1)."""
    p = np.asarray(params, dtype=np.float64)
    if p.shape[-1] != 8:
        raise ValueError("simulate_trials_scale takes 8 parameters (gamma last)")
    r = _run(engine.SINGLE_TRIAL, p, n_trials, 1.0, dt, max_steps, seed, set_offset, fast)
    return r["trials"][0].cpu().numpy().astype(np.float64)


def simulate_trials_scale2(params, n_trials, dt=.01, max_steps=400., seed=None, set_offset=None, fast=None):
    """(:1508-1519): z1 ~ N(2*bound_trial, sigma1)."""
    r = _run(engine.SINGLE_TRIAL, params, n_trials, 2.0, dt, max_steps, seed, set_offset, fast)
    return r["trials"][0].cpu().numpy().astype(np.float64)


def batch_simulate_trials(params, n_trials, dt=.01, max_steps=400., gamma=1.0, variant="single", seed=None,
                          set_offset=None, fast=None, as_numpy=True, with_summary=True):
    """Whole batch in one launch: params [B, 7] (or [B, 8] with gamma) -> {'sim_data': [B, n_trials, 2],
    'summary_stats': [B, 10]}.  variant: 'single' | 'alt'.  as_numpy: pinned host arrays, large batches chunk by chunk beside
    the simulation (engine.simulate_to_host)."""
    model = engine.SINGLE_TRIAL if variant == "single" else engine.SINGLE_TRIAL_ALT
    r = _run(model, params, n_trials, gamma, dt, max_steps, seed, set_offset, fast, want_summary=with_summary, to_host=as_numpy)
    out = {"sim_data": r["trials"]}
    if with_summary:
        out["summary_stats"] = r["summary"]
    return out


def configurator(sim_dict):
    """(:169-191): as basic_ddm_dc's, but 'parameters' is skipped when prior_draws is None (:189-190)."""
    if sim_dict.get('prior_draws', None) is None:
        tmp = dict(sim_dict)
        tmp['prior_draws'] = np.zeros((1, 1), dtype=np.float32)
        out = _basic_configurator(tmp)
        del out['parameters']
        return out
    return _basic_configurator(sim_dict)


def make_generative_model(batched=True, device_prior=False, fine=False, fast=None, as_numpy=True, seed=None,
                          skip_test=False):
    """The reference's wrapper block (:160-164); fine=True builds generative_model_fine (:1726-1728)."""
    dt, max_steps = (.001, 4000) if fine else (.01, 400.)
    experimental_context = ContextGenerator(non_batchable_context_fun=prior_N)
    prior = Prior(batch_prior_fun=DevicePrior("single", seed=2023 if seed is None else seed), param_names=PARAM_NAMES) \
        if device_prior else Prior(prior_fun=draw_prior, param_names=PARAM_NAMES)
    if batched:
        fun = lambda p, n: batch_simulate_trials(p, n, dt=dt, max_steps=max_steps, fast=fast, as_numpy=as_numpy)
        simulator = Simulator(batch_simulator_fun=fun, context_generator=experimental_context)
    else:
        fun = lambda p, n: simulate_trials(p, n, dt=dt, max_steps=max_steps, fast=fast)
        simulator = Simulator(simulator_fun=fun, context_generator=experimental_context)
    gm = GenerativeModel(prior, simulator, skip_test=skip_test, name="single_trial_alpha_not_scaled")
    # how a graph loop re-creates this model on the device (amortizer.Trainer(graph=True) -> graph_trainer.GraphTrainer)
    gm.graph_spec = dict(model="single", dt=dt, max_steps=max_steps, seed=2023 if seed is None else seed, n_min=60, n_max=300)
    return gm

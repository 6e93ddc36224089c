"""Drop-in for the generative-model section of the reference's basic_ddm_dc.py (lines 50-160): same function
names, argument order and return shapes; the simulator runs on the MI355X through the C ABI.

    from bayesflow_nddms_amd.basic_ddm_dc import draw_prior, prior_N, simulate_trials, configurator
    prior = Prior(prior_fun=draw_prior); ...; generative_model = GenerativeModel(prior, simulator)

or, batched (one kernel launch per batch, device-resident output):

    generative_model = make_generative_model(batched=True)
"""
import numpy as np

from . import engine
from .priors import RNG, DevicePrior, draw_prior_basic as draw_prior, prior_N, truncnorm_better  # noqa: F401
from .simulation import ContextGenerator, GenerativeModel, Prior, Simulator

MODEL = engine.BASIC_DDM_DC
PARAM_NAMES = ("drift", "boundary", "beta", "tau", "dc")   # basic_ddm_dc.py:118 -- the order is the ABI
num_params = 5


def diffusion_trial(drift, boundary, beta, tau, dc, dt=.01, max_steps=400., seed=None, set_offset=None, fast=None, state_f64=False):
    """One trial (basic_ddm_dc.py:85-112) -> (rt, choice).  choice is 0 on timeout (the reference leaves it unbound).
    state_f64=True (here and below): the evidence recurrence in the reference's float64 arithmetic (NDDM_STATE_F64)."""
    r = engine.simulate(MODEL, [drift, boundary, beta, tau, dc], 1, dt=dt, max_steps=max_steps, seed=seed,
                        set_offset=set_offset, fast=fast, want_summary=False, state_f64=state_f64)
    rt, choice = r["trials"][0, 0].tolist()
    return rt, int(choice)


def simulate_trials(params, n_trials, dt=.01, max_steps=400., seed=None, set_offset=None, fast=None, state_f64=False):
    """simulate_trials(params, n_trials) -> float64 [n_trials, 2] = (rt, choice)  (basic_ddm_dc.py:114-125)."""
    r = engine.simulate(MODEL, np.asarray(params, dtype=np.float64).reshape(1, 5), n_trials, dt=dt,
                        max_steps=max_steps, seed=seed, set_offset=set_offset, fast=fast, want_summary=False, state_f64=state_f64)
    return r["trials"][0].cpu().numpy().astype(np.float64)


def batch_simulate_trials(params, n_trials, dt=.01, max_steps=400., seed=None, set_offset=None, fast=None,
                          as_numpy=True, with_summary=True, state_f64=False):
    """Whole batch in one launch: params [B, 5] (numpy or device tensor) -> {'sim_data': [B, n_trials, 2] float32,
    'summary_stats': [B, 10]} (numpy by default -- large batches then travel to pinned host memory chunk by chunk beside the
    simulation of the next chunk, engine.simulate_to_host; device tensors with as_numpy=False)."""
    run = engine.simulate_to_host if as_numpy else engine.simulate
    r = run(MODEL, params, n_trials, dt=dt, max_steps=max_steps, seed=seed, set_offset=set_offset, fast=fast, want_summary=with_summary,
            state_f64=state_f64)
    out = {"sim_data": r["trials"]}
    if with_summary:
        out["summary_stats"] = r["summary"]
    return out


def configurator(sim_dict):
    """basic_ddm_dc.py:139-160: dict -> {'summary_conditions', 'direct_conditions', 'parameters'} (float32).
    Accepts numpy arrays or device tensors (tensors stay on the device)."""
    out = dict()
    data = sim_dict['sim_data']
    n_obs = np.log(sim_dict['sim_non_batchable_context'])
    if hasattr(data, "detach"):
        import torch
        data = data.to(torch.float32)
        out['summary_conditions'] = data
        out['direct_conditions'] = torch.full((data.shape[0], 1), float(n_obs), dtype=torch.float32, device=data.device)
        pd = sim_dict['prior_draws']
        out['parameters'] = pd.to(torch.float32) if hasattr(pd, "detach") else torch.as_tensor(
            np.asarray(pd), dtype=torch.float32, device=data.device)
        return out
    data = data.astype(np.float32)
    out['summary_conditions'] = data
    # float32 as on the reference's pinned NumPy 1.23.5 (value-based casting); NumPy >= 2 would promote to float64
    out['direct_conditions'] = (n_obs * np.ones((data.shape[0], 1), dtype=np.float32)).astype(np.float32)
    out['parameters'] = np.asarray(sim_dict['prior_draws']).astype(np.float32)
    return out


def make_generative_model(batched=True, device_prior=False, dt=.01, max_steps=400., fast=None, as_numpy=True,
                          seed=None, skip_test=False):
    """The reference's wrapper block (basic_ddm_dc.py:130-134).  batched=False keeps the per-set simulator_fun loop
    exactly as BayesFlow runs it; batched=True hands the whole batch to one kernel launch (and, with
    device_prior=True, also draws the parameters on the device)."""
    experimental_context = ContextGenerator(non_batchable_context_fun=prior_N)
    prior = Prior(batch_prior_fun=DevicePrior("basic", seed=2023 if seed is None else seed), param_names=PARAM_NAMES) \
        if device_prior else Prior(prior_fun=draw_prior, param_names=PARAM_NAMES)
    if batched:
        fun = lambda p, n: batch_simulate_trials(p, n, dt=dt, max_steps=max_steps, fast=fast, as_numpy=as_numpy)
        simulator = Simulator(batch_simulator_fun=fun, context_generator=experimental_context)
    else:
        fun = lambda p, n: simulate_trials(p, n, dt=dt, max_steps=max_steps, fast=fast)
        simulator = Simulator(simulator_fun=fun, context_generator=experimental_context)
    gm = GenerativeModel(prior, simulator, skip_test=skip_test, name="basic_ddm_dc")
    # how a graph loop re-creates this model on the device (amortizer.Trainer(graph=True) -> graph_trainer.GraphTrainer)
    gm.graph_spec = dict(model="basic", dt=dt, max_steps=max_steps, seed=2023 if seed is None else seed, n_min=60, n_max=300)
    return gm

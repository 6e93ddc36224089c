"""The generative-model boundary: a mirror of the BayesFlow 1.1 `bf.simulation` wrappers the reference instantiates
at basic_ddm_dc.py:130-134 / single_trial_alpha_not_scaled.py:160-164 (Prior, ContextGenerator, Simulator,
GenerativeModel), so reference scripts keep their call shape:

    prior = Prior(prior_fun=draw_prior)
    experimental_context = ContextGenerator(non_batchable_context_fun=prior_N)
    simulator = Simulator(simulator_fun=simulate_trials, context_generator=experimental_context)
    generative_model = GenerativeModel(prior, simulator)
    out = generative_model(32)   # dict: 'prior_draws' (B,P), 'sim_data' (B,N,2), 'sim_non_batchable_context' N, ...

BayesFlow itself is not importable in this image, so these classes reproduce the dictionary
contract the reference's call sites rely on (basic_ddm_dc.py:146,151,159; single_trial_alpha_not_scaled.py:884-885).
The objects are duck-type compatible with what bf.trainers.Trainer needs from a generative model: a callable
`generative_model(batch_size) -> dict`.  When the simulator is one of this package's HIP simulators in batched form,
the whole batch is ONE kernel launch, and the dict additionally carries 'summary_stats' (B, 10) from the fused
reduction (an additive key; the reference's keys are untouched).
"""
import numpy as np

DEFAULT_KEYS = {
    "prior_draws": "prior_draws",
    "prior_batchable_context": "prior_batchable_context",
    "prior_non_batchable_context": "prior_non_batchable_context",
    "sim_data": "sim_data",
    "sim_batchable_context": "sim_batchable_context",
    "sim_non_batchable_context": "sim_non_batchable_context",
    "summary_stats": "summary_stats",
}


class ContextGenerator:
    """bf.simulation.ContextGenerator: batchable and/or non-batchable context (the reference only uses the
    non-batchable form, prior_N, shared by the whole batch -- basic_ddm_dc.py:131)."""

    def __init__(self, batchable_context_fun=None, non_batchable_context_fun=None,
                 use_non_batchable_for_batchable=False):
        self.batchable_context_fun = batchable_context_fun
        self.non_batchable_context_fun = non_batchable_context_fun
        self.use_non_batchable_for_batchable = use_non_batchable_for_batchable

    def __call__(self, batch_size, *args, **kwargs):
        return self.generate_context(batch_size, *args, **kwargs)

    def non_batchable_context(self, *args, **kwargs):
        if self.non_batchable_context_fun is None:
            return None
        return self.non_batchable_context_fun(*args, **kwargs)

    def batchable_context(self, batch_size, *args, **kwargs):
        if self.batchable_context_fun is None:
            return None
        return [self.batchable_context_fun(*args, **kwargs) for _ in range(batch_size)]

    def generate_context(self, batch_size, *args, **kwargs):
        out = {"non_batchable_context": None, "batchable_context": None}
        if self.non_batchable_context_fun is not None:
            out["non_batchable_context"] = self.non_batchable_context_fun()
        if self.batchable_context_fun is not None:
            if self.use_non_batchable_for_batchable:
                out["batchable_context"] = [self.batchable_context_fun(out["non_batchable_context"], *args, **kwargs)
                                            for _ in range(batch_size)]
            else:
                out["batchable_context"] = self.batchable_context(batch_size, *args, **kwargs)
        return out


class Prior:
    """bf.simulation.Prior: `prior_fun() -> ndarray[P]` (called once per draw, as the reference's draw_prior) or
    `batch_prior_fun(batch_size) -> [B, P]` (numpy or device tensor; e.g. priors.DevicePrior)."""

    def __init__(self, batch_prior_fun=None, prior_fun=None, context_generator=None, param_names=None):
        if (batch_prior_fun is None) == (prior_fun is None):
            raise ValueError("Either batch_prior_fun or prior_fun should be provided, but not both!")
        self.prior = prior_fun if prior_fun is not None else batch_prior_fun
        self.is_batched = batch_prior_fun is not None
        self.context_gen = context_generator
        self.param_names = param_names

    def __call__(self, batch_size, *args, **kwargs):
        out = {"prior_draws": None, "batchable_context": None, "non_batchable_context": None}
        ctx = None
        if self.context_gen is not None:
            ctx = self.context_gen(batch_size, *args, **kwargs)
            out["non_batchable_context"] = ctx["non_batchable_context"]
            out["batchable_context"] = ctx["batchable_context"]
        if self.is_batched:
            out["prior_draws"] = self.prior(batch_size)
        else:
            out["prior_draws"] = np.array([self.prior() for _ in range(batch_size)])
        return out


class Simulator:
    """bf.simulation.Simulator.  `simulator_fun(params[P], context) -> [N, 2]` is looped over the batch exactly as
    BayesFlow does (basic_ddm_dc.py:132-134); `batch_simulator_fun(params[B, P], context)` takes the whole batch --
    the natural entry of the HIP simulators (one launch).  A batched simulator may return either the data array or
    a dict with 'sim_data' and optional 'summary_stats'."""

    def __init__(self, batch_simulator_fun=None, simulator_fun=None, context_generator=None):
        if (batch_simulator_fun is None) == (simulator_fun is None):
            raise ValueError("Either batch_simulator_fun or simulator_fun should be provided, but not both!")
        self.is_batched = batch_simulator_fun is not None
        self.simulator = batch_simulator_fun if self.is_batched else simulator_fun
        self.context_gen = context_generator

    def __call__(self, params, *args, **kwargs):
        batch_size = params.shape[0]
        out = {"sim_data": None, "batchable_context": None, "non_batchable_context": None}
        extra = []
        if self.context_gen is not None:
            ctx = self.context_gen.generate_context(batch_size, *args, **kwargs)
            out["non_batchable_context"] = ctx["non_batchable_context"]
            out["batchable_context"] = ctx["batchable_context"]
            if ctx["non_batchable_context"] is not None:
                extra.append(ctx["non_batchable_context"])
        if self.is_batched:
            if out["batchable_context"] is not None:
                res = self.simulator(params, out["batchable_context"], *extra, *args, **kwargs)
            else:
                res = self.simulator(params, *extra, *args, **kwargs)
            if isinstance(res, dict):
                out["sim_data"] = res["sim_data"]
                if "summary_stats" in res:
                    out["summary_stats"] = res["summary_stats"]
            else:
                out["sim_data"] = res
        else:
            host = np.asarray(params.detach().cpu().numpy() if hasattr(params, "detach") else params)
            if out["batchable_context"] is not None:
                rows = [self.simulator(host[b], out["batchable_context"][b], *extra, *args, **kwargs)
                        for b in range(batch_size)]
            else:
                rows = [self.simulator(host[b], *extra, *args, **kwargs) for b in range(batch_size)]
            out["sim_data"] = np.array(rows)
        return out


class GenerativeModel:
    """bf.simulation.GenerativeModel(prior, simulator): `generative_model(batch_size)` returns the dictionary the
    reference's configurators consume.  Like BayesFlow, construction runs a small self-test (batch of 2) unless
    skip_test=True."""

    _N_SIM_TEST = 2

    def __init__(self, prior, simulator, skip_test=False, prior_is_batched=False, simulator_is_batched=None,
                 name="anonymous"):
        if not isinstance(prior, Prior):
            prior = Prior(batch_prior_fun=prior) if prior_is_batched else Prior(prior_fun=prior)
        if not isinstance(simulator, Simulator):
            if simulator_is_batched is None:
                raise ValueError("simulator_is_batched must be given when simulator is a bare function")
            simulator = Simulator(batch_simulator_fun=simulator) if simulator_is_batched \
                else Simulator(simulator_fun=simulator)
        self.prior = prior
        self.simulator = simulator
        self.name = name
        self.param_names = prior.param_names
        if not skip_test:
            self._test()

    def __call__(self, batch_size, **kwargs):
        prior_out = self.prior(batch_size, **kwargs.pop("prior_args", {}))
        sim_out = self.simulator(prior_out["prior_draws"], **kwargs.pop("sim_args", {}))
        out = {
            DEFAULT_KEYS["prior_non_batchable_context"]: prior_out["non_batchable_context"],
            DEFAULT_KEYS["prior_batchable_context"]: prior_out["batchable_context"],
            DEFAULT_KEYS["prior_draws"]: prior_out["prior_draws"],
            DEFAULT_KEYS["sim_non_batchable_context"]: sim_out["non_batchable_context"],
            DEFAULT_KEYS["sim_batchable_context"]: sim_out["batchable_context"],
            DEFAULT_KEYS["sim_data"]: sim_out["sim_data"],
        }
        if "summary_stats" in sim_out:
            out[DEFAULT_KEYS["summary_stats"]] = sim_out["summary_stats"]
        return out

    def _test(self):
        out = self(self._N_SIM_TEST)
        p, d = out["prior_draws"], out["sim_data"]
        if p.shape[0] != self._N_SIM_TEST or d.shape[0] != self._N_SIM_TEST:
            raise ValueError(f"generative model self-test failed: prior_draws {tuple(p.shape)}, sim_data {tuple(d.shape)}")
        return {"prior_draws": tuple(p.shape), "sim_data": tuple(d.shape)}

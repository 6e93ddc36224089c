"""Batched device entry of the DDM trial simulators: Python adapter over the C ABI (include/nddm.h).

PyTorch is plumbing here (device memory and streams); all arithmetic happens in the HIP kernels.
There is no CPU path: calls raise if no ROCm device / HIP library is available.
"""
import math
import os
import threading

import numpy as np

from . import _lib

BASIC_DDM_DC, SINGLE_TRIAL, SINGLE_TRIAL_ALT, ALPHA_NOT_SCALED, EXPLICIT_BOUNDARY = range(5)
SUMMARY_K = 10
SUMMARY_COLS = ("n_upper", "n_lower", "n_missing", "mean_rt", "var_rt", "mean_rt_upper", "var_rt_upper",
                "mean_z", "var_z", "choice_mean")
NPARAMS = {BASIC_DDM_DC: 5, SINGLE_TRIAL: 8, SINGLE_TRIAL_ALT: 8, ALPHA_NOT_SCALED: 6, EXPLICIT_BOUNDARY: 4}
# columns that must be > 0 for the process to be defined (boundary / diffusion coefficient), per model
_POSITIVE_COLS = {BASIC_DDM_DC: (1, 4), SINGLE_TRIAL: (5,), SINGLE_TRIAL_ALT: (1,), ALPHA_NOT_SCALED: (1, 5),
                  EXPLICIT_BOUNDARY: (3,)}

# default Gaussian transform of the product path; tests pin the exact one against the oracle bit for bit
DEFAULT_FAST = True


def _torch():
    import torch
    return torch


def require_device():
    """Fail loudly when the HIP path cannot run (no silent CPU fallback)."""
    torch = _torch()
    _lib.lib()
    if not torch.cuda.is_available():
        raise RuntimeError("bayesflow_nddms_amd needs a ROCm GPU (torch.cuda.is_available() is False); "
                           "there is no CPU fallback")
    return torch


class StreamState:
    """Functional RNG position: (seed, next set index).  The simulators are stateless apart from this pair, so a
    resumed run continues the stream by restoring it (checkpoint/resume)."""

    def __init__(self, seed=0, offset=0):
        self._lock = threading.Lock()
        self.seed = int(seed)
        self.offset = int(offset)

    def take(self, n_sets):
        with self._lock:
            off = self.offset
            self.offset += int(n_sets)
            return self.seed, off

    def get_state(self):
        return {"seed": self.seed, "offset": self.offset}

    def set_state(self, state):
        with self._lock:
            self.seed, self.offset = int(state["seed"]), int(state["offset"])


GLOBAL_STREAM = StreamState(seed=0)


def seed(s):
    """Reset the package-level stream (the analogue of np.random.seed for the device simulators)."""
    GLOBAL_STREAM.set_state({"seed": int(s), "offset": 0})


def max_k_of(max_steps):
    """The reference loops `while ... n_steps < max_steps` with a float cap (basic_ddm_dc.py:87, 95): the largest
    step count reached is ceil(max_steps)."""
    return int(math.ceil(float(max_steps)))


def validate_params_host(model, params):
    """Host-side validation (only possible when parameters arrive on the host): the reference's error
    convention is ValueError (imputation_from_stahl_not_scaled.py:124-125)."""
    p = np.asarray(params)
    if not np.all(np.isfinite(p)):
        raise ValueError("parameters must be finite")
    for c in _POSITIVE_COLS[model]:
        if np.any(p[..., c] <= 0):
            raise ValueError(f"parameter column {c} (boundary / diffusion coefficient) must be > 0")
    if model in (SINGLE_TRIAL, SINGLE_TRIAL_ALT):
        lat = p[..., 1] if model == SINGLE_TRIAL else p[..., 5]
        if np.any(lat + 8.0 * np.abs(p[..., 4]) <= 0):
            raise ValueError("per-trial latent N(mean, std) > 0 is (numerically) never satisfied")


def simulate(model, params, n_trials, dt=0.01, max_steps=400.0, seed=None, set_offset=None, fast=None,
             bounds=None, ext_sigma=0.0, ext_mode=0, bridge=False, packed=False, want_trials=True, want_summary=True, want_ext=False,
             out_trials=None, out_summary=None, stream_state=None, device=None, set_offset_dev=None, want_codes=False,
             out_codes=None, state_f64=False):
    """Run one batched simulation on the current ROCm device.

    state_f64=True selects NDDM_STATE_F64 (include/nddm.h): the evidence is carried in float64 exactly as the reference's recurrence
    does (basic_ddm_dc.py:91-103) on the same normals -- basic_ddm_dc and single_trial only; with fast=False every trial's (step,
    choice) equals the float64 oracle's bit for bit.

    params: array-like or torch tensor [B, P] (or [P]) in the reference's parameter order.
    packed=True selects NDDM_GAUSS_PACKED (include/nddm.h): 8 normals per Philox block from 16 + 16 bit pairs, ~25 % faster,
    a different random stream; not with the bridge, max_steps < 2^14.

    set_offset_dev: optional device int64 tensor [1]; the global index of row 0 is then set_offset + its value WHEN THE
    LAUNCH RUNS (nddm_simulate_indirect) -- a launch captured into a hipGraph moves along the random stream by a captured
    `set_offset_dev += B` instead of new kernel arguments.

    want_codes / out_codes: also (or, with want_trials=False, only) write the trials in the 2-byte wire format, int16 [B, n_trials]
    holding uint16 (step index | code << 14) -- basic_ddm_dc and alpha_not_scaled without the bridge, max_steps < 2^14;
    decode_codes() gives the float pairs back (include/nddm.h: nddm_simulate_codes).

    Returns a dict of torch tensors on the device: 'trials' f32 [B, n_trials, 2], 'summary' f32 [B, 10],
    'ext' f32 [B] (alpha_not_scaled only), plus 'seed' / 'set_offset' actually used.
    """
    torch = require_device()
    L = _lib.lib()
    P = NPARAMS[model]
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    on_host = not (isinstance(params, torch.Tensor) and params.is_cuda)
    if on_host:
        p_np = np.ascontiguousarray(params.detach().cpu().numpy() if isinstance(params, torch.Tensor) else params,
                                    dtype=np.float64)
        if p_np.ndim == 1:
            p_np = p_np[None]
        if p_np.ndim != 2 or p_np.shape[1] != P:
            raise ValueError(f"params must have shape [B, {P}] for this model, got {p_np.shape}")
        validate_params_host(model, p_np)
        p_dev = torch.as_tensor(p_np, dtype=torch.float32).contiguous().to(dev)
    else:
        p_dev = params
        if p_dev.ndim == 1:
            p_dev = p_dev[None]
        if p_dev.ndim != 2 or p_dev.shape[1] != P:
            raise ValueError(f"params must have shape [B, {P}] for this model, got {tuple(p_dev.shape)}")
        p_dev = p_dev.to(dtype=torch.float32).contiguous()
    B = int(p_dev.shape[0])
    n_trials = int(n_trials)
    if n_trials <= 0:
        raise ValueError("n_trials must be positive")
    if not (dt > 0 and math.isfinite(dt)):
        raise ValueError("dt must be finite and > 0")
    max_k = max_k_of(max_steps)
    if max_k < 0:
        raise ValueError("max_steps must be >= 0")

    b_dev = None
    if model == EXPLICIT_BOUNDARY:
        if bounds is None:
            raise ValueError("explicit-boundary model needs `bounds`")
        if not (isinstance(bounds, torch.Tensor) and bounds.is_cuda):
            b_np = np.ascontiguousarray(np.asarray(bounds, dtype=np.float64).reshape(B, n_trials))
            if np.any(b_np < 0) or not np.all(np.isfinite(b_np)):
                raise ValueError("Trial-level boundary cannot be less than zero")
            b_dev = torch.as_tensor(b_np, dtype=torch.float32).contiguous().to(dev)
        else:
            b_dev = bounds.to(dtype=torch.float32).reshape(B, n_trials).contiguous()

    if seed is None or set_offset is None:
        s_seed, s_off = (stream_state or GLOBAL_STREAM).take(B)
        seed = s_seed if seed is None else seed
        set_offset = s_off if set_offset is None else set_offset
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    set_offset = int(set_offset) & 0xFFFFFFFFFFFFFFFF
    fast = DEFAULT_FAST if fast is None else bool(fast)
    flags = (_lib.GAUSS_FAST if fast else _lib.GAUSS_EXACT) | (_lib.BRIDGE if bridge else 0) | (_lib.GAUSS_PACKED if packed else 0) \
        | (_lib.STATE_F64 if state_f64 else 0)
    if bridge and model != ALPHA_NOT_SCALED:
        raise ValueError("the Brownian-bridge correction is only available for the alpha_not_scaled model")

    with torch.cuda.device(dev):
        if want_trials and out_trials is None:
            out_trials = torch.empty((B, n_trials, 2), dtype=torch.float32, device=dev)
        if want_summary and out_summary is None:
            out_summary = torch.empty((B, SUMMARY_K), dtype=torch.float32, device=dev)
        out_ext = torch.empty((B,), dtype=torch.float32, device=dev) if (want_ext and model == ALPHA_NOT_SCALED) else None
        if want_codes and out_codes is None:
            out_codes = torch.empty((B, n_trials), dtype=torch.int16, device=dev)
        if out_codes is not None and (tuple(out_codes.shape) != (B, n_trials) or out_codes.dtype != torch.int16
                                      or not out_codes.is_contiguous() or not out_codes.is_cuda):
            raise ValueError(f"out_codes must be a contiguous int16 device tensor of shape {(B, n_trials)}")
        for t, shape in ((out_trials, (B, n_trials, 2)), (out_summary, (B, SUMMARY_K))):
            if t is not None and (tuple(t.shape) != shape or t.dtype != torch.float32 or not t.is_contiguous()
                                  or not t.is_cuda):
                raise ValueError(f"output buffer must be a contiguous float32 device tensor of shape {shape}")
        st = torch.cuda.current_stream(dev).cuda_stream
        pt = lambda t: None if t is None else t.data_ptr()
        if B > 0:
            common = (B, n_trials, float(dt), max_k, seed, set_offset, flags)
            if set_offset_dev is not None and not (isinstance(set_offset_dev, torch.Tensor) and set_offset_dev.is_cuda
                                                   and set_offset_dev.dtype == torch.int64 and set_offset_dev.numel() >= 1):
                raise ValueError("set_offset_dev must be a device int64 tensor")
            if out_codes is not None:
                rc = L.nddm_simulate_codes(model, pt(p_dev), *common[:-1], pt(set_offset_dev), flags, pt(out_codes), pt(out_trials),
                                           pt(out_summary), st)
            elif set_offset_dev is not None:
                if not (isinstance(set_offset_dev, torch.Tensor) and set_offset_dev.is_cuda and set_offset_dev.dtype == torch.int64
                        and set_offset_dev.numel() >= 1):
                    raise ValueError("set_offset_dev must be a device int64 tensor")
                rc = L.nddm_simulate_indirect(model, pt(p_dev), pt(b_dev), *common[:-1], set_offset_dev.data_ptr(), flags,
                                              float(ext_sigma), int(ext_mode), pt(out_trials), pt(out_summary), pt(out_ext), st)
            elif model == BASIC_DDM_DC:
                rc = L.nddm_basic_ddm_dc_simulate(pt(p_dev), *common, pt(out_trials), pt(out_summary), st)
            elif model == SINGLE_TRIAL:
                rc = L.nddm_single_trial_simulate(pt(p_dev), *common, pt(out_trials), pt(out_summary), st)
            elif model == SINGLE_TRIAL_ALT:
                rc = L.nddm_single_trial_alt_simulate(pt(p_dev), *common, pt(out_trials), pt(out_summary), st)
            elif model == ALPHA_NOT_SCALED:
                rc = L.nddm_alpha_not_scaled_simulate(pt(p_dev), *common, float(ext_sigma), int(ext_mode),
                                                      pt(out_trials), pt(out_summary), pt(out_ext), st)
            elif model == EXPLICIT_BOUNDARY:
                rc = L.nddm_explicit_boundary_simulate(pt(p_dev), pt(b_dev), *common, pt(out_trials),
                                                       pt(out_summary), st)
            else:
                raise ValueError("unknown model")
            _lib.check(rc)
            # the kernel reads p_dev / b_dev asynchronously: tie their lifetime to the stream
            p_dev.record_stream(torch.cuda.current_stream(dev))
            if b_dev is not None:
                b_dev.record_stream(torch.cuda.current_stream(dev))
    res = {"seed": seed, "set_offset": set_offset, "params": p_dev}
    if out_trials is not None:
        res["trials"] = out_trials
    if out_summary is not None:
        res["summary"] = out_summary
    if out_ext is not None:
        res["ext"] = out_ext
    if out_codes is not None:
        res["codes"] = out_codes
    return res


def simulratcliff(params, n_trials, seed=None, set_offset=None, fast=None, ext_sigma=0.0, ext_mode=0, want_trials=True, want_summary=True,
                  want_ext=False, out_trials=None, out_summary=None, stream_state=None, device=None):
    """The EXACT first-passage sampler the reference generates alpha_not_scaled's data with -- simulratcliff, pyhddmjagsutils.py:47-176
    as called at alpha_not_scaled.py:95-108 -- batched on the device (include/nddm.h: nddm_simulratcliff): no step size.

    params: [B, 6] (or [6]) = Nu, Alpha, Beta, Tau, Eta, Varsigma.  Returns a dict of device tensors: 'trials' f32 [B, n_trials, 2] =
    (y, acc) with y = +-(Tau + decision time), 'summary' f32 [B, 10], 'ext' f32 [B], plus 'seed' / 'set_offset' / 'params'.
    fast=False: the reference's series term by term, bit-equal to the test suite's CPU restatement (its section D).  fast (the default,
    as for simulate()): hardware log / exp / reciprocal and the same acceptance function from three terms of its series or of the
    series' Jacobi-dual form -- on 6e6 trials no response differs from fast=False and no response time by more than 1e-6 s
    (profiles/r6_ratcliff_agreement.txt), at twice the rate."""
    torch = require_device()
    L = _lib.lib()
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    if not (isinstance(params, torch.Tensor) and params.is_cuda):
        p_np = np.ascontiguousarray(params.detach().cpu().numpy() if isinstance(params, torch.Tensor) else params, dtype=np.float64)
        if p_np.ndim == 1:
            p_np = p_np[None]
        if p_np.ndim != 2 or p_np.shape[1] != 6:
            raise ValueError(f"params must have shape [B, 6] (Nu, Alpha, Beta, Tau, Eta, Varsigma), got {p_np.shape}")
        validate_params_host(ALPHA_NOT_SCALED, p_np)
        if np.any(p_np[:, 2] < 0) or np.any(p_np[:, 2] > 1) or np.any(p_np[:, 4] < 0):
            raise ValueError("Beta must lie in [0, 1] and Eta must be >= 0")
        p_dev = torch.as_tensor(p_np, dtype=torch.float32).contiguous().to(dev)
    else:
        p_dev = params[None] if params.ndim == 1 else params
        if p_dev.ndim != 2 or p_dev.shape[1] != 6:
            raise ValueError(f"params must have shape [B, 6], got {tuple(p_dev.shape)}")
        p_dev = p_dev.to(dtype=torch.float32).contiguous()
    B, n_trials = int(p_dev.shape[0]), int(n_trials)
    if n_trials <= 0:
        raise ValueError("n_trials must be positive")
    if seed is None or set_offset is None:
        s_seed, s_off = (stream_state or GLOBAL_STREAM).take(B)
        seed = s_seed if seed is None else seed
        set_offset = s_off if set_offset is None else set_offset
    seed, set_offset = int(seed) & 0xFFFFFFFFFFFFFFFF, int(set_offset) & 0xFFFFFFFFFFFFFFFF
    fast = DEFAULT_FAST if fast is None else bool(fast)
    with torch.cuda.device(dev):
        if want_trials and out_trials is None:
            out_trials = torch.empty((B, n_trials, 2), dtype=torch.float32, device=dev)
        if want_summary and out_summary is None:
            out_summary = torch.empty((B, SUMMARY_K), dtype=torch.float32, device=dev)
        out_ext = torch.empty((B,), dtype=torch.float32, device=dev) if want_ext else None
        for t, shape in ((out_trials, (B, n_trials, 2)), (out_summary, (B, SUMMARY_K))):
            if t is not None and (tuple(t.shape) != shape or t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda):
                raise ValueError(f"output buffer must be a contiguous float32 device tensor of shape {shape}")
        pt = lambda t: None if t is None else t.data_ptr()
        if B > 0:
            _lib.check(L.nddm_simulratcliff(pt(p_dev), B, n_trials, seed, set_offset, _lib.GAUSS_FAST if fast else _lib.GAUSS_EXACT,
                                            float(ext_sigma), int(ext_mode), pt(out_trials), pt(out_summary), pt(out_ext),
                                            torch.cuda.current_stream(dev).cuda_stream))
            p_dev.record_stream(torch.cuda.current_stream(dev))
    res = {"seed": seed, "set_offset": set_offset, "params": p_dev}
    for k, v in (("trials", out_trials), ("summary", out_summary), ("ext", out_ext)):
        if v is not None:
            res[k] = v
    return res


def decode_codes(model, codes, params, dt, out_trials=None):
    """The 2-byte wire format back to the float pairs the simulator writes: codes int16 [B, n_trials] (uint16 content), params
    f32 [B, P] (tau is read from them), -> f32 [B, n_trials, 2], bit-identical to simulate()'s 'trials' (nddm_decode_codes)."""
    torch = require_device()
    B, n_trials = int(codes.shape[0]), int(codes.shape[1])
    if out_trials is None:
        out_trials = torch.empty((B, n_trials, 2), dtype=torch.float32, device=codes.device)
    p = params.to(dtype=torch.float32).contiguous()
    with torch.cuda.device(codes.device):
        _lib.check(_lib.lib().nddm_decode_codes(model, codes.contiguous().data_ptr(), p.data_ptr(), B, n_trials, float(dt),
                                                out_trials.data_ptr(), torch.cuda.current_stream(codes.device).cuda_stream))
    return out_trials


def draw_prior_device(model, batch_size, seed=0, set_offset=0, gamma=1.0, device=None, set_offset_dev=None, out=None):
    """On-device batched draw_prior (basic_ddm_dc.py:62-80 / single_trial_alpha_not_scaled.py:78-102): f32 [B, P].
    set_offset_dev: as in simulate() (nddm_draw_prior_indirect)."""
    torch = require_device()
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    P = NPARAMS[model]
    with torch.cuda.device(dev):
        if out is None:
            out = torch.empty((int(batch_size), P), dtype=torch.float32, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        if set_offset_dev is not None:
            rc = _lib.lib().nddm_draw_prior_indirect(model, int(batch_size), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                                     int(set_offset) & 0xFFFFFFFFFFFFFFFF, set_offset_dev.data_ptr(),
                                                     float(gamma), out.data_ptr(), st)
        else:
            rc = _lib.lib().nddm_draw_prior(model, int(batch_size), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                            int(set_offset) & 0xFFFFFFFFFFFFFFFF, float(gamma), out.data_ptr(), st)
        _lib.check(rc)
    return out


PINNED_FROM_BYTES = 1 << 20
# NDDM_PINNED_RESULTS=0 (or engine.PINNED_RESULTS = False): results come back in ordinary pageable memory (`.cpu()`), for callers
# that KEEP many large results -- see to_host()
PINNED_RESULTS = os.environ.get("NDDM_PINNED_RESULTS", "1") not in ("0", "false", "no")


def _pinned(pinned):
    return PINNED_RESULTS if pinned is None else bool(pinned)


def release_pinned_cache():
    """Hand the pinned host blocks of results that were dropped back to the OS (PyTorch caches them for reuse otherwise)."""
    torch = _torch()
    fn = getattr(torch._C, "_host_emptyCache", None)
    if fn is not None:
        fn()
        return True
    return False


def to_host(t, pinned=None):
    """Device tensor -> NumPy array (what the adapters' `as_numpy` forms return).  A result of a megabyte or more goes through
    PINNED host memory: the copy then runs at the link's rate (53 GB/s measured on the MI355X box against 6.4 GB/s into pageable
    memory -- 2.4 GB of trials in 45 ms instead of 380, `profiles/r4_pcie_rate.txt`).  The array owns its block (it returns to
    PyTorch's pinned-memory cache when the array is dropped).

    What that costs: PyTorch's pinned allocator rounds a block up to a power of two (the 2.4 GB headline result pins 4 GB) and
    keeps dropped blocks cached, so a caller that holds on to MANY large results (imputation loops, generate_data) can run out of
    lockable memory where `.cpu()` would not.  `pinned=False` / NDDM_PINNED_RESULTS=0 selects the pageable path;
    release_pinned_cache() returns the cached blocks of dropped results to the OS."""
    torch = _torch()
    if not isinstance(t, torch.Tensor):
        return np.asarray(t)
    if not t.is_cuda:
        return t.numpy()
    if t.numel() * t.element_size() < PINNED_FROM_BYTES or not _pinned(pinned):
        return t.cpu().numpy()
    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    h.copy_(t)
    return h.numpy()


HOST_CHUNK_BYTES = 128 << 20


def simulate_to_host(model, params, n_trials, seed=None, set_offset=None, stream_state=None, bounds=None, want_trials=True,
                     want_summary=True, want_ext=False, device=None, chunk_bytes=None, pinned=None, **kw):
    """simulate() for a caller who wants NumPy arrays back (the adapters' `as_numpy` forms): {'trials', 'summary', 'ext'} as
    float32 arrays in pinned host memory (pinned=False / NDDM_PINNED_RESULTS=0: pageable memory, see to_host), plus 'seed' /
    'set_offset'.

    A small batch is one launch and one copy.  A large one (more than `chunk_bytes` of trials, default 128 MB) is simulated in
    CHUNKS of parameter sets, and every chunk's results travel to the host on a second stream while the next chunk is simulated:
    the sets' random streams are keyed by their global index (set_offset + row), so the chunks reproduce the one launch bit for
    bit, the device holds two chunks instead of the whole result, and the 2.4 GB of the 1M x 300 workload are on the host
    47 ms after the call instead of 77 (one launch, then the copy) or 400 (`.cpu()`): profiles/r4_pcie_rate.txt."""
    torch = require_device()
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    on_host = not (isinstance(params, torch.Tensor) and params.is_cuda)
    if on_host:
        p_np = np.ascontiguousarray(params.detach().cpu().numpy() if isinstance(params, torch.Tensor) else params, dtype=np.float64)
        if p_np.ndim == 1:
            p_np = p_np[None]
        if p_np.ndim != 2 or p_np.shape[1] != NPARAMS[model]:
            raise ValueError(f"params must have shape [B, {NPARAMS[model]}] for this model, got {p_np.shape}")
        validate_params_host(model, p_np)
        params = torch.as_tensor(p_np, dtype=torch.float32).contiguous().to(dev)
    elif params.ndim == 1:
        params = params[None]
    B, n_trials = int(params.shape[0]), int(n_trials)
    if seed is None or set_offset is None:
        s_seed, s_off = (stream_state or GLOBAL_STREAM).take(B)
        seed = s_seed if seed is None else seed
        set_offset = s_off if set_offset is None else set_offset
    seed, set_offset = int(seed) & 0xFFFFFFFFFFFFFFFF, int(set_offset) & 0xFFFFFFFFFFFFFFFF
    if model == EXPLICIT_BOUNDARY and bounds is not None and not (isinstance(bounds, torch.Tensor) and bounds.is_cuda):
        b_np = np.ascontiguousarray(np.asarray(bounds, dtype=np.float64).reshape(B, n_trials))
        if np.any(b_np < 0) or not np.all(np.isfinite(b_np)):
            raise ValueError("Trial-level boundary cannot be less than zero")
        bounds = torch.as_tensor(b_np, dtype=torch.float32).contiguous().to(dev)
    elif bounds is not None and isinstance(bounds, torch.Tensor):
        bounds = bounds.reshape(B, n_trials)
    want_ext = bool(want_ext and model == ALPHA_NOT_SCALED)
    row_bytes = n_trials * 8 if want_trials else 4 * SUMMARY_K
    chunk_bytes = HOST_CHUNK_BYTES if chunk_bytes is None else int(chunk_bytes)
    rows = B if B * row_bytes <= chunk_bytes or not want_trials else max(1, chunk_bytes // row_bytes)
    common = dict(seed=seed, want_trials=want_trials, want_summary=want_summary, want_ext=want_ext, device=dev, **kw)
    res = {"seed": seed, "set_offset": set_offset}
    if rows >= B:                                            # one launch, one copy per output
        r = simulate(model, params, n_trials, set_offset=set_offset, bounds=bounds, **common)
        for k in ("trials", "summary", "ext"):
            if k in r:
                res[k] = to_host(r[k], pinned)
        return res
    with torch.cuda.device(dev):
        pin = lambda *shape: torch.empty(shape, dtype=torch.float32, pin_memory=_pinned(pinned))
        host = {"trials": pin(B, n_trials, 2) if want_trials else None, "summary": pin(B, SUMMARY_K) if want_summary else None,
                "ext": pin(B) if want_ext else None}
        cur, side = torch.cuda.current_stream(dev), torch.cuda.Stream(device=dev)
        bufs, copied = [None, None], [None, None]
        for c, lo in enumerate(range(0, B, rows)):
            hi, b = min(B, lo + rows), c & 1
            if copied[b] is not None:
                cur.wait_event(copied[b])                   # this buffer set's previous chunk is on the host
            if bufs[b] is None:
                bufs[b] = {"trials": torch.empty((rows, n_trials, 2), dtype=torch.float32, device=dev) if want_trials else None,
                           "summary": torch.empty((rows, SUMMARY_K), dtype=torch.float32, device=dev) if want_summary else None}
            n = hi - lo
            tr = bufs[b]["trials"][:n] if want_trials else None
            sm = bufs[b]["summary"][:n] if want_summary else None
            r = simulate(model, params[lo:hi], n_trials, set_offset=(set_offset + lo) & 0xFFFFFFFFFFFFFFFF,
                         bounds=None if bounds is None else bounds[lo:hi], out_trials=tr, out_summary=sm, **common)
            ev = torch.cuda.Event()
            ev.record(cur)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                for k in ("trials", "summary", "ext"):
                    if host[k] is not None:
                        host[k][lo:hi].copy_(r[k], non_blocking=True)
                        r[k].record_stream(side)
                copied[b] = torch.cuda.Event()
                copied[b].record(side)
        side.synchronize()
    for k, h in host.items():
        if h is not None:
            res[k] = h.numpy()
    return res


def debug_normals(counters, k0, k1, fast=False):
    """4 normals per Philox counter row (tests compare these with the oracle's)."""
    torch = require_device()
    c = torch.as_tensor(np.asarray(counters, dtype=np.uint32).view(np.int32)).cuda().contiguous()
    n = c.shape[0]
    out = torch.empty((n, 4), dtype=torch.float32, device=c.device)
    rc = _lib.lib().nddm_debug_normals(c.data_ptr(), n, int(k0), int(k1), 1 if fast else 0, out.data_ptr(),
                                       torch.cuda.current_stream().cuda_stream)
    _lib.check(rc)
    return out.cpu().numpy()


class debug_trace:
    """Developer aid (profiling): `with debug_trace() as t: simulate(...)`, then `t.read()`.  While active, every wave of
    the simulator kernels stores one record {step-loop blocks, refill phases, s_memtime cycles, lifetime / start / queue
    found empty / end in 100 MHz ticks} and, with chunks > 0, the tick at which each chunk was pulled from the work queue
    (include/nddm.h: nddm_set_debug_trace; plain stores, so the traced launch runs like any other)."""

    def __init__(self, waves=16384, chunks=0, device=None):
        torch = require_device()
        self.waves, self.chunks = int(waves), int(chunks)
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.buf = torch.zeros(8 * self.waves + self.chunks, dtype=torch.int64, device=dev)

    def __enter__(self):
        _lib.check(_lib.lib().nddm_set_debug_trace(self.buf.data_ptr(), self.waves, self.chunks))
        return self

    def __exit__(self, *exc):
        require_device().cuda.synchronize()
        _lib.lib().nddm_set_debug_trace(None, 0, 0)
        return False

    def read(self):
        """dict: totals over the waves that ran (blocks, refills, cycles, ticks, waves), the per-wave records [n, 8] and
        the chunks' pull ticks."""
        d = self.buf.cpu().numpy()
        rec = d[:8 * self.waves].reshape(self.waves, 8)
        rec = rec[rec[:, 7] == 1]
        pulls = d[8 * self.waves:]
        return {"blocks": float(rec[:, 0].sum()), "refills": float(rec[:, 1].sum()), "cycles": float(rec[:, 2].sum()),
                "ticks": float(rec[:, 3].sum()), "waves": int(rec.shape[0]), "records": rec, "pulls": pulls[pulls > 0]}


def release_graph_memory():
    """Free the OWNERLESS memory behind captured launches on the current device: what launches captured with no graph
    arena bound were given (include/nddm.h: nddm_release_graph_memory).  Memory charged to a GraphArena is never touched."""
    _lib.check(_lib.lib().nddm_release_graph_memory())


class GraphArena:
    """Owner of the library memory behind captured launches (include/nddm.h: nddm_graph_arena_*).  Every launch captured
    into a hipGraph pins an allocation of its own (queue words + scratch); launches captured inside `with arena.bound():`
    are charged to this arena, and `release()` frees exactly those -- another owner's graphs keep replaying.  The holder
    destroys its graphs first, then releases."""

    def __init__(self):
        import ctypes
        h = ctypes.c_uint64(0)
        _lib.check(_lib.lib().nddm_graph_arena_create(ctypes.byref(h)))
        self.handle = int(h.value)

    def bound(self):
        """Context manager: captured launches of THIS thread are charged to the arena inside the block."""
        return _ArenaBinding(self)

    def info(self):
        import ctypes
        b, n = ctypes.c_uint64(0), ctypes.c_int32(0)
        _lib.check(_lib.lib().nddm_graph_arena_info(self.handle, ctypes.byref(b), ctypes.byref(n)))
        return {"bytes": int(b.value), "allocations": int(n.value)}

    @property
    def released(self):
        return self.handle == 0

    def release(self):
        """Free the arena's memory (idempotent).  Call after its graphs have been destroyed and the device is idle with
        respect to them."""
        if self.handle:
            h, self.handle = self.handle, 0
            _lib.check(_lib.lib().nddm_graph_arena_release(h))


class _ArenaBinding:
    def __init__(self, arena):
        self.arena, self.prev = arena, None

    def __enter__(self):
        import ctypes
        if self.arena.released:
            raise RuntimeError("this GraphArena has been released")
        prev = ctypes.c_uint64(0)
        _lib.check(_lib.lib().nddm_graph_arena_bind(self.arena.handle, ctypes.byref(prev)))
        self.prev = int(prev.value)
        return self.arena

    def __exit__(self, *exc):
        L = _lib.lib()
        if L.nddm_graph_arena_bind(self.prev, None) != 0:      # the outer owner was released meanwhile: no owner
            L.nddm_graph_arena_bind(0, None)
        return False


class graph_memory:
    """`with engine.graph_memory(): ...capture, replay, delete the graphs...`: an arena of its own is bound for the block
    and released on exit -- only what was captured INSIDE the block is freed (a GraphTrainer alive beside it, or an outer
    graph_memory block, keeps its memory).  A loop that re-captures -- one graph per n_trials bucket, say -- grows
    without an owner."""

    def __enter__(self):
        self.arena = GraphArena()
        self._binding = self.arena.bound()
        self._binding.__enter__()
        return self.arena

    def __exit__(self, *exc):
        self._binding.__exit__(*exc)
        require_device().cuda.synchronize()
        self.arena.release()
        return False


def last_launch():
    """Developer aid: geometry of this thread's last simulator launch (include/nddm.h: nddm_debug_last_launch)."""
    import ctypes
    out = (ctypes.c_int32 * 8)()
    _lib.check(_lib.lib().nddm_debug_last_launch(out))
    keys = ("grid_waves", "vgpr_keys", "ring", "tile_trials", "tiles_per_set", "sets_per_chunk", "refill_thresh", "lds_bytes")
    return dict(zip(keys, (int(v) for v in out)))

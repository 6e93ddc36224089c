"""MI355X-native Euler-Maruyama drift-diffusion trial simulators behind the generative-model interface of
mdnunez/bayesflow_nddms (basic_ddm_dc / single_trial_alpha_not_scaled / alpha_not_scaled).

Host code is Python; the arithmetic lives in hand-written HIP kernels for gfx950 reached through a C ABI
(include/nddm.h).  There is no CPU fallback: without the built HIP library and a ROCm device the simulators raise.
"""
from .engine import (ALPHA_NOT_SCALED, BASIC_DDM_DC, EXPLICIT_BOUNDARY, SINGLE_TRIAL, SINGLE_TRIAL_ALT, SUMMARY_COLS,
                     SUMMARY_K, GLOBAL_STREAM, StreamState, draw_prior_device, seed, simulate)

__all__ = ["ALPHA_NOT_SCALED", "BASIC_DDM_DC", "EXPLICIT_BOUNDARY", "SINGLE_TRIAL", "SINGLE_TRIAL_ALT",
           "SUMMARY_COLS", "SUMMARY_K", "GLOBAL_STREAM", "StreamState", "draw_prior_device", "seed", "simulate"]
__version__ = "0.1.0"

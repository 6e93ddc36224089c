"""Drop-in for the choice-RT imputation loop of the reference's imputation_from_stahl_not_scaled.py: one trial per
EXPLICIT boundary value (the single-trial boundaries derived from EEG), `diffusion_trial(drift, bound_trial, beta,
ter, dc)` (lines 120-148) called once per trial (lines 207-213).  Here the whole vector is one kernel launch.
"""
import numpy as np

from . import engine


def diffusion_trial(drift, bound_trial, beta, ter, dc, dt=.01, max_steps=400., seed=None, set_offset=None, fast=None):
    """One trial with the given boundary -> choicert (:120-148).  Raises ValueError for a negative boundary (:124-125)."""
    if bound_trial < 0:
        raise ValueError("Trial-level boundary cannot be less than zero")
    r = engine.simulate(engine.EXPLICIT_BOUNDARY, [[drift, beta, ter, dc]], 1, dt=dt, max_steps=max_steps, seed=seed,
                        set_offset=set_offset, fast=fast, bounds=[[bound_trial]], want_summary=False)
    return float(r["trials"][0, 0, 0])


def impute_choicert(drift, bounds, beta, ter, dc, dt=.01, max_steps=400., seed=None, set_offset=None, fast=None):
    """Vectorised form of the loop at :207-213.

    drift, beta, ter, dc: scalars or arrays [B] (one row per participant); bounds: [B, n_trials] (or [n_trials] for
    one participant).  Returns choicert float64 with the shape of `bounds`."""
    b = np.asarray(bounds, dtype=np.float64)
    one = b.ndim == 1
    if one:
        b = b[None]
    B = b.shape[0]
    P = np.stack([np.broadcast_to(np.asarray(v, dtype=np.float64), (B,)) for v in (drift, beta, ter, dc)], axis=1)
    if np.any(b < 0):
        raise ValueError("Trial-level boundary cannot be less than zero")
    r = engine.simulate(engine.EXPLICIT_BOUNDARY, P, b.shape[1], dt=dt, max_steps=max_steps, seed=seed,
                        set_offset=set_offset, fast=fast, bounds=b, want_summary=False)
    out = engine.to_host(r["trials"][..., 0]).astype(np.float64)
    return out[0] if one else out

"""EZ-diffusion estimates from choice-RT data -- the consumer of the per-simulation summaries.

Mirrors ``ezdiff(rt, correct, s=1.0)`` of the reference's simulations/Basic_DDM_simulations.py:131-158 (same argument
meaning, same edge corrections, same return order ``[drift, boundary, ndt]``), and adds the batched form that the fused
summaries make possible: :func:`ez_from_summary` turns the ``summary_stats [B, 10]`` array the simulators write next to
(or instead of) the trials into ``[B, 3]`` estimates without touching the 8 bytes per trial.

The closed form (Wagenmakers, van der Maas & Grasman 2007): with Pc the proportion correct, VRT and MRT the variance and
mean of the correct response times and L = logit(Pc),

    v   = sign(Pc - 1/2) * s * ( L (L Pc^2 - L Pc + Pc - 1/2) / VRT )^(1/4)
    a   = s^2 L / v
    Ter = MRT - (a / 2v) (1 - e^y) / (1 + e^y),     y = -v a / s^2.

Edge corrections as in the reference (:141-145): Pc = 1 becomes 1 - 1/(2n), Pc = 1/2 becomes 1/2 + 1/(2n), n = number
of trials including missing ones.  "Correct" is the upper boundary (choice 1), missing trials (timeouts) count in n only.
"""
import numpy as np

from .engine import SUMMARY_COLS

_COL = {name: i for i, name in enumerate(SUMMARY_COLS)}


def _ez_core(pc, n, mrt, vrt, s, xp):
    """Vectorised closed form; every argument broadcastable, xp = numpy or torch.  Undefined cases give NaN."""
    one_half = 0.5
    half_trial = 1.0 / (2.0 * n)
    pc = xp.where(pc == 1.0, 1.0 - half_trial, pc)
    pc = xp.where(pc == one_half, one_half + half_trial, pc)
    logit = xp.log(pc / (1.0 - pc))
    x = logit * (logit * pc * pc - logit * pc + pc - one_half) / vrt
    drift = xp.sign(pc - one_half) * s * x ** 0.25
    boundary = s * s * logit / drift
    y = -drift * boundary / (s * s)
    ey = xp.exp(y)
    mdt = boundary / (2.0 * drift) * (1.0 - ey) / (1.0 + ey)
    return drift, boundary, mrt - mdt


def ezdiff(rt, correct, s=1.0):
    """``[drift, boundary, ndt]`` from response times and correctness (1 / 0 / NaN = missing), as the reference's ezdiff
    (simulations/Basic_DDM_simulations.py:131-158); its assertions are ValueErrors here."""
    rt = np.asarray(rt, dtype=np.float64)
    correct = np.asarray(correct, dtype=np.float64)
    if rt.size == 0 or rt.shape != correct.shape:
        raise ValueError("rt and correct must be non-empty and of equal length")
    if np.nanmax(correct) > 1 or np.nanmin(correct) < 0:
        raise ValueError("correct must lie in [0, 1]")
    pc = np.nanmean(correct)
    if not pc > 0:
        raise ValueError("no correct response: the EZ equations need Pc > 0")
    hits = rt[correct == 1]
    mrt, vrt = np.nanmean(hits), np.nanvar(hits)
    if not vrt > 0:
        raise ValueError("the variance of the correct response times must be positive")
    drift, boundary, ndt = _ez_core(np.float64(pc), float(correct.size), mrt, vrt, float(s), np)
    return [float(drift), float(boundary), float(ndt)]


def ez_from_summary(summary, s=1.0):
    """Batched EZ estimates ``[B, 3]`` = (drift, boundary, ndt) from the simulators' fused ``summary_stats [B, 10]``
    (NumPy array or torch tensor, any device; column names in engine.SUMMARY_COLS).  Rows for which the estimator is
    undefined (no correct response, no variance) are NaN -- a batch does not raise."""
    is_torch = hasattr(summary, "device") and not isinstance(summary, np.ndarray)
    if is_torch:
        import torch as xp
        sm = summary.to(xp.float64)
        stack = lambda cols: xp.stack(cols, dim=-1)
    else:
        xp = np
        sm = np.asarray(summary, dtype=np.float64)
        stack = lambda cols: np.stack(cols, axis=-1)
    n_up, n_lo, n_miss = sm[..., _COL["n_upper"]], sm[..., _COL["n_lower"]], sm[..., _COL["n_missing"]]
    n_resp = n_up + n_lo
    nan = float("nan")
    pc = xp.where(n_resp > 0, n_up / xp.where(n_resp > 0, n_resp, n_resp + 1.0), n_resp * nan)   # nanmean over responded trials
    vrt = sm[..., _COL["var_rt_upper"]]
    ok = (n_up > 0) & (vrt > 0)
    pc = xp.where(ok, pc, pc * nan)
    if is_torch:
        drift, boundary, ndt = _ez_core(pc, n_resp + n_miss, sm[..., _COL["mean_rt_upper"]], vrt, float(s), xp)
    else:
        with np.errstate(all="ignore"):                  # undefined rows are NaN by design
            drift, boundary, ndt = _ez_core(pc, n_resp + n_miss, sm[..., _COL["mean_rt_upper"]], vrt, float(s), xp)
    return stack([drift, boundary, ndt])


def accuracy_rt_moments(summary):
    """``(mean_accuracy, mean_rt, var_rt)`` per parameter set from the fused ``summary_stats [B, 10]`` -- the three numbers
    the reference's parameter sweeps plot (simulations/mean_RT_accuracy_effects.py:88-90: ``np.nanmean(correct)``,
    ``np.nanmean(rts)``, ``np.nanvar(rts)``; missing trials are NaN there, i.e. excluded).  NumPy array or torch tensor."""
    n_up, n_lo = summary[..., _COL["n_upper"]], summary[..., _COL["n_lower"]]
    return n_up / (n_up + n_lo), summary[..., _COL["mean_rt"]], summary[..., _COL["var_rt"]]

"""Host-side distribution diagnostics: histograms on the integer Euler-Maruyama step grid and the
Kolmogorov-Smirnov distance of signed response times (sign = choice, 0 = missing response).

Comparing on the integer step index avoids float32-vs-float64 tie artefacts: rt = k*dt + tau, so the
signed RT is a monotone function of choice*k and the KS distance of signed RTs is the KS distance of choice*k.
"""
import numpy as np


def step_hist_from_trials(trials, tau, dt, max_k, signed=False):
    """trials [..., 2] -> int64 hist [3, max_k+1] (rows: upper, lower, timeout) over the step index k.

    signed=False: columns (rt, choice) as basic_ddm_dc.py:124; signed=True: column 0 is choicert = +-(ter + rt) or 0
    (single_trial_alpha_not_scaled.py:136-141)."""
    t = np.asarray(trials, dtype=np.float64)
    if np.ndim(tau) > 0:                      # one non-decision time per set: trials is [B, N, 2]
        tau = np.broadcast_to(np.asarray(tau, dtype=np.float64).reshape(-1, 1), t.shape[:2]).reshape(-1)
    else:
        tau = float(tau)
    t = t.reshape(-1, 2)
    if signed:
        choice = np.sign(t[:, 0])
        k = np.where(choice == 0, max_k, np.rint((np.abs(t[:, 0]) - tau) / dt)).astype(np.int64)
    else:
        choice = t[:, 1]
        k = np.rint((t[:, 0] - tau) / dt).astype(np.int64)
    k = np.clip(k, 0, max_k)
    hist = np.zeros((3, max_k + 1), dtype=np.int64)
    for row, c in ((0, 1), (1, -1), (2, 0)):
        hist[row] = np.bincount(k[choice == c], minlength=max_k + 1)
    return hist


def signed_cdf(hist):
    """CDF over positions -K..K of choice*k (timeouts at 0)."""
    h = np.asarray(hist, dtype=np.float64)
    K = h.shape[1] - 1
    pmf = np.zeros(2 * K + 1)
    pmf[:K + 1] += h[1][::-1]          # lower boundary: position -k
    pmf[K] += h[2].sum()               # missing responses: position 0
    pmf[K:] += h[0]                    # upper boundary: position +k
    return np.cumsum(pmf) / pmf.sum()


def ks_signed(hist_a, hist_b):
    """Two-sample KS distance of the signed step index (== of the signed RT)."""
    return float(np.max(np.abs(signed_cdf(hist_a) - signed_cdf(hist_b))))


def ks_conditional(hist_a, hist_b, row):
    """KS of the step index given the choice (row 0 upper, 1 lower)."""
    a, b = np.asarray(hist_a[row], float), np.asarray(hist_b[row], float)
    if a.sum() == 0 or b.sum() == 0:
        return 0.0
    return float(np.max(np.abs(np.cumsum(a) / a.sum() - np.cumsum(b) / b.sum())))


def choice_probs(hist):
    h = np.asarray(hist, dtype=np.float64)
    return h.sum(axis=1) / h.sum()


def ks_quantile_table(sample, q_table):
    """KS distance between an empirical sample and a reference given as a dense quantile table
    (q_table[i] = quantile i/(len-1)).  Used for continuous outputs (z1, simulratcliff RTs)."""
    s = np.sort(np.asarray(sample, dtype=np.float64))
    q = np.asarray(q_table, dtype=np.float64)
    n, m = len(s), len(q) - 1
    # reference CDF evaluated at the sample points by inverting the quantile table
    f_ref = np.interp(s, q, np.linspace(0.0, 1.0, m + 1), left=0.0, right=1.0)
    f_emp_hi = np.arange(1, n + 1) / n
    f_emp_lo = np.arange(0, n) / n
    return float(max(np.max(np.abs(f_emp_hi - f_ref)), np.max(np.abs(f_emp_lo - f_ref))))

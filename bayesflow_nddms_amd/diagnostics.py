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


# ------------------------------------------------------------------------------------------------------------------
# The numbers the reference's recovery plots print (the consumer of amortizer.sample on the other side of the path)
def recovery_statistics(theta_true, theta_est):
    """Per parameter, what `recovery_scatter` writes into its panels (pyhddmjagsutils.py:609-623): R^2 = sklearn's r2_score(true,
    estimate) -- 1 - SS_res / SS_tot about the mean of the TRUE values, negative when the estimates are worse than that mean -- and
    Pearson's rho.  theta_true, theta_est: [n_datasets, P] (the reference passes posterior MEANS: basic_ddm_dc.py:236-250).
    -> {'r2': [P], 'rho': [P]}."""
    t, e = np.asarray(theta_true, dtype=np.float64), np.asarray(theta_est, dtype=np.float64)
    if t.shape != e.shape or t.ndim != 2:
        raise ValueError(f"theta_true and theta_est must both be [n_datasets, P]; got {t.shape} and {e.shape}")
    ss_res = ((t - e) ** 2).sum(axis=0)
    ss_tot = ((t - t.mean(axis=0)) ** 2).sum(axis=0)
    with np.errstate(divide="ignore", invalid="ignore"):
        r2 = np.where(ss_tot > 0, 1.0 - ss_res / ss_tot, np.where(ss_res == 0, 1.0, 0.0))      # (r2_score's convention for a constant target)
        tc, ec = t - t.mean(axis=0), e - e.mean(axis=0)
        rho = (tc * ec).sum(axis=0) / np.sqrt((tc ** 2).sum(axis=0) * (ec ** 2).sum(axis=0))
    return {"r2": r2, "rho": rho}


def converged_fits(param_means, index=3, low=0.0, high=1.0):
    """The reference's "clearly good" filter (basic_ddm_dc.py:239-241, single_trial_alpha_not_scaled.py:326-328): model fits whose
    posterior MEAN of the non-decision time (parameter 3 in both models) lies inside (0, 1) -> boolean [n_datasets]."""
    m = np.asarray(param_means, dtype=np.float64)
    return (m[:, index] > low) & (m[:, index] < high)

"""Pure-Python/NumPy restatement of the reference simulators, as the reference itself runs when numba is absent
(one np.random.normal() per Euler-Maruyama step on the global legacy stream).

TEST INFRASTRUCTURE ONLY (see oracle/ddm_oracle.c header).  Used (i) by bench.py's cpu_baseline leg as the
"NumPy/Python reference algorithm" timing BASELINE.md asks for -- the reference's .py files never travel to the
GPU box -- and (ii) by tests on tiny cases as a third, independent statement of the algorithm.

Restates: basic_ddm_dc.py:85-125; single_trial_alpha_not_scaled.py:107-155.
"""
import numpy as np


def basic_diffusion_trial(drift, boundary, beta, tau, dc, dt=.01, max_steps=400.):
    """basic_ddm_dc.py:85-112; the timeout branch returns choice 0 (the reference leaves `choice` unbound there)."""
    n_steps = 0.
    evidence = boundary * beta
    while (evidence > 0) and (evidence < boundary) and (n_steps < max_steps):
        evidence += drift * dt + np.sqrt(dt) * dc * np.random.normal()
        n_steps += 1.0
    rt = n_steps * dt + tau
    if evidence >= boundary:
        choice = 1
    elif evidence <= 0:
        choice = -1
    else:
        choice = 0
    return rt, choice


def basic_simulate_trials(params, n_trials, dt=.01, max_steps=400.):
    """basic_ddm_dc.py:114-125."""
    drift, boundary, beta, tau, dc = params
    out = np.empty((n_trials, 2))
    for i in range(n_trials):
        out[i] = basic_diffusion_trial(drift, boundary, beta, tau, dc, dt, max_steps)
    return out


def single_diffusion_trial(drift, mu_alpha, beta, ter, std_alpha, dc, sigma1, dt=.01, max_steps=400., gamma=1.0):
    """single_trial_alpha_not_scaled.py:107-142 (gamma: _scale :1262 / _scale2 :1496)."""
    while True:
        bound_trial = mu_alpha + std_alpha * np.random.normal()
        if bound_trial > 0:
            break
    n_steps = 0.
    evidence = bound_trial * beta
    while (evidence > 0) and (evidence < bound_trial) and (n_steps < max_steps):
        evidence += drift * dt + np.sqrt(dt) * dc * np.random.normal()
        n_steps += 1.0
    rt = n_steps * dt
    extdata1 = np.random.normal(gamma * bound_trial, sigma1)
    if evidence >= bound_trial:
        choicert = ter + rt
    elif evidence <= 0:
        choicert = -ter - rt
    else:
        choicert = 0
    return choicert, extdata1


def single_simulate_trials(params, n_trials, dt=.01, max_steps=400., gamma=1.0):
    """single_trial_alpha_not_scaled.py:144-155 / :1710-1722 (dt=.001, max_steps=4000)."""
    out = np.empty((n_trials, 2))
    for i in range(n_trials):
        out[i] = single_diffusion_trial(*params[:7], dt=dt, max_steps=max_steps, gamma=gamma)
    return out

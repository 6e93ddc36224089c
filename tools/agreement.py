"""Developer aid: per-trial agreement of the fast Gaussian transform with the exact one on the same stream."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesflow_nddms_amd import engine, priors
B, N = 200_000, 300
p = torch.as_tensor(priors.basic_prior_matrix(B, 2023)).cuda()
for dt, ms in ((0.001, 4000), (0.01, 400)):
    a = engine.simulate(0, p, N, dt=dt, max_steps=ms, seed=5, set_offset=0, fast=False)["trials"]
    b = engine.simulate(0, p, N, dt=dt, max_steps=ms, seed=5, set_offset=0, fast=True)["trials"]
    same = (a == b).all(dim=-1)
    dk = ((a[..., 0] - b[..., 0]).abs() / dt)[~same]
    print(f"dt={dt}: identical (rt, choice) in {same.float().mean().item():.6f} of {B*N} trials; "
          f"differing trials: median |dk| = {dk.median().item():.0f} steps, choice flips {(a[..., 1] != b[..., 1]).float().mean().item():.2e}")

#!/usr/bin/env python3
"""NDDM_STATE_F64 beside the default float32 state on the headline workload (1M x 300, dt=.001/4000) and at the reference default
(dt=.01/400): kernel time by events, per-trial agreement of (rt, choice).  Usage: python tools/f64_cost.py [sets]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bayesflow_nddms_amd import engine, priors  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
N = 300
for model, name, pm in ((engine.BASIC_DDM_DC, "basic", priors.basic_prior_matrix), (engine.SINGLE_TRIAL, "single", priors.single_prior_matrix)):
    p = torch.as_tensor(pm(B, 2023)).cuda()
    tr = torch.empty((B, N, 2), dtype=torch.float32, device="cuda")
    sm = torch.empty((B, 10), dtype=torch.float32, device="cuda")
    for dt, cap in ((0.001, 4000.0), (0.01, 400.0)):
        base = None
        for fast in (True, False):
            for f64 in (False, True):
                run = lambda i: engine.simulate(model, p, N, dt=dt, max_steps=cap, seed=2023, set_offset=i * B, fast=fast, state_f64=f64,
                                                out_trials=tr, out_summary=sm)
                run(0)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(3):
                    run(1 + i)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 3
                run(0)
                cur = tr[:20000].clone()
                if not f64:
                    base = cur
                agree = float((cur == base).all(dim=-1).float().mean().item()) if name == "basic" else float((cur[..., 0] == base[..., 0]).float().mean().item())
                print(f"{name:6s} dt={dt:<5g} {'fast ' if fast else 'exact'} {'f64 state' if f64 else 'f32 state'}: {ms:8.2f} ms  {B * N / ms / 1e6:8.2f} G trials/s"
                      f"   same (rt, choice) as the f32 state of this transform: {agree:.6f}", flush=True)

#!/usr/bin/env python3
"""nddm_simulratcliff at a fixed 3e8 trials per launch, cut into sets of different sizes: how much of the kernel's time is the tail of
a tile (lanes idle while the tile's last trials finish).  Usage: python tools/ratcliff_shapes.py [--one]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bayesflow_nddms_amd import engine, priors  # noqa: E402

from bayesflow_nddms_amd import _lib  # noqa: E402

print(f"# library source hash {_lib.lib().nddm_source_hash().decode()}", flush=True)
TOTAL = 300_000_000
SHAPES = (300,) if "--one" in sys.argv else (64, 128, 300, 512, 1024, 4096)       # --one: the bench leg's shape only (for the profiler)
for N in SHAPES:
    B = TOTAL // N
    p = torch.as_tensor(priors.alpha_ns_prior_matrix(B, 2023)).cuda()
    tr = torch.empty((B, N, 2), dtype=torch.float32, device="cuda")
    sm = torch.empty((B, 10), dtype=torch.float32, device="cuda")
    for fast in (True, False):
        run = lambda i: engine.simulratcliff(p, N, seed=2023, set_offset=i * B, fast=fast, out_trials=tr, out_summary=sm)
        run(0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(3):
            run(1 + i)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print(f"{B:9d} sets x {N:5d} trials, {'fast ' if fast else 'exact'}: {ms:7.2f} ms  {B * N / ms / 1e6:7.2f} G trials/s", flush=True)
    del p, tr, sm

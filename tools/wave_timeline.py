"""Developer aid: when do the simulator's waves start and end?  Runs one launch with the per-wave (start, end) trace of
nddm_set_debug_trace switched on and prints how many waves are alive over the kernel's duration (twentieths), the
spread of start and end times, and the share of wave-slot time that is empty at the head and at the tail.

usage: python tools/wave_timeline.py [model:B:N:dt:max_steps[:packed] ...]   (default: the headline shape at both step sizes)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import prior_util
from bayesflow_nddms_amd import engine, _lib

NW = 8192 * 2


def run(model, B, N, dt, ms, packed=False, tune=None):
    p = {0: prior_util.basic_prior, 1: prior_util.single_prior, 3: prior_util.alpha_ns_prior}[model](B, 2023)
    pd = torch.as_tensor(p).cuda()
    out = torch.empty((B, N, 2), dtype=torch.float32, device="cuda")
    summ = torch.empty((B, 10), dtype=torch.float32, device="cuda")
    kw = dict(dt=dt, max_steps=ms, set_offset=0, fast=True, out_trials=out, out_summary=summ, packed=packed)
    if tune:
        _lib.check(_lib.lib().nddm_set_tuning(*tune))
    engine.simulate(model, pd, N, seed=1, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with engine.debug_trace(waves=NW, chunks=1 << 20) as tr:
        e0.record()
        engine.simulate(model, pd, N, seed=2, **kw)
        e1.record()
    _lib.lib().nddm_set_tuning(0, 0, 0, 0, 0, 0)
    t = tr.read()
    rec = t["records"]
    nw = t["waves"]
    blocks = rec[:, 0]
    se = rec[:, 4:7].astype(np.float64) * 1e-5                           # start, queue dry, end in ms (100 MHz ticks)
    t0 = se[:, 0].min()
    st, dry, en = se[:, 0] - t0, se[:, 1] - t0, se[:, 2] - t0
    T = en.max()
    print(f"model={model} B={B} N={N} dt={dt} cap={int(ms)} packed={packed} tune={tune}: {nw} waves, events {e0.elapsed_time(e1):.3f} ms, first start -> last end {T:.3f} ms")
    q = [0, 1, 5, 25, 50, 75, 95, 99, 100]
    print("   start ms   percentiles " + " ".join(f"p{x}={np.percentile(st, x):.3f}" for x in q))
    print("   end ms     percentiles " + " ".join(f"p{x}={np.percentile(en, x):.3f}" for x in q))
    print("   queue dry  percentiles " + " ".join(f"p{x}={np.percentile(dry, x):.3f}" for x in q))
    print("   end - dry  percentiles " + " ".join(f"p{x}={np.percentile(en - dry, x):.3f}" for x in q))
    print("   blocks per wave percentiles " + " ".join(f"p{x}={np.percentile(blocks, x):.0f}" for x in q) +
          f"   (corr. with workgroup id {np.corrcoef(np.arange(nw), blocks)[0, 1]:.2f})")
    print("   lifetime   percentiles " + " ".join(f"p{x}={np.percentile(en - st, x):.3f}" for x in q))
    edges = np.linspace(0, T, 21)
    alive = [(np.clip(np.minimum(en, b) - np.maximum(st, a), 0, None)).sum() / (b - a) for a, b in zip(edges[:-1], edges[1:])]
    print("   waves alive per twentieth of the kernel: " + " ".join(f"{a:.0f}" for a in alive))
    pulls = t["pulls"].astype(np.float64) * 1e-5 - t0                    # pull time of chunk c, in queue order
    nc = len(pulls)
    if nc:
        print(f"   {nc} chunks; pull time (ms) of the chunk at queue position 0 %, 10 %, ... 100 %: " +
              " ".join(f"{pulls[min(nc - 1, int(f * nc))]:.3f}" for f in np.linspace(0, 1, 11)))
        print("   position in the queue (% of chunks) reached at each twentieth of the kernel: " +
              " ".join(f"{100.0 * np.searchsorted(np.maximum.accumulate(pulls), b) / nc:.1f}" for b in edges[1:]))
    print(f"   empty wave-slot time: head {st.mean() / T:.3%}, tail {(T - en).mean() / T:.3%} of {nw} x {T:.3f} ms")


if __name__ == "__main__":
    specs = sys.argv[1:] or ["0:1000000:300:0.001:4000", "0:1000000:300:0.01:400"]
    for spec in specs:
        f = spec.split(":")
        flags = f[5].split(",") if len(f) > 5 else []
        tune = None
        for fl in flags:
            if fl.startswith("tune="):
                tune = tuple(int(x) for x in fl[5:].split("/"))
        run(int(f[0]), int(f[1]), int(f[2]), float(f[3]), float(f[4]), packed="packed" in flags, tune=tune)

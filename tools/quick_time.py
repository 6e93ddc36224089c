"""Developer aid: time the simulator kernels at a few sizes / tunings (not the bench contract).

usage: python tools/quick_time.py [case ...]      case = model:B:N:dt:max_steps[:flag,...]   flags: exact, notrials, lockstep, bridge
       python tools/quick_time.py                 the standard sweep
Prints trials/s, E-M steps/s, SIMD cycles per useful wave-block (at 2.4 GHz x 1024 SIMDs), lane efficiency, blocks per
refill, in-kernel clock and resident waves per SIMD (in-kernel wave lifetimes)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import prior_util
from bayesflow_nddms_amd import engine, _lib

LOCK = {0: [0.0, 50.0, 0.5, 0.3, 1.0], 1: [0.0, 50.0, 0.5, 0.3, 0.01, 1.0, 1.0, 1.0], 3: [0.0, 50.0, 0.5, 0.3, 0.0, 1.0]}


def run(B, N, dt, ms, fast=True, tune=None, model=0, reps=3, trials_out=True, lockstep=False, bridge=False, packed=False):
    p = {0: prior_util.basic_prior, 1: prior_util.single_prior, 3: prior_util.alpha_ns_prior}[model](B, 2023)
    if lockstep:   # every trial runs to the cap: all lanes busy, no refill -> pure step-loop cost
        p[:] = np.array(LOCK[model], dtype=np.float32)
        tune = tune or (1, 0, 64, 1, 32 * torch.cuda.get_device_properties(0).multi_processor_count, 0)   # LDS-keys variant, full grid
    pd = torch.as_tensor(p).cuda()
    if tune:
        _lib.check(_lib.lib().nddm_set_tuning(*tune))
    out = torch.empty((B, N, 2), dtype=torch.float32, device="cuda") if trials_out else None
    summ = torch.empty((B, 10), dtype=torch.float32, device="cuda")
    kw = dict(dt=dt, max_steps=ms, set_offset=0, fast=fast, out_trials=out, out_summary=summ, want_trials=trials_out, bridge=bridge, packed=packed)
    engine.simulate(model, pd, N, seed=1, **kw)
    torch.cuda.synchronize()
    with engine.debug_trace() as tr:
        engine.simulate(model, pd, N, seed=2, **kw)
    t = tr.read()
    rec = t["records"]
    d = [t["blocks"], t["refills"], t["cycles"], t["ticks"], t["waves"]]
    span_ms = float(rec[:, 6].max() - rec[:, 4].min()) * 1e-5
    best = 1e9
    for r in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        engine.simulate(model, pd, N, seed=2 + r, **kw)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    s = summ.cpu().numpy()
    nresp = s[:, 0] + s[:, 1]
    tau = p[:, 3] - (0.5 * dt if bridge else 0.0)
    steps = float(((s[:, 3] - tau) / dt * nresp)[nresp > 0].sum() + s[:, 2].sum() * int(ms))
    spb = 8 if (packed or bridge) else 4
    cyc = best * 1e-3 * 2.4e9 * 1024 / (steps / 256)            # per 4 steps x 64 lanes, whatever the block size
    resident = d[3] * 1e-5 / span_ms / 1024.0
    print(f"model={model} B={B} N={N} dt={dt} cap={int(ms)} fast={fast} tune={tune} trials_out={trials_out} lockstep={lockstep} bridge={bridge} packed={packed}: "
          f"{best:.3f} ms  {B*N/best*1e3:.3e} trials/s  {steps/best*1e3:.3e} steps/s  {cyc:.0f} cyc/useful-block | lane-eff {steps/(d[0]*64*spb):.3f} "
          f"blocks/refill {d[0]/max(d[1],1):.1f} clock {d[2]/max(d[3],1)*0.1:.3f} GHz waves {d[4]:.0f} resident/SIMD {resident:.2f}", flush=True)
    _lib.lib().nddm_set_tuning(0, 0, 0, 0, 0, 0)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        for spec in sys.argv[1:]:
            f = spec.split(":")
            flags = f[5].split(",") if len(f) > 5 else []
            tune = None
            for fl in flags:
                if fl.startswith("tune="):
                    tune = tuple(int(x) for x in fl[5:].split("/"))
            run(int(f[1]), int(f[2]), float(f[3]), float(f[4]), fast="exact" not in flags, model=int(f[0]), trials_out="notrials" not in flags,
                lockstep="lockstep" in flags, bridge="bridge" in flags, packed="packed" in flags, tune=tune)
        sys.exit(0)
    B = 1000000
    run(B, 300, 0.001, 4000)
    run(B, 300, 0.001, 4000, fast=False)
    run(B, 300, 0.01, 400)
    run(B, 60, 0.01, 400)
    run(B, 300, 0.001, 4000, model=1)
    run(B, 300, 0.01, 400, model=1)
    run(B, 300, 0.001, 4000, model=3)
    run(B, 300, 0.001, 4000, model=3, bridge=True)
    run(B, 60, 0.001, 4000)
    for b in (3000000, 300000, 100000, 50000, 30000, 20000, 10000, 5000, 3000, 1000):
        run(b, 300, 0.001, 4000)
    for b in (300000, 100000, 30000, 10000, 3000, 1000):
        run(b, 300, 0.01, 400)
    run(100000, 60, 0.01, 400)
    run(32, 180, 0.01, 400); run(32, 180, 0.001, 4000); run(256, 180, 0.01, 400)
    run(40000, 300, 0.001, 4000, lockstep=True)     # every lane busy: pure step-loop cost
    run(400000, 300, 0.01, 400, lockstep=True)
    run(40000, 300, 0.001, 4000, lockstep=True, model=1)

"""Developer aid: time the basic_ddm_dc kernel at a few sizes / tunings (not the bench contract)."""
import sys
import numpy as np
import torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import prior_util
from bayesflow_nddms_amd import engine, _lib

def run(B, N, dt, ms, fast, tune=None, model=0, reps=3, trials_out=True, lockstep=False, bridge=False):
    p = {0: prior_util.basic_prior, 1: prior_util.single_prior, 3: prior_util.alpha_ns_prior}[model](B, 2023)
    if lockstep == 'typical':   # identical, typical parameters: no slow sets
        p[:] = np.array([1.5, 1.2, 0.5, 0.35, 1.0], dtype=np.float32)
    elif lockstep:   # every trial runs to the cap: all lanes busy, no refill -> pure step-loop cost
        p[:] = np.array([0.0, 50.0, 0.5, 0.3, 1.0], dtype=np.float32)
    pd = torch.as_tensor(p).cuda()
    if tune:
        _lib.check(_lib.lib().nddm_set_tuning(*tune))
    out = torch.empty((B, N, 2), dtype=torch.float32, device="cuda") if trials_out else None
    summ = torch.empty((B, 10), dtype=torch.float32, device="cuda")
    engine.simulate(model, pd, N, dt=dt, max_steps=ms, seed=1, set_offset=0, fast=fast, out_trials=out, out_summary=summ, want_trials=trials_out, bridge=bridge)
    torch.cuda.synchronize()
    dbg = torch.zeros(8, dtype=torch.int64, device='cuda')
    _lib.lib().nddm_set_debug_counters(dbg.data_ptr())
    engine.simulate(model, pd, N, dt=dt, max_steps=ms, seed=2, set_offset=0, fast=fast, out_trials=out, out_summary=summ, want_trials=trials_out, bridge=bridge)
    torch.cuda.synchronize()
    _lib.lib().nddm_set_debug_counters(None)
    d = dbg.cpu().numpy().astype(float)
    best = 1e9
    for r in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        engine.simulate(model, pd, N, dt=dt, max_steps=ms, seed=2 + r, set_offset=0, fast=fast, out_trials=out, out_summary=summ, want_trials=trials_out, bridge=bridge)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    s = summ.cpu().numpy()
    nresp = s[:, 0] + s[:, 1]
    steps = float(((s[:, 3] - p[:, 3 if model != 4 else 2]) / dt * nresp)[nresp > 0].sum() + s[:, 2].sum() * int(ms)) if not bridge else float("nan")
    cyc = best * 1e-3 * 2.35e9 * 1024 / (steps / 256) if steps == steps else float("nan")
    print(f"model={model} B={B} N={N} dt={dt} fast={fast} tune={tune} trials_out={trials_out} lockstep={lockstep} bridge={bridge}: {best:.2f} ms  {B*N/best*1e3:.3e} trials/s  {steps/best*1e3:.3e} steps/s  ~{cyc:.0f} SIMD-cycles/wave-block@2.35GHz | lane-eff {steps/(d[0]*256):.3f} blocks/refill {d[0]/d[1]:.1f} clock {d[2]/d[3]*0.1:.3f} GHz waves {d[4]:.0f}", flush=True)
    _lib.lib().nddm_set_tuning(0, 0, 0, 0, 0, 0)

if __name__ == "__main__":
    # usage: python tools/quick_time.py [sets]   -- throughput sweep over models / step sizes / batch sizes
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    run(B, 300, 0.001, 4000, True)
    run(B, 300, 0.001, 4000, False)
    run(B, 300, 0.01, 400, True)
    run(B, 300, 0.001, 4000, True, model=1)
    run(B, 300, 0.001, 4000, True, model=3)
    run(B, 300, 0.001, 4000, True, model=3, bridge=True)
    run(B, 60, 0.001, 4000, True)
    for b in (3000000, 300000, 100000, 30000, 10000, 1000):
        run(b, 300, 0.001, 4000, True)
    run(40000, 300, 0.001, 4000, True, (1, 0, 64, 64, 0, 0), lockstep=True)     # every lane busy: pure step-loop cost

"""Developer aid: wall-clock latency of the per-set drop-in calls (includes host validation, H2D, launch, D2H)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bayesflow_nddms_amd import basic_ddm_dc, engine
p = [1.5, 1.2, 0.5, 0.35, 1.0]
for _ in range(20):
    basic_ddm_dc.simulate_trials(p, 300)
for name, fn in (("simulate_trials(p, 300)  dt=.01", lambda: basic_ddm_dc.simulate_trials(p, 300)),
                 ("simulate_trials(p, 300)  dt=.001", lambda: basic_ddm_dc.simulate_trials(p, 300, dt=.001, max_steps=4000)),
                 ("generative_model(1) batched", None), ("generative_model(32) batched", None),
                 ("generative_model(32) per-set loop", None)):
    if fn is None:
        np.random.seed(1)
        if "per-set" in name:
            gm = basic_ddm_dc.make_generative_model(batched=False)
        else:
            gm = basic_ddm_dc.make_generative_model(batched=True)
        B = 1 if "(1)" in name else 32
        fn = lambda gm=gm, B=B: gm(B)
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 200
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print(f"{name:40s} {(time.perf_counter()-t0)/n*1e6:8.0f} us per call")
p_dev = torch.tensor([p] * 32, device="cuda")
out = torch.empty((32, 180, 2), device="cuda"); summ = torch.empty((32, 10), device="cuda")
for _ in range(10):
    engine.simulate(0, p_dev, 180, out_trials=out, out_summary=summ, seed=1, set_offset=0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(1000):
    engine.simulate(0, p_dev, 180, out_trials=out, out_summary=summ, seed=1, set_offset=0)
torch.cuda.synchronize()
print(f"{'engine.simulate device-resident 32x180':40s} {(time.perf_counter()-t0)/1000*1e6:8.0f} us per call")

"""The host-inclusive rate of the headline workload (DESIGN.md §6 "Host-inclusive"; HISTORY.md §C.6 "PCIe note"): simulate 1M sets x 300 trials on the device, then bring
the 2.4 GB of (rt, choice) pairs to the host -- what a caller pays who wants NumPy arrays back (the per-set drop-in form).  Three
forms: a copy into PINNED host memory, a copy into pageable memory (torch's `.cpu()`), and the copy CHUNKED and overlapped with the
simulation of the next chunk (two streams, pinned memory).  Never the bench line's `value`.   usage: python tools/pcie_rate.py [sets]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import prior_util
from bayesflow_nddms_amd import engine


def main():
    B, N = (int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000), 300
    p = torch.as_tensor(prior_util.basic_prior(B, 2023)).cuda()
    out = torch.empty((B, N, 2), dtype=torch.float32, device="cuda")
    summ = torch.empty((B, 10), dtype=torch.float32, device="cuda")
    pinned = torch.empty((B, N, 2), dtype=torch.float32).pin_memory()
    kw = dict(dt=0.001, max_steps=4000, set_offset=0, fast=True, out_summary=summ)

    def timed(fn, reps=3):
        fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        return best

    def kernel_only():
        engine.simulate(0, p, N, seed=1, out_trials=out, **kw)

    def to_pinned():
        engine.simulate(0, p, N, seed=1, out_trials=out, **kw)
        pinned.copy_(out, non_blocking=True)

    def to_pageable():
        engine.simulate(0, p, N, seed=1, out_trials=out, **kw)
        return out.cpu()

    chunks = 8
    cs = B // chunks
    side = torch.cuda.Stream()

    def chunked():
        evs = []
        for c in range(chunks):
            sl = slice(c * cs, (c + 1) * cs if c + 1 < chunks else B)
            engine.simulate(0, p[sl], N, seed=1, dt=0.001, max_steps=4000, set_offset=c * cs, fast=True, out_trials=out[sl], out_summary=summ[sl])
            ev = torch.cuda.Event(); ev.record()
            side.wait_event(ev)
            with torch.cuda.stream(side):
                pinned[sl].copy_(out[sl], non_blocking=True)
        torch.cuda.current_stream().wait_stream(side)

    from bayesflow_nddms_amd import basic_ddm_dc

    def adapter():                                   # the drop-in's batched form returning NumPy (chunks of 128 MB, pinned, copy beside compute)
        return basic_ddm_dc.batch_simulate_trials(p, N, dt=0.001, max_steps=4000, seed=1, set_offset=0, with_summary=False)["sim_data"]

    tk, tp, tg, tc, ta = timed(kernel_only), timed(to_pinned), timed(to_pageable, 2), timed(chunked), timed(adapter)
    gb = B * N * 8 / 1e9
    print(f"{B} sets x {N} trials, dt=.001 (basic_ddm_dc, fast), {gb:.2f} GB of (rt, choice) pairs")
    for name, t in (("device only (the bench line's form)", tk), ("+ copy to pinned host memory", tp), ("+ copy to pageable host memory (.cpu())", tg),
                    (f"{chunks} chunks, copy of chunk i beside the simulation of chunk i + 1 (pinned)", tc),
                    ("basic_ddm_dc.batch_simulate_trials(..., as_numpy=True): the adapter (engine.simulate_to_host)", ta)):
        print(f"  {name:82s} {t * 1e3:8.1f} ms   {B * N / t:.3e} trials/s" + ("" if t is tk else f"   copy alone ~ {gb / max(t - tk, 1e-9):.1f} GB/s"))
    chk = np.array_equal(pinned.numpy()[:1000], out[:1000].cpu().numpy())
    print("  chunked output == one-launch output (first 1000 sets):", chk)


if __name__ == "__main__":
    main()

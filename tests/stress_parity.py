"""Larger-scale bit parity of the GPU exact mode against the oracle than the small unit cases (ordering pre-pass
active, many queue chunks, ring wrap-around, split sets), every model, both bit layouts, default tuning.  `python tests/stress_parity.py`
runs all sizes (~80M trials, ~20 s on one MI355X + 16 host threads); tests/test_gpu_fuzz.py runs a subset."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402  (before the HIP library)
import oracle  # noqa: E402
import prior_util  # noqa: E402
from bayesflow_nddms_amd import engine  # noqa: E402

SIZES = ((20000, 300), (5000, 60), (3000, 700), (2049, 513), (40000, 64))


def params_for(model, B):
    if model == 0:
        return prior_util.basic_prior(B, 11)
    if model in (1, 2):
        p = prior_util.single_prior(B, 12, gamma=1.0)
        if model == 2:
            p[:, 4] = np.minimum(p[:, 4], 1.0)
        return p
    if model == 3:
        return prior_util.alpha_ns_prior(B, 13)
    return prior_util.basic_prior(B, 14)[:, [0, 2, 3, 4]]


def run(sizes=SIZES, models=range(5), verbose=True, threads=16):
    """Returns the list of mismatching cases (empty = parity)."""
    rng = np.random.default_rng(7)
    bad = []
    for model in models:
        for B, N in sizes:
            for bridge, packed in (((False, False), (False, True), (True, False)) if model == 3 else ((False, False), (False, True))):
                p = params_for(model, B)
                bounds = np.abs(rng.normal(1.2, 0.5, size=(B, N))).astype(np.float32) if model == 4 else None
                kw = dict(dt=0.001, max_steps=4000, seed=2024 + model, set_offset=123456789, bounds=bounds,
                          ext_sigma=0.2, ext_mode=0, want_ext=(model == 3), bridge=bridge, packed=packed)
                t0 = time.time()
                g = engine.simulate(model, p, N, fast=False, **kw)
                gt, gs = g["trials"].cpu().numpy(), g["summary"].cpu().numpy()
                t1 = time.time()
                o = oracle.philox_simulate(model, p, N, threads=threads, **kw)
                t2 = time.time()
                ok = (np.array_equal(gt.view(np.uint32), o["trials"].view(np.uint32))
                      and np.array_equal(np.nan_to_num(gs).view(np.uint32), np.nan_to_num(o["summary"]).view(np.uint32))
                      and (model != 3 or np.array_equal(g["ext"].cpu().numpy().view(np.uint32), o["ext"].view(np.uint32))))
                if not ok:
                    bad.append((model, B, N, bridge, packed))
                if verbose:
                    print(f"model {model} B={B} N={N} bridge={bridge} packed={packed}: {'OK' if ok else 'MISMATCH'}  ({B * N} trials; "
                          f"gpu {t1 - t0:.2f}s oracle {t2 - t1:.2f}s)", flush=True)
    # the device paths of round 6 at the same scale: NDDM_STATE_F64 (basic, single) against the float64 restatement, nddm_simulratcliff
    # against section D (hundreds of groups of sets per launch, tiled sets)
    for model in (m for m in (0, 1) if m in models):
        for B, N in sizes:
            p = params_for(model, B)
            kw = dict(dt=0.001, max_steps=4000, seed=3024 + model, set_offset=987654321)
            t0 = time.time()
            g = engine.simulate(model, p, N, fast=False, state_f64=True, **kw)
            gt, gs = g["trials"].cpu().numpy(), g["summary"].cpu().numpy()
            t1 = time.time()
            o = oracle.philox_simulate_f64(model, p, N, threads=threads, want_outputs=True, **kw)
            ok = (np.array_equal(gt.view(np.uint32), o["trials"].view(np.uint32))
                  and np.array_equal(np.nan_to_num(gs).view(np.uint32), np.nan_to_num(o["summary"]).view(np.uint32)))
            if not ok:
                bad.append((model, B, N, "state_f64"))
            if verbose:
                print(f"model {model} B={B} N={N} NDDM_STATE_F64: {'OK' if ok else 'MISMATCH'}  ({B * N} trials; gpu {t1 - t0:.2f}s oracle {time.time() - t1:.2f}s)", flush=True)
    if 3 in models:
        for B, N in sizes:
            p = params_for(3, B)
            t0 = time.time()
            g = engine.simulratcliff(p, N, seed=4024, set_offset=55555555555, fast=False, ext_sigma=0.2, ext_mode=0, want_ext=True)
            got = {k: g[k].cpu().numpy() for k in ("trials", "summary", "ext")}
            t1 = time.time()
            o = oracle.philox_ratcliff(p, N, seed=4024, set_offset=55555555555, ext_sigma=0.2, ext_mode=0, want_ext=True, threads=threads)
            ok = all(np.array_equal(np.nan_to_num(got[k]).view(np.uint32), np.nan_to_num(o[k]).view(np.uint32)) for k in got)
            if not ok:
                bad.append((3, B, N, "simulratcliff"))
            if verbose:
                print(f"nddm_simulratcliff B={B} N={N}: {'OK' if ok else 'MISMATCH'}  ({B * N} trials; gpu {t1 - t0:.2f}s oracle {time.time() - t1:.2f}s)", flush=True)
    return bad


if __name__ == "__main__":
    bad = run()
    print("ALL OK" if not bad else f"MISMATCHES: {bad}")
    sys.exit(1 if bad else 0)

"""A long run of the randomised bit-parity check (tests/test_gpu_fuzz.py draws 90 cases; this draws cases for a given number of
minutes from seeds the suite does not use): GPU exact mode vs the CPU oracle, every bit of trials, summaries and external datum.
usage: python tests/fuzz_long.py [minutes=8] [first_seed=5000]     (prints progress; exit code 1 at the first mismatch)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    import oracle
    import prior_util
    from test_gpu_fuzz import _case, round6_case
    from bayesflow_nddms_amd import _lib, engine
    t0, n, per_model, chunk, n6 = time.time(), 0, [0] * 5, first, 0
    try:
        while time.time() - t0 < 60 * minutes:
            rng = np.random.default_rng(chunk)
            for _ in range(10):
                model, B, N, dt, max_steps, seed, off, tune, bridge = _case(rng)
                packed = chunk % 3 == 2
                bridge = bridge and not packed
                pseed = int(rng.integers(0, 10**6))
                if model == 0:
                    p = prior_util.basic_prior(B, pseed)
                elif model in (1, 2):
                    p = prior_util.single_prior(B, pseed, gamma=float(rng.choice([1.0, 2.0, 0.3])))
                    if model == 2:
                        p[:, 4] = np.minimum(p[:, 4], 1.0)
                elif model == 3:
                    p = prior_util.alpha_ns_prior(B, pseed)
                else:
                    p = prior_util.basic_prior(B, pseed)[:, [0, 2, 3, 4]]
                bounds = np.abs(rng.normal(1.2, 0.5, size=(B, N))).astype(np.float32) if model == 4 else None
                _lib.check(_lib.lib().nddm_set_tuning(*tune))
                kw = dict(dt=dt, max_steps=max_steps, seed=seed, set_offset=off, bounds=bounds, ext_sigma=0.2, ext_mode=0,
                          want_ext=(model == 3), bridge=bridge, packed=packed)
                g = engine.simulate(model, p, N, fast=False, **kw)
                o = oracle.philox_simulate(model, p, N, threads=16, **kw)
                ctx = (chunk, model, B, N, dt, max_steps, seed, off, tune, bridge, packed)
                ok = (np.array_equal(g["trials"].cpu().numpy().view(np.uint32), o["trials"].view(np.uint32))
                      and np.array_equal(np.nan_to_num(g["summary"].cpu().numpy()).view(np.uint32), np.nan_to_num(o["summary"]).view(np.uint32))
                      and (model != 3 or np.array_equal(g["ext"].cpu().numpy().view(np.uint32), o["ext"].view(np.uint32))))
                if not ok:
                    print("MISMATCH", ctx, flush=True)
                    return 1
                n += 1
                per_model[model] += 1
            try:                                              # ... and one case of each round-6 path per chunk (NDDM_STATE_F64, nddm_simulratcliff)
                round6_case(rng)
                n6 += 2
            except AssertionError as e:
                print("MISMATCH (round-6 paths)", chunk, str(e)[:300], flush=True)
                return 1
            chunk += 1
            if (chunk - first) % 20 == 0:
                print(f"{n} cases bit-equal ({time.time() - t0:.0f} s)", flush=True)
    finally:
        _lib.lib().nddm_set_tuning(0, 0, 0, 0, 0, 0)
    print(f"fuzz_long: {n} cases (seeds {first}..{chunk - 1}; per model {per_model}) + {n6} cases of the round-6 paths (NDDM_STATE_F64 against the "
          f"float64 restatement, nddm_simulratcliff against section D) bit-equal to the oracle in {time.time() - t0:.0f} s: no mismatch")
    return 0


if __name__ == "__main__":
    sys.exit(main())

"""A battery over every entry point of the CPU oracle, printed as one sha256: run in-process against the normal build and in a child
process against an AddressSanitizer + UndefinedBehaviorSanitizer build of the same source (tests/test_oracle_golden.py)."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def battery():
    import oracle
    h = hashlib.sha256()

    def take(*arrays):
        for a in arrays:
            h.update(np.ascontiguousarray(a).tobytes())

    rng = np.random.default_rng(12)
    basic = np.stack([rng.normal(0, 2, 7), rng.uniform(.4, 2.5, 7), rng.uniform(.2, .8, 7), rng.uniform(.1, .9, 7), rng.uniform(.3, 2, 7)], 1).astype(np.float32)
    single = np.stack([rng.normal(0, 2, 5), rng.uniform(.6, 2, 5), rng.uniform(.2, .8, 5), rng.uniform(.1, .9, 5), rng.uniform(.1, 1.5, 5),
                       rng.uniform(.3, 2, 5), rng.uniform(.05, 3, 5), np.ones(5)], 1).astype(np.float32)
    alpha = np.stack([rng.uniform(-4, 4, 6), rng.uniform(.8, 1.4, 6), rng.uniform(.3, .7, 6), rng.uniform(.15, .6, 6), rng.uniform(0, 2, 6),
                      rng.uniform(.8, 1.4, 6)], 1).astype(np.float32)
    for n in (1, 3, 64, 257):                               # ragged trial counts, caps not divisible by the block size
        for dt, cap in ((0.01, 400.0), (0.001, 1003.0)):
            for packed in (False, True):
                r = oracle.philox_simulate(oracle.M_BASIC, basic, n, dt=dt, max_steps=cap, seed=5, set_offset=(1 << 40) + 3, packed=packed, want_k=True)
                take(r["trials"], r["k"], r["summary"])
            r = oracle.philox_simulate(oracle.M_SINGLE, single, n, dt=dt, max_steps=cap, seed=6, want_k=True)
            take(r["trials"], r["k"], r["summary"])
            r = oracle.philox_simulate(oracle.M_ALT, single, n, dt=dt, max_steps=cap, seed=7)
            take(r["trials"], r["summary"])
            for bridge in (False, True):
                r = oracle.philox_simulate(oracle.M_ALPHA_NS, alpha, n, dt=dt, max_steps=cap, seed=8, bridge=bridge, ext_sigma=0.1, want_ext=True)
                take(r["trials"], r["summary"], r["ext"])
            bounds = rng.uniform(0.3, 2.5, (4, n)).astype(np.float32)
            r = oracle.philox_simulate(oracle.M_EXPLICIT, basic[:4, [0, 2, 3, 4]], n, dt=dt, max_steps=cap, seed=9, bounds=bounds)
            take(r["trials"], r["summary"])
    take(oracle.philox_simulate(oracle.M_BASIC, basic, 1200, dt=0.01, max_steps=400.0, seed=1, threads=3)["summary"])      # > 512 trials per set, OpenMP
    take(*oracle.philox_simulate_f64(oracle.M_BASIC, basic, 50, dt=0.001, max_steps=4000.0, seed=2))
    for m, par in ((oracle.M_BASIC, basic), (oracle.M_SINGLE, single)):               # ... with the outputs NDDM_STATE_F64 writes
        f = oracle.philox_simulate_f64(m, par, 33, dt=0.01, max_steps=403.0, seed=3, want_outputs=True)
        take(f["k"], f["choice"], f["trials"], f["summary"])
    for n in (1, 64, 257, 700):                              # section D: simulratcliff on the device stream
        r = oracle.philox_ratcliff(alpha, n, seed=11, set_offset=(1 << 35) + 5, ext_sigma=0.2, ext_mode=n & 1, want_ext=True, threads=3)
        take(r["trials"], r["summary"], r["ext"])
    take(oracle.philox_normals4(1, 2, 3, 4, 5, 6), oracle.philox_block(1, 2, 3, 4, 5, 6))
    oracle.mt_seed(2023)
    take(np.array([oracle.mt_gauss(), oracle.mt_double()]), oracle.mt_basic([1.5, 1.2, .5, .35, 1.0], 40),
         oracle.mt_basic([0, 9, .5, .3, .2], 3), oracle.mt_single([3, 1.5, .5, .4, 1, 1, .1], 40, dt=0.001, max_steps=4000.0),
         oracle.mt_explicit(1.0, np.linspace(.5, 2, 30), .5, .3, 1.0), oracle.mt_ratcliff(N=50, Alpha=1.1, Tau=.3, Nu=2, Beta=.5, Eta=1.0, Varsigma=1.2))
    return h.hexdigest()


if __name__ == "__main__":
    print("BATTERY", battery())

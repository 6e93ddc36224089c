"""The drop-in boundary used from plain C (examples/c_abi_demo.c: include/nddm.h + the HIP runtime API, no Python and
no PyTorch in that process): its exact-mode output equals the CPU oracle's bit for bit."""
import json
import os
import subprocess

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "examples", "c_abi_demo")


@pytest.mark.parametrize("flags", [0, 1])
def test_c_host_program_matches_oracle(tmp_path, flags):
    if not os.path.exists(DEMO):
        import __graft_entry__
        __graft_entry__.build()
    B, N, dt, ms, seed = 257, 300, 0.001, 4000, 77
    out = tmp_path / "demo.bin"
    r = subprocess.run([DEMO, str(B), str(N), str(dt), str(ms), str(seed), str(flags), str(out)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert info["sets"] == B and info["bad_shape_status"] == 2          # NDDM_ERR_SHAPE for n_trials = 0
    # ABI 3 from plain C: a launch captured under a graph arena replays bit-identically, the arena held its memory and is released
    # exactly once, and a destroyed stream's handle is an error code (NDDM_ERR_HIP), not a fault
    assert info["graph_replay_equal"] == 1 and info["arena_allocations"] == 1 and info["arena_bytes"] >= 256
    assert info["arena_released_twice_status"] == 3 and info["dead_stream_status"] == 4
    info4 = json.loads(r.stdout.strip().splitlines()[-2])           # ABI 4: the build record, and a flag the exact sampler does not take
    assert info4["build"].startswith("hipcc=") and info4["ratcliff_bad_flag_status"] == 3
    raw = np.fromfile(out, dtype=np.float32)
    cuts = np.cumsum([B * 5, B * N * 2, B * 10, B * N * 2, B * 6, B * N * 2])
    p, t, s, t64, q, tr, sr = np.split(raw, cuts)
    p, t, s = p.reshape(B, 5), t.reshape(B, N, 2), s.reshape(B, 10)
    t64, q, tr, sr = t64.reshape(B, N, 2), q.reshape(B, 6), tr.reshape(B, N, 2), sr.reshape(B, 10)
    # ABI 4 from plain C: NDDM_STATE_F64 against the float64 restatement, nddm_simulratcliff against the oracle's section D
    o64 = oracle.philox_simulate_f64(oracle.M_BASIC, p, N, dt=dt, max_steps=ms, seed=seed, set_offset=0, threads=8, want_outputs=True)
    orat = oracle.philox_ratcliff(q, N, seed=seed, set_offset=0, threads=8)
    if flags == 0:
        assert np.array_equal(t64.view(np.uint32), o64["trials"].view(np.uint32))
        assert np.array_equal(tr.view(np.uint32), orat["trials"].view(np.uint32))
        assert np.array_equal(np.nan_to_num(sr).view(np.uint32), np.nan_to_num(orat["summary"]).view(np.uint32))
    else:
        assert (t64[..., 0] == o64["trials"][..., 0]).mean() > 0.99
        assert (np.sign(tr[..., 0]) == np.sign(orat["trials"][..., 0])).mean() > 0.995
    o = oracle.philox_simulate(oracle.M_BASIC, p, N, dt=dt, max_steps=ms, seed=seed, set_offset=0, threads=8)
    if flags == 0:      # exact transform: bit parity
        assert np.array_equal(t.view(np.uint32), o["trials"].view(np.uint32))
        assert np.array_equal(np.nan_to_num(s).view(np.uint32), np.nan_to_num(o["summary"]).view(np.uint32))
    else:               # hardware transcendentals: same stream, a trial only differs when its path grazes a boundary
        assert (t[..., 0] == o["trials"][..., 0]).mean() > 0.995
        assert (t[..., 1] == o["trials"][..., 1]).mean() > 0.999

"""CPU tests of the host side: the C-ABI library loads and exports every symbol include/nddm.h declares, the
product path fails loudly without a GPU (no CPU fallback), the BayesFlow-style wrappers and the configurator honour
the reference's dictionary contract."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "nddm.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nddm_[a-z0-9_]+)\s*\(", txt)))


def test_library_builds_loads_and_exports_every_declared_symbol():
    from bayesflow_nddms_amd import _lib, build
    build.build_hip()                      # hipcc cross-compiles gfx950 without a GPU
    L = ctypes.CDLL(build.SO_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 13
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/nddm.h but not exported"
    assert set(syms) <= set(_lib.EXPORTS)
    assert _lib.lib().nddm_abi_version() == _lib.ABI_VERSION == 4
    assert _lib.lib().nddm_source_hash().decode() == build.source_hash()
    assert _lib.lib().nddm_summary_k() == 10
    assert [_lib.lib().nddm_model_nparams(m) for m in range(6)] == [5, 8, 8, 6, 4, -1]


def test_c_abi_argument_errors_without_gpu():
    """Argument validation happens before any HIP call, so it is checkable on a CPU-only box."""
    from bayesflow_nddms_amd import _lib
    L = _lib.lib()
    dummy = ctypes.c_void_p(16)
    assert L.nddm_basic_ddm_dc_simulate(dummy, -1, 10, 0.01, 400, 0, 0, 0, dummy, None, None) == _lib.NDDM_ERR_SHAPE
    assert L.nddm_basic_ddm_dc_simulate(dummy, 4, 0, 0.01, 400, 0, 0, 0, dummy, None, None) == _lib.NDDM_ERR_SHAPE
    assert L.nddm_basic_ddm_dc_simulate(dummy, 4, 10, -0.01, 400, 0, 0, 0, dummy, None, None) == _lib.NDDM_ERR_PARAM
    assert L.nddm_basic_ddm_dc_simulate(dummy, 4, 10, 0.01, 400, 0, 0, 7, dummy, None, None) == _lib.NDDM_ERR_PARAM
    assert L.nddm_basic_ddm_dc_simulate(dummy, 4, 10, 0.01, 400, 0, 0, 16, dummy, None, None) == _lib.NDDM_ERR_PARAM     # unknown flag
    assert L.nddm_basic_ddm_dc_simulate(dummy, 4, 10, 0.01, 400, 0, 0, 8 | 4, dummy, None, None) == _lib.NDDM_ERR_PARAM  # STATE_F64 + packed
    assert L.nddm_alpha_not_scaled_simulate(dummy, 4, 10, 0.01, 400, 0, 0, 8, 0.1, 0, dummy, None, None, None) == _lib.NDDM_ERR_PARAM
    assert b"NDDM_STATE_F64" in L.nddm_last_error()
    assert L.nddm_build_info().startswith(b"hipcc=") and L.nddm_build_info() != b"hipcc=unknown"
    assert L.nddm_basic_ddm_dc_simulate(None, 4, 10, 0.01, 400, 0, 0, 0, dummy, None, None) == _lib.NDDM_ERR_NULL
    assert L.nddm_basic_ddm_dc_simulate(dummy, 4, 10, 0.01, 400, 0, 0, 0, None, None, None) == _lib.NDDM_ERR_NULL
    assert L.nddm_explicit_boundary_simulate(dummy, None, 4, 10, 0.01, 400, 0, 0, 0, dummy, None, None) == _lib.NDDM_ERR_NULL
    assert L.nddm_basic_ddm_dc_simulate(dummy, 2**30, 100000, 0.01, 400, 0, 0, 0, dummy, None, None) == _lib.NDDM_ERR_SHAPE
    assert b"< 2^31" in L.nddm_last_error()
    assert L.nddm_basic_ddm_dc_simulate(dummy, 0, 10, 0.01, 400, 0, 0, 0, dummy, None, None) == _lib.NDDM_OK  # empty batch
    assert L.nddm_set_debug_trace(dummy, -1, 0) == _lib.NDDM_ERR_PARAM                                         # developer aids
    assert L.nddm_set_debug_trace(None, 0, 0) == _lib.NDDM_OK
    assert L.nddm_debug_last_launch(None) == _lib.NDDM_ERR_NULL
    geo = (ctypes.c_int32 * 8)()
    assert L.nddm_debug_last_launch(geo) == _lib.NDDM_OK and list(geo) == [0] * 8      # nothing launched on this thread
    assert L.nddm_set_tuning(0, 1, 0, 0, 0, 0) == _lib.NDDM_ERR_PARAM                   # a ring has at least two slots
    assert L.nddm_set_tuning(0, 0, 0, 0, 0, 0) == _lib.NDDM_OK
    with pytest.raises(ValueError):
        _lib.check(_lib.NDDM_ERR_SHAPE)
    with pytest.raises(RuntimeError):
        _lib.check(_lib.NDDM_ERR_HIP)


def test_graph_arena_handles_without_gpu():
    """The owner handles of captured-launch memory (include/nddm.h: nddm_graph_arena_*) are host bookkeeping: create / bind /
    info / release and their error cases need no device.  Bindings are per thread."""
    import threading
    from bayesflow_nddms_amd import _lib
    L = _lib.lib()
    a, b, prev = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_uint64(99)
    assert L.nddm_graph_arena_create(None) == _lib.NDDM_ERR_NULL
    assert L.nddm_graph_arena_create(ctypes.byref(a)) == _lib.NDDM_OK and a.value != 0
    assert L.nddm_graph_arena_create(ctypes.byref(b)) == _lib.NDDM_OK and b.value not in (0, a.value)
    assert L.nddm_graph_arena_bind(a.value, ctypes.byref(prev)) == _lib.NDDM_OK and prev.value == 0
    assert L.nddm_graph_arena_bind(b.value, ctypes.byref(prev)) == _lib.NDDM_OK and prev.value == a.value
    seen = []

    def other_thread():                     # a fresh thread has no arena bound, whatever this one has
        p = ctypes.c_uint64(99)
        seen.append((L.nddm_graph_arena_bind(0, ctypes.byref(p)), p.value))

    t = threading.Thread(target=other_thread)
    t.start(); t.join()
    assert seen == [(_lib.NDDM_OK, 0)]
    nbytes, n = ctypes.c_uint64(7), ctypes.c_int32(7)
    assert L.nddm_graph_arena_info(a.value, ctypes.byref(nbytes), ctypes.byref(n)) == _lib.NDDM_OK and (nbytes.value, n.value) == (0, 0)
    assert L.nddm_graph_arena_info(0, None, None) == _lib.NDDM_OK                      # 0 = the ownerless list
    assert L.nddm_graph_arena_release(0) == _lib.NDDM_ERR_PARAM
    assert L.nddm_graph_arena_release(b.value) == _lib.NDDM_OK                          # (bound to this thread: the binding is dropped)
    assert L.nddm_graph_arena_bind(0, ctypes.byref(prev)) == _lib.NDDM_OK and prev.value == 0
    assert L.nddm_graph_arena_release(b.value) == _lib.NDDM_ERR_PARAM                   # already released
    assert L.nddm_graph_arena_bind(b.value, None) == _lib.NDDM_ERR_PARAM
    assert L.nddm_graph_arena_info(b.value, None, None) == _lib.NDDM_ERR_PARAM
    assert L.nddm_graph_arena_release(a.value) == _lib.NDDM_OK
    assert b"arena" in L.nddm_last_error() or L.nddm_last_error() == b""


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from bayesflow_nddms_amd import basic_ddm_dc, engine
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        engine.simulate(engine.BASIC_DDM_DC, np.array([[1.5, 1.2, .5, .35, 1.0]]), 10)
    with pytest.raises(RuntimeError):
        basic_ddm_dc.simulate_trials([1.5, 1.2, .5, .35, 1.0], 10)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under bayesflow_nddms_amd/ may reference it."""
    pkg = os.path.join(ROOT, "bayesflow_nddms_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M), f
                assert "liboracle" not in src and "ddm_oracle" not in src, f
    # ... nor may tools/ or examples/ (checker programs that need the oracle -- the long fuzz run, the stress parity run -- live under
    # tests/); bench.py uses it in cpu_baseline() only and __graft_entry__ in smoke() / build() only
    for sub in ("tools", "examples", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, sub)):
            for f in files:
                if f.endswith((".py", ".sh", ".c", ".h", ".hip")):
                    src = open(os.path.join(dirpath, f), errors="replace").read()
                    assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M) and "liboracle" not in src, os.path.join(sub, f)
    bench = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"^\s*(import|from)\s+oracle\b", bench, flags=re.M)]
    lo, hi = bench.index("def cpu_baseline("), bench.index("def ks_vs_golden(")
    assert uses and all(lo < u < hi for u in uses), "bench.py may touch the oracle inside cpu_baseline() only"


def test_validation_and_stream_state():
    from bayesflow_nddms_amd import engine
    engine.validate_params_host(engine.BASIC_DDM_DC, np.array([[1.5, 1.2, .5, .35, 1.0]]))
    for bad in ([[1.5, -1.2, .5, .35, 1.0]], [[1.5, 1.2, .5, .35, 0.0]], [[np.inf, 1.2, .5, .35, 1.0]]):
        with pytest.raises(ValueError):
            engine.validate_params_host(engine.BASIC_DDM_DC, np.array(bad))
    with pytest.raises(ValueError):   # per-trial boundary that can never be positive
        engine.validate_params_host(engine.SINGLE_TRIAL, np.array([[0., -5., .5, .3, .1, 1., 1., 1.]]))
    s = engine.StreamState(seed=5)
    assert s.take(10) == (5, 0) and s.take(3) == (5, 10) and s.get_state() == {"seed": 5, "offset": 13}
    s.set_state({"seed": 5, "offset": 10})
    assert s.take(1) == (5, 10)                      # resume continues the stream
    assert engine.max_k_of(400.) == 400 and engine.max_k_of(400.5) == 401 and engine.max_k_of(4000) == 4000


def test_generative_model_dict_contract(kat):
    """The wrappers reproduce the keys / shapes the reference's call sites read (basic_ddm_dc.py:146,151,159) with
    both the per-set simulator_fun loop and the batched form, and carry 'summary_stats' additively."""
    from bayesflow_nddms_amd.simulation import ContextGenerator, GenerativeModel, Prior, Simulator
    calls = []

    def draw_prior():
        return np.arange(5, dtype=np.float64)

    def prior_N():
        return 77

    def simulate_trials(params, n_trials):
        calls.append(("set", tuple(np.shape(params)), n_trials))
        return np.zeros((n_trials, 2))

    def batch_sim(params, n_trials):
        calls.append(("batch", tuple(np.shape(params)), n_trials))
        return {"sim_data": np.ones((params.shape[0], n_trials, 2), np.float32),
                "summary_stats": np.zeros((params.shape[0], 10), np.float32)}

    ctx = ContextGenerator(non_batchable_context_fun=prior_N)
    gm = GenerativeModel(Prior(prior_fun=draw_prior), Simulator(simulator_fun=simulate_trials, context_generator=ctx))
    assert calls[:2] == [("set", (5,), 77)] * 2      # construction self-test: batch of 2, per-set calls
    out = gm(4)
    assert out["prior_draws"].shape == (4, 5) and out["sim_data"].shape == (4, 77, 2)
    assert out["sim_non_batchable_context"] == 77 and "summary_stats" not in out
    assert set(out) >= {"prior_draws", "sim_data", "sim_non_batchable_context", "sim_batchable_context",
                        "prior_batchable_context", "prior_non_batchable_context"}
    calls.clear()
    gmb = GenerativeModel(Prior(prior_fun=draw_prior), Simulator(batch_simulator_fun=batch_sim, context_generator=ctx),
                          skip_test=True)
    out = gmb(6)
    assert calls == [("batch", (6, 5), 77)]           # ONE call for the whole batch
    assert out["sim_data"].shape == (6, 77, 2) and out["summary_stats"].shape == (6, 10)
    with pytest.raises(ValueError):
        Prior()
    with pytest.raises(ValueError):
        Simulator(batch_simulator_fun=batch_sim, simulator_fun=simulate_trials)


def test_configurator_matches_reference(kat):
    from bayesflow_nddms_amd import basic_ddm_dc, single_trial_alpha_not_scaled as st
    sim = {"sim_data": kat["basic_seed2023_p0_n300"][None], "sim_non_batchable_context": 300,
           "prior_draws": kat["basic_sets"][0][None]}
    c = basic_ddm_dc.configurator(sim)
    assert c["summary_conditions"].dtype == np.float32 and c["summary_conditions"].shape == (1, 300, 2)
    assert c["direct_conditions"].dtype == np.float32
    # the fixture was made under NumPy 2 (float64 by NEP 50); the reference's pinned NumPy 1.23.5 gives float32
    assert np.allclose(c["direct_conditions"], kat["conf_direct_conditions"], rtol=1e-7)
    assert c["parameters"].dtype == np.float32 and c["parameters"].shape == (1, 5)
    # hand-built dict with prior_draws=None (fitting_stahl_data.py:201-202): 'parameters' is skipped
    c2 = st.configurator({"sim_data": kat["single_seed2024_p0_n300"][None], "sim_non_batchable_context": 300,
                          "prior_draws": None})
    assert "parameters" not in c2 and c2["summary_conditions"].shape == (1, 300, 2)
    # torch tensors stay tensors
    import torch
    c3 = basic_ddm_dc.configurator({k: (torch.as_tensor(v) if isinstance(v, np.ndarray) else v) for k, v in sim.items()})
    assert isinstance(c3["summary_conditions"], torch.Tensor) and c3["direct_conditions"].shape == (1, 1)
    assert np.allclose(c3["direct_conditions"].numpy(), np.log(300))


@pytest.mark.skipif(not os.path.exists(os.path.join(GOLDEN, "priors.npz")), reason="priors.npz not generated yet")
def test_host_priors_bit_exact():
    """draw_prior / prior_N restated call for call: same seeds -> the reference's numbers."""
    from bayesflow_nddms_amd import priors
    g = np.load(os.path.join(GOLDEN, "priors.npz"))
    for name, fn in (("basic", priors.draw_prior_basic), ("single", priors.draw_prior_single)):
        np.random.seed(2023)
        priors.reset_host_rng(2023)
        d = np.array([fn() for _ in range(16)])
        n = np.array([priors.prior_N() for _ in range(16)])
        assert np.array_equal(d, g[f"{name}_draws16"]), name
        assert np.array_equal(n, g[f"{name}_priorN16"]), name
    assert priors.draw_prior_scale().shape == (8,)


def test_alpha_not_scaled_participants_bit_exact(kat):
    """alpha_not_scaled.py:63-72, 82-88: participant-level parameters on the global NumPy stream, seed 2021."""
    from bayesflow_nddms_amd import alpha_not_scaled
    par = alpha_not_scaled.draw_participants(100, seed=2021)
    for key in ("ndt", "alpha", "beta", "delta", "varsigma", "deltatrialsd"):
        assert np.array_equal(par[key], kat[f"alpha_ns_part_{key}"]), key
    assert alpha_not_scaled.SIGMA_OF_TEST == {1: .5, 2: .1, 3: .01, 4: .2}


def test_diagnostics_ks():
    from bayesflow_nddms_amd import diagnostics as dg
    rng = np.random.default_rng(0)
    K = 50
    k = rng.integers(1, K, 20000)
    ch = rng.choice([1, -1, 0], 20000, p=[.6, .3, .1])
    tr = np.stack([k * 0.01 + 0.3, ch], axis=1)
    tr[ch == 0, 0] = K * 0.01 + 0.3
    h = dg.step_hist_from_trials(tr, 0.3, 0.01, K)
    assert h.sum() == 20000 and h[2, K] == (ch == 0).sum()
    assert dg.ks_signed(h, h) == 0.0
    h2 = h.copy(); h2[0] = np.roll(h2[0], 3)
    assert 0.01 < dg.ks_signed(h, h2) < 0.2
    signed = np.stack([np.where(ch == 0, 0, ch * (k * 0.01 + 0.3)), np.zeros(20000)], axis=1)
    assert np.array_equal(dg.step_hist_from_trials(signed, 0.3, 0.01, K, signed=True)[:2], h[:2])
    s = rng.normal(size=50000)
    q = np.quantile(rng.normal(size=400000), np.linspace(0, 1, 2001))
    assert dg.ks_quantile_table(s, q) < 0.01 and dg.ks_quantile_table(s + 0.2, q) > 0.05


def test_shard_bounds_cover_and_partition():
    from bayesflow_nddms_amd.distributed import shard_bounds, shared_prior_N
    for B in (0, 1, 7, 8, 9, 1000, 1_000_000):
        for G in (1, 2, 3, 8):
            spans = [shard_bounds(B, G, r) for r in range(G)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert shared_prior_N(3, 5) == shared_prior_N(3, 5) and 60 <= shared_prior_N(3, 6) <= 300


def test_bench_launcher_without_gpu_fails_loudly():
    """bench.py --gpus N with more ranks than the node has GPUs is refused by the PARENT, before any rank is started (it asks
    a throwaway child for the device count and never loads HIP itself); with --share-device two fresh ranks start, and on
    a box without a GPU both refuse loudly (no CPU fallback) and the launcher exits non-zero; a WORLD_SIZE that contradicts
    --gpus is refused before anything else happens."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--sets", "10", "--steps", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "--gpus 2 but this node shows 0 GPU(s); nothing was started" in r.stderr and "{" not in r.stdout
    assert "needs a ROCm GPU" not in r.stderr                       # no rank ever ran
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-device", "--backend", "gloo",
                        "--sets", "10", "--steps", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and r.stderr.count("needs a ROCm GPU") >= 1 and "{" not in r.stdout
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--sets", "10", "--steps", "1"],
                       capture_output=True, text=True, timeout=300, env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in r.stderr


def test_ezdiff_matches_reference_known_answers():
    """bayesflow_nddms_amd.ezdiff.ezdiff against the reference's ezdiff() run on reference choice-RT data
    (tests/golden/ezdiff.npz, made by make_golden.py --ezdiff), and the batched summary form against it."""
    from bayesflow_nddms_amd import ezdiff as ez
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ezdiff.npz"))
    for ci in range(int(g["n_cases"])):
        rt, correct, want = g[f"rt_{ci}"], g[f"correct_{ci}"], g[f"est_{ci}"]
        got = np.array(ez.ezdiff(rt, correct))
        assert np.allclose(got, want, rtol=1e-12, atol=0), (ci, got, want)
        # the same data as a fused summary row (upper = correct): counts, mean / variance of the correct RTs
        hits = rt[correct == 1]
        row = np.full(10, np.nan)
        row[0], row[1], row[2] = (correct == 1).sum(), (correct == 0).sum(), np.isnan(correct).sum()
        row[5], row[6] = hits.mean(), hits.var()
        assert np.allclose(ez.ez_from_summary(row[None])[0], want, rtol=1e-9), ci
    with pytest.raises(ValueError):
        ez.ezdiff(np.array([0.5, 0.6]), np.array([0.0, 0.0]))              # no correct response
    with pytest.raises(ValueError):
        ez.ezdiff(np.array([0.5]), np.array([1.0, 1.0]))                   # length mismatch
    bad = np.zeros((2, 10)); bad[1, 0] = 5; bad[1, 5] = 0.4                 # row 0: nothing; row 1: zero variance
    assert np.isnan(ez.ez_from_summary(bad)).all()


def test_staleness_is_decided_by_content_not_by_file_times(tmp_path):
    """build.is_stale() compares the sha256 of the sources (and flags) with the hash compiled into the library: touching a
    source leaves the library fresh, a snapshot with reordered mtimes does not trigger a rebuild."""
    from bayesflow_nddms_amd import build
    assert build.embedded_hash() == build.source_hash() and not build.is_stale()
    src = build.SOURCES[0]
    st = os.stat(src)
    try:
        os.utime(src, (st.st_atime, os.path.getmtime(build.SO_PATH) + 1000.0))      # newer than the library
        assert not build.is_stale()
    finally:
        os.utime(src, (st.st_atime, st.st_mtime))
    fake = tmp_path / "lib.so"
    fake.write_bytes(b"\x7fELF....NDDM_SRC_HASH=" + b"0" * 64 + b"\0")
    assert build.embedded_hash(str(fake)) == "0" * 64 and build.embedded_hash(str(tmp_path / "absent.so")) is None


def test_graph_trainer_buckets_and_masked_pooling_equal_the_unpadded_batch():
    """The n_trials buckets of graph_trainer (one hipGraph per bucket): 16 buckets cover 60..300, every N maps to a top >=
    N within one bucket width; and the summary network on a batch PADDED to the bucket top with (mask, 1/N) equals the
    network on the unpadded batch."""
    import torch
    from bayesflow_nddms_amd.amortizer import InvariantNetwork
    from bayesflow_nddms_amd.graph_trainer import GraphTrainer
    gt = GraphTrainer.__new__(GraphTrainer)
    gt.n_min, gt.n_max, gt.n_buckets = 60, 300, 16
    gt.width = -(-(gt.n_max - gt.n_min + 1) // gt.n_buckets)
    tops = sorted({gt.bucket_top(n) for n in range(60, 301)})
    assert len(tops) == 16 and tops[-1] == 300
    assert all(0 <= gt.bucket_top(n) - n < gt.width for n in range(60, 301))
    torch.manual_seed(1)
    net = InvariantNetwork()
    x = torch.randn(5, 91, 2)
    n = 77
    mask = (torch.arange(91) < n).float().view(1, 91, 1)
    x_pad = x.clone()
    x_pad[:, n:] = 1e3 * torch.randn(5, 91 - n, 2)                      # whatever the padding holds
    assert torch.allclose(net(x_pad, mask, torch.tensor(1.0 / n)), net(x[:, :n]), atol=2e-5)


def test_training_library_builds_and_exports_its_entry_points():
    """libnddm_train.so (the amortizer's flow, summary-network and optimizer kernels: csrc/train_*.hip) cross-compiles for
    gfx950 without a GPU, loads, and exports what _train_lib binds; the shape predicates answer on the host."""
    import ctypes
    from bayesflow_nddms_amd import build
    L = ctypes.CDLL(build.build_train())
    for name in ("nddm_train_flow_supported", "nddm_train_flow_fwd", "nddm_train_flow_bwd", "nddm_deepset_supported",
                 "nddm_deepset_mlp_fwd", "nddm_deepset_mlp_bwd", "nddm_deepset_reduce",
                 "nddm_train_adam_step"):
        assert hasattr(L, name), name
    L.nddm_train_flow_supported.argtypes = [ctypes.c_int] * 5
    assert L.nddm_train_flow_supported(128, 6, 5, 2, 11) == 1          # the reference's network: 6 layers, 5 parameters, 10 + 1 conditions
    assert L.nddm_train_flow_supported(64, 6, 5, 2, 11) == 0 and L.nddm_train_flow_supported(128, 9, 5, 2, 11) == 0
    L.nddm_deepset_supported.argtypes = [ctypes.c_int] * 2
    assert L.nddm_deepset_supported(64, 2) == 1 and L.nddm_deepset_supported(64, 64) == 1 and L.nddm_deepset_supported(64, 7) == 0
    assert build.train_source_hash() == open(build.TRAIN_SO_PATH + ".srchash").read().strip()


_RACE_WORKER = r"""
import os, stat, sys, time
sys.path.insert(0, sys.argv[1])
from bayesflow_nddms_amd import build
tmp = sys.argv[2]
build.TRAIN_SO_PATH = os.path.join(tmp, "libfake_train.so")
build._hipcc = lambda: os.path.join(tmp, "fake_hipcc.sh")
while not os.path.exists(os.path.join(tmp, "go")):      # both processes leave the gate together
    time.sleep(0.005)
build.build_train()
assert not build.train_is_stale()
"""


def test_two_processes_racing_on_a_stale_training_library_compile_once(tmp_path):
    """build_train() re-checks freshness AFTER taking the lock (as build_hip does): of two ranks that both found the library stale,
    one compiles and the other finds it fresh when its turn comes.  (The compiler is a stand-in that logs each invocation and
    takes half a second, writing to a scratch path: the tree's own library is not touched.)"""
    import subprocess
    import sys
    import time
    fake = tmp_path / "fake_hipcc.sh"
    fake.write_text('#!/bin/bash\necho run >> "%s/calls.log"\nsleep 0.5\nwhile [ "$1" != "-o" ]; do shift; done\necho lib > "$2"\n' % tmp_path)
    fake.chmod(0o755)
    (tmp_path / "libfake_train.so.srchash").write_text("stale")
    worker = tmp_path / "worker.py"
    worker.write_text(_RACE_WORKER)
    procs = [subprocess.Popen([sys.executable, str(worker), ROOT, str(tmp_path)]) for _ in range(2)]
    time.sleep(1.0)                                       # (imports done, both spinning at the gate)
    (tmp_path / "go").write_text("")
    assert [p.wait(timeout=60) for p in procs] == [0, 0]
    assert (tmp_path / "calls.log").read_text().count("run") == 1


def test_all_gather_form_follows_the_backend_and_errors_surface(monkeypatch):
    """distributed.all_gather_rows picks the collective by dist.get_backend() -- RCCL: ONE all_gather_into_tensor, gloo: the list form
    -- and never by catching an error: a failing RCCL collective raises its own message and no second collective follows on the
    same communicator."""
    import torch
    from bayesflow_nddms_amd import distributed as D

    class FakeDist:
        def __init__(self, backend, fail=False):
            self.backend, self.fail, self.calls = backend, fail, []

        def get_world_size(self, group=None):
            return 2

        def get_backend(self, group=None):
            return self.backend

        def all_gather_into_tensor(self, full, local, group=None):
            self.calls.append("flat")
            if self.fail:
                raise RuntimeError("NCCL error: unhandled system error")
            full.copy_(torch.cat([local, local]))

        def all_gather(self, parts, local, group=None):
            self.calls.append("list")
            for p in parts:
                p.copy_(local)

    x = torch.arange(6.0).view(3, 2)
    for backend, form in (("nccl", "flat"), ("gloo", "list"), ("cpu:gloo,cuda:nccl", "flat")):
        fd = FakeDist(backend)
        monkeypatch.setattr(D, "_dist", lambda fd=fd: fd)
        out = D.all_gather_rows(x, 5)
        assert fd.calls == [form] and out.shape == (5, 2) and torch.equal(out[:3], x) and torch.equal(out[3:], x[:2])
    fd = FakeDist("nccl", fail=True)
    monkeypatch.setattr(D, "_dist", lambda: fd)
    with pytest.raises(RuntimeError, match="unhandled system error"):
        D.all_gather_rows(x, 5)
    assert fd.calls == ["flat"]                            # nothing was attempted after the failure


def test_recovery_statistics_are_the_reference_plots_numbers():
    """diagnostics.recovery_statistics == what pyhddmjagsutils.recovery_scatter prints (sklearn r2_score and scipy pearsonr per
    parameter, :609-623), including R^2 far below zero when one estimate is carried off; converged_fits == basic_ddm_dc.py:239-241."""
    from scipy import stats
    from sklearn.metrics import r2_score
    from bayesflow_nddms_amd import diagnostics as dg
    rng = np.random.default_rng(3)
    true = rng.normal(size=(200, 5)) * np.array([2.0, 0.5, 0.2, 0.25, 0.5]) + np.array([0.0, 1.0, 0.5, 0.5, 1.0])
    est = true + 0.3 * rng.normal(size=true.shape) * np.array([2.0, 0.5, 0.2, 0.25, 0.5])
    est[17, 1] = -1.4e6                                                 # one posterior mean carried off by a tail draw
    got = dg.recovery_statistics(true, est)
    for j in range(5):
        assert abs(got["r2"][j] - r2_score(true[:, j], est[:, j])) <= 1e-9 * max(1.0, abs(got["r2"][j]))
        assert abs(got["rho"][j] - stats.pearsonr(true[:, j], est[:, j])[0]) < 1e-12
    assert got["r2"][1] < -1e6 and got["r2"][0] > 0.8
    conv = dg.converged_fits(est)
    assert conv.dtype == bool and conv.sum() == ((est[:, 3] > 0) & (est[:, 3] < 1)).sum()
    with pytest.raises(ValueError):
        dg.recovery_statistics(true, est[:, :4])


def test_bench_plan_for_eight_gpus_touches_no_gpu():
    """`bench.py --plan --gpus 8`: the pre-flight of the N > 1 run the builder cannot make -- what every rank allocates and moves per step
    under each gather mode, printed without importing torch or touching a GPU (this box has none), so that the first 8-GPU run does
    not discover an out-of-memory or a link-bound default."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, PYTHONPATH="")
    r = subprocess.run([sys.executable, "-X", "importtime", os.path.join(ROOT, "bench.py"), "--plan", "--gpus", "8"], capture_output=True,
                       text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert " torch" not in r.stderr                                                     # (-X importtime lists every import on stderr)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["plan"] is True and d["n_gpus"] == 8 and d["sets_per_gpu"] == 1_000_000 and d["n_trials"] == 300
    g = d["gather"]
    # gather=trials: 2.4 GB per shard, 8 shards gathered, two buffers of it; 16.8 GB received per rank per step
    assert g["trials"]["buffers"]["gathered trials x2"] == 2 * 8 * 2_400_000_000
    assert g["trials"]["received_bytes_per_rank_per_step"] == 7 * 2_400_000_000
    assert g["trials"]["hidden_behind_simulate"]["if_one_ring"] is False and g["trials"]["link_bound_even_on_all_links"] is True
    # the 2-byte codes: a quarter of the bytes on the wire (+ the parameter rows), the decoded floats exist on every rank
    assert g["codes"]["sent_bytes_per_rank_per_step"] == 1_000_000 * (300 * 2 + 5 * 4)
    assert g["codes"]["buffers"]["decoded trials f32[W,B,N,2] x2"] == 2 * 8 * 2_400_000_000
    assert g["summary"]["received_bytes_per_rank_per_step"] == 7 * 40_000_000 and g["summary"]["hidden_behind_simulate"]["if_one_ring"] is True
    assert all(v["fits_hbm"] for v in g.values()) and max(v["hbm_fraction"] for v in g.values()) < 0.25
    # the driver's plain command gathers nothing; its side legs are the two forms that fit and stay hidden
    pc = d["plain_command"]
    assert pc["gather"] == "none" and set(pc["side_legs"]) == {"summary", "codes"} and pc["strong"]["sets_per_gpu"] == 125_000
    assert all(v["runs_if_free_memory_exceeds"] < 0.3 * d["hbm_bytes_per_gpu"] for v in pc["side_legs"].values())
    # a shape that does NOT fit says so
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--plan", "--gpus", "8", "--sets", "8000000"], capture_output=True,
                       text=True, timeout=120, env=env)
    big = json.loads(r.stdout.strip().splitlines()[-1])["gather"]
    assert big["trials"]["fits_hbm"] is False and big["none"]["fits_hbm"] is True


def test_host_shim_is_clean_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """SURVEY section 5's `-fsanitize=address` host shim, on the PRODUCT: the host side of csrc/nddm_kernels.hip -- launch slots, graph
    arenas, the per-thread bind, free_list, the knob snapshots under one mutex, every entry point's validation and its no-device error
    path -- built with AddressSanitizer + UndefinedBehaviorSanitizer (`-Xarch_host`: host code only, the gfx950 code object is the
    product's; sanitizers never go to the GPU pool) and driven by tests/host_shim_battery.py in a child process that loads the
    library the way a C program would: no report, and the same return codes and error strings as the normal build."""
    import glob
    import subprocess
    import sys
    from bayesflow_nddms_amd import build
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import host_shim_battery
    build.build_hip()
    want, n_checks = host_shim_battery.battery(build.SO_PATH)
    assert n_checks > 800
    rts = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    if not rts:
        pytest.skip("this ROCm has no clang AddressSanitizer runtime")
    so = str(tmp_path / "libnddm_hip_san.so")
    flags = [f for f in build.HIPCC_FLAGS if f != "-O3"] + ["-O1", "-fno-omit-frame-pointer"]
    for f in ("-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fsanitize-address-use-after-scope"):
        flags += ["-Xarch_host", f]
    cc = subprocess.run([build._hipcc()] + flags + [f'-DNDDM_SOURCE_HASH="{build.source_hash()}"', "-o", so] + build.SOURCES,
                        capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-3000:]
    syms = subprocess.run(["nm", "-D", so], capture_output=True, text=True).stdout
    assert "__asan_init" in syms and "__ubsan_handle" in syms                   # the host code really is instrumented
    env = dict(os.environ, LD_PRELOAD=rts[0], NDDM_HIP_LIB=so, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "host_shim_battery.py")], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    assert r.stdout.strip().splitlines()[-1] == f"SHIM {want} ({n_checks} checks)"


def test_three_term_acceptance_function_of_the_exact_samplers_fast_mode():
    """nddm_simulratcliff's fast mode accepts an attempt when s2 e^-a <= theta(a), theta(a) = sum_{k odd} (-1)^((k-1)/2) k e^{-a k^2}, and
    evaluates theta from THREE terms of that series (a >= pi/4) or of its Jacobi-dual form (pi/4a)^{3/2} theta(pi^2/16a) (a < pi/4) --
    csrc/nddm_ratcliff.h.  The same float32 arithmetic in NumPy, with the kernel's constants, against 400 terms of the series in
    float64 over the whole range the kernel evaluates (a >= 2^-6): the acceptance probability theta(a) e^a equal to 5e-7 (the reference
    sums the series until a term no longer changes its sum; pyhddmjagsutils.py:147-159)."""
    f = np.float32
    a = np.exp(np.linspace(np.log(2.0 ** -6), np.log(60.0), 40001)).astype(np.float32)
    ea = np.exp(-a.astype(np.float64)).astype(np.float32)
    dual = a < f(0.785398163)
    rs = (1.0 / np.sqrt(a.astype(np.float64))).astype(np.float32)
    ra = rs * rs
    E = np.where(dual, np.exp(-(f(0.616850275) * ra).astype(np.float64)).astype(np.float32), ea)
    E2 = E * E; E4 = E2 * E2; E8 = E4 * E4
    P = E8 * (f(5.0) * E8 * E8 + f(-3.0)) + f(1.0)
    theta3 = np.where(dual, (f(0.696040999) * (rs * ra)) * (E * P), ea * P).astype(np.float64)
    k = np.arange(1, 801, 2, dtype=np.float64)
    sgn = np.where(((k - 1) // 2) % 2 == 0, 1.0, -1.0)
    a64 = a.astype(np.float64)
    small = a64 < 0.3                                        # the alternating float64 sum cancels below ~0.3: its exact value is the dual's
    ref = np.empty_like(a64)
    ref[~small] = (sgn * k * np.exp(-a64[~small, None] * k * k)).sum(1)
    b = (np.pi ** 2 / 16.0) / a64[small]
    ref[small] = (np.pi / (4.0 * a64[small])) ** 1.5 * (sgn[:8] * k[:8] * np.exp(-b[:, None] * k[:8] * k[:8])).sum(1)
    # what the test decides is P(accept) = theta(a) e^a: equal to 5e-7 absolute everywhere (one float32 ulp of a probability); theta
    # itself to 1e-5 relative (at a = 2^-6, where theta ~ 1e-15, the float32 rounding of the exponent pi^2/16a ~ 38 is what is left)
    p_err = np.abs(theta3 - ref) * np.exp(a64)
    rel = np.abs(theta3 - ref) / ref
    assert p_err.max() < 5e-7 and rel.max() < 1e-5, (p_err.max(), a[p_err.argmax()], rel.max(), a[rel.argmax()])
    # ... and where both float64 forms are accurate (0.3 <= a <= 2) they are the same function: the identity itself
    mid = (a64 >= 0.3) & (a64 <= 2.0)
    bm = (np.pi ** 2 / 16.0) / a64[mid]
    dual64 = (np.pi / (4.0 * a64[mid])) ** 1.5 * (sgn[:12] * k[:12] * np.exp(-bm[:, None] * k[:12] * k[:12])).sum(1)
    ser64 = (sgn * k * np.exp(-a64[mid, None] * k * k)).sum(1)
    assert np.abs(dual64 / ser64 - 1.0).max() < 1e-12

"""GPU tests of the OWNER of captured-launch memory (include/nddm.h: nddm_graph_arena_*, ABI 3).  A launch captured into a
hipGraph pins a library allocation (queue words + scratch) that the graph replays into; up to ABI 2 one device-wide call freed
all of them, whoever's graph they belonged to.  Now: releasing one owner leaves every other owner's graphs replaying -- two live
GraphTrainers (the checkpointed, re-entered training of basic_ddm_dc.py:169-176, 199-207 keeps more than one alive in a
notebook), a user's graph beside a trainer, nested `engine.graph_memory()` blocks.  And: a `stream` the HIP runtime does not know
is refused before anything is enqueued on it."""
import ctypes

import numpy as np
import pytest

import prior_util

pytestmark = pytest.mark.gpu


def _capture(torch, engine, p_dev, B, N, seed):
    out = torch.empty((B, N, 2), device="cuda")
    summ = torch.empty((B, 10), device="cuda")
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side), torch.cuda.graph(g, stream=side):
        engine.simulate(engine.BASIC_DDM_DC, p_dev[:B], N, dt=.01, max_steps=400, seed=seed, set_offset=0, fast=True,
                        out_trials=out, out_summary=summ)
    torch.cuda.synchronize()
    return g, out, summ


def test_releasing_one_owner_leaves_the_others_replaying():
    import torch
    from bayesflow_nddms_amd import engine
    p_dev = torch.as_tensor(prior_util.basic_prior(4096, 3)).cuda()
    shapes = [(32, 150, 5), (3000, 64, 6), (200, 300, 7)]              # (B, N, seed); the second takes the longest-first pre-pass too
    ref = []
    for B, N, seed in shapes:
        r = engine.simulate(engine.BASIC_DDM_DC, p_dev[:B], N, dt=.01, max_steps=400, seed=seed, set_offset=0, fast=True)
        ref.append((r["trials"].clone(), torch.nan_to_num(r["summary"]).clone()))
    torch.cuda.synchronize()
    mine = engine.GraphArena()
    with mine.bound():
        kept = [_capture(torch, engine, p_dev, *s) for s in shapes]
    assert mine.info()["allocations"] == len(shapes) and mine.info()["bytes"] >= 256 * len(shapes)
    ownerless = _capture(torch, engine, p_dev, *shapes[0])             # no arena bound: the ownerless list
    assert mine.info()["allocations"] == len(shapes)

    def check(graphs, tag, which=None):
        which = range(len(graphs)) if which is None else which          # the shape (index into `ref`) each graph captured
        for g, out, summ in graphs:
            out.zero_(); summ.fill_(-7.0)
            g.replay()
        torch.cuda.synchronize()
        for (g, out, summ), i in zip(graphs, which):
            assert torch.equal(out, ref[i][0]) and torch.equal(torch.nan_to_num(summ), ref[i][1]), (tag, i)

    # other owners come and go -- each releases ITS memory, new captures re-use the freed blocks' addresses
    for rep in range(4):
        with engine.graph_memory() as inner:
            theirs = [_capture(torch, engine, p_dev, *s) for s in shapes]
            assert inner.info()["allocations"] == len(shapes)
            check(theirs, ("inner", rep))
            with engine.graph_memory() as innermost:                   # nested owners: the inner block's exit leaves the outer's alone
                g2 = _capture(torch, engine, p_dev, *shapes[1])
                check([g2], ("innermost", rep), which=[1])
                assert innermost.info()["allocations"] == 1
                del g2
            check(theirs, ("inner after innermost", rep))
            del theirs
        scribble = [torch.full((1 << 20,), float("nan"), device="cuda") for _ in range(8)]     # whatever re-uses freed memory writes NaNs
        check(kept, ("kept", rep))
        check([ownerless], ("ownerless", rep))
        del scribble
    # the ownerless release frees only the ownerless list
    del ownerless
    torch.cuda.synchronize()
    engine.release_graph_memory()
    check(kept, "kept after the ownerless release")
    assert mine.info()["allocations"] == len(shapes)
    # a capture under a released arena is refused, not charged to nobody
    dead = engine.GraphArena()
    binding = dead.bound()
    binding.__enter__()
    dead.release()
    from bayesflow_nddms_amd import _lib
    assert _lib.lib().nddm_graph_arena_bind(0, None) == 0
    with pytest.raises(RuntimeError):
        dead.bound().__enter__()
    del kept
    torch.cuda.synchronize()
    mine.release()
    mine.release()                                                     # idempotent
    r = engine.simulate(engine.BASIC_DDM_DC, p_dev[:32], 150, dt=.01, max_steps=400, seed=5, set_offset=0, fast=True)
    assert torch.equal(r["trials"], ref[0][0])                          # the library stays usable


def test_capture_under_a_released_arena_is_refused():
    import torch
    from bayesflow_nddms_amd import _lib, engine
    L = _lib.lib()
    p_dev = torch.as_tensor(prior_util.basic_prior(64, 3)).cuda()
    out = torch.empty((32, 100, 2), device="cuda")
    a, other = ctypes.c_uint64(0), ctypes.c_uint64(0)
    _lib.check(L.nddm_graph_arena_create(ctypes.byref(a)))
    _lib.check(L.nddm_graph_arena_bind(a.value, None))
    # released from ANOTHER thread while this thread still has it bound: this thread's next captured launch must fail cleanly
    import threading
    t = threading.Thread(target=lambda: _lib.check(L.nddm_graph_arena_release(a.value)))
    t.start(); t.join()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    failed = False
    try:
        with torch.cuda.stream(side), torch.cuda.graph(g, stream=side):
            try:
                engine.simulate(engine.BASIC_DDM_DC, p_dev[:32], 100, dt=.01, max_steps=400, seed=1, set_offset=0, fast=True,
                                out_trials=out, want_summary=False)
            except ValueError as e:
                failed = "arena" in str(e)
    finally:
        _lib.check(L.nddm_graph_arena_bind(0, None))
    assert failed
    del g
    torch.cuda.synchronize()
    r = engine.simulate(engine.BASIC_DDM_DC, p_dev[:32], 100, dt=.01, max_steps=400, seed=1, set_offset=0, fast=True)
    assert r["trials"].shape == (32, 100, 2)


def test_a_stream_the_runtime_does_not_know_is_refused():
    """A destroyed stream's handle (and an integer that never was a stream) comes back as NDDM_ERR_HIP with a text that says so;
    nothing is enqueued and the library stays usable.  (C callers keep raw handles; a stale one used to go straight into the
    launch calls.)"""
    import torch
    from test_gpu_threads import _hip_runtime
    from bayesflow_nddms_amd import _lib, engine
    hip, L = _hip_runtime(), _lib.lib()
    p_dev = torch.as_tensor(prior_util.basic_prior(64, 3)).cuda()
    out = torch.empty((32, 100, 2), device="cuda")
    ref = engine.simulate(engine.BASIC_DDM_DC, p_dev[:32], 100, dt=.01, max_steps=400, seed=1, set_offset=0, fast=True)["trials"].clone()
    torch.cuda.synchronize()
    h = ctypes.c_void_p()
    assert hip.hipStreamCreate(ctypes.byref(h)) == 0
    live = h.value
    assert L.nddm_basic_ddm_dc_simulate(p_dev.data_ptr(), 32, 100, 0.01, 400, 1, 0, 1, out.data_ptr(), None, ctypes.c_void_p(live)) == 0
    assert hip.hipStreamDestroy(ctypes.c_void_p(live)) == 0
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    for bad in (live, 0x1230):
        rc = L.nddm_basic_ddm_dc_simulate(p_dev.data_ptr(), 32, 100, 0.01, 400, 1, 0, 1, out.data_ptr(), None, ctypes.c_void_p(bad))
        assert rc == _lib.NDDM_ERR_HIP and b"not a live stream" in L.nddm_last_error(), (hex(bad), rc, L.nddm_last_error())
    out.zero_()
    assert L.nddm_basic_ddm_dc_simulate(p_dev.data_ptr(), 32, 100, 0.01, 400, 1, 0, 1, out.data_ptr(), None, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(out, ref)


def test_two_live_graph_trainers_close_one_the_other_keeps_training():
    """Two GraphTrainers alive at once; the first is closed (its graphs destroyed, its arena released) in the middle of the
    second's run, a third is created and trained on the freed memory, the ownerless release is called for good measure -- the
    second's loss history equals its solo run."""
    import torch
    from bayesflow_nddms_amd import engine
    from bayesflow_nddms_amd.amortizer import AmortizedPosterior, InvariantNetwork, InvertibleNetwork
    from bayesflow_nddms_amd.graph_trainer import GraphTrainer

    def make(seed):
        torch.manual_seed(seed)
        return GraphTrainer(AmortizedPosterior(InvertibleNetwork(num_params=5), InvariantNetwork()), batch_size=32,
                            total_steps=40, seed=seed, learning_rate=1e-3)

    with make(2024) as solo:
        solo.train_online(40)
        h_solo = solo.loss_history()
    first, second = make(2023), make(2024)
    first.train_online(15)
    second.train_online(18)
    assert first._arena.info()["allocations"] > 0 and second._arena.info()["allocations"] > 0
    h_first = first.loss_history()
    first.close()
    assert first._arena.released and len(first.loss_history()) == 15 and first.loss_history() == h_first
    engine.release_graph_memory()                       # (ABI 2's device-wide release would have freed `second`'s memory here)
    third = make(7)
    third.train_online(12)                              # new captures: their allocations re-use what `first` gave back
    second.train_online(22)
    third.close()
    h = second.loss_history()
    second.close()
    assert len(h) == 40 and np.all(np.isfinite(h))
    assert np.allclose(h, h_solo, rtol=1e-5, atol=1e-6), np.abs(np.array(h) - np.array(h_solo)).max()

"""GPU tests of the C ABI's threading contract (include/nddm.h: "re-entrant and thread-safe given distinct output
buffers"): concurrent host threads on several streams, a stalled stream while another stream cycles through many
launches, and a captured hipGraph replayed while eager launches go on.  Every result must equal the serial run bit for
bit -- the device memory a launch borrows from the library (queue words, scratch) is keyed on completion, not on a
launch count."""
import threading

import numpy as np
import pytest

import prior_util

pytestmark = pytest.mark.gpu


def _cases():
    """Small launches of different shapes (training-loop sized), with and without summaries / trials."""
    rng = np.random.default_rng(5)
    cases = []
    for i in range(8):
        B = int(rng.integers(8, 160))
        N = int(rng.integers(20, 301))
        dt, ms = ((0.01, 400.0), (0.001, 4000.0))[i % 2]
        cases.append(dict(B=B, N=N, dt=dt, max_steps=ms, seed=100 + i, set_offset=1000 * i,
                          want_trials=(i % 4 != 3), want_summary=(i % 3 != 2)))
    cases.append(dict(B=3000, N=64, dt=0.01, max_steps=400.0, seed=77, set_offset=5, want_trials=True, want_summary=True))
    return cases


def _run(engine, torch, p_dev, c):
    r = engine.simulate(engine.BASIC_DDM_DC, p_dev[:c["B"]], c["N"], dt=c["dt"], max_steps=c["max_steps"], seed=c["seed"],
                        set_offset=c["set_offset"], fast=True, want_trials=c["want_trials"] or not c["want_summary"],
                        want_summary=c["want_summary"])
    return r.get("trials"), r.get("summary")


def _same(torch, a, b):
    if a is None or b is None:
        return a is None and b is None
    return bool(torch.equal(torch.nan_to_num(a), torch.nan_to_num(b)))


def test_two_threads_two_streams_each_bit_equal_to_serial():
    """2 host threads x 2 streams each x 200+ small launches (mixed shapes, with and without summaries), all in flight
    together: every result equals the serial reference."""
    import torch
    from bayesflow_nddms_amd import engine
    p_dev = torch.as_tensor(prior_util.basic_prior(4096, 11)).cuda()
    cases = _cases()
    ref = [_run(engine, torch, p_dev, c) for c in cases]
    torch.cuda.synchronize()
    errors, results = [], {}

    def worker(tid):
        try:
            streams = [torch.cuda.Stream(), torch.cuda.Stream()]
            keep = []
            for it in range(210):
                ci = (it * 3 + tid) % len(cases)
                with torch.cuda.stream(streams[it % 2]):
                    keep.append((ci, _run(engine, torch, p_dev, cases[ci])))
            for s in streams:
                s.synchronize()
            results[tid] = keep
        except Exception as e:          # noqa: BLE001 -- surfaced in the main thread
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    n = 0
    for tid, keep in results.items():
        for ci, (t, s) in keep:
            assert _same(torch, t, ref[ci][0]) and _same(torch, s, ref[ci][1]), (tid, ci)
            n += 1
    assert n >= 400


def test_stalled_stream_does_not_lose_its_launch_resources():
    """Stream A is stalled (a long spin kernel) with launches queued behind the stall; meanwhile stream B goes through
    300 launches -- more than any pool the library keeps.  A's results must still be right (a round-robin hand-out keyed
    on the launch count would have given A's queue words and scratch to B's launches while A's kernels were pending)."""
    import torch
    from bayesflow_nddms_amd import engine
    p_dev = torch.as_tensor(prior_util.basic_prior(4096, 12)).cuda()
    cases = _cases()
    ref = [_run(engine, torch, p_dev, c) for c in cases]
    torch.cuda.synchronize()
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(a):
        torch.cuda._sleep(int(3e8))                     # ~0.1-0.15 s of GPU time
        got_a = [_run(engine, torch, p_dev, c) for c in cases]
    with torch.cuda.stream(b):
        got_b = [(i % len(cases), _run(engine, torch, p_dev, cases[i % len(cases)])) for i in range(300)]
    torch.cuda.synchronize()
    for ci, (t, s) in enumerate(got_a):
        assert _same(torch, t, ref[ci][0]) and _same(torch, s, ref[ci][1]), ("A", ci)
    for ci, (t, s) in got_b:
        assert _same(torch, t, ref[ci][0]) and _same(torch, s, ref[ci][1]), ("B", ci)


@pytest.mark.parametrize("B", [32, 3000])
def test_graph_replay_while_eager_launches_cycle(B):
    """A captured launch owns its queue words and scratch: replaying the graph on one stream while 300 eager launches run on
    another leaves both right.  Also: the FIRST call of a shape may itself be the captured one."""
    import torch
    from bayesflow_nddms_amd import _lib, engine
    p_dev = torch.as_tensor(prior_util.basic_prior(4096, 13)).cuda()
    out = torch.empty((B, 150, 2), device="cuda")
    summ = torch.empty((B, 10), device="cuda")
    kw = dict(dt=.01, max_steps=400, seed=9, set_offset=3, fast=True)
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            engine.simulate(engine.BASIC_DDM_DC, p_dev[:B], 150, out_trials=out, out_summary=summ, **kw)
    torch.cuda.synchronize()
    r = engine.simulate(engine.BASIC_DDM_DC, p_dev[:B], 150, **kw)
    ref_t, ref_s = r["trials"].clone(), r["summary"].clone()
    cases = _cases()
    ref = [_run(engine, torch, p_dev, c) for c in cases]
    torch.cuda.synchronize()
    for rep in range(6):
        out.zero_(); summ.fill_(-7.0)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            g.replay()
        eager = [(i % len(cases), _run(engine, torch, p_dev, cases[i % len(cases)])) for i in range(50)]
        torch.cuda.synchronize()
        assert torch.equal(out, ref_t), rep
        assert torch.equal(torch.nan_to_num(summ), torch.nan_to_num(ref_s)), rep
        for ci, (t, s) in eager:
            assert _same(torch, t, ref[ci][0]) and _same(torch, s, ref[ci][1]), (rep, ci)
    del g
    _lib.check(_lib.lib().nddm_release_graph_memory())
    r2 = engine.simulate(engine.BASIC_DDM_DC, p_dev[:B], 150, **kw)          # the library stays usable
    assert torch.equal(r2["trials"], ref_t)


def test_tuning_knobs_are_snapshotted_per_call():
    """nddm_set_tuning from one thread while another thread launches: results never depend on the knobs (they only move
    work around), and nothing crashes."""
    import torch
    from bayesflow_nddms_amd import _lib, engine
    p_dev = torch.as_tensor(prior_util.basic_prior(512, 14)).cuda()
    c = dict(B=300, N=100, dt=0.01, max_steps=400.0, seed=3, set_offset=0, want_trials=True, want_summary=True)
    ref = _run(engine, torch, p_dev, c)
    torch.cuda.synchronize()
    stop = threading.Event()

    def flipper():
        L = _lib.lib()
        i = 0
        while not stop.is_set():
            L.nddm_set_tuning(1 + i % 3, (2, 4, 8)[i % 3], 4 + 4 * (i % 4), 8 + 8 * (i % 2), 0, 0)
            i += 1
        L.nddm_set_tuning(0, 0, 0, 0, 0, 0)

    th = threading.Thread(target=flipper)
    th.start()
    try:
        for _ in range(150):
            t, s = _run(engine, torch, p_dev, c)
            assert _same(torch, t, ref[0]) and _same(torch, s, ref[1])
    finally:
        stop.set()
        th.join()


def test_more_streams_than_launch_slots():
    """With the slot limit forced down to the slots that already exist, launches on six fresh streams must queue behind
    in-flight launches of other streams (a stream-side wait on the slot's completion event) -- and still be right."""
    import torch
    from bayesflow_nddms_amd import _lib, engine
    p_dev = torch.as_tensor(prior_util.basic_prior(4096, 15)).cuda()
    cases = _cases()
    ref = [_run(engine, torch, p_dev, c) for c in cases]
    torch.cuda.synchronize()
    L = _lib.lib()
    L.nddm_debug_set_slot_limit(1)
    try:
        streams = [torch.cuda.Stream() for _ in range(6)]
        got = []
        for it in range(120):
            ci = it % len(cases)
            with torch.cuda.stream(streams[it % 6]):
                if it % 17 == 0:
                    torch.cuda._sleep(int(2e7))
                got.append((ci, _run(engine, torch, p_dev, cases[ci])))
        torch.cuda.synchronize()
    finally:
        L.nddm_debug_set_slot_limit(256)
    for ci, (t, s) in got:
        assert _same(torch, t, ref[ci][0]) and _same(torch, s, ref[ci][1]), ci


def test_per_thread_default_stream_handle():
    """hipStreamPerThread is ONE handle value for a different stream in every host thread: the library must not treat two
    such launches as ordered with each other.  Three threads call the C ABI directly with that handle, concurrently."""
    import ctypes
    import torch
    from bayesflow_nddms_amd import _lib, engine
    L = _lib.lib()
    B, N = 600, 200
    p_dev = torch.as_tensor(prior_util.basic_prior(B, 16)).cuda()
    ref = engine.simulate(engine.BASIC_DDM_DC, p_dev, N, dt=0.01, max_steps=400, seed=21, set_offset=0, fast=True)
    torch.cuda.synchronize()
    outs = [(torch.empty((B, N, 2), device="cuda"), torch.empty((B, 10), device="cuda")) for _ in range(3)]
    torch.cuda.synchronize()
    per_thread = ctypes.c_void_p(2)                      # hipStreamPerThread
    errors = []

    def worker(i):
        try:
            t, s = outs[i]
            for _ in range(60):
                rc = L.nddm_basic_ddm_dc_simulate(p_dev.data_ptr(), B, N, 0.01, 400, 21, 0, 1, t.data_ptr(), s.data_ptr(), per_thread)
                assert rc == 0, L.nddm_last_error()
            torch.cuda.synchronize()
        except Exception as e:          # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    for t, s in outs:
        assert torch.equal(t, ref["trials"]) and torch.equal(torch.nan_to_num(s), torch.nan_to_num(ref["summary"]))


def _hip_runtime():
    """The HIP runtime this process already uses (torch's copy), through ctypes: raw hipStreamCreate / hipStreamDestroy."""
    import ctypes
    path = None
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line:
                path = line.split()[-1]
                break
    assert path, "libamdhip64 is not mapped"
    hip = ctypes.CDLL(path)
    hip.hipStreamCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
    hip.hipStreamDestroy.argtypes = [ctypes.c_void_p]
    return hip


def test_destroyed_stream_with_work_in_flight_and_a_recycled_handle():
    """A C-ABI caller may destroy a stream whose last launches are still in flight and create a new one: handle values
    are recycled.  The library's "same stream as last time" slot reuse must not hand the old stream's queue words to
    the new stream while the old launch runs (it now waits on the slot's completion event).  Raw HIP streams and direct
    C-ABI calls, as a C caller would make them (no torch object ever refers to the destroyed stream): a stream gets a long
    launch and small ones queued behind it and is destroyed at once; new streams are created until the handle value
    repeats (or 64 times) and launches go out on each; every result equals the serial reference."""
    import ctypes
    import torch
    from bayesflow_nddms_amd import _lib, engine
    hip, L = _hip_runtime(), _lib.lib()
    p_dev = torch.as_tensor(prior_util.basic_prior(200000, 13)).cuda()
    shapes = [(24 + 8 * i, 60 + 20 * i) for i in range(8)]                      # (B, N) of the small launches

    def launch(stream, B, N, seed, out, summ):
        _lib.check(L.nddm_basic_ddm_dc_simulate(p_dev.data_ptr(), B, N, 0.01, 400, seed, 0, 1, out.data_ptr(), summ.data_ptr(),
                                                ctypes.c_void_p(stream)))

    def buffers(B, N):
        return torch.empty((B, N, 2), device="cuda"), torch.empty((B, 10), device="cuda")

    ref = []
    for i, (B, N) in enumerate(shapes):
        o, s = buffers(B, N)
        launch(None, B, N, 50 + i, o, s)
        ref.append((o, s))
    big_o, big_s = buffers(200000, 300)
    torch.cuda.synchronize()
    h = ctypes.c_void_p()
    assert hip.hipStreamCreate(ctypes.byref(h)) == 0
    old = h.value
    _lib.check(L.nddm_basic_ddm_dc_simulate(p_dev.data_ptr(), 200000, 300, 0.001, 4000, 1, 0, 1, big_o.data_ptr(), big_s.data_ptr(),
                                            ctypes.c_void_p(old)))                # ~7 ms of GPU time in front of the small launches
    got_old = []
    for i, (B, N) in enumerate(shapes):
        o, s = buffers(B, N)
        launch(old, B, N, 50 + i, o, s)
        got_old.append((o, s))
    assert hip.hipStreamDestroy(ctypes.c_void_p(old)) == 0      # work still pending: HIP finishes it, the handle is gone
    made, got_new, recycled = [], [], False
    for j in range(64):
        h2 = ctypes.c_void_p()
        assert hip.hipStreamCreate(ctypes.byref(h2)) == 0
        made.append(h2.value)
        for i in ((j % len(shapes),) if h2.value != old else range(len(shapes))):
            o, s = buffers(*shapes[i])
            launch(h2.value, shapes[i][0], shapes[i][1], 50 + i, o, s)
            got_new.append((i, o, s))
        if h2.value == old:
            recycled = True
            break
    torch.cuda.synchronize()
    for i, (o, s) in enumerate(got_old):
        assert _same(torch, o, ref[i][0]) and _same(torch, s, ref[i][1]), ("destroyed stream", i)
    for i, o, s in got_new:
        assert _same(torch, o, ref[i][0]) and _same(torch, s, ref[i][1]), ("new stream", i, recycled)
    for v in made:
        assert hip.hipStreamDestroy(ctypes.c_void_p(v)) == 0
    # the library keeps working on torch's streams afterwards
    r = engine.simulate(engine.BASIC_DDM_DC, p_dev[:shapes[0][0]], shapes[0][1], dt=0.01, max_steps=400, seed=50, set_offset=0, fast=True)
    torch.cuda.synchronize()
    assert _same(torch, r["trials"], ref[0][0]) and _same(torch, r["summary"], ref[0][1])

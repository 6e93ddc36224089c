"""A battery over the HOST side of libnddm_hip.so that needs no GPU (argument validation, error text, the graph-arena bookkeeping, the
developer knobs under one mutex, the no-device error paths of every entry point), run against the library named by NDDM_HIP_LIB -- in a
child process against an AddressSanitizer + UndefinedBehaviorSanitizer build of the product's own host code
(tests/test_host_logic.py::test_host_shim_is_clean_under_address_and_undefined_behaviour_sanitizers).  It never imports torch: the
library is loaded directly, as a C program would.  Prints one line, `SHIM <sha256 of every return code and error string>`."""
import ctypes
import hashlib
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def battery(path):
    from bayesflow_nddms_amd import _lib
    L = ctypes.CDLL(path)
    _lib._declare(L)
    h = hashlib.sha256()
    log = []

    def note(*vals):
        for v in vals:
            log.append(v)
            h.update(repr(v).encode())

    err = lambda: L.nddm_last_error()
    note(L.nddm_abi_version(), L.nddm_summary_k(), [L.nddm_model_nparams(m) for m in range(-2, 8)])
    assert L.nddm_build_info().startswith(b"hipcc=") and len(L.nddm_source_hash()) in (7, 64)
    n = ctypes.c_int(-1)
    note(L.nddm_device_count(None), b"NULL" in err())
    rc = L.nddm_device_count(ctypes.byref(n))
    no_gpu = rc != 0 or n.value == 0
    d = ctypes.c_void_p(4096)                                   # a "device pointer" that is never dereferenced on the host
    # ---- argument validation of every simulator entry (all before the first HIP call)
    for B, N, dt, cap, flags in ((-1, 10, .01, 400, 0), (4, 0, .01, 400, 0), (4, -3, .01, 400, 0), (4, 10, -.01, 400, 0), (4, 10, float("nan"), 400, 0),
                                 (4, 10, float("inf"), 400, 0), (4, 10, .01, -1, 0), (4, 10, .01, 1 << 30, 0), (4, 10, .01, 400, 16), (4, 10, .01, 400, 7),
                                 (4, 10, .01, 400, 6), (4, 10, .01, 20000, 4), (4, 10, .01, 400, 8 | 4), (4, 10, .01, 400, 8 | 2),
                                 (1 << 30, 100000, .01, 400, 0), (0, 10, .01, 400, 0)):
        note(L.nddm_basic_ddm_dc_simulate(d, B, N, dt, cap, 0, 0, flags, d, None, None), err())
        note(L.nddm_single_trial_simulate(d, B, N, dt, cap, 0, 0, flags, d, d, None), err())
        note(L.nddm_single_trial_alt_simulate(d, B, N, dt, cap, 0, 0, flags, None, d, None), err())
        note(L.nddm_alpha_not_scaled_simulate(d, B, N, dt, cap, 0, 0, flags, 0.1, 0, d, None, d, None), err())
        note(L.nddm_explicit_boundary_simulate(d, d, B, N, dt, cap, 0, 0, flags, d, None, None), err())
        for model in (-1, 0, 1, 2, 3, 4, 5):
            note(L.nddm_simulate(model, d, d, B, N, dt, cap, 0, 0, flags, 0.1, 0, d, d, d, None), err())
            note(L.nddm_simulate_indirect(model, d, d, B, N, dt, cap, 0, 0, d, flags, 0.1, 0, d, d, d, None), err())
            note(L.nddm_simulate_codes(model, d, B, N, dt, cap, 0, 0, None, flags, d, None, None, None), err())
    for so in (1 << 60, (1 << 60) - 2, (1 << 64) - 1):          # the 60-bit set index
        note(L.nddm_basic_ddm_dc_simulate(d, 4, 10, .01, 400, 5, so, 0, d, None, None), err())
    note(L.nddm_basic_ddm_dc_simulate(None, 4, 10, .01, 400, 0, 0, 0, d, None, None), err())
    note(L.nddm_basic_ddm_dc_simulate(d, 4, 10, .01, 400, 0, 0, 0, None, None, None), err())
    note(L.nddm_explicit_boundary_simulate(d, None, 4, 10, .01, 400, 0, 0, 0, d, None, None), err())
    note(L.nddm_simulate_codes(0, d, 4, 10, .01, 400, 0, 0, None, 0, None, d, d, None), err())
    note(L.nddm_simulate_codes(1, d, 4, 10, .01, 400, 0, 0, None, 0, d, d, d, None), err())
    note(L.nddm_simulate_codes(0, d, 4, 10, .0001, 20000, 0, 0, None, 0, d, d, d, None), err())
    for model, B, N, dt in ((1, 4, 10, .01), (0, -1, 10, .01), (0, 4, 0, .01), (0, 4, 10, 0.0), (0, 0, 10, .01)):
        note(L.nddm_decode_codes(model, d, d, B, N, dt, d, None), err())
    note(L.nddm_decode_codes(0, None, d, 4, 10, .01, d, None), err())
    for model, B in ((3, 4), (4, 4), (9, 4), (0, -1), (0, 0), (1, 0)):
        note(L.nddm_draw_prior(model, B, 1, 0, 1.0, d, None), err())
        note(L.nddm_draw_prior_indirect(model, B, 1, 0, d, 1.0, d, None), err())
    note(L.nddm_draw_prior(0, 4, 1, 0, 1.0, None, None), err())
    for B, N, flags in ((-1, 10, 0), (4, 0, 0), (4, 1 << 30, 0), (4, 10, 2), (4, 10, 8), (1 << 30, 100000, 1), (0, 10, 1)):
        note(L.nddm_simulratcliff(d, B, N, 1, 0, flags, 0.1, 0, d, d, d, None), err())
    note(L.nddm_simulratcliff(None, 4, 10, 1, 0, 0, 0.1, 0, d, d, d, None), err(), L.nddm_simulratcliff(d, 4, 10, 1, 0, 0, 0.1, 0, None, None, None, None), err(),
         L.nddm_simulratcliff(d, 4, 10, 1, 1 << 60, 0, 0.1, 0, d, None, None, None), err())
    note(L.nddm_debug_normals(d, -1, 1, 2, 0, d, None), L.nddm_debug_normals(d, 0, 1, 2, 0, d, None), L.nddm_debug_normals(None, 3, 1, 2, 1, d, None), err())
    # ---- valid arguments: the call reaches the HIP runtime, which has no device here -> an error code and its text, never a crash
    if no_gpu:
        rcs = [L.nddm_basic_ddm_dc_simulate(d, 4, 10, .01, 400, 0, 0, 1, d, d, None), L.nddm_single_trial_simulate(d, 4, 10, .001, 4000, 0, 0, 8, d, d, None),
               L.nddm_alpha_not_scaled_simulate(d, 4, 10, .01, 400, 0, 0, 3, .1, 0, d, d, d, None), L.nddm_decode_codes(0, d, d, 4, 10, .01, d, None),
               L.nddm_draw_prior(0, 4, 1, 0, 1.0, d, None), L.nddm_debug_normals(d, 3, 1, 2, 1, d, None), L.nddm_release_graph_memory(),
               L.nddm_simulratcliff(d, 4, 600, 1, 0, 1, 0.1, 0, d, d, d, None),
               L.nddm_set_device(0)]
        assert all(rc in (_lib.NDDM_ERR_HIP, _lib.NDDM_ERR_NO_DEVICE) for rc in rcs), rcs
        assert len(err()) > 0
        note(rcs)
    # ---- developer knobs (one mutex) and the per-thread launch record
    note(L.nddm_set_tuning(0, 1, 0, 0, 0, 0), err(), L.nddm_set_tuning(0, -2, 0, 0, 0, 0), L.nddm_set_tuning(3, 4, 16, -5, 1024, 64), L.nddm_set_tuning(0, 0, 0, 0, 0, 0))
    note(L.nddm_set_debug_trace(d, -1, 0), L.nddm_set_debug_trace(d, 0, -1), L.nddm_set_debug_trace(d, 8, 8), L.nddm_set_debug_trace(None, 5, 5))
    note(L.nddm_set_ordering(0), L.nddm_set_ordering(1), L.nddm_debug_set_slot_limit(-4), L.nddm_debug_set_slot_limit(10 ** 6), L.nddm_debug_set_slot_limit(256))
    geo = (ctypes.c_int32 * 8)(*range(8))
    note(L.nddm_debug_last_launch(None), L.nddm_debug_last_launch(geo), list(geo))
    # ---- graph arenas: create / bind / info / release, error cases, bindings per thread
    a, b, prev = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_uint64(99)
    note(L.nddm_graph_arena_create(None), L.nddm_graph_arena_create(ctypes.byref(a)), L.nddm_graph_arena_create(ctypes.byref(b)), a.value != 0, b.value not in (0, a.value))
    note(L.nddm_graph_arena_bind(a.value, ctypes.byref(prev)), prev.value, L.nddm_graph_arena_bind(b.value, ctypes.byref(prev)), prev.value == a.value)
    nb, na = ctypes.c_uint64(7), ctypes.c_int32(7)
    note(L.nddm_graph_arena_info(a.value, ctypes.byref(nb), ctypes.byref(na)), nb.value, na.value, L.nddm_graph_arena_info(0, None, None),
         L.nddm_graph_arena_info(12345678, None, None), err())
    note(L.nddm_graph_arena_release(0), err(), L.nddm_graph_arena_release(b.value), L.nddm_graph_arena_bind(0, ctypes.byref(prev)), prev.value,
         L.nddm_graph_arena_release(b.value), L.nddm_graph_arena_bind(b.value, None), L.nddm_graph_arena_info(b.value, None, None), L.nddm_graph_arena_release(a.value))
    # many threads at once: arenas created, bound, queried and released while others do the same and turn the knobs
    results, errors = [], []

    def worker(seed):
        try:
            mine = []
            for i in range(150):
                x, p = ctypes.c_uint64(0), ctypes.c_uint64(0)
                assert L.nddm_graph_arena_create(ctypes.byref(x)) == 0
                mine.append(x.value)
                assert L.nddm_graph_arena_bind(x.value, ctypes.byref(p)) == 0
                assert L.nddm_graph_arena_info(x.value, ctypes.byref(nb2 := ctypes.c_uint64(1)), None) == 0 and nb2.value == 0
                L.nddm_set_tuning(0, 0, (seed + i) % 3 * 8, 0, 0, 0)
                L.nddm_set_debug_trace(None, 0, 0)
                L.nddm_debug_set_slot_limit(1 + (seed + i) % 9)
                g = (ctypes.c_int32 * 8)()
                L.nddm_debug_last_launch(g)
                L.nddm_basic_ddm_dc_simulate(d, -1, 10, .01, 400, 0, 0, 0, d, None, None)        # (thread-local error text)
                assert b"B < 0" in L.nddm_last_error()
                if i % 3 == 2:
                    for v in mine[:-1]:
                        assert L.nddm_graph_arena_release(v) == 0
                    mine = mine[-1:]
            for v in mine:
                assert L.nddm_graph_arena_release(v) == 0
            assert L.nddm_graph_arena_bind(0, None) == 0
            results.append(seed)
        except Exception as e:                                              # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=worker, args=(s,)) for s in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors and sorted(results) == list(range(8)), errors
    L.nddm_set_tuning(0, 0, 0, 0, 0, 0); L.nddm_debug_set_slot_limit(256)
    return h.hexdigest(), len(log)


if __name__ == "__main__":
    from bayesflow_nddms_amd import build
    digest, n = battery(os.environ.get("NDDM_HIP_LIB") or build.SO_PATH)
    print(f"SHIM {digest} ({n} checks)")

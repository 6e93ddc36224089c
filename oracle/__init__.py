"""ctypes front-end of the CPU oracle (oracle/ddm_oracle.c).

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; bayesflow_nddms_amd never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_SRC = os.path.join(_HERE, "ddm_oracle.c")

M_BASIC, M_SINGLE, M_ALT, M_ALPHA_NS, M_EXPLICIT = 0, 1, 2, 3, 4
SUMMARY_K = 10

_lib = None


def _src_hash():
    import hashlib
    with open(_SRC, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def build(force=False):
    """gcc the oracle into oracle/liboracle.so (contraction off: every fma is spelled out).  Stale = built from other source CONTENT
    than the tree's (the hash is kept beside the library: a snapshot copied to the GPU box has arbitrary file times)."""
    stamp = _SO + ".srchash"
    if not force and os.path.exists(_SO) and os.path.exists(_SRC):
        try:
            if open(stamp).read().strip() == _src_hash():
                return _SO
        except OSError:
            pass
    elif not force and os.path.exists(_SO):
        return _SO                                           # (a prebuilt library without its source beside it)
    tmp = f"{_SO}.{os.getpid()}.tmp"
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-mfma", "-fno-math-errno", "-fopenmp", "-fPIC", "-shared",
           "-o", tmp, _SRC, "-lm"]
    subprocess.check_call(cmd)
    os.replace(tmp, _SO)
    with open(stamp, "w") as f:
        f.write(_src_hash())
    return _SO


def lib():
    global _lib
    if _lib is None:
        # NDDM_ORACLE_LIB: another build of the same source (tests/test_oracle_golden.py runs the oracle's battery under
        # AddressSanitizer + UndefinedBehaviorSanitizer in a child process)
        L = ctypes.CDLL(os.environ.get("NDDM_ORACLE_LIB") or build())
        L.oracle_mt_seed.argtypes = [ctypes.c_uint32]
        L.oracle_mt_double.restype = ctypes.c_double
        L.oracle_mt_gauss.restype = ctypes.c_double
        dp = ctypes.POINTER(ctypes.c_double)
        L.oracle_mt_basic.argtypes = [dp, ctypes.c_int, ctypes.c_double, ctypes.c_double, dp]
        L.oracle_mt_single.argtypes = [dp, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                       ctypes.c_int, dp]
        L.oracle_mt_explicit.argtypes = [dp, dp, ctypes.c_int, ctypes.c_double, ctypes.c_double, dp]
        L.oracle_mt_explicit.restype = ctypes.c_int
        L.oracle_mt_ratcliff.argtypes = [ctypes.c_int] + [ctypes.c_double] * 8 + [dp]
        L.oracle_model_nparams.argtypes = [ctypes.c_int]
        L.oracle_model_nparams.restype = ctypes.c_int
        fp = ctypes.POINTER(ctypes.c_float)
        L.oracle_philox_simulate.argtypes = [
            ctypes.c_int, ctypes.c_int, fp, fp, ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_uint64,
            ctypes.c_uint64, ctypes.c_float, ctypes.c_int, fp, ctypes.POINTER(ctypes.c_int32), fp, fp,
            ctypes.c_int]
        L.oracle_philox_simulate.restype = ctypes.c_int
        ip = ctypes.POINTER(ctypes.c_int32)
        L.oracle_philox_simulate_f64.argtypes = [ctypes.c_int, fp, ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_int,
                                                 ctypes.c_uint64, ctypes.c_uint64, ip, ip, fp, fp, ctypes.c_int]
        L.oracle_philox_simulate_f64.restype = ctypes.c_int
        L.oracle_philox_ratcliff.argtypes = [fp, ctypes.c_int64, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_float, ctypes.c_int,
                                             fp, fp, fp, ctypes.c_int]
        L.oracle_philox_ratcliff.restype = ctypes.c_int
        L.oracle_philox_normals4.argtypes = [ctypes.c_uint32] * 6 + [fp]
        L.oracle_philox_block.argtypes = [ctypes.c_uint32] * 6 + [ctypes.POINTER(ctypes.c_uint32)]
        _lib = L
    return _lib


def _dptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _fptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


# ----------------------------------------------------------------------------- MT19937 mode
def mt_seed(seed):
    """np.random.seed(int) equivalent for the oracle's global legacy stream."""
    lib().oracle_mt_seed(int(seed) & 0xFFFFFFFF)


def mt_gauss():
    return lib().oracle_mt_gauss()


def mt_double():
    return lib().oracle_mt_double()


def mt_basic(params, n_trials, dt=0.01, max_steps=400.0):
    """basic_ddm_dc.py:114-125 simulate_trials on the NumPy legacy stream (float64)."""
    p = np.ascontiguousarray(params, dtype=np.float64)
    out = np.empty((n_trials, 2), dtype=np.float64)
    lib().oracle_mt_basic(_dptr(p), n_trials, dt, max_steps, _dptr(out))
    return out


def mt_single(params, n_trials, dt=0.01, max_steps=400.0, gamma=1.0, variant=0):
    """single_trial_alpha_not_scaled.py:144-155 (variant 0; gamma for _scale/_scale2) / :963-974 (variant 1, _alt)."""
    p = np.ascontiguousarray(params, dtype=np.float64)
    assert p.shape == (7,)
    out = np.empty((n_trials, 2), dtype=np.float64)
    lib().oracle_mt_single(_dptr(p), n_trials, dt, max_steps, gamma, variant, _dptr(out))
    return out


def mt_explicit(drift, bounds, beta, ter, dc, dt=0.01, max_steps=400.0):
    """imputation_from_stahl_not_scaled.py:120-148 over a boundary vector."""
    p = np.array([drift, beta, ter, dc], dtype=np.float64)
    b = np.ascontiguousarray(bounds, dtype=np.float64)
    out = np.empty(len(b), dtype=np.float64)
    rc = lib().oracle_mt_explicit(_dptr(p), _dptr(b), len(b), dt, max_steps, _dptr(out))
    if rc != 0:
        raise ValueError("Trial-level boundary cannot be less than zero")
    return out


def mt_ratcliff(N=100, Alpha=1, Tau=.4, Nu=1, Beta=.5, rangeTau=0, rangeBeta=0, Eta=.3, Varsigma=1):
    """pyhddmjagsutils.py:47-176 simulratcliff on the NumPy legacy stream."""
    out = np.empty(N, dtype=np.float64)
    lib().oracle_mt_ratcliff(N, Alpha, Tau, Nu, Beta, rangeTau, rangeBeta, Eta, Varsigma, _dptr(out))
    return out


# ----------------------------------------------------------------------------- Philox mode
def philox_simulate(model, params, n_trials, dt=0.01, max_steps=400.0, seed=0, set_offset=0, bounds=None,
                    ext_sigma=0.0, ext_mode=0, bridge=False, packed=False, want_trials=True, want_k=False, want_summary=True,
                    want_ext=False, threads=1):
    """The device stream on the CPU: element-wise checker of the HIP kernels.

    Returns dict(trials f32[B,N,2], k i32[B,N], summary f32[B,10], ext f32[B]) (requested keys only)."""
    L = lib()
    P = L.oracle_model_nparams(model)
    p = np.ascontiguousarray(params, dtype=np.float32)
    if p.ndim == 1:
        p = p[None]
    assert p.shape[1] == P, f"model {model} takes {P} parameters per set, got {p.shape}"
    B = p.shape[0]
    max_k = int(np.ceil(max_steps))
    bnd = None
    if model == M_EXPLICIT:
        bnd = np.ascontiguousarray(bounds, dtype=np.float32).reshape(B, n_trials)
    res = {}
    trials = np.empty((B, n_trials, 2), np.float32) if want_trials else None
    k = np.empty((B, n_trials), np.int32) if want_k else None
    summ = np.empty((B, SUMMARY_K), np.float32) if want_summary else None
    ext = np.empty((B,), np.float32) if want_ext else None
    rc = L.oracle_philox_simulate(
        model, int(bool(bridge)) | (2 if packed else 0), _fptr(p), _fptr(bnd), B, n_trials, np.float32(dt), max_k, seed, set_offset,
        np.float32(ext_sigma), ext_mode, _fptr(trials),
        None if k is None else k.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), _fptr(summ), _fptr(ext),
        threads)
    if rc != 0:
        raise ValueError(f"oracle_philox_simulate rc={rc}")
    if want_trials:
        res["trials"] = trials
    if want_k:
        res["k"] = k
    if want_summary:
        res["summary"] = summ
    if want_ext:
        res["ext"] = ext
    return res


def philox_ratcliff(params, n_trials, seed=0, set_offset=0, ext_sigma=0.0, ext_mode=0, want_summary=True, want_ext=False, threads=1):
    """Section D: simulratcliff (pyhddmjagsutils.py:47-176) on the device stream in float32 -- the element-wise checker of
    nddm_simulratcliff.  params [B, 6] = Nu, Alpha, Beta, Tau, Eta, Varsigma.  dict(trials f32[B,N,2] = (y, acc), summary, ext)."""
    p = np.ascontiguousarray(params, dtype=np.float32)
    if p.ndim == 1:
        p = p[None]
    assert p.shape[1] == 6
    B = p.shape[0]
    trials = np.empty((B, n_trials, 2), np.float32)
    summ = np.empty((B, SUMMARY_K), np.float32) if want_summary else None
    ext = np.empty((B,), np.float32) if want_ext else None
    rc = lib().oracle_philox_ratcliff(_fptr(p), B, int(n_trials), seed, set_offset, np.float32(ext_sigma), int(ext_mode), _fptr(trials),
                                      _fptr(summ), _fptr(ext), threads)
    if rc != 0:
        raise ValueError(f"oracle_philox_ratcliff rc={rc}")
    res = {"trials": trials}
    if want_summary:
        res["summary"] = summ
    if want_ext:
        res["ext"] = ext
    return res


def philox_simulate_f64(model, params, n_trials, dt=0.01, max_steps=400.0, seed=0, set_offset=0, threads=1, want_outputs=False):
    """The REFERENCE'S float64 recurrence (basic_ddm_dc.py:91-103; single_trial_alpha_not_scaled.py:113-128) on the device
    stream's normals: (k i32[B,N], choice i32[B,N]) -- compare with philox_simulate(want_k=True) trial by trial to state how
    often the float32 integrator of the product ends on another (step, choice) than the float64 one.
    want_outputs=True: a dict instead -- 'k', 'choice', and 'trials' f32[B,N,2] / 'summary' f32[B,10] exactly as the device writes
    them under NDDM_STATE_F64 (the checker of that option)."""
    L = lib()
    P = L.oracle_model_nparams(model)
    p = np.ascontiguousarray(params, dtype=np.float32)
    if p.ndim == 1:
        p = p[None]
    assert p.shape[1] == P
    B = p.shape[0]
    k = np.empty((B, n_trials), np.int32)
    c = np.empty((B, n_trials), np.int32)
    ip = ctypes.POINTER(ctypes.c_int32)
    trials = np.empty((B, n_trials, 2), np.float32) if want_outputs else None
    summ = np.empty((B, SUMMARY_K), np.float32) if want_outputs else None
    rc = L.oracle_philox_simulate_f64(model, _fptr(p), B, n_trials, np.float32(dt), int(np.ceil(max_steps)), seed, set_offset,
                                      k.ctypes.data_as(ip), c.ctypes.data_as(ip), _fptr(trials), _fptr(summ), threads)
    if rc != 0:
        raise ValueError(f"oracle_philox_simulate_f64 rc={rc}")
    if want_outputs:
        return {"k": k, "choice": c, "trials": trials, "summary": summ}
    return k, c


def integrator_disagreement(model, params, n_trials, dt, max_steps, seed=0, set_offset=0, threads=1):
    """float32 (product arithmetic) against float64 (reference arithmetic) on the same normals: dict with the number of
    trials, the fraction whose (k, choice) differ, the fraction whose choice differs, and the largest |delta k|."""
    r = philox_simulate(model, params, n_trials, dt=dt, max_steps=max_steps, seed=seed, set_offset=set_offset, want_k=True,
                        want_summary=False, threads=threads)
    k32 = r["k"]
    t = r["trials"]
    c32 = (t[..., 1] if model == M_BASIC else np.sign(t[..., 0])).astype(np.int32)
    k64, c64 = philox_simulate_f64(model, params, n_trials, dt=dt, max_steps=max_steps, seed=seed, set_offset=set_offset, threads=threads)
    differ = (k32 != k64) | (c32 != c64)
    return {"trials": int(k32.size), "differ": float(differ.mean()), "choice_differs": float((c32 != c64).mean()),
            "max_abs_dk": int(np.abs(k32.astype(np.int64) - k64).max()), "n_differ": int(differ.sum())}


def philox_normals4(c0, c1, c2, c3, k0, k1):
    z = np.empty(4, np.float32)
    lib().oracle_philox_normals4(c0, c1, c2, c3, k0, k1, _fptr(z))
    return z


def philox_block(c0, c1, c2, c3, k0, k1):
    x = np.empty(4, np.uint32)
    lib().oracle_philox_block(c0, c1, c2, c3, k0, k1, x.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)))
    return x

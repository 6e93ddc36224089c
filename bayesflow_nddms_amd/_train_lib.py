"""ctypes binding of libnddm_train.so: the fused coupling half-layer of the amortizer's flow (csrc/train_kernels.hip).
Optional: `lib()` returns None when the library cannot be built or loaded, and the amortizer then runs its PyTorch path."""
import ctypes
import os

_lib = None
_tried = False


def lib():
    global _lib, _tried
    if _tried:
        return _lib
    _tried = True
    if os.environ.get("NDDM_NO_FUSED_COUPLING"):
        return None
    try:
        from .build import build_train
        import torch  # noqa: F401  (its HIP runtime first)
        L = ctypes.CDLL(build_train())
    except Exception:      # noqa: BLE001 -- no hipcc / no runtime: the PyTorch path serves
        return None
    c = ctypes
    fp, vp, i32, f32 = c.c_void_p, c.c_void_p, c.c_int, c.c_float
    L.nddm_train_coupling_supported.argtypes = [i32] * 4
    L.nddm_train_coupling_supported.restype = i32
    L.nddm_train_coupling_fwd.argtypes = [fp, i32, i32, fp, i32, fp, i32, i32, fp, fp, fp, fp, fp, fp, f32, i32, fp, i32, fp, i32, fp, fp, vp]
    L.nddm_train_coupling_fwd.restype = i32
    L.nddm_train_coupling_bwd.argtypes = [fp, i32, i32, fp, i32, fp, i32, i32, fp, fp, fp, f32, i32, fp, i32, fp, fp, fp, i32, fp, i32,
                                          fp, i32, fp, i32, i32, fp, i32, fp, i32, fp, fp, fp, fp, fp, fp, vp]
    L.nddm_train_coupling_bwd.restype = i32
    _lib = L
    return _lib

"""ctypes binding of libnddm_train.so: the amortizer's flow as one kernel each way (csrc/train_kernels.hip) and the summary
network's per-trial MLPs (csrc/train_deepset.hip), and the optimizer step on flat buffers (csrc/train_update.hip).
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
    L.nddm_train_flow_supported.argtypes = [i32] * 5
    L.nddm_train_flow_supported.restype = i32
    #                               L    R    D    d1   C    clamp params perm theta cond z_all out_all s_all h_all ld nll stream
    L.nddm_train_flow_fwd.argtypes = [i32, i32, i32, i32, i32, f32, vp, vp, fp, fp, fp, fp, fp, fp, fp, fp, vp]
    L.nddm_train_flow_fwd.restype = i32
    #                               ... params perm grads theta cond z_all out_all s_all h_all g_z g_ld g_nll gz_all gx gcond work stream
    L.nddm_train_flow_bwd.argtypes = [i32, i32, i32, i32, i32, f32, vp, vp, vp, fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, vp]
    L.nddm_train_flow_bwd.restype = i32
    L.nddm_deepset_supported.argtypes = [i32, i32]
    L.nddm_deepset_supported.restype = i32
    common = [fp, i32, i32, i32, i32, i32, fp, i32, fp, f32, fp, i32, fp, i32, fp, fp, fp, fp, fp, i32, fp, i32]   # x .. S_x (csrc/train_deepset.hip)
    L.nddm_deepset_mlp_fwd.argtypes = common + [fp, fp, fp, fp, i32, fp, i32, i32, vp]       # h1 h2 y pool_part | ldy extra n_extra extra_stride
    L.nddm_deepset_mlp_fwd.restype = i32
    L.nddm_deepset_mlp2_fwd.argtypes = common + [fp, fp, fp] + [fp] * 6 + [fp, fp, fp, vp]
    L.nddm_deepset_mlp2_fwd.restype = i32
    #                                        h1y h2y gx dctx wpart_y x1 | W1x..b3x | h1x h2x gpool gp_S gp_W gp_ldw gx_prev wpart_x ld_part stream
    L.nddm_deepset_mlp2_bwd.argtypes = common + [fp, fp, fp, fp, fp, fp] + [fp] * 6 + [fp, fp, fp, i32, fp, i32, fp, fp, i32, vp]
    L.nddm_deepset_mlp2_bwd.restype = i32
    L.nddm_deepset_mlp_bwd.argtypes = common + [fp, fp, fp, i32, fp, i32, fp, i32, fp, i32, fp, fp, i32, vp]   # h1 h2 gy ldgy gpool ...
    L.nddm_deepset_mlp_bwd.restype = i32
    L.nddm_deepset_reduce.argtypes = [fp, i32, i32, i32, i32, fp, vp]
    L.nddm_deepset_reduce.restype = i32
    i64 = c.c_longlong
    L.nddm_train_adam_step.argtypes = [fp, fp, fp, fp, i64, fp, f32, f32, f32, f32, f32, f32, f32, fp, fp, fp, fp, i32, fp, vp]
    L.nddm_train_adam_step.restype = i32
    L.nddm_train_stage.argtypes = [fp, fp, i64, fp, fp, i64, fp, f32, f32, vp]
    L.nddm_train_stage.restype = i32
    L.nddm_train_set2.argtypes = [fp, f32, f32, vp]
    L.nddm_train_set2.restype = i32
    _lib = L
    return _lib

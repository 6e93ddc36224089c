"""ctypes binding of libnddm_hip.so (include/nddm.h).  No CPU fallback exists: if the HIP
library is missing or cannot be loaded, importing the simulators fails loudly."""
import ctypes
import os

from .build import SO_PATH

NDDM_OK, NDDM_ERR_NULL, NDDM_ERR_SHAPE, NDDM_ERR_PARAM, NDDM_ERR_HIP, NDDM_ERR_NO_DEVICE = range(6)
GAUSS_EXACT, GAUSS_FAST, BRIDGE, GAUSS_PACKED, STATE_F64 = 0, 1, 2, 4, 8
ABI_VERSION = 4

_lib = None


class NddmLibraryError(ImportError):
    pass


def _declare(L):
    c = ctypes
    fp, vp = c.c_void_p, c.c_void_p      # device pointers travel as integers
    L.nddm_abi_version.restype = c.c_int
    L.nddm_last_error.restype = c.c_char_p
    L.nddm_device_count.argtypes = [c.POINTER(c.c_int)]
    L.nddm_set_device.argtypes = [c.c_int]
    L.nddm_summary_k.restype = c.c_int
    L.nddm_model_nparams.argtypes = [c.c_int]
    L.nddm_set_tuning.argtypes = [c.c_int] * 6
    L.nddm_set_debug_trace.argtypes = [c.c_void_p, c.c_int, c.c_int]
    L.nddm_set_ordering.argtypes = [c.c_int]
    L.nddm_debug_set_slot_limit.argtypes = [c.c_int]
    L.nddm_debug_last_launch.argtypes = [c.POINTER(c.c_int32)]
    common = [c.c_int64, c.c_int32, c.c_float, c.c_int32, c.c_uint64, c.c_uint64, c.c_uint32]
    for name in ("nddm_basic_ddm_dc_simulate", "nddm_single_trial_simulate", "nddm_single_trial_alt_simulate"):
        getattr(L, name).argtypes = [fp] + common + [fp, fp, vp]
    L.nddm_alpha_not_scaled_simulate.argtypes = [fp] + common + [c.c_float, c.c_int32, fp, fp, fp, vp]
    L.nddm_explicit_boundary_simulate.argtypes = [fp, fp] + common + [fp, fp, vp]
    L.nddm_simulratcliff.argtypes = [fp, c.c_int64, c.c_int32, c.c_uint64, c.c_uint64, c.c_uint32, c.c_float, c.c_int32, fp, fp, fp, vp]
    L.nddm_simulate.argtypes = [c.c_int32, fp, fp] + common + [c.c_float, c.c_int32, fp, fp, fp, vp]
    L.nddm_simulate_indirect.argtypes = [c.c_int32, fp, fp] + common[:-1] + [fp, c.c_uint32, c.c_float, c.c_int32, fp, fp, fp, vp]
    L.nddm_simulate_codes.argtypes = [c.c_int32, fp] + common[:-1] + [fp, c.c_uint32, fp, fp, fp, vp]
    L.nddm_decode_codes.argtypes = [c.c_int32, fp, fp, c.c_int64, c.c_int32, c.c_float, fp, vp]
    L.nddm_draw_prior.argtypes = [c.c_int32, c.c_int64, c.c_uint64, c.c_uint64, c.c_float, fp, vp]
    L.nddm_draw_prior_indirect.argtypes = [c.c_int32, c.c_int64, c.c_uint64, c.c_uint64, fp, c.c_float, fp, vp]
    L.nddm_graph_arena_create.argtypes = [c.POINTER(c.c_uint64)]
    L.nddm_graph_arena_bind.argtypes = [c.c_uint64, c.POINTER(c.c_uint64)]
    L.nddm_graph_arena_info.argtypes = [c.c_uint64, c.POINTER(c.c_uint64), c.POINTER(c.c_int32)]
    L.nddm_graph_arena_release.argtypes = [c.c_uint64]
    L.nddm_source_hash.restype = c.c_char_p
    L.nddm_build_info.restype = c.c_char_p
    L.nddm_debug_normals.argtypes = [fp, c.c_int64, c.c_uint32, c.c_uint32, c.c_uint32, fp, vp]
    for name in EXPORTS:
        if name not in ("nddm_last_error", "nddm_source_hash", "nddm_build_info"):
            getattr(L, name).restype = c.c_int


# every symbol include/nddm.h declares (+ the tuning aid); tests check they all resolve
EXPORTS = [
    "nddm_abi_version", "nddm_last_error", "nddm_device_count", "nddm_set_device", "nddm_summary_k",
    "nddm_model_nparams", "nddm_basic_ddm_dc_simulate", "nddm_single_trial_simulate",
    "nddm_single_trial_alt_simulate", "nddm_alpha_not_scaled_simulate", "nddm_explicit_boundary_simulate",
    "nddm_simulate", "nddm_draw_prior", "nddm_debug_normals", "nddm_set_tuning", "nddm_set_debug_trace", "nddm_set_ordering",
    "nddm_release_graph_memory", "nddm_debug_set_slot_limit", "nddm_debug_last_launch",
    "nddm_simulate_indirect", "nddm_draw_prior_indirect", "nddm_source_hash", "nddm_simulate_codes", "nddm_decode_codes",
    "nddm_graph_arena_create", "nddm_graph_arena_bind", "nddm_graph_arena_info", "nddm_graph_arena_release", "nddm_build_info",
    "nddm_simulratcliff",
]


def lib():
    """The loaded library; raises NddmLibraryError (an ImportError) when it is absent, was built from other sources than
    the tree holds and cannot be rebuilt, or does not export the ABI this binding declares.  Staleness is decided by a
    content hash compiled into the library (build.source_hash), never by file times."""
    global _lib
    if _lib is None:
        from .build import build_hip, embedded_hash, is_stale, source_hash
        override = bool(os.environ.get("NDDM_HIP_LIB"))
        if is_stale():                 # no library, or one built from other sources: compile in-tree (hipcc, gfx950)
            try:
                build_hip()
            except Exception as e:     # noqa: BLE001 -- reported below
                why = "is missing" if not os.path.exists(SO_PATH) else \
                    f"was built from other sources (library {embedded_hash()}, tree {source_hash()})"
                raise NddmLibraryError(
                    f"{SO_PATH} {why} and could not be rebuilt ({e!r}): build it with "
                    "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950).  "
                    "There is no CPU fallback.") from e
        if not os.path.exists(SO_PATH):
            raise NddmLibraryError(f"{SO_PATH} does not exist (NDDM_HIP_LIB points to a missing file?).  There is no CPU fallback.")
        # PyTorch (the plumbing for device memory / streams) bundles its own ROCm runtime: it must be the first HIP
        # runtime loaded into the process, otherwise torch and this library end up on different libamdhip64 copies
        import torch  # noqa: F401
        try:
            L = ctypes.CDLL(SO_PATH)
        except OSError as e:   # e.g. libamdhip64 not found
            raise NddmLibraryError(f"cannot load {SO_PATH}: {e}") from e
        missing = [name for name in EXPORTS if not hasattr(L, name)]
        if missing:
            raise NddmLibraryError(f"{SO_PATH} does not export {missing}: it is older than this binding -- rebuild it "
                                   "(`python -c 'import __graft_entry__ as g; g.build()'`)")
        _declare(L)
        if L.nddm_abi_version() != ABI_VERSION:
            raise NddmLibraryError(f"ABI mismatch: library {L.nddm_abi_version()} != binding {ABI_VERSION} "
                                   "(the random stream differs between ABI versions): rebuild the library")
        if not override and L.nddm_source_hash().decode() != source_hash():
            raise NddmLibraryError(f"{SO_PATH} reports source hash {L.nddm_source_hash().decode()} but the tree hashes to "
                                   f"{source_hash()}: rebuild the library")
        _lib = L
    return _lib


def check(rc):
    """Map an nddm_status to the reference's exception convention (ValueError for bad input)."""
    if rc == NDDM_OK:
        return
    msg = lib().nddm_last_error().decode() or f"nddm status {rc}"
    if rc in (NDDM_ERR_NULL, NDDM_ERR_SHAPE, NDDM_ERR_PARAM):
        raise ValueError(msg)
    raise RuntimeError(msg)

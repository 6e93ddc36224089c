"""Build the HIP library in-tree: bayesflow_nddms_amd/libnddm_hip.so (gfx950 only).

hipcc cross-compiles without a GPU, so this runs in the build container; the built .so
travels to the GPU box with the repository snapshot.
"""
import hashlib
import os
import re
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# NDDM_HIP_LIB: developer override (A/B runs of differently built libraries); the product path is the in-tree library
SO_PATH = os.environ.get("NDDM_HIP_LIB") or os.path.join(_HERE, "libnddm_hip.so")
SOURCES = [os.path.join(_HERE, "csrc", "nddm_kernels.hip")]
HEADERS = [os.path.join(_HERE, "csrc", "nddm_rng.h"), os.path.join(_HERE, "csrc", "nddm_sim.h"),
           os.path.join(_HERE, "csrc", "nddm_prepass.h"), os.path.join(_HERE, "csrc", "nddm_ratcliff.h"),
           os.path.join(os.path.dirname(_HERE), "include", "nddm.h")]
# -ffp-contract=off: the exact Gaussian transform spells out every fma; contraction would change roundings
HIPCC_FLAGS = ["-O3", "-ffp-contract=off", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built")
    return exe


def source_hash():
    """sha256 over the CONTENT of every source and header (in a fixed order) and the compiler flags: what the library is
    a function of.  It is compiled into the library (-DNDDM_SOURCE_HASH, exported as nddm_source_hash() and findable in the
    file as `NDDM_SRC_HASH=<hex>`), so staleness does not depend on file times -- a snapshot copied to the GPU box has
    arbitrary ones."""
    h = hashlib.sha256()
    for p in SOURCES + HEADERS:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()


def embedded_hash(path=None):
    """The source hash compiled into a built library, read from the file (no dlopen); None if there is none."""
    path = path or SO_PATH
    try:
        with open(path, "rb") as f:
            m = re.search(rb"NDDM_SRC_HASH=([0-9a-f]{64})", f.read())
    except OSError:
        return None
    return m.group(1).decode() if m else None


def is_stale():
    """True when there is no library or it was built from other sources than the ones in the tree (by content hash)."""
    if os.environ.get("NDDM_HIP_LIB"):
        return False
    if not os.path.exists(SO_PATH):
        return True
    return embedded_hash() != source_hash()


def hipcc_version():
    """HIP version of the compiler (`hipcc --version`), recorded in the library at build time (nddm_build_info) so that nothing has
    to start the compiler at run time to report it."""
    try:
        m = re.search(r"HIP version:\s*(\S+)", subprocess.run([_hipcc(), "--version"], capture_output=True, text=True, timeout=60).stdout)
        return m.group(1) if m else "unknown"
    except Exception:                                                   # noqa: BLE001
        return "unknown"


def build_hip(force=False, verbose=False):
    """Compile csrc/*.hip for gfx950 into libnddm_hip.so; returns the path."""
    if not force and not is_stale():
        return SO_PATH
    import fcntl
    # several ranks may import the package at once: one builds (to a temporary name, renamed into place), the others wait
    with open(SO_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if force or is_stale():
            tmp = f"{SO_PATH}.{os.getpid()}.tmp"
            cmd = [_hipcc()] + HIPCC_FLAGS + [f'-DNDDM_SOURCE_HASH="{source_hash()}"', f'-DNDDM_HIPCC_VERSION="{hipcc_version()}"', "-o", tmp] + SOURCES
            if verbose:
                print(" ".join(cmd))
            try:
                subprocess.check_call(cmd)
                os.replace(tmp, SO_PATH)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
    return SO_PATH


# ---- the amortizer's kernels (csrc/train_kernels.hip: the flow; csrc/train_deepset.hip: the summary network's per-trial MLPs; csrc/train_update.hip: Adam
# -> libnddm_train.so): not part of the simulator's C ABI; the PyTorch path is the fallback wherever this library is absent or the
# shape is not covered
TRAIN_SO_PATH = os.path.join(_HERE, "libnddm_train.so")
TRAIN_SOURCES = [os.path.join(_HERE, "csrc", f) for f in ("train_kernels.hip", "train_deepset.hip", "train_update.hip")]
TRAIN_SOURCE = TRAIN_SOURCES[0]


def train_source_hash():
    h = hashlib.sha256()
    for path in TRAIN_SOURCES:
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def train_is_stale():
    """True when libnddm_train.so is missing or was built from other source content than the one recorded beside it."""
    stamp = TRAIN_SO_PATH + ".srchash"
    try:
        with open(stamp) as f:
            return not os.path.exists(TRAIN_SO_PATH) or f.read().strip() != train_source_hash()
    except OSError:
        return True


def build_train(force=False, verbose=False):
    """Compile the amortizer's kernels for gfx950; stale = other source content than the one recorded beside the library."""
    if not force and not train_is_stale():
        return TRAIN_SO_PATH
    import fcntl
    # several ranks may import a stale tree at once: one compiles, the others wait for the lock and then find the library fresh
    with open(TRAIN_SO_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if force or train_is_stale():
            tmp = f"{TRAIN_SO_PATH}.{os.getpid()}.tmp"
            cmd = [_hipcc(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-o", tmp] + TRAIN_SOURCES
            if verbose:
                print(" ".join(cmd))
            try:
                subprocess.check_call(cmd)
                os.replace(tmp, TRAIN_SO_PATH)
                with open(TRAIN_SO_PATH + ".srchash", "w") as f:
                    f.write(train_source_hash())
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
    return TRAIN_SO_PATH


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
    print(build_train(force=True, verbose=True))

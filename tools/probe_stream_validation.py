#!/usr/bin/env python3
"""Which HIP stream calls VALIDATE their stream handle, and which dereference it?  (Round 4: a main-thread SIGSEGV inside the
library turned out to be hipStreamIsCapturing() on a handle the runtime does not know.)  Every probe runs in a child process of
its own -- a crash there is a host-side signal in that child, nothing is in flight on the GPU -- and reports the return code or
the signal.  Handles probed: a destroyed stream, a small integer that never was a pointer, zeroed host memory that is not a
stream.  Output: one line per (call, handle).  Usage: python tools/probe_stream_validation.py > profiles/r4_stream_validation.txt"""
import subprocess
import sys

CHILD = r'''
import ctypes, sys
import torch
torch.cuda.init(); torch.zeros(1, device="cuda")
path = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][0]
hip = ctypes.CDLL(path)
vp = ctypes.c_void_p
hip.hipStreamCreate.argtypes = [ctypes.POINTER(vp)]
hip.hipStreamDestroy.argtypes = [vp]
call, kind = sys.argv[1], sys.argv[2]
if kind == "destroyed":
    h = vp(); assert hip.hipStreamCreate(ctypes.byref(h)) == 0
    assert hip.hipStreamDestroy(h) == 0
    bad = h.value
elif kind == "small_int":
    bad = 0x1230
elif kind == "zeroed_memory":
    buf = ctypes.create_string_buffer(4096)
    bad = ctypes.addressof(buf)
elif kind == "live":
    h = vp(); assert hip.hipStreamCreate(ctypes.byref(h)) == 0
    bad = h.value
s = vp(bad)
if call == "hipStreamIsCapturing":
    st = ctypes.c_int(-1); hip.hipStreamIsCapturing.argtypes = [vp, ctypes.POINTER(ctypes.c_int)]
    rc = hip.hipStreamIsCapturing(s, ctypes.byref(st))
elif call == "hipStreamQuery":
    hip.hipStreamQuery.argtypes = [vp]; rc = hip.hipStreamQuery(s)
elif call == "hipStreamGetFlags":
    f = ctypes.c_uint(0); hip.hipStreamGetFlags.argtypes = [vp, ctypes.POINTER(ctypes.c_uint)]; rc = hip.hipStreamGetFlags(s, ctypes.byref(f))
elif call == "hipStreamGetPriority":
    f = ctypes.c_int(0); hip.hipStreamGetPriority.argtypes = [vp, ctypes.POINTER(ctypes.c_int)]; rc = hip.hipStreamGetPriority(s, ctypes.byref(f))
elif call == "hipStreamGetDevice":
    f = ctypes.c_int(0); hip.hipStreamGetDevice.argtypes = [vp, ctypes.POINTER(ctypes.c_int)]; rc = hip.hipStreamGetDevice(s, ctypes.byref(f))
elif call == "hipStreamGetCaptureInfo":
    st = ctypes.c_int(-1); i = ctypes.c_ulonglong(0)
    hip.hipStreamGetCaptureInfo.argtypes = [vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_ulonglong)]
    rc = hip.hipStreamGetCaptureInfo(s, ctypes.byref(st), ctypes.byref(i))
elif call == "hipStreamSynchronize":
    hip.hipStreamSynchronize.argtypes = [vp]; rc = hip.hipStreamSynchronize(s)
elif call == "hipEventRecord":
    e = vp(); hip.hipEventCreate.argtypes = [ctypes.POINTER(vp)]; assert hip.hipEventCreate(ctypes.byref(e)) == 0
    hip.hipEventRecord.argtypes = [vp, vp]; rc = hip.hipEventRecord(e, s)
hip.hipGetErrorName.restype = ctypes.c_char_p
print("rc=%d (%s)" % (rc, hip.hipGetErrorName(rc).decode()))
'''

CALLS = ["hipStreamIsCapturing", "hipStreamQuery", "hipStreamGetFlags", "hipStreamGetPriority", "hipStreamGetDevice",
         "hipStreamGetCaptureInfo", "hipStreamSynchronize", "hipEventRecord"]
KINDS = ["live", "destroyed", "small_int", "zeroed_memory"]


def main():
    import torch
    print(f"# torch {torch.__version__}, HIP {torch.version.hip}; one child process per probe")
    for call in CALLS:
        for kind in KINDS:
            r = subprocess.run([sys.executable, "-c", CHILD, call, kind], capture_output=True, text=True, timeout=300)
            out = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
            verdict = out if r.returncode == 0 else f"CHILD DIED: return code {r.returncode}" + (" (SIGSEGV)" if r.returncode == -11 else "")
            print(f"{call:26s} {kind:14s} {verdict}", flush=True)


if __name__ == "__main__":
    main()

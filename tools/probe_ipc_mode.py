#!/usr/bin/env python3
"""Probe of HSA_ENABLE_IPC_MODE_LEGACY (bench.py::launch_ranks sets it to 0 for the ranks it starts).

RCCL's intra-node transport and PyTorch's CUDA-tensor sharing both hand device buffers to another PROCESS with
hipIpcGetMemHandle / hipIpcOpenMemHandle.  One GPU is enough to exercise that call: a fresh child process allocates a device
tensor, exports it (storage._share_cuda_ -> hipIpcGetMemHandle), and a grandchild on the same card opens the handle and reads the
values back.  Done once with the variable set to 0 (dmabuf handles) and once with it unset / 1 (legacy handles).

Usage: python tools/probe_ipc_mode.py            (prints one line per mode; never touches the GPU in THIS process)
"""
import os
import subprocess
import sys
import tempfile

CHILD = r'''
import os, sys
import torch
import torch.multiprocessing as mp

def reader(q, out):
    t = q.get()
    out.put((float(t.sum().item()), tuple(t.shape)))
    del t

if __name__ == "__main__":
    mp.set_start_method("spawn")
    t = torch.arange(1 << 20, dtype=torch.float32, device="cuda")
    try:
        t.untyped_storage()._share_cuda_()                      # hipIpcGetMemHandle
        print("export ok", flush=True)
    except Exception as e:                                      # noqa: BLE001
        print("export FAILED:", str(e).splitlines()[0][:200], flush=True)
        sys.exit(0)
    q, out = mp.Queue(), mp.Queue()
    p = mp.Process(target=reader, args=(q, out))
    p.start()
    q.put(t)
    try:
        s, shape = out.get(timeout=120)
        print("open in a second process ok:", s == float(t.sum().item()), shape, flush=True)
    except Exception as e:                                      # noqa: BLE001
        print("open FAILED:", repr(e)[:200], flush=True)
    p.join(30)
'''


def main():
    print(f"environment of this shell: HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')!r}")
    r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True)
    print("visible GPUs:", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "?")
    td = tempfile.mkdtemp()
    child = os.path.join(td, "ipc_child.py")               # (a file, not `-c`: the grandchild is spawned and must import `reader`)
    with open(child, "w") as f:
        f.write(CHILD)
    for mode in ("0", "1", None):
        env = {k: v for k, v in os.environ.items() if k != "HSA_ENABLE_IPC_MODE_LEGACY"}
        if mode is not None:
            env["HSA_ENABLE_IPC_MODE_LEGACY"] = mode
        try:
            r = subprocess.run([sys.executable, child], capture_output=True, text=True, env=env, timeout=300)
            lines = [l for l in r.stdout.splitlines() if l.strip()]
            err = [l for l in r.stderr.splitlines() if "hipIpc" in l or "Error" in l][:2]
            print(f"HSA_ENABLE_IPC_MODE_LEGACY={'unset' if mode is None else mode}: rc={r.returncode}; " + "; ".join(lines + err))
        except subprocess.TimeoutExpired:
            print(f"HSA_ENABLE_IPC_MODE_LEGACY={'unset' if mode is None else mode}: timed out")


if __name__ == "__main__":
    main()

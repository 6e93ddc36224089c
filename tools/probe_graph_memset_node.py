#!/usr/bin/env python3
"""Probe behind the workaround at csrc/nddm_kernels.hip (zero_words_kernel: "a kernel, not hipMemsetAsync"): does a MEMSET NODE of a
captured stream take effect on every replay of the graph?

The simulator's launch zeroes 64 counting-sort counters / its queue words before the kernels that use them.  With hipMemsetAsync in a
launch that torch.cuda.graph captures, replays were observed reading stale counters (round 3; results of replay k depended on replay
k - 1).  This probe captures   hipMemsetAsync(buf, 0, n) ; buf += 1   on a stream, replays the graph R times and reports what the
buffer holds after each replay: 1 everywhere every time if the memset node zeroes per replay.  (A node that simply did nothing would
leave k after replay k; what the MI355X runs of rounds 5 and 6 show for nodes of <= 4 KB is neither: from the second replay on the
buffer holds CONSTANT GARBAGE -- min -2147483648, max 434269841 -- i.e. the replayed node writes a wrong value; the probe prints
the first words.)  Sizes: the 256 bytes of the counters, 4 KB, 1 MB.  Run on the MI355X:  python tools/probe_graph_memset_node.py  (output: profiles/r6_probe_graph_memset_node.txt)."""
import ctypes
import sys

import torch


def loaded_hip():
    with open("/proc/self/maps") as f:
        paths = sorted({l.split()[-1] for l in f if "libamdhip64" in l})
    return ctypes.CDLL(paths[0]), paths[0]


def main():
    dev = torch.device("cuda")
    torch.zeros(1, device=dev)
    hip, path = loaded_hip()
    v = ctypes.c_int(0)
    hip.hipRuntimeGetVersion(ctypes.byref(v))
    print(f"# torch {torch.__version__}, HIP runtime {v.value // 10000000}.{v.value // 100000 % 100}.{v.value % 100000} ({path})")
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    hip.hipMemsetAsync.restype = ctypes.c_int
    stale = 0
    for nbytes in (256, 4096, 1 << 20):
        for mode in ("global", "thread_local", "relaxed"):
            buf = torch.zeros(nbytes // 4, dtype=torch.int32, device=dev)
            side = torch.cuda.Stream()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                buf.add_(1)                                            # (warm-up outside the capture)
                torch.cuda.synchronize()
                buf.zero_()
                with torch.cuda.graph(g, stream=side, capture_error_mode=mode):
                    rc = hip.hipMemsetAsync(buf.data_ptr(), 0, nbytes, torch.cuda.current_stream().cuda_stream)
                    buf.add_(1)
            after, words = [], None
            for _ in range(4):
                g.replay()
                torch.cuda.synchronize()
                after.append((int(buf.min()), int(buf.max())))
                if after[-1] != (1, 1) and words is None:
                    words = buf[:6].tolist()
            ok = all(a == (1, 1) for a in after)
            stale += not ok
            print(f"memset node of {nbytes:8d} bytes, capture mode {mode:12s}: hipMemsetAsync rc {rc}; (min, max) of the buffer after replays 1..4: {after}"
                  f"  -> {'the memset zeroed on every replay' if ok else f'WRONG: the replayed memset node left other values than zero behind; first words after the first bad replay: {words}'}")
            del g
    print("=> " + ("memset nodes of a captured stream are NOT reliable on this runtime: the product zeroes with a kernel (csrc/nddm_kernels.hip: zero_words_kernel)"
                   if stale else "memset nodes took effect on every replay in this probe: the observation behind zero_words_kernel did not reproduce here "
                                 "(the kernel form stays: it costs the same one node, and needs no per-runtime check)"))
    return 0


if __name__ == "__main__":
    sys.exit(main())

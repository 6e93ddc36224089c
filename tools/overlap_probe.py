#!/usr/bin/env python3
"""Does a collective's kernel get to run beside the simulator's persistent grid, and what does it cost?  

The simulator launches as many single-wave workgroups as stay resident (8 per SIMD) and keeps them until its queue is empty,
so a kernel that arrives later finds no free wave slot before the launch ends.  bench.py therefore issues the minibatch
all-gather on a HIGH-PRIORITY communication stream, enqueued before the next simulate becomes runnable.  On one GPU the
collective of a one-rank group is a copy, so this probe stands a kernel in for it (tools/overlap_probe.hip: W workgroups that
copy and hold their slots for T ms, like an all-gather paced by xGMI) and measures, per configuration:
    sim alone | sim + stand-in on a high-priority stream (enqueued first) | the same on a normal-priority stream (enqueued after)
    | sim on a grid of 7 waves per SIMD with the stand-in
and reports the simulate time and WHEN the stand-in actually started and ended relative to the simulate launch.

usage: python tools/overlap_probe.py [--wgs 32] [--threads 512] [--hold-ms 30] [--sets 1000000]"""
import argparse
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesflow_nddms_amd import _lib, engine, priors  # noqa: E402


def build():
    src, so = os.path.join(ROOT, "tools", "overlap_probe.hip"), os.path.join(ROOT, "tools", "liboverlap_probe.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", so, src])
    L = ctypes.CDLL(so)
    L.overlap_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_void_p]
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--wgs", type=int, default=32)
    ap.add_argument("--threads", type=int, default=512)
    ap.add_argument("--hold-ms", type=float, default=30.0)
    ap.add_argument("--sets", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    L = build()
    dev = torch.device("cuda", 0)
    B, N = a.sets, 300
    p = torch.as_tensor(priors.basic_prior_matrix(B, 2023)).to(dev)
    out = [torch.empty((B, N, 2), device=dev) for _ in range(2)]
    summ = [torch.empty((B, 10), device=dev) for _ in range(2)]
    per_wg = 4 << 20
    src = torch.empty(a.wgs * per_wg // 4, device=dev)
    dst = torch.empty_like(src)
    simds = 4 * torch.cuda.get_device_properties(dev).multi_processor_count
    hi = torch.cuda.Stream(device=dev, priority=-1)
    lo = torch.cuda.Stream(device=dev)

    def sim(i):
        engine.simulate(engine.BASIC_DDM_DC, p, N, dt=0.001, max_steps=4000, seed=2023, set_offset=i * B, fast=True,
                        out_trials=out[i % 2], out_summary=summ[i % 2])

    def run(tag, comm, comm_first, grid, delay_ms=0.0):
        _lib.check(_lib.lib().nddm_set_tuning(0, 0, 0, 0, grid, 0))
        main_s = torch.cuda.current_stream(dev)
        sim(0); torch.cuda.synchronize()
        stamps = torch.tensor([2**62, 0], dtype=torch.int64, device=dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t_rel = []
        for i in range(a.steps):
            stamps.copy_(torch.tensor([2**62, 0], dtype=torch.int64)); torch.cuda.synchronize()
            gate = torch.cuda.Event(); gate.record(main_s)       # (the stand-in waits for what precedes the simulate, as a gather does)

            def launch_comm():
                if comm is not None:
                    comm.wait_event(gate)
                    L.overlap_probe_launch(src.data_ptr(), dst.data_ptr(), per_wg, a.wgs, a.threads, int(a.hold_ms * 1e5),
                                           stamps.data_ptr(), ctypes.c_void_p(comm.cuda_stream))
            if comm_first:
                launch_comm()
            with engine.debug_trace(device=dev) as tr:
                ev0.record(); sim(1 + i); ev1.record()
                if not comm_first:
                    if delay_ms:                         # the stand-in arrives while the persistent grid is resident
                        import time
                        time.sleep(delay_ms * 1e-3)
                    launch_comm()
            torch.cuda.synchronize()
            rec = tr.read()["records"]
            s0, s1 = int(rec[:, 4].min()), int(rec[:, 6].max())                    # first wave start, last wave end (100 MHz ticks)
            st = stamps.cpu().tolist()
            t_rel.append((ev0.elapsed_time(ev1), (s1 - s0) * 1e-5, (st[0] - s0) * 1e-5 if comm is not None else None,
                          (st[1] - s0) * 1e-5 if comm is not None else None))
        _lib.lib().nddm_set_tuning(0, 0, 0, 0, 0, 0)
        t_rel = t_rel[1:]
        ms = sum(t[0] for t in t_rel) / len(t_rel)
        line = f"{tag:58s} simulate {ms:7.2f} ms (waves' span {sum(t[1] for t in t_rel) / len(t_rel):6.2f})"
        if comm is not None:
            line += (f"   stand-in started {sum(t[2] for t in t_rel) / len(t_rel):+7.2f} ms, ended "
                     f"{sum(t[3] for t in t_rel) / len(t_rel):+7.2f} ms after the first simulator wave")
        print(line, flush=True)
        return ms

    print(f"stand-in collective: {a.wgs} workgroups x {a.threads} threads holding their wave slots for {a.hold_ms} ms "
          f"({a.wgs * a.threads // 64} of the chip's {8 * simds} wave slots); simulate = basic_ddm_dc {B} x {N}, dt=.001", flush=True)
    base = run("simulate alone, full grid (8 waves per SIMD)", None, False, 0)
    run("simulate alone, grid of 7 waves per SIMD", None, False, 7 * simds)
    run("+ stand-in, HIGH-priority stream, enqueued before the simulate", hi, True, 0)
    run("+ stand-in, normal-priority stream, enqueued before the simulate", lo, True, 0)
    run("+ stand-in, HIGH-priority stream, enqueued after the simulate", hi, False, 0)
    run("+ stand-in, normal-priority stream, enqueued after the simulate", lo, False, 0)
    run("+ stand-in (high, before), simulate on 7 waves per SIMD", hi, True, 7 * simds)
    run("+ stand-in (high, after), simulate on 7 waves per SIMD", hi, False, 7 * simds)
    run("+ stand-in (high) launched ~10 ms INTO the simulate", hi, False, 0, delay_ms=10.0)
    run("+ stand-in (high) launched ~10 ms into it, 7 waves per SIMD", hi, False, 7 * simds, delay_ms=10.0)
    print(f"(baseline {base:.2f} ms)")


if __name__ == "__main__":
    main()

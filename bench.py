#!/usr/bin/env python3
"""bench.py -- simulated DDM trials per second (BASELINE.json metric) on N MI355X GPUs of one node.

One "step" = one pass of the hot path over one batch: `--sets` parameter sets x `--trials` trials of the
Euler-Maruyama simulator at dt=0.001 / max_steps=4000 (BASELINE.json configs[1]: basic_ddm_dc, 1M x 300; `--model
single` = configs[3], `--model alpha_ns_bridge` = configs[2] (`alpha_ns`: the same model without the bridge correction, a side
figure that fails the KS bar)), parameters already resident in HBM, output =
float32 pairs [B, 300, 2] + fused per-set summaries [B, 10].

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL).  `python bench.py --gpus N` starts the N ranks
itself (fresh child processes, before anything touches the GPU); under `torch.distributed.run --nproc-per-node N` it
joins the ranks it is given.  The batch shards embarrassingly -- rank r simulates global set indices [r*B, (r+1)*B) of
each step with no data-path collective (weak scaling); `--gather summary|trials` adds the RCCL all-gather that
reassembles a training minibatch (north_star) to the timed region.

Rank 0 prints ONE JSON line (contract in the task description) with extra objects:
  roofline      HBM view of the dominant kernel (algorithmic bytes / measured kernel time vs 8 TB/s) -- tiny by
                construction: 8 B are written per trial for ~246 Gaussian draws
  roofline_valu the binding resource, vector-ALU issue, against two ceilings: (i) the same kernel's step loop with
                every lane useful, MEASURED in this run (a lockstep workload, outside the timed region); (ii) the ISA-level
                issue model of the shipped library (tools/isa_mix.py x tools/ubench_valu, profiles/*_issue_model.json)
  cpu_baseline  the CPU oracle (C restatement, same Philox stream) timed on this box's host cores on a bounded
                sample of the same workload; plus the pure-Python/NumPy port the reference runs without numba

  legs          (N = 1, default command) the OTHER BASELINE configs under the same command and clock, each outside the headline's timed
                region: `single` = configs[3] (trials + fused summaries, and summaries alone), `alpha_ns_bridge` = configs[2], `train` =
                configs[4] (GraphTrainer at one rank and in the RCCL all-gather form, both step sizes); each with rate, kernel time,
                KS + bar, roofline_valu
  dist          (any line with a process group) who ran it: backend, RCCL version, one entry per rank (device, PCI bus id, kernel ms,
                elapsed, host prior seconds), imbalance = slowest / fastest
  side_legs     (process group, default command) short timed passes the weak-scaling headline leaves out: the minibatch all-gather of
                the summaries and of the 2-byte codes (north_star's collective), and the strong-scaling point (1M sets in total)
  toolchain     HIP runtime actually loaded, hipcc, torch, RCCL, rocRAND

`--train` measures BASELINE config 5 instead (online simulation feeding the PyTorch-ROCm amortizer): iterations/s,
microseconds per simulate launch (eager and hipGraph), simulator share of a step, prefetch on/off.
"""
import argparse
import glob
import json
import math
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
CLOCK_GHZ = 2.4                # nominal engine clock the cycle figures are quoted at
ARITHMETIC = ("f32 state and Gaussian transform, integer step index (rt = k*dt + tau exact in k); the reference integrates in f64: on the "
              "same normals the two end a trial on another (step, choice) in 7.6e-5 of 2e6 prior-mixture trials at dt=.001/4000 and 4e-6 at "
              "dt=.01/400, no choice differs (tests/test_oracle_golden.py::test_f32_integrator_against_the_reference_f64_recurrence)")

MODELS = {  # bench name -> (engine model attribute, bridge, host prior matrix, oracle model attribute, index of tau)
    "basic": ("BASIC_DDM_DC", False, "basic_prior_matrix", "M_BASIC", 3),
    "single": ("SINGLE_TRIAL", False, "single_prior_matrix", "M_SINGLE", 3),
    "alpha_ns": ("ALPHA_NOT_SCALED", False, "alpha_ns_prior_matrix", "M_ALPHA_NS", 3),
    "alpha_ns_bridge": ("ALPHA_NOT_SCALED", True, "alpha_ns_prior_matrix", "M_ALPHA_NS", 3),
}
# every trial runs to the step cap (zero drift, boundary 50): no refill, no idle lane -- the step loop alone
LOCKSTEP_ROW = {"basic": [0.0, 50.0, 0.5, 0.3, 1.0], "single": [0.0, 50.0, 0.5, 0.3, 0.01, 1.0, 1.0, 1.0],
                "alpha_ns": [0.0, 50.0, 0.5, 0.3, 0.0, 1.0], "alpha_ns_bridge": [0.0, 50.0, 0.5, 0.3, 0.0, 1.0]}
KERNEL_NAME = {"basic": "nddm::sim_kernel<0 (basic_ddm_dc), %s>", "single": "nddm::sim_kernel<1 (single_trial), %s>",
               "alpha_ns": "nddm::sim_kernel<3 (alpha_not_scaled), %s>", "alpha_ns_bridge": "nddm::sim_kernel<3 (alpha_not_scaled), %s, bridge>"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (defaults: 6.6 s of GPU time at the headline shape -- 32 ms per step -- so that the timed region is long against launch
    #  jitter and visible to a coarse utilisation sampler; the CPU-baseline legs beside it take ~55 s)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--sets", type=int, default=1_000_000, help="parameter sets per GPU per step")
    ap.add_argument("--trials", type=int, default=300)
    ap.add_argument("--dt", type=float, default=0.001)
    ap.add_argument("--max-steps", type=float, default=4000.0)
    ap.add_argument("--gauss", choices=["fast", "exact", "packed"], default="fast",
                    help="fast (product default) | exact (bit-reproducible on a CPU) | packed = fast transform on the opt-in "
                         "NDDM_GAUSS_PACKED layout (8 normals per Philox block from 16 + 16 bit pairs; include/nddm.h)")
    ap.add_argument("--model", choices=list(MODELS), default="basic",
                    help="basic = BASELINE configs[1] (the headline); single = configs[3]; alpha_ns_bridge = configs[2]; alpha_ns = "
                         "the same model with plain Euler-Maruyama (fails the KS bar against the exact sampler: a side figure)")
    ap.add_argument("--gather", choices=["none", "summary", "trials", "codes"], default="none",
                    help="the minibatch all-gather in the timed region: fused summaries (40 B per set), float trials (8 B per trial), "
                         "or `codes` = the trials in the 2-byte wire format (+ the parameter rows), decoded to the same floats on "
                         "every rank after the gather (include/nddm.h: nddm_simulate_codes / nddm_decode_codes)")
    ap.add_argument("--summary-only", action="store_true", help="do not write the 8 B/trial (fused summaries only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the oracle baseline sample")
    ap.add_argument("--cpu-sample", action="store_true",
                    help="time the NumPy port (BASELINE configs[0]) on 12 / 60 parameter sets instead of its stated 500 sets x 300 "
                         "trials at dt=.001/4000 and dt=.01/400 (the default; about 35 s of CPU time)")
    ap.add_argument("--cpu-full", action="store_true", help="(the default since round 3; kept for old command lines)")
    ap.add_argument("--no-ks", action="store_true")
    ap.add_argument("--no-ceiling", action="store_true", help="skip the lockstep run that measures the VALU ceiling")
    ap.add_argument("--no-legs", action="store_true",
                    help="skip the side legs of the default line: at N = 1 the other BASELINE configs (single-trial + fused summaries, "
                         "alpha_not_scaled with the bridge, the config-5 training loop), at N > 1 the all-gather and strong-scaling legs")
    ap.add_argument("--leg-launches", type=int, default=4, help="timed launches per simulator side leg (a leg whose launch is under 15 ms times four times as many: the first launches of a short "
                         "sample run while the clocks settle; every launch's time is in the leg's launch_ms)")
    ap.add_argument("--leg-train-iters", type=int, default=150, help="timed iterations per training side leg")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo + --share-device rehearse the multi-process path on a one-GPU box")
    ap.add_argument("--share-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--dist", action="store_true",
                    help="take the multi-rank code path whatever the world size: process group (RCCL for --backend nccl) with "
                         "device_id, barrier(device_ids), the all-gather of --gather, the device-side all_reduce(MAX) of the "
                         "elapsed time.  With --gpus 1 this runs every distributed call of the 8-GPU job on one GPU")
    ap.add_argument("--no-overlap", action="store_true",
                    help="issue the minibatch all-gather on the simulate stream (serialised) instead of on a communication "
                         "stream with double-buffered outputs")
    ap.add_argument("--train-mode", choices=["graph", "eager", "both"], default="both",
                    help="--train: one hipGraph replay per iteration (GraphTrainer), the eager PyTorch loop, or both side by side")
    ap.add_argument("--train-parallel", choices=["gather", "ddp"], default="gather",
                    help="--train at world > 1: 'gather' = all-gather the simulated shards and train the same minibatch on every "
                         "rank (replicated training, north_star's all-gather); 'ddp' = every rank trains on its own shard and the "
                         "gradients are all-reduced (sharded training)")
    ap.add_argument("--plan", action="store_true",
                    help="touch no GPU: print, for --gpus / --sets / --trials, what every rank allocates and what it moves over xGMI per "
                         "step under each --gather mode (and which of them the plain command and its side legs use), against 288 GB of HBM")
    ap.add_argument("--no-plain-compare", action="store_true",
                    help="--dist at world 1 with a gather: skip the interleaved passes that report the gathered rate over the plain one")
    ap.add_argument("--compare-plain", action="store_true",
                    help="with a process group of ANY size and a gather: after the timed region, interleave passes with and without the "
                         "gather in this same job and report gathered / plain (dist.gathered_over_plain_same_process; default at --dist world 1)")
    ap.add_argument("--train", action="store_true", help="BASELINE config 5: online simulation feeding the amortizer")
    ap.add_argument("--train-iters", type=int, default=150)
    ap.add_argument("--batch", type=int, default=32, help="--train: parameter sets per rank per training step")
    return ap.parse_args()


# --------------------------------------------------------------------------------------------------- launcher
def free_port():
    import socket
    with socket.socket() as s:                       # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def init_own_group(dist, backend, dev, tries=5):
    """init_process_group for a world of ONE that this process rendezvouses with itself: on a port that was free a moment ago
    (free_port() closes its probe socket before the store binds: another socket of this host can take the port in between -- seen once
    as EADDRINUSE on the GPU box), so a failed bind is retried on a fresh port."""
    last = None
    for _ in range(tries):
        os.environ["MASTER_PORT"] = str(free_port())
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            else:
                dist.init_process_group("gloo", rank=0, world_size=1)
            return
        except Exception as e:                                   # noqa: BLE001 -- DistNetworkError / RuntimeError, by version
            last = e
            if "EADDRINUSE" not in str(e) and "address already in use" not in str(e).lower():
                raise
    raise last


def visible_gpus():
    """Number of GPUs a worker would see, asked of a THROWAWAY child process: the launcher itself never loads the HIP runtime
    (its children must be fresh processes, and a process that has touched the GPU must not be re-executed)."""
    import subprocess
    r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True)
    try:
        return int(r.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        return 0


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh worker processes of this same command, one per GPU
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, as torch.distributed.run would set them), wait for
    them and exit with the worst status.  Nothing in THIS process has touched the GPU or imported torch: the workers
    are children, never an exec of a process that initialised HIP.  Rank 0 prints the JSON line on the inherited
    stdout."""
    import subprocess
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        # dmabuf IPC: RCCL's intra-node transport shares device buffers between the ranks' processes with hipIpcGetMemHandle /
        # hipIpcOpenMemHandle, and the hosts this runs on support only the dmabuf form of it -- with the legacy mode the first
        # communicator of a world > 1 fails with `hipIpcGetMemHandle: invalid argument`.  The image exports the variable already
        # (setdefault: an operator's own setting wins); it is set here so that a rank started from a scrubbed environment still
        # has it.  One rank never opens an IPC handle, so world 1 cannot probe it: tools/probe_ipc_mode.py reports what a box's
        # environment holds and, given two GPUs, tries the two-process handle exchange both ways.
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # as torch.distributed.run does: N ranks each starting os.cpu_count() host threads oversubscribe the box (measured:
        # a gloo all-gather of 6 MB takes 230 ms instead of 5 with 2 x 256 threads on a 16-core share)
        env.setdefault("OMP_NUM_THREADS", "1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    worst, alive = 0, list(procs)
    while alive:
        for p in list(alive):
            rc = p.poll()
            if rc is None:
                continue
            alive.remove(p)
            if rc != 0:
                worst = worst or rc
                for q in alive:                      # one rank died: the others would wait for it until a timeout
                    q.terminate()
        time.sleep(0.05)
    sys.exit(worst)


# --------------------------------------------------------------------------------------------------- plan (no GPU)
HBM_BYTES = 288e9              # per MI355X
XGMI_LINK_GBS = 76.5           # one xGMI link, ONE direction (7 links per GPU, ~153 GB/s each both ways: MI355X_MICROARCH.md)
MODEL_NPARAMS = {"basic": 5, "single": 8, "alpha_ns": 6, "alpha_ns_bridge": 6}


def pass_buffers(world, B, N, P, gather, overlap=True, summary_only=False):
    """Every device buffer one simulate_pass() allocates on a rank, by name -> bytes (the same shapes, in the same order): what
    `--plan` prints and what the side legs' collective memory check asks for.  tests/test_gpu_bench_contract.py holds the two
    together (the pass reports what it allocated)."""
    K = 10
    gather_on = world >= 1 and gather != "none"
    codes = gather == "codes"
    nbuf = 2 if (gather_on and overlap) else 1
    b = {"params f32[B,P]": B * P * 4}
    if not (summary_only or codes):
        b[f"trials f32[B,N,2] x{nbuf}"] = nbuf * B * N * 8
    if codes:
        b[f"codes u16[B,N] x{nbuf}"] = nbuf * B * N * 2
    b[f"summary f32[B,10] x{nbuf}"] = nbuf * B * K * 4
    if gather_on and codes:
        b[f"gathered codes u16[W,B,N] x{nbuf}"] = nbuf * world * B * N * 2
        b[f"gathered params f32[W,B,P] x{nbuf}"] = nbuf * world * B * P * 4
        b[f"decoded trials f32[W,B,N,2] x{nbuf}"] = nbuf * world * B * N * 8
    elif gather_on:
        b[f"gathered {gather} x{nbuf}"] = nbuf * world * (B * K * 4 if gather == "summary" else B * N * 8)
    # the library's own scratch per launch (nddm_kernels.hip: hand-out records 48 B per set + 64 counters, integer partial sums
    # 5 u64 per tile; stream-ordered, reused from the pool)
    b["library scratch (records + partials)"] = B * 48 + 512 + B * 5 * 8
    return b


def wire_bytes(world, B, N, P, gather):
    """bytes one rank SENDS per step into the all-gather (its shard; each peer receives it once)"""
    return {"none": 0, "summary": B * 40, "trials": B * N * 8, "codes": B * N * 2 + B * P * 4}[gather]


def plan(a):
    """`bench.py --plan --gpus N [--sets B --trials T]`: no torch import, no GPU.  One JSON line."""
    W, B, N, P = a.gpus, a.sets, a.trials, MODEL_NPARAMS[a.model]
    sim_ms = 32.2 * (B * N / 3e8)                      # the measured one-GPU step of the headline shape, scaled by trials (dt=.001/4000)
    out = {"plan": True, "n_gpus": W, "sets_per_gpu": B, "n_trials": N, "model": a.model, "hbm_bytes_per_gpu": HBM_BYTES,
           "xgmi": {"links_per_gpu": 7, "gb_per_s_per_link_per_direction": XGMI_LINK_GBS,
                    "note": "point-to-point: a ring all-gather is bound by ONE link (W-1 hops of one shard each), a direct all-gather "
                            "pushes the shard down min(W-1, 7) links at once"},
           "simulate_ms_per_step_estimate": sim_ms, "gather": {}}
    for g in ("none", "summary", "trials", "codes"):
        if g == "codes" and a.model not in ("basic", "alpha_ns"):
            continue
        bufs = pass_buffers(W, B, N, P, g)
        total = sum(bufs.values())
        shard = wire_bytes(W, B, N, P, g)
        recv = (W - 1) * shard
        ring_ms = recv / (XGMI_LINK_GBS * 1e9) * 1e3
        direct_ms = shard * max(1, -(-(W - 1) // 7)) / (XGMI_LINK_GBS * 1e9) * 1e3 if W > 1 else 0.0
        decode_ms = (W * B * N * 10) / 3.6e12 * 1e3 if g == "codes" else 0.0        # 2 B read + 8 B written per trial at the measured 3.6 TB/s
        e = {"allocated_bytes_per_rank": total, "buffers": bufs, "fits_hbm": bool(total < 0.9 * HBM_BYTES),
             "hbm_fraction": total / HBM_BYTES, "sent_bytes_per_rank_per_step": shard, "received_bytes_per_rank_per_step": recv,
             # the two ends of what RCCL can do on point-to-point links: ONE ring (every hop of a shard crosses one link: W-1 shard
             # times) and every link at once (the shard goes down min(W-1, 7) links in parallel: one shard time); RCCL builds several
             # rings over different links, so a real collective lies between the two
             "all_gather_ms_one_ring": ring_ms, "all_gather_ms_all_links": direct_ms, "decode_ms": decode_ms,
             "hidden_behind_simulate": {"if_all_links": bool(direct_ms + decode_ms < sim_ms), "if_one_ring": bool(ring_ms + decode_ms < sim_ms)},
             "link_bound_even_on_all_links": bool(direct_ms + decode_ms > 0.9 * sim_ms)}
        out["gather"][g] = e
    out["plain_command"] = {
        "gather": "none",
        "why": "the path shards with no data-path collective (basic_ddm_dc.py:121-122: sets are independent): the driver's plain `bench.py "
               "--gpus N` measures weak scaling of the simulator itself; north_star's minibatch all-gather is measured by the line's "
               "side_legs (gather_summary, gather_codes: the two forms that stay hidden behind the simulate) and by --gather",
        "side_legs": {g: {"allocated_bytes_per_rank": out["gather"][g]["allocated_bytes_per_rank"] if g in out["gather"] else None,
                          "runs_if_free_memory_exceeds": None if g not in out["gather"] else
                          1.25 * side_leg_need(W, B, N, g) + (1 << 30)} for g in ("summary", "codes")},
        "strong": {"sets_per_gpu": -(-B // W), "sets_total": -(-B // W) * W}}
    t = out["gather"]["trials"]
    out["recommendation"] = ("gather=summary, or gather=codes where the trials themselves are needed; gather=trials receives %.1f GB per rank per "
                             "step: %.0f ms with every link busy, %.0f ms on one ring, against a %.0f ms simulate -- at best as long as the "
                             "simulate it would have to hide behind" % (t["received_bytes_per_rank_per_step"] / 1e9, t["all_gather_ms_all_links"],
                                                                        t["all_gather_ms_one_ring"], sim_ms))
    print(json.dumps(out), flush=True)


def side_leg_need(world, B, N, g):
    """device bytes a default-line side leg (gather_summary / gather_codes) allocates on top of the headline's buffers"""
    return sum(v for k, v in pass_buffers(world, B, N, 5, g).items() if not k.startswith(("params", "library")))


# --------------------------------------------------------------------------------------------------- CPU legs
def cpu_baseline(a, params, model_name, n_trials, dt, max_steps, target_s):
    """Time the CPU oracle (kind 'port': our C restatement on the same Philox stream) on a bounded sample of the
    same workload: the first S parameter sets, S sized from a pilot so that the single-thread run takes ~target_s."""
    import oracle
    from oracle import numpy_port
    oracle.build()
    om = getattr(oracle, MODELS[model_name][3])
    bridge = MODELS[model_name][1]
    sim = lambda p, threads: oracle.philox_simulate(om, p, n_trials, dt=dt, max_steps=max_steps, seed=1, bridge=bridge,
                                                    packed=a.gauss == "packed", threads=threads)
    # threads actually used for the multi-core figure: the GPU box grants a CPU share of 16 cores per GPU, whatever
    # os.cpu_count() says (256 there)
    cores = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), 16)
    pilot = 64
    t0 = time.perf_counter()
    sim(params[:pilot], 1)
    t_pilot = time.perf_counter() - t0
    S = int(max(pilot, min(len(params), pilot * target_s / max(t_pilot, 1e-6))))
    t0 = time.perf_counter()
    sim(params[:S], 1)
    t1 = time.perf_counter() - t0
    v1 = S * n_trials / t1
    Sm = int(min(len(params), S * cores))            # all host cores, same sample scaled up
    t0 = time.perf_counter()
    sim(params[:Sm], cores)
    tm = time.perf_counter() - t0
    vm = Sm * n_trials / tm
    out = {"value": v1, "unit": "trials/s", "cores": 1, "kind": "port",
           "sample": f"first {S} of the step's parameter sets x {n_trials} trials, C oracle of the {model_name} model "
                     f"(Philox stream, f32), {t1:.1f} s single thread",
           "all_cores": {"value": vm, "cores": cores, "host_cpu_count": os.cpu_count(),
                         "sample": f"{Sm} sets, OpenMP over sets with {cores} threads, {tm:.1f} s"}}
    # the pure-Python/NumPy algorithm the reference runs when numba is absent (numba is not installed here): BASELINE
    # configs[0].  Default: the stated 500 sets x 300 trials, at the bench's dt and at the reference default dt=.01/400
    # (about 35 s); --cpu-sample: 12 / 60 sets.
    port = {"basic": numpy_port.basic_simulate_trials, "single": numpy_port.single_simulate_trials}.get(model_name)
    if port is not None:
        legs = {}
        for tag, (pdt, pms, n_sets) in {"bench_dt": (dt, max_steps, 12 if a.cpu_sample else 500),
                                        "reference_default_dt.01_max400": (0.01, 400.0, 60 if a.cpu_sample else 500)}.items():
            np.random.seed(2023)
            rows = params[:n_sets, :7].astype(np.float64) if model_name == "single" else params[:n_sets].astype(np.float64)
            t0 = time.perf_counter()
            for row in rows:
                port(row, n_trials, dt=pdt, max_steps=pms)
            tp = time.perf_counter() - t0
            legs[tag] = {"value": len(rows) * n_trials / tp, "cores": 1, "dt": pdt, "max_steps": pms,
                         "sample": f"{len(rows)} sets x {n_trials} trials, {tp:.1f} s"}
        out["numpy_port"] = dict(legs["bench_dt"], what="pure-Python/NumPy statement of the reference simulator "
                                 "(basic_ddm_dc.py:85-125 / single_trial_alpha_not_scaled.py:107-155; numba not installed)",
                                 host_cpu_count=os.cpu_count(), reference_default=legs["reference_default_dt.01_max400"],
                                 full_config_1=not a.cpu_sample)
    return out


def ks_vs_golden(engine, model_name, dt, max_steps, fast, packed=False, state_f64=False):
    """KS distance vs the golden fixtures made from the reference (tests/golden/): the signed step index against
    ks_hist.npz (basic, single; >= 4e5 trials per side per parameter set), the signed RT against the exact sampler's
    quantile tables ratcliff.npz (alpha_not_scaled; 2e5 reference draws per set)."""
    from bayesflow_nddms_amd import diagnostics as dg
    K = int(max_steps)
    per = []
    if model_name in ("basic", "single"):
        path = os.path.join(ROOT, "tests", "golden", "ks_hist.npz")
        if not os.path.exists(path):
            return None
        gold = np.load(path)
        dts = list(gold["dt"])
        if dt not in dts or float(gold["max_steps"][dts.index(dt)]) != float(max_steps):
            return None
        ci = dts.index(dt)
        for si, p in enumerate(gold[f"{model_name}_sets"]):
            row = p if model_name == "basic" else np.append(p, 1.0)
            r = engine.simulate(getattr(engine, MODELS[model_name][0]), np.tile(row, (2048, 1)), 200, dt=dt,
                                max_steps=max_steps, seed=777, set_offset=si * 4096, fast=fast, packed=packed, want_summary=False,
                                state_f64=state_f64)
            h = dg.step_hist_from_trials(r["trials"].cpu().numpy(), float(np.float32(p[3])), dt, K, signed=model_name == "single")
            per.append(round(dg.ks_signed(h, gold[f"{model_name}_hist_s{si}_c{ci}"]), 5))
        return {"max": max(per), "per_set": per, "n_trials_per_side": 409600, "bar": 0.01, "meets_bar": bool(max(per) < 0.01),
                "reference": f"NumPy reference simulator run on {len(per)} fixed parameter sets, tests/golden/ks_hist.npz"}
    path = os.path.join(ROOT, "tests", "golden", "ratcliff.npz")
    if not os.path.exists(path):
        return None
    gold = np.load(path)
    bridge = MODELS[model_name][1]
    for si, p in enumerate(gold["sets"]):
        r = engine.simulate(engine.ALPHA_NOT_SCALED, np.tile(p, (2048, 1)), 200, dt=dt, max_steps=8.0 / dt, seed=777,
                            set_offset=si * 4096, fast=fast, bridge=bridge, packed=packed, want_summary=False)
        per.append(round(dg.ks_quantile_table(r["trials"][..., 0].cpu().numpy().ravel(), gold[f"yq_s{si}"]), 5))
    return {"max": max(per), "per_set": per, "n_trials_per_side": "409600 vs 2e5", "bar": 0.01, "meets_bar": bool(max(per) < 0.01),
            "reference": "simulratcliff (pyhddmjagsutils.py:47-176, the EXACT first-passage sampler alpha_not_scaled.py runs), "
                         "tests/golden/ratcliff.npz" + ("" if bridge else "; plain Euler-Maruyama detects crossings O(sqrt(dt)) late and "
                                                        "FAILS the 0.01 bar (0.070): a side figure only -- `--model alpha_ns_bridge` is "
                                                        "the BASELINE configs[2] measurement")}


def _round_key(path):
    m = re.match(r"r(\d+)_", os.path.basename(path))
    return (int(m.group(1)) if m else -1, os.path.basename(path))


def pmc_traffic(model_name, B, N, dt=0.001, gauss="fast"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/*_pmc.json, made by tools/gpu_profile.sh + tools/summarize_profile.py): WRITE_SIZE is exact for streaming
    stores; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950.  A file counts only if it was collected from THIS
    library: it records the nddm_source_hash() of the library it profiled, and a file with another hash (or none: rounds 1-5) is
    refused -- the line then says `traffic: null` and why (`traffic_refused`) instead of quoting counters of other code."""
    from bayesflow_nddms_amd.build import source_hash
    want, stale = source_hash(), []
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")), key=_round_key, reverse=True):
        try:
            d = json.load(open(path))
            if (d.get("model", "basic") == model_name and d["sets_per_gpu"] == B and d["n_trials"] == N
                    and abs(d.get("dt", 0.001) - dt) < 1e-12 and d.get("gauss", "fast") == gauss and "WRITE_SIZE" in d["pmc_per_launch"]):
                if d.get("source_hash") != want:
                    stale.append(os.path.basename(path))
                    continue
                c = d["pmc_per_launch"]
                return {"bytes": (2.0 * c.get("FETCH_SIZE", 0.0) + c["WRITE_SIZE"]) * 1024.0, "source": os.path.basename(path),
                        "source_hash": want[:16]}
        except (OSError, KeyError, ValueError):
            continue
    return {"bytes": None, "refused": stale[:3]} if stale else None


def issue_model(model_name, gauss, f64=False, key=None):
    """ISA-level ceiling of the shipped library's step loop (tools/isa_mix.py), if the committed file matches the .so."""
    from bayesflow_nddms_amd.build import SO_PATH
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import isa_mix
        digest = isa_mix.code_object(SO_PATH)[1]                # hash of the library's gfx950 code object
    except Exception:                                           # noqa: BLE001 -- reported as "does not match"
        digest = None
    key = key or model_name + {"fast": "", "exact": "_exact", "packed": "_packed"}[gauss] + ("_f64" if f64 else "")
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_issue_model.json")), key=_round_key, reverse=True):
        try:
            d = json.load(open(path))
            k = d["kernels"][key]
            same = digest is not None and d.get("library_sha256_16") == digest
            return {"cycles_per_block": k.get("cycles_per_block", k.get("cycles_per_trip")), "valu_per_block": k["valu"], "steps_per_block": k.get("steps_per_block", 4),
                    "source": os.path.basename(path),
                    "issue_costs_from": d.get("issue_costs_from"), "library_matches": same}
        except (OSError, KeyError, ValueError):
            continue
    return None


# --------------------------------------------------------------------------------------------------- worker
def em_steps_of(summary, tau, dt, max_k, bridge):
    """Executed Euler-Maruyama steps of one launch from the fused summaries (exact integer sums inside)."""
    import torch
    s = summary.double()
    tau = tau.double() - (0.5 * dt if bridge else 0.0)   # RTs carry a uniform sub-step jitter in bridge mode (mean -dt/2)
    n_resp = s[:, 0] + s[:, 1]
    mean_k = torch.where(n_resp > 0, (s[:, 3] - tau) / dt, torch.zeros_like(tau))
    return float((mean_k * n_resp + s[:, 2] * max_k).sum().item())


def measure_ceiling(a, engine, _lib, torch, dev, model_id, bridge, fast, packed, geometry, model_name=None, dt=None, max_steps=None,
                    state_f64=False):
    """The step loop with every lane useful: the SAME kernel (variant and grid of the timed launch: `geometry` =
    engine.last_launch() after a timed step) on a workload whose trials all run to the step cap (no refill, no idle lanes,
    same residency).  Runs outside the timed region.  Returns E-M steps/s and the lane efficiency it was measured at
    (executed-block counter of the kernel)."""
    dt = a.dt if dt is None else dt
    max_steps = a.max_steps if max_steps is None else max_steps
    max_k = engine.max_k_of(max_steps)
    N = a.trials
    B = int(max(2048, min(400_000, (1.6e10 if (state_f64 or not fast) else 4.8e10) / (N * max_k))))
    p = torch.tensor([LOCKSTEP_ROW[model_name or a.model]] * B, dtype=torch.float32, device=dev)
    summ = torch.empty((B, engine.SUMMARY_K), dtype=torch.float32, device=dev)
    L = _lib.lib()
    # never leave the loop early (refill only when all lanes are done); the timed launch's kernel variant and grid
    _lib.check(L.nddm_set_tuning(1, 0, 64, 2 if geometry["vgpr_keys"] else 1, geometry["grid_waves"], 0))
    try:
        run = lambda: engine.simulate(model_id, p, N, dt=dt, max_steps=max_steps, seed=7, set_offset=0, fast=fast,
                                      out_summary=summ, want_trials=False, bridge=bridge, packed=packed, state_f64=state_f64)
        run()
        with engine.debug_trace(device=dev) as tr:
            run(); run()                              # back to back: the second launch's records are the ones that stay
        t = tr.read()
        blocks, clock = t["blocks"], 0.1 * t["cycles"] / max(t["ticks"], 1.0)      # s_memtime cycles per 100 MHz tick
        best = 1e30
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
    finally:
        L.nddm_set_tuning(0, 0, 0, 0, 0, 0)
    steps = float(B) * N * max_k
    lane_eff = steps / (blocks * 64.0 * (8 if (packed or bridge) else 4))       # steps per pass of the step loop
    sps = steps / (best * 1e-3)
    return {"steps_per_s": sps, "lane_efficiency": lane_eff, "steps_per_s_all_lanes_useful": sps / lane_eff,
            "kernel_ms": best, "clock_ghz_in_kernel": clock, "launch": engine.last_launch(),
            "workload": f"{B} sets x {N} trials, every trial runs to the cap of {max_k} steps"}


def valu_roofline(achieved_steps, simds, spb, ceiling, im):
    """The binding resource's roofline object.  `frac` is the fraction of the HARDWARE-derived ceiling: the ISA issue model of the
    shipped library's step loop (every VALU instruction of the loop body at its isolated issue cost, tools/isa_mix.py x
    tools/ubench_valu: what the SIMDs could issue at the nominal clock if every lane of every block were useful and nothing but the
    loop body ran).  `frac_vs_lockstep` is the same rate against this kernel variant's own lockstep run measured in this process --
    a ceiling that shares the kernel's code and therefore measures divergence and refill loss, not distance from the hardware."""
    rv = {"bound": "valu", "achieved": achieved_steps / 1e9, "unit": "G E-M steps/s", "clock_ghz": CLOCK_GHZ}
    if im:
        peak_im = simds * CLOCK_GHZ * 1e9 / im["cycles_per_block"] * 64.0 * im["steps_per_block"]
        rv.update({"peak": peak_im / 1e9, "frac": achieved_steps / peak_im,
                   "ceiling": "ISA issue model of the shipped library's step loop at the nominal clock (hardware-derived; see frac_vs_lockstep "
                              "for the kernel's own lockstep run)", "issue_model": im})
    if ceiling:
        peak = ceiling["steps_per_s_all_lanes_useful"]
        rv.update({"peak_lockstep": peak / 1e9, "frac_vs_lockstep": achieved_steps / peak,
                   "lockstep": "this kernel variant's step loop with every lane useful, measured in this run (a workload whose trials all run "
                               "to the step cap): shares the kernel's code, so it prices divergence and refills, not the hardware",
                   "ceiling_measured_steps_per_s": ceiling["steps_per_s"], "ceiling_lane_efficiency": ceiling["lane_efficiency"],
                   "ceiling_kernel_ms": ceiling["kernel_ms"], "ceiling_workload": ceiling["workload"],
                   "ceiling_clock_ghz_in_kernel": ceiling["clock_ghz_in_kernel"], "ceiling_launch": ceiling["launch"],
                   "issue_cycles_per_block": simds * CLOCK_GHZ * 1e9 * 64.0 * spb / peak, "steps_per_block": spb})
        if "frac" not in rv:
            rv.update({"peak": peak / 1e9, "frac": achieved_steps / peak,
                       "ceiling": "no issue model matches this library: frac falls back to the lockstep ceiling (frac_vs_lockstep)"})
    return rv


_JSON_OUT = None      # the process's real standard output, kept for the ONE JSON line


def emit(obj):
    print(json.dumps(obj), file=_JSON_OUT or sys.stdout, flush=True)


def worker(a):
    # libraries write to file descriptor 1 too (RCCL prints a five-line version banner there when its first communicator
    # comes up): everything but the JSON line goes to standard error, so standard output carries exactly one line
    global _JSON_OUT
    if _JSON_OUT is None:
        sys.stdout.flush()
        _JSON_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {a.gpus} "
                 f"(or let `python bench.py --gpus N` start the ranks itself); refusing to measure a different job")
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a ROCm GPU (no CPU fallback)")
    if a.share_device:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        sys.exit(f"bench.py: rank {rank} has LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) are visible "
                 f"(--share-device puts every rank on cuda:0)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist_on = world > 1 or a.dist                     # --dist: the multi-rank code path at any world size
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")      # the process group's own streams: high priority as well
        if world == 1 and os.environ.get("NDDM_BENCH_OWN_LAUNCHER") == "1":
            init_own_group(dist, a.backend, dev)                 # (main() made this process its own launcher: the port is ours to choose)
        elif a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        # rank <-> device: whatever the process group did at start-up, THIS rank's work must go to its own card
        if torch.cuda.current_device() != local_rank:
            sys.exit(f"bench.py: rank {rank}: current device {torch.cuda.current_device()} != LOCAL_RANK {local_rank} after init_process_group")
    from bayesflow_nddms_amd import _lib, engine
    from bayesflow_nddms_amd import priors as prior_util
    ctx = dict(world=world, rank=rank, local_rank=local_rank, dev=dev, torch=torch, dist=dist, engine=engine, _lib=_lib,
               prior_util=prior_util, dist_on=dist_on)
    if a.train:
        train_bench(a, ctx)
    else:
        simulate_bench(a, ctx)
    if dist_on:
        barrier(a, ctx)
        dist.destroy_process_group()


def barrier(a, ctx):
    if ctx["dist_on"]:
        if a.backend == "nccl":
            ctx["dist"].barrier(device_ids=[ctx["local_rank"]])
        else:
            ctx["dist"].barrier()
    ctx["torch"].cuda.synchronize()


def simulate_pass(a, ctx, p_dev, B, gather, steps, warmup, first_step=0, summary_only=False):
    """`warmup` untimed + `steps` timed passes of the hot path over batches of B parameter sets per rank (rows of p_dev), bracketed by
    barrier + synchronize on both sides; with a process group and gather != 'none' the minibatch all-gather is inside the timed
    region.  Returns {'elapsed': seconds, MAX over ranks; 'elapsed_local'; 'kernel_ms': mean simulate-kernel time on this rank (events
    on the launch stream); 'trials' / 'summary': the buffers the last step wrote; 'geometry'; 'overlap'; 'step': the step function}."""
    world, rank, dev, torch, dist, engine = (ctx[k] for k in ("world", "rank", "dev", "torch", "dist", "engine"))
    N = a.trials
    fast, packed = a.gauss != "exact", a.gauss == "packed"
    model_attr, bridge, _, _, _ = MODELS[a.model]
    model_id = getattr(engine, model_attr)
    # Output buffers.  With a minibatch all-gather (north_star's reassembly step) the collective runs on a COMMUNICATION
    # stream and the outputs are double-buffered: step i+1's simulate is enqueued before step i's gather is waited on, so the
    # two overlap (at weak scale the gather of the trials moves as many bytes over xGMI as the simulate
    # takes time for).  A buffer is handed to the simulator again only after the gather that reads it has completed.
    dist_on = ctx["dist_on"]
    gather_on = dist_on and gather != "none"
    if gather in ("trials", "codes") and summary_only:
        sys.exit("--gather trials / codes needs the trials: drop --summary-only")
    codes = gather == "codes"                         # the simulator writes 2-byte codes; floats appear after the gather, by decoding
    if codes and (a.model not in ("basic", "alpha_ns") or a.max_steps >= 16384):
        sys.exit("--gather codes: basic / alpha_ns (no bridge) with max_steps < 2^14")
    overlap = gather_on and not a.no_overlap
    nbuf = 2 if overlap else 1
    p_dev = p_dev[:B]
    buf_trials = [None if (summary_only or codes) else torch.empty((B, N, 2), dtype=torch.float32, device=dev) for _ in range(nbuf)]
    buf_codes = [torch.empty((B, N), dtype=torch.int16, device=dev) for _ in range(nbuf)] if codes else None
    buf_summary = [torch.empty((B, engine.SUMMARY_K), dtype=torch.float32, device=dev) for _ in range(nbuf)]
    gathered = g_codes = g_params = None
    if gather_on and codes:
        g_codes = [torch.empty((world, B, N), dtype=torch.int16, device=dev) for _ in range(nbuf)]
        g_params = [torch.empty((world,) + tuple(p_dev.shape), dtype=torch.float32, device=dev) for _ in range(nbuf)]
        gathered = [torch.empty((world, B, N, 2), dtype=torch.float32, device=dev) for _ in range(nbuf)]
    elif gather_on:
        shape = tuple((buf_summary if gather == "summary" else buf_trials)[0].shape)
        gathered = [torch.empty((world,) + shape, dtype=torch.float32, device=dev) for _ in range(nbuf)]
    # the communication stream has the higher priority: when a gather and the next simulate become runnable together the
    # collective's few workgroups are placed first, and the persistent simulator grid (which otherwise holds every wave slot
    # of the chip until its queue is empty) fills what is left
    comm = torch.cuda.Stream(device=dev, priority=-1) if overlap else None
    pending = [None] * nbuf                              # the gather in flight on buffer b
    allocated = sum(t.numel() * t.element_size() for grp in (buf_trials, buf_codes, buf_summary, gathered, g_codes, g_params)
                    for t in (grp or []) if t is not None)

    def all_gather(dst, src, async_op):
        if a.backend == "nccl":
            return dist.all_gather_into_tensor(dst, src, async_op=async_op)     # one RCCL all-gather per batch
        return dist.all_gather(list(dst.unbind(0)), src, async_op=async_op)     # gloo rehearsal: list form

    def step(i, ev=None):
        # every step is a fresh batch: global set index = (i*world + rank)*B + row, one seed
        b = i % nbuf
        if pending[b] is not None:
            pending[b].wait()                           # (RCCL: the simulate stream waits for an event, the host does not)
            pending[b] = None
        if ev is not None:
            ev[0].record()                              # torch's current stream == the stream the kernel is launched on
        engine.simulate(model_id, p_dev, N, dt=a.dt, max_steps=a.max_steps, seed=2023,
                        set_offset=(i * world + rank) * B, fast=fast, out_trials=buf_trials[b], out_summary=buf_summary[b],
                        want_trials=not (summary_only or codes), bridge=bridge, packed=packed,
                        out_codes=buf_codes[b] if codes else None)
        if ev is not None:
            ev[1].record()

        def exchange(blocking):
            """the collective(s) of this step on the current stream; returns the last one's handle (non-blocking form)"""
            if not codes:
                return all_gather(gathered[b], buf_summary[b] if gather == "summary" else buf_trials[b], not blocking)
            all_gather(g_params[b], p_dev, False)                # tau travels with the codes (20 B per set)
            all_gather(g_codes[b].view(torch.uint8), buf_codes[b].view(torch.uint8), False)     # (as bytes: RCCL has no 16-bit integer type;
                                                                                                  #  always the blocking form: the decode needs them)
            for r in range(world):                               # 2 B read + 8 B written per trial, beside the next simulate
                engine.decode_codes(model_id, g_codes[b][r], g_params[b][r], a.dt, out_trials=gathered[b][r])
            return None

        if gather_on:
            if overlap:
                done = torch.cuda.Event()
                done.record()
                with torch.cuda.stream(comm):
                    comm.wait_event(done)
                    if a.backend == "nccl" or codes:
                        # the blocking form runs the collective ON the current stream -- the high-priority communication
                        # stream, which has a hardware queue of its own; the process group's internal stream (async_op=True)
                        # was observed sharing the simulate stream's hardware queue, i.e. serialised with the next simulate
                        # (profiles/r3_dist_overlap.md).  The simulate stream later waits for the event recorded behind it.
                        exchange(True)
                        pending[b] = torch.cuda.Event()
                        pending[b].record(comm)
                    else:
                        pending[b] = exchange(False)
            else:
                exchange(True)

    def drain():
        for b in range(nbuf):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    for i in range(warmup):
        step(first_step + i)
    drain()
    barrier(a, ctx)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for i in range(steps):
        step(first_step + warmup + i, ev[i])
    drain()
    barrier(a, ctx)
    elapsed_local = elapsed = time.perf_counter() - t0
    geometry = engine.last_launch()          # the timed launches' kernel variant, grid, ring, tiles (outside the timed region)
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    last = (first_step + warmup + steps - 1) % nbuf       # the buffers the last step wrote
    return {"elapsed": elapsed, "elapsed_local": elapsed_local, "kernel_ms": float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev])),
            "trials": buf_trials[last], "summary": buf_summary[last], "geometry": geometry, "overlap": overlap, "codes": codes,
            "step": step, "last_step": first_step + warmup + steps - 1, "allocated_bytes": allocated}


def rank_identity(a, ctx, t_prior, run):
    """Who ran this line: one entry per rank -- the device it used (index, name, PCI bus id), its simulate-kernel time and its own
    elapsed time of the timed region -- gathered with all_gather_object AFTER the timed region; plus the backend the process group
    reports and the RCCL version.  `imbalance` = slowest / fastest rank.  (With --share-device every rank truthfully reports the same card.)"""
    world, rank, dev, torch, dist = (ctx[k] for k in ("world", "rank", "dev", "torch", "dist"))
    props = torch.cuda.get_device_properties(dev)
    bus = None
    if all(hasattr(props, k) for k in ("pci_bus_id", "pci_device_id", "pci_domain_id")):
        bus = f"{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0"
    me = {"rank": rank, "local_rank": ctx["local_rank"], "device": f"cuda:{torch.cuda.current_device()}", "device_name": props.name,
          "pci_bus_id": bus, "uuid": str(getattr(props, "uuid", "")) or None, "pid": os.getpid(),
          "kernel_ms": run["kernel_ms"], "elapsed_s": run["elapsed_local"], "host_prior_s": t_prior}
    ranks = [None] * world
    dist.all_gather_object(ranks, me)
    ranks.sort(key=lambda r: r["rank"])
    try:
        rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:                                               # noqa: BLE001 -- a build without it
        rccl = None
    km, el = [r["kernel_ms"] for r in ranks], [r["elapsed_s"] for r in ranks]
    distinct = len({(r["pci_bus_id"], r["device"]) for r in ranks})
    if distinct < world and not a.share_device:
        print(f"bench.py: WARNING: {world} ranks ran on {distinct} distinct device(s)", file=sys.stderr, flush=True)
    return {"backend": str(dist.get_backend()), "world": dist.get_world_size(), "rccl_version": rccl,
            "distinct_devices": distinct, "every_rank_on_a_device_of_its_own": bool(distinct == world), "ranks": ranks,
            "imbalance": {"kernel_ms_max_over_min": max(km) / min(km), "elapsed_max_over_min": max(el) / min(el)}}


def toolchain(torch):
    """What runs and what built this line: the HIP runtime the process actually LOADED (PyTorch's wheel brings its own copy, which is
    the one the in-tree libraries then bind to), the compiler that built them, torch, RCCL, and the rocRAND release whose device API
    the vendor yardstick (tools/ubench_rocrand.hip) was built against."""
    import ctypes
    import re
    out = {"torch": torch.__version__, "torch_hip": torch.version.hip, "hip_runtime": None, "hip_runtime_library": None, "hipcc": None,
           "rccl": None, "rocrand": None}
    try:
        with open("/proc/self/maps") as f:
            paths = sorted({l.split()[-1] for l in f if "libamdhip64" in l})
        if paths:
            v = ctypes.c_int(0)
            ctypes.CDLL(paths[0]).hipRuntimeGetVersion(ctypes.byref(v))      # (already loaded: this returns the same handle)
            out["hip_runtime"] = f"{v.value // 10000000}.{v.value // 100000 % 100}.{v.value % 100000}"
            out["hip_runtime_library"] = paths[0]
    except Exception:                                               # noqa: BLE001
        pass
    try:                                         # the compiler that built the library, from the library's own build record -- nothing is
        from bayesflow_nddms_amd import _lib     # started from this (GPU-initialised, possibly profiled) process to find out
        out["hipcc"] = _lib.lib().nddm_build_info().decode().split("hipcc=", 1)[1] or None
    except Exception:                                               # noqa: BLE001
        pass
    try:
        out["rccl"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:                                               # noqa: BLE001
        pass
    try:
        m = re.search(r"#define\s+ROCRAND_VERSION\s+(\d+)", open("/opt/rocm/include/rocrand/rocrand_version.h").read())
        if m:
            v = int(m.group(1))
            out["rocrand"] = f"{v // 100000}.{v // 100 % 1000}.{v % 100}"
    except Exception:                                               # noqa: BLE001
        pass
    return out


def simulator_leg(a, ctx, name, out_trials, out_summary, with_summary_only=False, dt=None, max_steps=None, gauss="fast", state_f64=False,
                  ks=True, ceiling=True):
    """One of the other BASELINE simulator configs -- or another SHAPE of the headline's model -- as a SIDE LEG of the default line,
    outside the headline's timed region: the same 1M x 300 workload shape on model `name` (parameters from ITS reference prior,
    resident in HBM) at step size `dt` / cap `max_steps` (default: the bench's) with Gaussian transform `gauss` and, optionally, the
    float64 state arithmetic; `--leg-launches` launches timed one by one with events on the launch stream; the same objects the
    model's own `--model` line carries -- rate, kernel time, HBM and VALU rooflines (frac = issue-model fraction, the lockstep
    ceiling of THIS kernel variant measured here beside it), KS against the reference fixtures with its bar."""
    dev, torch, engine, _lib, prior_util = (ctx[k] for k in ("dev", "torch", "engine", "_lib", "prior_util"))
    B, N, L = a.sets, a.trials, max(1, a.leg_launches)
    dt = a.dt if dt is None else dt
    max_steps = a.max_steps if max_steps is None else max_steps
    fast = gauss != "exact"
    model_attr, bridge, prior_fn, _, tau_i = MODELS[name]
    model_id = getattr(engine, model_attr)
    p_host = getattr(prior_util, prior_fn)(B, 2023)
    p_dev = torch.as_tensor(p_host).to(dev)
    max_k = engine.max_k_of(max_steps)

    def timed(want_trials):
        run = lambda i: engine.simulate(model_id, p_dev, N, dt=dt, max_steps=max_steps, seed=2023, set_offset=i * B, fast=fast,
                                        out_trials=out_trials if want_trials else None, out_summary=out_summary,
                                        want_trials=want_trials, bridge=bridge, state_f64=state_f64)
        run(0)
        torch.cuda.synchronize()
        times, done = [], 0
        for n_more in (L, 3 * L):                        # a launch of a few ms is sampled 4 L times: a short sample is at the mercy of one hiccup
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_more)]
            for i, (e0, e1) in enumerate(ev):
                e0.record(); run(1 + done + i); e1.record()
            torch.cuda.synchronize()
            times += [e0.elapsed_time(e1) for e0, e1 in ev]
            done += n_more
            if np.mean(times) >= 15.0:
                break
        launch_ms[:] = [round(t, 3) for t in times]
        return float(np.mean(times))

    launch_ms = []
    ms = timed(True)
    launch_ms_trials = list(launch_ms)
    geometry = engine.last_launch()
    em_steps = em_steps_of(out_summary, p_dev[:, tau_i], dt, max_k, bridge)          # of the last launch
    alg_bytes = B * N * 8 + B * (p_host.shape[1] * 4 + engine.SUMMARY_K * 4)
    achieved = alg_bytes / (ms * 1e-3) / 1e9
    nm = "single_trial_alpha_not_scaled" if name == "single" else ("basic_ddm_dc" if name == "basic" else name)
    variant = gauss + (", state_f64" if state_f64 else "")
    leg = {"metric": f"simulated DDM trials/sec at n_trials={N} dt={dt:g} ({nm}, max_steps={max_steps:g})",
           "value": B * N / (ms * 1e-3), "unit": "trials/s", "kernel_ms": ms, "launches": len(launch_ms_trials), "launch_ms": launch_ms_trials,
           "kernel": KERNEL_NAME[name] % variant,
           "workload": f"{nm} HIP simulator, {B} parameter sets x {N} trials per launch, dt={dt}, max_steps={max_steps:g}, params ~ its "
                       f"reference prior (default_rng 2023); trials f32[B,N,2] + fused summaries f32[B,10]",
           "gauss": gauss, "state_f64": bool(state_f64), "dt": dt, "max_steps": max_steps,
           "em_steps_per_trial": em_steps / (B * N), "p_missing": float((out_summary[:, 2].sum() / (B * N)).item()), "launch": geometry,
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_launch": alg_bytes, "traffic": None}}
    tr = None if state_f64 else pmc_traffic(name, B, N, dt, gauss)
    if tr and tr.get("bytes") is not None:
        leg["roofline"].update(traffic=tr["bytes"], traffic_source=tr["source"], traffic_source_hash=tr["source_hash"])
    elif tr:
        leg["roofline"]["traffic_refused"] = tr["refused"]
    achieved_steps = em_steps / (ms * 1e-3)
    simds = 4 * torch.cuda.get_device_properties(dev).multi_processor_count
    c = None
    if ceiling and not a.no_ceiling:
        c = measure_ceiling(a, engine, _lib, torch, dev, model_id, bridge, fast, False, geometry, model_name=name, dt=dt,
                            max_steps=max_steps, state_f64=state_f64)
    leg["roofline_valu"] = valu_roofline(achieved_steps, simds, 8 if bridge else 4, c, issue_model(name, gauss, state_f64))
    # where the distance to the ceiling goes, from the kernel's own per-wave counters (two more launches of the same batch, untimed):
    # lane efficiency = useful lane-steps / executed lane-steps (idle lanes waiting for a refill + the unused steps of a trial's last
    # block), and how many Philox blocks a wave runs between two refills
    with engine.debug_trace(device=dev) as trc:
        for i in (1, 2):
            engine.simulate(model_id, p_dev, N, dt=dt, max_steps=max_steps, seed=2023, set_offset=L * B, fast=fast, out_trials=out_trials,
                            out_summary=out_summary, bridge=bridge, state_f64=state_f64)
    d = trc.read()
    if d["blocks"] > 0:
        lane_steps = d["blocks"] * 64.0 * (8 if bridge else 4)
        leg["roofline_valu"].update({"lane_efficiency": em_steps_of(out_summary, p_dev[:, tau_i], dt, max_k, bridge) / lane_steps,
                                     "philox_blocks_per_refill": d["blocks"] / max(d["refills"], 1.0),
                                     "clock_ghz_in_kernel": 0.1 * d["cycles"] / max(d["ticks"], 1.0)})
    if with_summary_only:                                # configs[3]'s second form: the fused reduction alone, no 8 B per trial
        ms_s = timed(False)
        leg["summary_only"] = {"value": B * N / (ms_s * 1e-3), "unit": "trials/s", "kernel_ms": ms_s,
                               "outputs": "fused summaries f32[B,10] only", "launches": L}
    if ks and not a.no_ks:
        leg["ks_vs_ref"] = ks_vs_golden(engine, name, dt, max_steps, fast, False, state_f64=state_f64)
    return leg


def ratcliff_leg(a, ctx, out_trials, out_summary):
    """BASELINE configs[2] with the reference's OWN generator: pyhddmjagsutils.simulratcliff (:47-176, the exact first-passage sampler
    alpha_not_scaled.py:95-108 calls) on the device -- nddm_simulratcliff, no step size -- at the same 1M x 300 shape, parameters from
    the generator's ranges (alpha_not_scaled.py:66-72): rate, kernel time by events, HBM roofline (8 B written per trial), KS against
    the reference's own draws with its bar, and the exact transform (bit-equal to oracle section D) beside the fast one."""
    dev, torch, engine, prior_util = (ctx[k] for k in ("dev", "torch", "engine", "prior_util"))
    B, N, L = a.sets, a.trials, max(1, a.leg_launches)
    p_host = prior_util.alpha_ns_prior_matrix(B, 2023)
    p_dev = torch.as_tensor(p_host).to(dev)

    def timed(fast):
        run = lambda i: engine.simulratcliff(p_dev, N, seed=2023, set_offset=i * B, fast=fast, out_trials=out_trials, out_summary=out_summary)
        run(0)
        torch.cuda.synchronize()
        times, done = [], 0
        for n_more in (L, 3 * L):                        # (a few ms per launch: 4 L launches, as simulator_leg samples its short kernels)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_more)]
            for i, (e0, e1) in enumerate(ev):
                e0.record(); run(1 + done + i); e1.record()
            torch.cuda.synchronize()
            times += [e0.elapsed_time(e1) for e0, e1 in ev]
            done += n_more
            if np.mean(times) >= 15.0:
                break
        launch_ms[:] = [round(t, 3) for t in times]
        return float(np.mean(times))

    launch_ms = []
    ms = timed(True)
    launch_ms_fast = list(launch_ms)
    mean_rt = float(out_summary[:, 3].double().mean().item())
    alg_bytes = B * N * 8 + B * (6 * 4 + engine.SUMMARY_K * 4)
    achieved = alg_bytes / (ms * 1e-3) / 1e9
    leg = {"metric": f"simulated DDM trials/sec at n_trials={N}, exact first-passage sampler (alpha_not_scaled's own generator, simulratcliff; no dt)",
           "value": B * N / (ms * 1e-3), "unit": "trials/s", "kernel_ms": ms, "launches": len(launch_ms_fast), "launch_ms": launch_ms_fast,
           "kernel": "nddm::ratcliff_kernel<fast>",
           "workload": f"simulratcliff on the device, {B} parameter sets x {N} trials per launch, params ~ alpha_not_scaled.py:66-72 (default_rng 2023); "
                       f"(y, acc) f32[B,N,2] + fused summaries f32[B,10]",
           "mean_rt_s": mean_rt, "p_missing": 0.0,
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_launch": alg_bytes, "traffic": None,
                        "note": "VALU-issue bound (a rejection sampler: ~2.6 attempts, ~1.8 spheres and ~3.2 Philox blocks per trial); `valu` below "
                                "states the measured instruction stream of this build (rocprofv3 PMC) instead of an ISA issue model: the loop's "
                                "trip counts are data-dependent"}}
    # HBM bytes from the committed rocprofv3 PMC passes of this kernel at this shape (tools/gpu_profile_ratcliff.sh), quoted only if they
    # were collected from the library that is running
    from bayesflow_nddms_amd.build import source_hash
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_ratcliff_pmc.json")), key=_round_key, reverse=True):
        try:
            d = json.load(open(path))
            c = next(v["pmc_per_launch"] for k, v in d["kernels"].items() if "ratcliff_kernel<true>" in k)
            if d["sets"] == B and d["n_trials"] == N and "WRITE_SIZE" in c:
                if d.get("source_hash") == source_hash():
                    leg["roofline"].update(traffic=(2.0 * c.get("FETCH_SIZE", 0.0) + c["WRITE_SIZE"]) * 1024.0, traffic_source=os.path.basename(path),
                                           traffic_source_hash=d["source_hash"][:16])
                    if all(k in c for k in ("SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE")):
                        # the profiled launches' own instruction stream: wave-instructions per trial, the share of their 64 lanes that
                        # was switched on (the flattened loop's lane efficiency), and how often a SIMD issued one (XCD-summed GUI cycles / 8
                        # = the launch's cycles; 1024 SIMDs)
                        v = {"wave_insts_per_trial": c["SQ_INSTS_VALU"] / (B * N),
                             "exec_mask_utilisation": c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]),
                             "simd_cycles_per_valu_inst": 1024.0 * (c["GRBM_GUI_ACTIVE"] / 8.0) / c["SQ_INSTS_VALU"],
                             "source": os.path.basename(path)}
                        leg["roofline"]["valu"] = v
                        # ... against the loop's ISA mix priced with the measured per-instruction issue costs (tools/ratcliff_isa_mix.py, in
                        # the tracked issue-model file while it is of the running code object): how busy the VALU pipe is, and -- with the
                        # exec mask -- what share of the pipe's lane-cycles is the sampler's arithmetic
                        im = issue_model("ratcliff", "fast", key="ratcliff_fast")
                        if im and im["library_matches"]:
                            mean = im["cycles_per_block"] / im["valu_per_block"]
                            busy = mean / v["simd_cycles_per_valu_inst"]
                            leg["roofline_valu"] = {"bound": "valu_issue", "mean_issue_cycles_per_valu_inst": mean, "valu_pipe_busy": busy,
                                                    "exec_mask_utilisation": v["exec_mask_utilisation"],
                                                    "frac": min(1.0, busy) * v["exec_mask_utilisation"],
                                                    "what": "frac = (VALU pipe busy, capped at 1) x (exec-mask utilisation): the share of the vector "
                                                            "pipe's lane-cycles spent on switched-on lanes -- lanes that hold a trial, plus (~0.015) "
                                                            "idle lanes that run the fast mode's attempt along because a region around it would "
                                                            "cost more; the instruction stream itself (trip counts of a rejection sampler) is "
                                                            "data-dependent, so there is no lockstep run to compare with",
                                                    "issue_model": {k: im[k] for k in ("source", "issue_costs_from", "library_matches")}}
                else:
                    leg["roofline"]["traffic_refused"] = [os.path.basename(path)]
                break
        except (OSError, KeyError, ValueError, StopIteration):
            continue
    ms_x = timed(False)
    leg["exact_transform"] = {"value": B * N / (ms_x * 1e-3), "unit": "trials/s", "kernel_ms": ms_x,
                              "what": "NDDM_GAUSS_EXACT: every rounding spelled out, bit-equal to oracle/ddm_oracle.c section D "
                                      "(tests/test_gpu_parity.py::test_simulratcliff_bit_parity)"}
    if not a.no_ks:
        from bayesflow_nddms_amd import diagnostics as dg
        path = os.path.join(ROOT, "tests", "golden", "ratcliff.npz")
        if os.path.exists(path):
            gold, per = np.load(path), []
            for si, p in enumerate(gold["sets"]):
                r = engine.simulratcliff(np.tile(p, (2048, 1)), 200, seed=777, set_offset=si * 4096, fast=True, want_summary=False)
                per.append(round(dg.ks_quantile_table(r["trials"][..., 0].cpu().numpy().ravel(), gold[f"yq_s{si}"]), 5))
            leg["ks_vs_ref"] = {"max": max(per), "per_set": per, "n_trials_per_side": "409600 vs 2e5", "bar": 0.01, "meets_bar": bool(max(per) < 0.01),
                                "reference": "simulratcliff (pyhddmjagsutils.py:47-176) itself, tests/golden/ratcliff.npz: the same algorithm on "
                                             "both sides, no discretisation in between"}
    return leg


def training_leg(a, ctx, gather_rccl):
    """BASELINE configs[4] as a side leg: graph_trainer.GraphTrainer on the reference's loop shape (basic_ddm_dc.py:199-202: batch 32,
    N ~ U{60..300} per batch), 10 + `--leg-train-iters` iterations with the graph captures, then `--leg-train-iters` timed, at the
    reference's default step (dt=.01 / 400) and at the bench's (dt=.001 / 4000).  gather_rccl: the multi-rank form on this one GPU --
    a process group over RCCL at world 1, simulate graph | all_gather_into_tensor on the communication stream | training graph,
    pipelined -- i.e. every call the 8-GPU feed makes."""
    dev, torch, dist = (ctx[k] for k in ("dev", "torch", "dist"))
    from bayesflow_nddms_amd.amortizer import AmortizedPosterior, InvariantNetwork, InvertibleNetwork
    from bayesflow_nddms_amd.graph_trainer import GraphTrainer
    own_group = False
    if gather_rccl and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        init_own_group(dist, "nccl", dev)
        own_group = True
    iters, warm, out = max(1, a.leg_train_iters), 10, {}
    try:
        for tag, dt, ms in (("dt.01_max400", 0.01, 400.0), ("dt.001_max4000", 0.001, 4000.0)):
            torch.manual_seed(0)
            amortizer = AmortizedPosterior(InvertibleNetwork(num_params=5), InvariantNetwork())
            with GraphTrainer(amortizer, batch_size=a.batch, total_steps=warm + 2 * iters, dt=dt, max_steps=ms, seed=2023, device=dev,
                              world=1, rank=0, parallel="gather", backend="nccl", split=bool(gather_rccl)) as gt:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                gt.train_online(warm + iters)                              # captures a graph whenever N falls into a new bucket
                torch.cuda.synchronize()
                t_first = time.perf_counter() - t0
                t0 = time.perf_counter()
                gt.train_online(iters)
                t_host = time.perf_counter() - t0
                torch.cuda.synchronize()
                el = time.perf_counter() - t0
                h = gt.loss_history()
                out[tag] = {"iterations_per_s": iters / el, "ms_per_iteration": el / iters * 1e3, "iterations": iters,
                            "host_enqueue_ms_per_iteration": t_host / iters * 1e3, "gpu_bound": bool(t_host < 0.9 * el),
                            "first_pass_ms_per_iteration_with_captures": t_first / (warm + iters) * 1e3, "graphs_captured": gt.n_graphs,
                            "loss_first10": float(np.mean(h[:10])), "loss_last10": float(np.mean(h[-10:])),
                            "all_losses_finite": bool(np.all(np.isfinite(h))), "pipelined_feed": bool(gt.overlap),
                            "pipelined_feed_independent_queues": list(gt.independent_queues),
                            "collective": "all_gather_into_tensor (RCCL, world 1) between the simulate graph and the training graph"
                                          if gather_rccl else None}
    finally:
        if own_group:
            torch.cuda.synchronize()
            dist.destroy_process_group()
    return out


def simulate_bench(a, ctx):
    world, rank, dev, torch, dist, engine, _lib, prior_util = (ctx[k] for k in ("world", "rank", "dev", "torch", "dist", "engine",
                                                                                   "_lib", "prior_util"))
    B, N = a.sets, a.trials
    fast, packed = a.gauss != "exact", a.gauss == "packed"
    if packed and MODELS[a.model][1]:
        sys.exit("--gauss packed cannot be combined with the bridge correction")
    model_attr, bridge, prior_fn, _, tau_i = MODELS[a.model]
    model_id = getattr(engine, model_attr)
    dist_on = ctx["dist_on"]
    # synthetic inputs: the reference prior (basic_ddm_dc.py:62-80 / single_trial_alpha_not_scaled.py:78-102 /
    # alpha_not_scaled.py:66-72), default_rng(2023 + rank), resident in HBM
    t_prior = time.perf_counter()
    p_host = getattr(prior_util, prior_fn)(B, 2023 + rank)
    p_dev = torch.as_tensor(p_host).to(dev)
    t_prior = time.perf_counter() - t_prior
    # per-rank host work before the timed region: N ranks share the node's host cores (vectorised draws: ~0.2 s per 1M rows on one)
    print(f"bench.py: rank {rank}/{world} on cuda:{torch.cuda.current_device()}: {B} parameter rows drawn on the host and copied in "
          f"{t_prior:.2f} s", file=sys.stderr, flush=True)
    run = simulate_pass(a, ctx, p_dev, B, a.gather, a.steps, a.warmup, summary_only=a.summary_only)
    elapsed, kern_ms, geometry, overlap, codes, step = (run[k] for k in ("elapsed", "kernel_ms", "geometry", "overlap", "codes", "step"))
    out_trials, out_summary = run["trials"], run["summary"]
    # ---- everything below is OUTSIDE the headline's timed region
    ident = rank_identity(a, ctx, t_prior, run) if dist_on else None            # (a collective: every rank takes part)
    pass_allocated = run["allocated_bytes"]
    if dist_on and a.gather != "none" and ((a.dist and world == 1 and not a.no_plain_compare) or a.compare_plain):
        # The gathered pass against the plain one IN THIS PROCESS, interleaved (gathered, plain, gathered, plain, ...): the ratio
        # the correctness suite used to take between two separately launched processes -- where one hiccup of the box read as a
        # 27 % gap in round 5 (gpurun_out/r5/gpu_suite1.log) -- reported here as a number, asserted nowhere.
        k, rates, nxt0 = max(3, min(a.steps, 10)), {"gathered": [], "plain": []}, a.warmup + a.steps
        for rep_i in range(3):
            for tag, g in (("gathered", a.gather), ("plain", "none")):
                r = simulate_pass(a, ctx, p_dev, B, g, k, 1, first_step=nxt0, summary_only=a.summary_only and g == "none")
                nxt0 += k + 1
                rates[tag].append(world * B * N * k / r["elapsed"])
                del r
        med = lambda v: float(np.median(v))
        ident["gathered_over_plain_same_process"] = {"ratio": med(rates["gathered"]) / med(rates["plain"]), "gathered_trials_per_s": rates["gathered"],
                                                     "plain_trials_per_s": rates["plain"], "steps_per_pass": k, "passes": 3,
                                                     "what": "this line's pass (process group, all-gather on the communication stream) and the "
                                                             "plain pass, interleaved in one process; median over median"}
    side = {}
    if dist_on and not a.no_legs and a.model == "basic" and not a.summary_only and a.gather == "none":
        # What the default multi-GPU command leaves out, as short timed passes of the same code (every rank takes part): north_star's
        # minibatch all-gather in its two practical forms, and the strong-scaling point (a FIXED 1M-set batch split over the ranks)
        del run
        k_steps, k_warm, nxt = 4, 1, a.warmup + a.steps

        def every_rank_has(nbytes):
            """A side leg runs only if EVERY rank has the device memory for it (decided collectively: a leg that one rank skipped
            and another entered would wait in its collective forever)."""
            free = torch.cuda.mem_get_info(dev)[0]
            ok = torch.tensor([1 if free > nbytes * 1.25 + (1 << 30) else 0], dtype=torch.int32, device=dev if a.backend == "nccl" else "cpu")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            return bool(ok.item())

        for g in ("summary", "codes"):
            need = side_leg_need(world, B, N, g)
            if not every_rank_has(need):
                side["gather_" + g] = {"skipped": f"needs {need / 1e9:.1f} GB of device memory on every rank"}
                continue
            r = simulate_pass(a, ctx, p_dev, B, g, k_steps, k_warm, first_step=nxt)
            nxt += k_steps + k_warm
            side["gather_" + g] = {"value": world * B * N * k_steps / r["elapsed"], "unit": "trials/s", "ms_per_step": r["elapsed"] / k_steps * 1e3,
                                   "kernel_ms": r["kernel_ms"], "steps": k_steps, "warmup": k_warm, "scaling": "weak",
                                   "bytes_gathered_per_rank_per_step": world * B * (engine.SUMMARY_K * 4 if g == "summary" else (N * 2 + p_host.shape[1] * 4)),
                                   "what": ("all-gather of the fused summaries f32[B,10]" if g == "summary" else
                                            "all-gather of the trials as 2-byte codes + the parameter rows, decoded to f32[world*B,N,2] on every rank")
                                           + (" on a communication stream, double-buffered (inside the timed region)" if r["overlap"] else " (serialised)")}
            del r
        Bs = -(-B // world)                                  # the headline's per-GPU batch (1M sets by default) as the FIXED total
        r = simulate_pass(a, ctx, p_dev, Bs, "none", k_steps, k_warm, first_step=nxt)
        side["strong"] = {"sets_total": Bs * world, "sets_per_gpu": Bs, "value": world * Bs * N * k_steps / r["elapsed"], "unit": "trials/s",
                          "ms_per_step": r["elapsed"] / k_steps * 1e3, "kernel_ms": r["kernel_ms"], "steps": k_steps, "warmup": k_warm,
                          "scaling": "strong", "what": f"a fixed batch of {Bs * world} parameter sets x {N} trials split over the ranks, no gather"}
        del r
    if rank != 0:
        return

    max_k = engine.max_k_of(a.max_steps)
    em_steps = em_steps_of(out_summary, p_dev[:, tau_i], a.dt, max_k, bridge)      # of the last step's launch
    p_missing = float((out_summary[:, 2].sum() / (B * N)).item())
    trials_per_step = world * B * N
    value = trials_per_step * a.steps / elapsed
    P = p_host.shape[1]
    alg_bytes = B * N * (0 if a.summary_only else (2 if codes else 8)) + B * (P * 4 + engine.SUMMARY_K * 4)
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    name = "basic_ddm_dc" if a.model == "basic" else a.model
    res = {
        "metric": f"simulated DDM trials/sec at n_trials={N} dt={a.dt:g} ({name}, max_steps={a.max_steps:g})",
        "value": value, "unit": "trials/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{name} HIP simulator, {B} parameter sets x {N} trials per GPU per step, "
                               f"dt={a.dt}, max_steps={a.max_steps:g}, params ~ reference prior (default_rng 2023)",
                   "sets_per_gpu": B, "n_trials": N, "dt": a.dt, "max_steps": a.max_steps,
                   "gauss": a.gauss, "arithmetic": ARITHMETIC,
                   "outputs": "summaries only" if a.summary_only else ("trials as 2-byte codes u16[B,N] + summaries f32[B,10]; floats by "
                                                                                "decoding after the gather" if codes else
                                                                                "trials f32[B,N,2] + summaries f32[B,10]"),
                   "parallelism": f"dp{world} over parameter sets, gather={a.gather}"
                                  + (", all-gather on a communication stream, double-buffered outputs" if overlap else "")
                                  + (f", distributed code path forced at world {world} ({a.backend})" if a.dist else "")},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": KERNEL_NAME[a.model] % a.gauss,
                     "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": alg_bytes,
                     "note": "path is VALU-bound, not HBM- or MFMA-bound: see roofline_valu"},
        "launch": geometry,
        "em_steps_per_trial": em_steps / (B * N), "em_steps_per_s_per_gpu": em_steps / (kern_ms * 1e-3),
        "p_missing": p_missing,
    }
    tr = None if (a.summary_only or codes) else pmc_traffic(a.model, B, N, a.dt, a.gauss)
    if tr and tr.get("bytes") is not None:
        res["roofline"].update(traffic=tr["bytes"], traffic_source=tr["source"], traffic_source_hash=tr["source_hash"])
    elif tr:
        res["roofline"]["traffic_refused"] = tr["refused"]      # counters of another build of the library: not quoted
    res["host_prior_seconds_rank0"] = t_prior
    res["library"] = {"source_hash": _lib.lib().nddm_source_hash().decode(), "build": _lib.lib().nddm_build_info().decode(),
                      "abi": int(_lib.lib().nddm_abi_version())}      # the library that RAN (profiles/*_pmc.json are keyed by this hash)
    res["pass_allocated_bytes"] = pass_allocated          # what simulate_pass() allocated on this rank (== pass_buffers(): --plan's figure)
    res["toolchain"] = toolchain(torch)
    if ident is not None:
        res["dist"] = ident
    if side:
        res["side_legs"] = side
    if world > 1:
        # the per-kernel analysis belongs to the one-GPU line: these objects are measured at N = 1 only (their absence here is
        # by design, not a failure); `roofline` (the contract's object) and `launch` are on every line
        res["n1_only"] = ["roofline_valu", "occupancy", "ks_vs_ref", "packed_gauss", "cpu_baseline", "gpu_over_cpu_1core", "legs"]
    if world == 1:
        achieved_steps = em_steps / (kern_ms * 1e-3)
        simds = 4 * torch.cuda.get_device_properties(dev).multi_processor_count
        spb = 8 if (packed or bridge) else 4         # Euler-Maruyama steps per pass of the step loop (packed: one Philox block; bridge: two + one of crossing uniforms)
        c = None if a.no_ceiling else measure_ceiling(a, engine, _lib, torch, dev, model_id, bridge, fast, packed, geometry)
        rv = valu_roofline(achieved_steps, simds, spb, c, issue_model(a.model, a.gauss))
        if "issue_model" in rv:                      # (the names rounds 2-5 used, kept for readers of older lines)
            rv.update({"peak_issue_model": rv["peak"], "frac_vs_issue_model": rv["frac"]})
        # an OUTSIDE yardstick beside the two self-measured ceilings: the vendor's device API doing what north_star names
        # ("hiprandStatePhilox per lane": rocrand_state_philox4x32_10 + rocrand_normal4 in a loop), measured on this chip by
        # tools/ubench_rocrand.hip and read from the tracked file, like `traffic`
        try:
            y = json.load(open(os.path.join(ROOT, "profiles", "r4_ubench_rocrand.json")))
            rv["vendor_philox_normals_per_s"] = y["fast_math"]["rocrand_normal4_normals_per_s"]
            rv["vendor_philox"] = {"normals_per_s_default_build": y["rocrand_normal4_normals_per_s"],
                                   "normals_per_s_fast_math": y["fast_math"]["rocrand_normal4_normals_per_s"],
                                   "normals_with_an_em_step_per_s_fast_math": y["fast_math"]["rocrand_normal4_plus_step_normals_per_s"],
                                   "raw_u32_per_s": y["rocrand4_u32_per_s"], "source": y["source"],
                                   # (one Gaussian per plain Euler-Maruyama step: comparable; a bridged step also draws crossing uniforms)
                                   "this_kernel_over_vendor_with_step": None if bridge else
                                   achieved_steps / y["fast_math"]["rocrand_normal4_plus_step_normals_per_s"]}
        except (OSError, KeyError, ValueError):
            pass
        res["roofline_valu"] = rv
        # steps the lanes actually EXECUTED (incl. lanes idling on a finished trial until the next refill): one more
        # launch of the last batch, outside the timed region, with the kernel's debug counters switched on
        with engine.debug_trace(device=dev) as tr:
            for _ in range(3):                        # back to back, like the timed steps; the last launch's records stay
                step(a.warmup + a.steps - 1)
        d = tr.read()
        lane_steps = d["blocks"] * 64.0 * spb
        clock = 0.1 * d["cycles"] / max(d["ticks"], 1.0)
        rv.update({"executed_lane_steps_per_launch": lane_steps, "lane_efficiency": em_steps / lane_steps,
                   "philox_blocks_per_refill": d["blocks"] / max(d["refills"], 1.0), "waves": d["waves"],
                   "clock_ghz_in_kernel": clock})
        if "ceiling_clock_ghz_in_kernel" in rv and clock > 0:
            # the shader clock the two launches actually ran at (s_memtime / s_memrealtime inside the kernel): short
            # launches of the refill-heavy mix run at a lower clock than the long lockstep launch, which a ratio of rates
            # counts against the kernel; this is the lockstep ratio in SIMD cycles
            rv["frac_vs_lockstep_in_cycles"] = rv["frac_vs_lockstep"] * rv["ceiling_clock_ghz_in_kernel"] / clock
        if "issue_model" in rv and clock > 0:
            # ... and the issue-model fraction at the clock the kernel actually ran at instead of the nominal 2.4 GHz
            rv["frac_at_measured_clock"] = rv["frac"] * CLOCK_GHZ / clock
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        # waves actually resident: sum of the waves' lifetimes (100 MHz s_memrealtime) over kernel time x SIMDs
        rec = d["records"]
        span = float(rec[:, 6].max() - rec[:, 4].min())             # first wave start -> last wave end, 100 MHz ticks
        resident = d["ticks"] / max(span, 1.0) / (4.0 * cus)
        res["occupancy"] = {"resident_waves_per_simd": resident, "hardware_max": 8, "grid_waves": d["waves"],
                            "limit": "SGPR file (800 per SIMD, a wave is charged its SGPRs + 22 rounded up to 16): <= 74 SGPRs -> 8 wave64 "
                                     "per SIMD; tools/resource_table.py lists every kernel",
                            "note": "sum of the waves' lifetimes / (first start -> last end) / SIMDs, from per-wave s_memrealtime records (nddm_set_debug_trace)"}
        if not a.no_ks:
            res["ks_vs_ref"] = ks_vs_golden(engine, a.model, a.dt, a.max_steps, fast, packed)
        if not packed and not bridge and fast and not a.no_ks:
            # the opt-in NDDM_GAUSS_PACKED layout on the same batch, outside the timed region: reported beside the
            # headline, never as the headline
            def pk(i):
                engine.simulate(model_id, p_dev, N, dt=a.dt, max_steps=a.max_steps, seed=2023, set_offset=i * B, fast=True,
                                out_trials=out_trials, out_summary=out_summary, want_trials=not a.summary_only, packed=True)
            pk(0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(3):
                pk(1 + i)
            e1.record()
            torch.cuda.synchronize()
            pk_ms = e0.elapsed_time(e1) / 3
            res["packed_gauss"] = {"value": B * N / (pk_ms * 1e-3), "unit": "trials/s", "kernel_ms": pk_ms,
                                   "em_steps_per_s": em_steps_of(out_summary, p_dev[:, tau_i], a.dt, max_k, False) / (pk_ms * 1e-3),
                                   "ks_vs_ref": ks_vs_golden(engine, a.model, a.dt, a.max_steps, True, True),
                                   "what": "same workload with flags |= NDDM_GAUSS_PACKED (opt-in: 8 normals per Philox block from "
                                           "16 + 16 bit Box-Muller pairs; include/nddm.h), 3 launches outside the timed region"}
        if not a.no_legs and a.model == "basic" and a.gauss == "fast" and not a.summary_only:
            # the OTHER BASELINE configs under this same command (and the driver's clock), each outside the headline's timed
            # region: configs[3] single-trial + fused summaries (and summaries alone), configs[2] alpha_not_scaled with the bridge,
            # configs[4] the training loop at one rank and in its RCCL all-gather form
            t_legs = time.perf_counter()

            def guarded(fn, *args, **kw):
                """A side leg that raises must not cost the line its headline: the leg then reads {"error": ...} (and the traceback
                goes to standard error); tests/test_gpu_bench_contract.py asserts that no leg does."""
                try:
                    return fn(*args, **kw)
                except Exception as e:                                      # noqa: BLE001
                    import traceback
                    traceback.print_exc(file=sys.stderr)
                    return {"error": f"{type(e).__name__}: {e}"[:400]}

            legs = {"single": guarded(simulator_leg, a, ctx, "single", out_trials, out_summary, with_summary_only=True),
                    "alpha_ns_bridge": guarded(simulator_leg, a, ctx, "alpha_ns_bridge", out_trials, out_summary),
                    # the REFERENCE'S OWN default shape (basic_ddm_dc.py:87: dt=.01, max_steps=400; every training run of the
                    # reference uses it): the kernel's worst -- a trial lasts 7 blocks, so every per-trial instruction weighs
                    # nine times what it does at dt=.001
                    "basic_dt01": guarded(simulator_leg, a, ctx, "basic", out_trials, out_summary, dt=0.01, max_steps=400.0),
                    # the headline workload with NDDM_GAUSS_EXACT: the ONLY mode that is bit-equal to the oracle (every -m gpu
                    # parity test runs it); the headline's fast transform is pinned to it by per-trial agreement and KS
                    "exact_gauss": guarded(simulator_leg, a, ctx, "basic", out_trials, out_summary, gauss="exact")}
            # configs[2] as the reference itself generates it: simulratcliff, the exact sampler, on the device
            legs["alpha_ns_exact_sampler"] = guarded(ratcliff_leg, a, ctx, out_trials, out_summary)
            # NDDM_STATE_F64: the reference's float64 recurrence on the device (bit-equal to the float64 oracle with the exact
            # transform), as a side number with its cost against the float32 state of the same transform
            f64 = {g: guarded(simulator_leg, a, ctx, "basic", out_trials, out_summary, gauss=g, state_f64=True, ceiling=False, ks=(g == "exact"))
                   for g in ("exact", "fast")}
            for g, base in (("exact", legs["exact_gauss"].get("value")), ("fast", value)):
                if "value" in f64[g] and base:
                    f64[g]["rate_vs_f32_state_same_transform"] = f64[g]["value"] / base
            legs["state_f64"] = dict(f64["exact"], fast_transform=f64["fast"],
                                     what="flags |= NDDM_STATE_F64 (include/nddm.h) on the headline workload: evidence and range test in float64 "
                                          "exactly as basic_ddm_dc.py:91-103; with the exact transform every (step, choice) equals "
                                          "oracle_philox_simulate_f64's (tests/test_gpu_parity.py::test_state_f64_bit_parity)")
            from bayesflow_nddms_amd import _train_lib
            one, gat = guarded(training_leg, a, ctx, False), guarded(training_leg, a, ctx, True)
            legs["train"] = {"metric": "training iterations/sec, online simulation feeding the amortizer (BASELINE configs[4])",
                             "value": one.get("dt.01_max400", {}).get("iterations_per_s"), "unit": "iterations/s",
                             "workload": f"basic_ddm_dc online training feed: {a.batch} sets per step, N ~ U{{60..300}} per batch, device prior -> "
                                         "simulate -> DeepSet + 6-layer coupling flow -> Adam, one hipGraph replay per iteration "
                                         "(graph_trainer.GraphTrainer); value = one rank at the reference's default dt=.01 / 400",
                             "training_kernels": "libnddm_train.so" if _train_lib.lib() is not None else "PyTorch (library not built)",
                             "one_rank": one, "gather_rccl_world1": gat}
            legs["wall_seconds"] = time.perf_counter() - t_legs
            res["legs"] = legs
        if not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(a, p_host, a.model, N, a.dt, a.max_steps, a.cpu_seconds)
            res["gpu_over_cpu_1core"] = value / res["cpu_baseline"]["value"]
    emit(res)


# --------------------------------------------------------------------------------------------------- config 5
def train_bench(a, ctx):
    """BASELINE config 5: basic_ddm_dc online simulation feeding the amortizer (PyTorch-ROCm DeepSet + coupling flow),
    the reference's training loop shape (basic_ddm_dc.py:199-202: batch 32, N ~ U{60..300} shared by the batch, dt=.01
    / max_steps=400), simulation sharded over the ranks.  Per rank and step: draw `--batch` parameter sets on the device
    (counter-based prior), simulate them, then either all-gather trials + parameters and run the same Adam step on the
    gathered minibatch on every rank (--train-parallel gather: replicated training, north_star's all-gather) or train on
    the local shard and all-reduce the gradients (--train-parallel ddp: sharded training).

    Two drivers of the same loop are measured side by side (--train-mode):
      eager  amortizer.Trainer: eager PyTorch, one loss read-back per step (round 2's figure; prefetch off / on)
      graph  graph_trainer.GraphTrainer: one hipGraph replay per iteration (prior -> simulate -> forward -> backward ->
             clip -> Adam captured; one graph per n_trials bucket), losses read back once at the end
    plus the simulate launch in isolation (eager and as a hipGraph replay)."""
    world, rank, dev, torch, dist, engine = (ctx[k] for k in ("world", "rank", "dev", "torch", "dist", "engine"))
    dist_on = ctx["dist_on"]
    from bayesflow_nddms_amd import basic_ddm_dc
    from bayesflow_nddms_amd.amortizer import AmortizedPosterior, InvariantNetwork, InvertibleNetwork, Trainer
    from bayesflow_nddms_amd.distributed import shared_prior_N
    from bayesflow_nddms_amd import _train_lib
    from bayesflow_nddms_amd.graph_trainer import GraphTrainer
    from bayesflow_nddms_amd.priors import DevicePrior
    Bl, warm = a.batch, 10
    if a.train_parallel == "ddp" and a.train_mode != "graph":
        sys.exit("--train-parallel ddp is implemented by the graph trainer: use --train-mode graph")
    results = {}
    for tag, dt, ms in (("dt.01_max400", 0.01, 400.0), ("dt.001_max4000", 0.001, 4000.0)):
        leg = {}
        if a.train_mode in ("eager", "both"):
            prior = DevicePrior("basic", seed=2023)
            counter = {"i": 0}

            def generative_model(batch_size, dt=dt, ms=ms, prior=prior, counter=counter):
                i = counter["i"]; counter["i"] += 1
                n = shared_prior_N(2023, i)                                   # batch-shared N, no communication
                base = i * batch_size * world
                p = prior(batch_size, set_offset=base + rank * batch_size)    # this rank's rows of the global batch
                r = engine.simulate(engine.BASIC_DDM_DC, p, n, dt=dt, max_steps=ms, seed=2023,
                                    set_offset=base + rank * batch_size, fast=True, want_summary=False)
                data, pd = r["trials"], p
                if dist_on:
                    gd = torch.empty((world,) + tuple(data.shape), dtype=torch.float32, device=dev)
                    gp = torch.empty((world,) + tuple(p.shape), dtype=torch.float32, device=dev)
                    if a.backend == "nccl":
                        dist.all_gather_into_tensor(gd, data); dist.all_gather_into_tensor(gp, p)
                    else:
                        dist.all_gather(list(gd.unbind(0)), data); dist.all_gather(list(gp.unbind(0)), p)
                    data, pd = gd.reshape(-1, n, 2), gp.reshape(-1, p.shape[1])
                return {"prior_draws": pd, "sim_data": data, "sim_non_batchable_context": n}

            for prefetch in (False, True):
                torch.manual_seed(0)
                amortizer = AmortizedPosterior(InvertibleNetwork(num_params=5), InvariantNetwork())
                trainer = Trainer(amortizer, generative_model, basic_ddm_dc.configurator, checkpoint_path=None, device=dev)
                counter["i"] = 0
                # the same run as the graph leg below: `warm + train_iters` iterations untimed, `train_iters` timed, ONE cosine
                # schedule of warm + 2 train_iters steps over both calls (the Trainer's own default is a schedule per call), so
                # the two drivers' loss columns are comparable at equal iteration counts
                total = warm + 2 * a.train_iters
                opt = torch.optim.Adam(amortizer.parameters(), lr=trainer.lr)
                sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s, total=total: 0.5 * (1.0 + math.cos(math.pi * min(s, total) / total)))
                trainer.train_online(epochs=1, iterations_per_epoch=warm + a.train_iters, batch_size=Bl, save_checkpoint=False,
                                     prefetch=prefetch, optimizer=opt, scheduler=sched)
                barrier(a, ctx)
                t0 = time.perf_counter()
                trainer.train_online(epochs=1, iterations_per_epoch=a.train_iters, batch_size=Bl, save_checkpoint=False,
                                     prefetch=prefetch, optimizer=opt, scheduler=sched)
                barrier(a, ctx)
                el = time.perf_counter() - t0
                h = trainer.loss_history
                leg["eager_prefetch_on" if prefetch else "eager_prefetch_off"] = {
                    "iterations_per_s": a.train_iters / el, "ms_per_iteration": el / a.train_iters * 1e3,
                    "loss_first10": float(np.mean(h[:10])), "loss_last10": float(np.mean(h[-10:]))}
            # training step alone on a fixed batch (no simulation)
            fixed = basic_ddm_dc.configurator(generative_model(Bl))
            for _ in range(5):
                trainer._step(fixed)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.train_iters):
                trainer._step(fixed)
            torch.cuda.synchronize()
            leg["eager_train_step_alone_ms"] = (time.perf_counter() - t0) / a.train_iters * 1e3
        if a.train_mode in ("graph", "both"):
            torch.manual_seed(0)
            amortizer = AmortizedPosterior(InvertibleNetwork(num_params=5), InvariantNetwork())
            with GraphTrainer(amortizer, batch_size=Bl, total_steps=warm + 2 * a.train_iters, dt=dt, max_steps=ms, seed=2023,
                              device=dev, world=world, rank=rank, parallel=a.train_parallel, backend=a.backend,
                              split=dist_on) as gt:
                # (1) the first pass captures a graph whenever N falls into a new bucket: timed as "with capture"
                barrier(a, ctx)
                t0 = time.perf_counter()
                gt.train_online(warm + a.train_iters)
                barrier(a, ctx)
                t_first = time.perf_counter() - t0
                n_graphs = gt.n_graphs
                # (2) steady state: the same number of iterations again (most buckets exist by now)
                t0 = time.perf_counter()
                gt.train_online(a.train_iters)
                t_host = time.perf_counter() - t0                        # host time to enqueue the iterations
                barrier(a, ctx)
                el = time.perf_counter() - t0
                h = gt.loss_history()
                leg["graph"] = {"iterations_per_s": a.train_iters / el, "ms_per_iteration": el / a.train_iters * 1e3,
                                "host_enqueue_ms_per_iteration": t_host / a.train_iters * 1e3,
                                "gpu_bound": bool(t_host < 0.9 * el),
                                "first_pass_ms_per_iteration_with_captures": t_first / (warm + a.train_iters) * 1e3,
                                "graphs_captured": gt.n_graphs, "graphs_captured_in_first_pass": n_graphs,
                                "buckets": gt.n_buckets, "loss_first10": float(np.mean(h[:10])), "loss_last10": float(np.mean(h[-10:])),
                                "parallel": a.train_parallel if world > 1 else "one rank",
                                "two_graphs_with_collective_between": bool(dist_on),
                                # simulate (+ all-gather) of batch i + 1 on their own streams beside the training graph(s) of batch i
                                "pipelined_feed": bool(gt.overlap)}
            # the reference's own call, trainer.train_experience_replay (basic_ddm_dc.py:199-202): simulate graph | buffer of
            # 100 batches | training graph of the drawn batch's bucket
            torch.manual_seed(0)
            amortizer = AmortizedPosterior(InvertibleNetwork(num_params=5), InvariantNetwork())
            with GraphTrainer(amortizer, batch_size=Bl, total_steps=warm + 2 * a.train_iters, dt=dt, max_steps=ms, seed=2023,
                              device=dev, world=world, rank=rank, parallel=a.train_parallel, backend=a.backend,
                              split=dist_on) as gt:
                gt.train_experience_replay(warm + a.train_iters)
                barrier(a, ctx)
                t0 = time.perf_counter()
                gt.train_experience_replay(a.train_iters)
                barrier(a, ctx)
                el = time.perf_counter() - t0
                h = gt.loss_history()
                leg["graph_experience_replay"] = {"iterations_per_s": a.train_iters / el, "ms_per_iteration": el / a.train_iters * 1e3,
                                                  "loss_first10": float(np.mean(h[:10])), "loss_last10": float(np.mean(h[-10:])),
                                                  "graphs_captured": gt.n_graphs}
            if "eager_prefetch_on" in leg:
                leg["graph_over_eager"] = leg["graph"]["iterations_per_s"] / max(leg["eager_prefetch_on"]["iterations_per_s"],
                                                                                leg["eager_prefetch_off"]["iterations_per_s"])
        # the simulate launch in isolation at the mean shape (Bl sets x 180 trials, device-resident): eager vs graph replay
        p = DevicePrior("basic", seed=2023)(Bl, set_offset=0)
        out = torch.empty((Bl, 180, 2), dtype=torch.float32, device=dev)
        kw = dict(dt=dt, max_steps=ms, seed=2023, set_offset=0, fast=True, out_trials=out, want_summary=False)
        for _ in range(20):
            engine.simulate(engine.BASIC_DDM_DC, p, 180, **kw)
        torch.cuda.synchronize()
        n_rep = 300
        t0 = time.perf_counter()
        for _ in range(n_rep):
            engine.simulate(engine.BASIC_DDM_DC, p, 180, **kw)
        t_host = (time.perf_counter() - t0) / n_rep                     # host time to enqueue
        torch.cuda.synchronize()
        t_eager = (time.perf_counter() - t0) / n_rep
        with engine.graph_memory():                                     # owns the library memory behind the captured launch
            side = torch.cuda.Stream(device=dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):      # (a process group's watchdog may be polling)
                    engine.simulate(engine.BASIC_DDM_DC, p, 180, **kw)
            torch.cuda.synchronize()
            for _ in range(20):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n_rep):
                g.replay()
            torch.cuda.synchronize()
            t_graph = (time.perf_counter() - t0) / n_rep
            del g
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            engine.simulate(engine.BASIC_DDM_DC, p, 180, **kw)
        e1.record()
        torch.cuda.synchronize()
        leg["simulate_launch_us"] = {"shape": f"{Bl} sets x 180 trials, device-resident parameters", "eager_back_to_back": t_eager * 1e6,
                                     "eager_host_enqueue": t_host * 1e6, "hipgraph_replay": t_graph * 1e6,
                                     "gpu_time_back_to_back": e0.elapsed_time(e1) / 50 * 1e3}
        if "graph" in leg:
            leg["simulator_share_of_graph_iteration"] = leg["simulate_launch_us"]["gpu_time_back_to_back"] * 1e-3 / leg["graph"]["ms_per_iteration"]
        results[tag] = leg
    if rank == 0:
        main_leg = results["dt.01_max400"]
        ref = main_leg.get("graph") or main_leg["eager_prefetch_on"]
        driver = "one hipGraph replay per iteration (GraphTrainer)" if "graph" in main_leg else "eager PyTorch loop (Trainer)"
        if world == 1:
            par = "one rank"
        elif a.train_parallel == "gather" or "graph" not in main_leg:
            par = (f"dp{world}: simulation sharded, one all-gather per minibatch, training step REPLICATED on every rank "
                   f"(each trains the same {Bl * world} sets)")
        else:
            par = f"dp{world}: simulation AND training sharded ({Bl} sets per rank), flat gradient all-reduce"
        emit({
            "metric": "training iterations/sec, online simulation feeding the amortizer (BASELINE config 5)",
            "value": ref["iterations_per_s"], "unit": "iterations/s", "n_gpus": world, "steps": a.train_iters, "warmup": warm,
            "ms_per_step": ref["ms_per_iteration"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"basic_ddm_dc online training feed: {Bl} sets per rank per step (minibatch {Bl * world}), "
                                   f"N ~ U{{60..300}} per batch, dt=.01/max 400 (reference default; dt=.001/4000 also reported), "
                                   f"device prior -> simulate -> DeepSet + 6-layer coupling flow, Adam; {driver}",
                       "arithmetic": ARITHMETIC, "parallelism": par, "backend": a.backend,
                       "train_mode": a.train_mode,
                       # libnddm_train.so (flow, summary network, optimizer step as HIP kernels) or the PyTorch composition
                       "training_kernels": "libnddm_train.so" if _train_lib.lib() is not None else "PyTorch (library not built)"},
            "loss_first10": ref["loss_first10"], "loss_last10": ref["loss_last10"], "toolchain": toolchain(torch),
            "train": results})


def main():
    a = parse()
    if a.gpus < 1:
        sys.exit("--gpus must be >= 1")
    if a.plan:
        return plan(a)
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        if not a.share_device:
            have = visible_gpus()
            if a.gpus > have:
                sys.exit(f"bench.py: --gpus {a.gpus} but this node shows {have} GPU(s); nothing was started "
                         f"(--share-device --backend gloo rehearses the multi-rank path on one GPU)")
        launch_ranks(a.gpus)                         # does not return
    if "WORLD_SIZE" not in os.environ and a.dist:    # one rank, distributed code path: be our own launcher
        os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
                          NDDM_BENCH_OWN_LAUNCHER="1")
    worker(a)


if __name__ == "__main__":
    main()

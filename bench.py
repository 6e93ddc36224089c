#!/usr/bin/env python3
"""bench.py -- simulated DDM trials per second (BASELINE.json metric) on N MI355X GPUs of one node.

One "step" = one pass of the hot path over one batch: `--sets` parameter sets x `--trials` trials of the
basic_ddm_dc Euler-Maruyama simulator at dt=0.001 / max_steps=4000 (BASELINE.json configs[1]: 1M x 300), parameters
already resident in HBM, output = float32 (rt, choice) pairs [B, 300, 2] + fused per-set summaries [B, 10].
Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL); the batch shards embarrassingly --
rank r simulates global set indices [r*B, (r+1)*B) of each step with no data-path collective (weak scaling);
`--gather summary|trials` adds the RCCL all-gather that reassembles a training minibatch (north_star) to the
timed region.

Rank 0 prints ONE JSON line (contract in the task description) with two extra objects:
  roofline      HBM view of the dominant kernel (algorithmic bytes / measured kernel time vs 8 TB/s) -- tiny by
                construction: 8 B are written per trial for ~246 Gaussian draws
  roofline_valu the binding resource: vector-ALU issue cycles (instruction mix of the step loop x measured
                per-instruction issue cost) -- see DESIGN.md section 6
  cpu_baseline  the CPU oracle (C restatement, same Philox stream) timed on this box's host cores on a bounded
                sample of the same workload; plus the pure-Python/NumPy port the reference runs without numba
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# VALU ceiling model: issue cycles per wave64 for one Philox block = 4 E-M steps, fast Gaussian
# mode, from the ISA of sim_kernel<basic, fast> x the measured per-instruction issue cost (tools/ubench_valu)
VALU_MODEL = {
    "clock_ghz": 2.4, "simds": 1024,
    # SIMD cycles (at 2.4 GHz) one wave64 needs per Philox block (4 E-M steps x 64 lanes) when EVERY lane is useful:
    # measured with tools/quick_time.py's lockstep run (all trials run to the step cap: no refill, no idle lanes, same
    # kernel, same residency): 2.48e12 E-M steps/s at lane efficiency 0.981 = 2.53e12 with every lane useful = 249 cycles
    # per block (fast); 1.231e12 at 0.979 = 500 cycles (exact) -- profiles/r1_summary.md.  The sum of the isolated
    # per-instruction issue costs of the loop (tools/isa_mix.py x profiles/r1_ubench_valu.txt) is 274 / 595: the real
    # loop issues better than that sum, so the measured figure is the tighter ceiling.
    "cycles_per_block_fast": 249.0, "cycles_per_block_exact": 500.0,
    "sum_of_issue_costs_fast": 274.0, "sum_of_issue_costs_exact": 595.0,
}


def pmc_traffic(B, N):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/*_pmc.json, made by tools/gpu_profile.sh + tools/summarize_profile.py): WRITE_SIZE is exact for streaming
    stores; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950.  None if no matching profile."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")), reverse=True):
        try:
            d = json.load(open(path))
            if d["sets_per_gpu"] == B and d["n_trials"] == N and "WRITE_SIZE" in d["pmc_per_launch"]:
                c = d["pmc_per_launch"]
                return {"bytes": (2.0 * c.get("FETCH_SIZE", 0.0) + c["WRITE_SIZE"]) * 1024.0, "source": os.path.basename(path)}
        except (OSError, KeyError, ValueError):
            continue
    return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--sets", type=int, default=1_000_000, help="parameter sets per GPU per step")
    ap.add_argument("--trials", type=int, default=300)
    ap.add_argument("--dt", type=float, default=0.001)
    ap.add_argument("--max-steps", type=float, default=4000.0)
    ap.add_argument("--gauss", choices=["fast", "exact"], default="fast")
    ap.add_argument("--model", choices=["basic", "single", "alpha_ns", "alpha_ns_bridge"], default="basic",
                    help="basic = BASELINE configs[1] (the headline); single = configs[3]; alpha_ns* = configs[2]")
    ap.add_argument("--gather", choices=["none", "summary", "trials"], default="none")
    ap.add_argument("--summary-only", action="store_true", help="do not write the 8 B/trial (fused summaries only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the oracle baseline sample")
    ap.add_argument("--no-ks", action="store_true")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo + --share-device rehearse the multi-process path on a one-GPU box")
    ap.add_argument("--share-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    return ap.parse_args()


def cpu_baseline(params, n_trials, dt, max_steps, target_s):
    """Time the CPU oracle (kind 'port': our C restatement on the same Philox stream) on a bounded sample of the
    same workload: the first S parameter sets, S sized from a pilot so that the single-thread run takes ~target_s."""
    import oracle
    from oracle import numpy_port
    oracle.build()
    # threads actually used for the multi-core figure: the GPU box grants a CPU share of 16 cores per GPU, whatever
    # os.cpu_count() says (256 there)
    cores = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), 16)
    pilot = 64
    t0 = time.perf_counter()
    oracle.philox_simulate(oracle.M_BASIC, params[:pilot], n_trials, dt=dt, max_steps=max_steps, seed=1, threads=1)
    t_pilot = time.perf_counter() - t0
    S = int(max(pilot, min(len(params), pilot * target_s / max(t_pilot, 1e-6))))
    t0 = time.perf_counter()
    oracle.philox_simulate(oracle.M_BASIC, params[:S], n_trials, dt=dt, max_steps=max_steps, seed=1, threads=1)
    t1 = time.perf_counter() - t0
    v1 = S * n_trials / t1
    # all host cores, same sample scaled up
    Sm = int(min(len(params), S * cores))
    t0 = time.perf_counter()
    oracle.philox_simulate(oracle.M_BASIC, params[:Sm], n_trials, dt=dt, max_steps=max_steps, seed=1, threads=cores)
    tm = time.perf_counter() - t0
    vm = Sm * n_trials / tm
    # the pure-Python/NumPy algorithm the reference runs when numba is absent (numba is not installed here)
    Sp = 12
    np.random.seed(2023)
    t0 = time.perf_counter()
    for i in range(Sp):
        numpy_port.basic_simulate_trials(params[i].astype(np.float64), n_trials, dt=dt, max_steps=max_steps)
    tp = time.perf_counter() - t0
    return {"value": v1, "unit": "trials/s", "cores": 1, "kind": "port",
            "sample": f"first {S} of the step's parameter sets x {n_trials} trials, C oracle (Philox stream), "
                      f"{t1:.1f} s single thread",
            "all_cores": {"value": vm, "cores": cores, "host_cpu_count": os.cpu_count(),
                          "sample": f"{Sm} sets, OpenMP over sets with {cores} threads, {tm:.1f} s"},
            "numpy_port": {"value": Sp * n_trials / tp, "cores": 1,
                           "sample": f"{Sp} sets x {n_trials} trials, pure-Python/NumPy statement of "
                                     f"basic_ddm_dc.py:85-125 (numba not installed), {tp:.1f} s"}}


def ks_vs_golden(engine, dt, max_steps, fast):
    """KS distance of the signed RT distribution vs the golden histograms made from the reference's NumPy
    simulator (tests/golden/ks_hist.npz), >= 4e5 trials per side, all fixed basic_ddm_dc parameter sets."""
    from bayesflow_nddms_amd import diagnostics as dg
    path = os.path.join(ROOT, "tests", "golden", "ks_hist.npz")
    if not os.path.exists(path):
        return None
    gold = np.load(path)
    dts = list(gold["dt"])
    if dt not in dts:
        return None
    ci = dts.index(dt)
    if float(gold["max_steps"][ci]) != float(max_steps):
        return None
    K = int(max_steps)
    worst, per = 0.0, []
    for si, p in enumerate(gold["basic_sets"]):
        key = f"basic_hist_s{si}_c{ci}"
        if key not in gold:
            continue
        r = engine.simulate(engine.BASIC_DDM_DC, np.tile(p, (2048, 1)), 200, dt=dt, max_steps=max_steps,
                            seed=777, set_offset=si * 4096, fast=fast, want_summary=False)
        h = dg.step_hist_from_trials(r["trials"].cpu().numpy(), float(np.float32(p[3])), dt, K)
        ks = dg.ks_signed(h, gold[key])
        per.append(round(ks, 5))
        worst = max(worst, ks)
    return {"max": worst, "per_set": per, "n_trials_per_side": 409600, "bar": 0.01,
            "reference": "NumPy reference simulator (basic_ddm_dc.py:85-125), tests/golden/ks_hist.npz"}


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh worker processes of this same command, one per GPU
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, as torch.distributed.run would set them), wait for
    them and exit with the worst status.  Nothing in THIS process has touched the GPU or imported torch: the workers
    are children, never an exec of a process that initialised HIP.  Rank 0 prints the JSON line on the inherited
    stdout."""
    import socket
    import subprocess
    with socket.socket() as s:                       # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), NDDM_BENCH_WORKER="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
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


def main():
    a = parse()
    if a.gpus < 1:
        sys.exit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        launch_ranks(a.gpus)                         # does not return
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
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    from bayesflow_nddms_amd import engine
    from bayesflow_nddms_amd import priors as prior_util

    B, N = a.sets, a.trials
    fast = a.gauss == "fast"
    # synthetic inputs: the reference prior (basic_ddm_dc.py:62-80 / single_trial_alpha_not_scaled.py:78-102 /
    # alpha_not_scaled.py:66-72), default_rng(2023 + rank), resident in HBM
    model_id = {"basic": engine.BASIC_DDM_DC, "single": engine.SINGLE_TRIAL, "alpha_ns": engine.ALPHA_NOT_SCALED,
                "alpha_ns_bridge": engine.ALPHA_NOT_SCALED}[a.model]
    bridge = a.model == "alpha_ns_bridge"
    p_host = {"basic": prior_util.basic_prior_matrix, "single": prior_util.single_prior_matrix,
              "alpha_ns": prior_util.alpha_ns_prior_matrix, "alpha_ns_bridge": prior_util.alpha_ns_prior_matrix}[a.model](B, 2023 + rank)
    p_dev = torch.as_tensor(p_host).to(dev)
    out_trials = None if a.summary_only else torch.empty((B, N, 2), dtype=torch.float32, device=dev)
    out_summary = torch.empty((B, engine.SUMMARY_K), dtype=torch.float32, device=dev)
    gathered = None
    if world > 1 and a.gather != "none":
        src = out_summary if a.gather == "summary" else out_trials
        gathered = torch.empty((world,) + tuple(src.shape), dtype=torch.float32, device=dev)

    def step(i):
        # every step is a fresh batch: global set index = (i*world + rank)*B + row, one seed
        engine.simulate(model_id, p_dev, N, dt=a.dt, max_steps=a.max_steps, seed=2023,
                        set_offset=(i * world + rank) * B, fast=fast, out_trials=out_trials, out_summary=out_summary,
                        want_trials=not a.summary_only, bridge=bridge)
        if gathered is not None:
            src = out_summary if a.gather == "summary" else out_trials
            if a.backend == "nccl":
                dist.all_gather_into_tensor(gathered, src)              # one RCCL all-gather per batch
            else:                                                       # gloo rehearsal: list form
                dist.all_gather(list(gathered.unbind(0)), src)

    def barrier():
        if world > 1:
            if a.backend == "nccl":
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        step(i)
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    t0 = time.perf_counter()
    for i in range(a.steps):
        ev[i][0].record()              # torch's current stream == the stream the kernel is launched on
        step(a.warmup + i)
        ev[i][1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev]))

    # executed Euler-Maruyama steps of the last step, from the fused summaries (exact integer sums)
    s = out_summary.double()
    tau = p_dev[:, 3].double()
    if bridge:
        tau = tau - 0.5 * a.dt      # RTs carry a uniform sub-step jitter in bridge mode (mean -dt/2)
    n_resp = s[:, 0] + s[:, 1]
    max_k = engine.max_k_of(a.max_steps)
    mean_k = torch.where(n_resp > 0, (s[:, 3] - tau) / a.dt, torch.zeros_like(tau))
    em_steps = float((mean_k * n_resp + s[:, 2] * max_k).sum().item())
    p_missing = float((s[:, 2].sum() / (B * N)).item())

    if rank == 0:
        trials_per_step = world * B * N
        value = trials_per_step * a.steps / elapsed
        alg_bytes = B * N * (0 if a.summary_only else 8) + B * (p_host.shape[1] * 4 + engine.SUMMARY_K * 4)
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        res = {
            "metric": f"simulated DDM trials/sec at n_trials={N} dt={a.dt:g} ({a.model if a.model != 'basic' else 'basic_ddm_dc'}, max_steps={a.max_steps:g})",
            "value": value, "unit": "trials/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{'basic_ddm_dc' if a.model == 'basic' else a.model} HIP simulator, {B} parameter sets x {N} trials per GPU per step, "
                                   f"dt={a.dt}, max_steps={a.max_steps:g}, params ~ reference prior (default_rng 2023)",
                       "sets_per_gpu": B, "n_trials": N, "dt": a.dt, "max_steps": a.max_steps,
                       "gauss": a.gauss, "outputs": "summaries only" if a.summary_only else "trials f32[B,N,2] + summaries f32[B,10]",
                       "parallelism": f"dp{world} over parameter sets, gather={a.gather}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "nddm::sim_kernel<%s, %s>" % (a.model, "fast" if fast else "exact"),
                         "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "path is VALU-bound, not HBM- or MFMA-bound: see roofline_valu"},
            "em_steps_per_trial": em_steps / (B * N), "em_steps_per_s_per_gpu": em_steps / (kern_ms * 1e-3),
            "p_missing": p_missing,
        }
        tr = None if (a.summary_only or a.model != "basic") else pmc_traffic(B, N)
        if tr:
            res["roofline"]["traffic"] = tr["bytes"]
            res["roofline"]["traffic_source"] = tr["source"]
        cpb = VALU_MODEL["cycles_per_block_fast" if fast else "cycles_per_block_exact"]
        if cpb and a.model == "basic":
            # a wave64 advances 64 lanes x 4 steps per block; ceiling assumes every lane useful
            peak_steps = VALU_MODEL["simds"] * VALU_MODEL["clock_ghz"] * 1e9 / cpb * 64 * 4
            res["roofline_valu"] = {"bound": "valu", "achieved": em_steps / (kern_ms * 1e-3) / 1e9,
                                    "peak": peak_steps / 1e9, "unit": "G E-M steps/s",
                                    "frac": em_steps / (kern_ms * 1e-3) / peak_steps,
                                    "issue_cycles_per_block": cpb, "clock_ghz": VALU_MODEL["clock_ghz"],
                                    "ceiling": "step loop with every lane useful (measured lockstep run)",
                                    "sum_of_isolated_issue_costs": VALU_MODEL["sum_of_issue_costs_fast" if fast
                                                                              else "sum_of_issue_costs_exact"]}
        if world == 1 and "roofline_valu" in res:
            # steps the lanes actually EXECUTED (incl. lanes idling on a finished trial until the next refill): one more
            # launch of the last batch, outside the timed region, with the kernel's debug counters switched on
            from bayesflow_nddms_amd import _lib
            dbg = torch.zeros(8, dtype=torch.int64, device=dev)
            _lib.lib().nddm_set_debug_counters(dbg.data_ptr())
            d0, d1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            d0.record()
            step(a.warmup + a.steps - 1)
            d1.record()
            torch.cuda.synchronize()
            dbg_ms = d0.elapsed_time(d1)
            _lib.lib().nddm_set_debug_counters(None)
            d = dbg.cpu().numpy().astype(np.float64)
            res["roofline_valu"].update({"executed_lane_steps_per_launch": d[0] * 256.0,
                                         "lane_efficiency": em_steps / (d[0] * 256.0),
                                         "philox_blocks_per_refill": d[0] / max(d[1], 1.0), "waves": int(d[4])})
            cus = torch.cuda.get_device_properties(dev).multi_processor_count
            # waves actually resident: sum of the waves' lifetimes (100 MHz s_memrealtime) over kernel time x SIMDs
            resident = d[3] * 1e-8 / (dbg_ms * 1e-3) / (4.0 * cus)
            res["occupancy"] = {"resident_waves_per_simd": resident, "hardware_max": 8, "grid_waves": int(d[4]),
                                "limit": "SGPR file: 73 SGPRs per wave -> 8 wave64 per SIMD for this kernel, 75-83 -> 7 for the other models (DESIGN.md 5.1)",
                                "note": "from in-kernel wave lifetimes; PMC SQ_WAVE_CYCLES agrees (profiles/r1_summary.md)"}
        if world == 1 and not a.no_ks and a.model == "basic":
            res["ks_vs_ref"] = ks_vs_golden(engine, a.dt, a.max_steps, fast)
        if world == 1 and not a.no_cpu_baseline and a.model == "basic":
            res["cpu_baseline"] = cpu_baseline(p_host, N, a.dt, a.max_steps, a.cpu_seconds)
            res["gpu_over_cpu_1core"] = value / res["cpu_baseline"]["value"]
        print(json.dumps(res), flush=True)
    if world > 1:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""bench.py's one-line JSON contract, exercised on the GPU at a reduced size (a second python process)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--sets", "30000", "--cpu-seconds", "0.5", "--cpu-sample", "--leg-launches", "2", "--leg-train-iters", "40"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # exactly ONE JSON line
    d = json.loads(lines[0])
    _check_legs(d)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "trials/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 30000 * 300 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert d["ks_vs_ref"]["max"] < 0.01
    rv = d["roofline_valu"]
    # `frac` is the fraction of the HARDWARE-derived ceiling (the ISA issue model of the shipped library: it must match the library
    # that ran); the kernel's own lockstep run, measured in the same process, stands beside it
    assert rv["issue_model"]["library_matches"] is True and abs(rv["frac"] - rv["achieved"] / rv["peak"]) < 1e-9
    assert 0 < rv["frac"] < 1.05 and 0 < rv["frac_vs_lockstep"] < 1.05 and rv["ceiling_measured_steps_per_s"] > 0
    assert d["pass_allocated_bytes"] == 30000 * 300 * 8 + 30000 * 40                   # == bench.pass_buffers(): --plan's figure
    assert "arithmetic" in d["config"] and d["cpu_baseline"]["numpy_port"]["reference_default"]["dt"] == 0.01
    assert d["config"]["gauss"] == "fast" and d["packed_gauss"]["ks_vs_ref"]["max"] < 0.01     # opt-in mode: beside, not as, the headline


def _check_legs(d):
    """The default N = 1 line carries EVERY BASELINE config, not only configs[1]: `legs.single` (configs[3]: trials + fused summaries,
    and summaries alone), `legs.alpha_ns_bridge` (configs[2]) and `legs.train` (configs[4]: the graph trainer at one rank and in its
    RCCL all-gather form, at both step sizes) -- each with its rate, kernel time, KS distance against the reference fixtures WITH its
    bar, and the VALU roofline against the lockstep ceiling of its own kernel variant, all measured in this run."""
    legs = d["legs"]
    assert not any("error" in (legs[k] if isinstance(legs[k], dict) else {}) for k in legs) and "error" not in legs["train"]["one_rank"] \
        and "error" not in legs["train"]["gather_rccl_world1"], legs
    for name, kernel in (("single", "sim_kernel<1 (single_trial)"), ("alpha_ns_bridge", "bridge>"), ("basic_dt01", "sim_kernel<0 (basic_ddm_dc), fast>"),
                         ("exact_gauss", "sim_kernel<0 (basic_ddm_dc), exact>")):
        leg = legs[name]
        assert leg["unit"] == "trials/s" and leg["value"] > 1e8 and leg["kernel_ms"] > 0 and kernel in leg["kernel"], (name, leg["value"])
        assert abs(leg["value"] - 30000 * 300 / (leg["kernel_ms"] * 1e-3)) / leg["value"] < 1e-6
        ks = leg["ks_vs_ref"]
        assert ks["bar"] == 0.01 and ks["max"] < ks["bar"] and ks["meets_bar"] is True, (name, ks["max"])
        rv = leg["roofline_valu"]
        assert 0 < rv["frac"] < 1.05 and 0 < rv["frac_vs_lockstep"] < 1.05 and rv["ceiling_lane_efficiency"] > 0.9
        assert rv["issue_model"]["library_matches"] is True and "ISA issue model" in rv["ceiling"]
        assert 0.5 < rv["lane_efficiency"] <= 1.0 and rv["philox_blocks_per_refill"] > 1            # the kernel's own counters, per leg
        assert leg["roofline"]["bound"] == "hbm" and 0 < leg["roofline"]["frac"] < 1
        assert leg["em_steps_per_trial"] > (20 if name == "basic_dt01" else 50)
    # the reference's own default shape (dt=.01 / max_steps 400, basic_ddm_dc.py:87) and the bit-pinned transform are under this
    # command's clock, each with its own KS against the reference histograms of ITS step size
    assert legs["basic_dt01"]["dt"] == 0.01 and legs["basic_dt01"]["max_steps"] == 400.0 and legs["basic_dt01"]["gauss"] == "fast"
    assert legs["exact_gauss"]["dt"] == d["config"]["dt"] and legs["exact_gauss"]["gauss"] == "exact"
    assert legs["exact_gauss"]["value"] < d["value"]                                   # the polynomial transform costs what it costs
    # NDDM_STATE_F64 beside it, with its cost against the float32 state of the same transform
    f64 = legs["state_f64"]
    assert f64["state_f64"] is True and f64["gauss"] == "exact" and f64["ks_vs_ref"]["max"] < 0.01
    assert 0.3 < f64["rate_vs_f32_state_same_transform"] < 1.02 and 0.3 < f64["fast_transform"]["rate_vs_f32_state_same_transform"] < 1.02
    # configs[2] with the reference's own generator (simulratcliff) on the device: no step size, KS against the reference's draws
    ex = legs["alpha_ns_exact_sampler"]
    assert ex["unit"] == "trials/s" and ex["value"] > 1e8 and ex["ks_vs_ref"]["max"] < ex["ks_vs_ref"]["bar"] == 0.01
    assert 0 < ex["exact_transform"]["value"] <= 1.05 * ex["value"] and 0.2 < ex["mean_rt_s"] < 2.0
    # its VALU statement: the measured instruction stream (PMC file of THIS library, else absent) against the ISA mix of its loop
    if "valu" in ex["roofline"]:
        rv = ex["roofline_valu"]
        assert rv["issue_model"]["library_matches"] is True and 0.3 < rv["exec_mask_utilisation"] <= 1.0 and 0.5 < rv["valu_pipe_busy"] < 1.2
        assert abs(rv["frac"] - min(1.0, rv["valu_pipe_busy"]) * rv["exec_mask_utilisation"]) < 1e-9
    so = legs["single"]["summary_only"]
    assert so["value"] > 0 and so["kernel_ms"] > 0                                         # (no 8 B per trial: its rate is in the line, not asserted)
    tr = legs["train"]
    assert tr["unit"] == "iterations/s" and tr["training_kernels"] == "libnddm_train.so"
    assert tr["value"] == tr["one_rank"]["dt.01_max400"]["iterations_per_s"]
    for form in ("one_rank", "gather_rccl_world1"):
        for tag in ("dt.01_max400", "dt.001_max4000"):
            t = tr[form][tag]
            assert t["iterations_per_s"] > 500 and t["all_losses_finite"] and t["loss_last10"] < t["loss_first10"], (form, tag, t)
            assert t["pipelined_feed"] is True and t["graphs_captured"] >= 2
            assert (t["collective"] is not None) == (form == "gather_rccl_world1")
    assert legs["wall_seconds"] < 60
    tc = d["toolchain"]
    assert tc["hip_runtime"] and tc["hipcc"] and tc["torch"] and tc["rccl"] and "libamdhip64" in tc["hip_runtime_library"]


@pytest.mark.parametrize("model", ["single", "alpha_ns_bridge"])
def test_bench_other_models_carry_the_same_objects(model):
    """configs[2] / configs[3] are measured the way configs[1] is: KS against the reference fixtures, CPU oracle
    baseline of the same model, lockstep ceiling."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--model", model, "--steps", "2", "--warmup", "1",
                        "--sets", "30000", "--cpu-seconds", "0.5", "--cpu-sample"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert model in d["metric"] and d["ks_vs_ref"]["max"] < 0.01 and d["cpu_baseline"]["value"] > 0
    assert 0 < d["roofline_valu"]["frac"] < 1.05 and d["occupancy"]["resident_waves_per_simd"] > 3


def _bench(*args, timeout=600, env=None):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True,
                          timeout=timeout, cwd=ROOT, env=env)


@pytest.mark.parametrize("gather", ["none", "summary"])
def test_bench_gpus_2_starts_two_ranks(gather):
    """`python bench.py --gpus 2` starts its two ranks itself (fresh processes) and rank 0 prints ONE line with n_gpus 2
    and the aggregate of both ranks.  On a one-GPU box the ranks share cuda:0 over gloo (the rehearsal mode)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = _bench("--gpus", "2", "--share-device", "--backend", "gloo", "--sets", "20000", "--steps", "2", "--warmup", "1",
               "--gather", gather, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and f"gather={gather}" in d["config"]["parallelism"]
    assert abs(d["value"] - 2 * 20000 * 300 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6


def test_bench_many_ranks_rehearsal_names_what_an_n_gpu_line_omits():
    """The driver's 8-GPU command, rehearsed with FIVE ranks sharing cuda:0 over gloo (this test process holds the card as well
    and the box allows six GPU processes at once): the parent starts the ranks, every rank reports its device and its host-side
    parameter draw on standard error, rank 0 prints ONE line with n_gpus 5 -- and, because the per-kernel analysis
    (roofline_valu, KS, CPU baseline ...) is measured at N = 1 only, the line NAMES the objects it omits (`n1_only`)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = _bench("--gpus", "5", "--share-device", "--backend", "gloo", "--sets", "20000", "--steps", "3", "--warmup", "1", env=env)
    d = _one_line(r)
    assert d["n_gpus"] == 5 and d["scaling"] == "weak" and "roofline" in d and "launch" in d
    # the line says WHO ran it: the backend the process group reports, one entry per rank with its device and its own times
    ds = d["dist"]
    assert ds["backend"] == "gloo" and ds["world"] == 5 and [r_["rank"] for r_ in ds["ranks"]] == [0, 1, 2, 3, 4]
    assert len({r_["pid"] for r_ in ds["ranks"]}) == 5 and all(r_["device"] == "cuda:0" and r_["kernel_ms"] > 0 and r_["elapsed_s"] > 0
                                                                and r_["host_prior_s"] >= 0 for r_ in ds["ranks"])
    assert ds["distinct_devices"] == 1 and ds["imbalance"]["kernel_ms_max_over_min"] >= 1.0       # (the rehearsal: five ranks, ONE card)
    # ... and carries what the collective-free weak-scaling headline leaves out: north_star's all-gather (two forms) and the strong point
    sl = d["side_legs"]
    for k in ("gather_summary", "gather_codes"):
        assert sl[k]["scaling"] == "weak" and sl[k]["value"] > 0 and sl[k]["bytes_gathered_per_rank_per_step"] > 0
        assert abs(sl[k]["value"] - 5 * 20000 * 300 / (sl[k]["ms_per_step"] * 1e-3)) / sl[k]["value"] < 1e-6
    assert sl["strong"]["scaling"] == "strong" and sl["strong"]["sets_total"] == 20_000 and sl["strong"]["sets_per_gpu"] == 4_000
    assert abs(d["value"] - 5 * 20000 * 300 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert set(d["n1_only"]) >= {"roofline_valu", "ks_vs_ref", "cpu_baseline"} and not (set(d["n1_only"]) & set(d))
    import re                          # (five processes share one stderr pipe: a line may start in the middle of another's)
    assert sorted(int(m) for m in re.findall(r"bench\.py: rank (\d+)/5 on cuda:0", r.stderr)) == [0, 1, 2, 3, 4]
    one = _one_line(_bench("--sets", "20000", "--steps", "2", "--warmup", "1", "--no-ceiling", "--no-ks", "--no-cpu-baseline", env=env))
    assert "n1_only" not in one and one["host_prior_seconds_rank0"] < 5.0 and "dist" not in one and "side_legs" not in one


def _one_line(r):
    """Standard output is EXACTLY the JSON line: what libraries print to file descriptor 1 (RCCL's version banner) is sent to
    standard error by bench.py."""
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[:500]
    return json.loads(lines[0])


@pytest.mark.parametrize("gather", ["summary", "trials", "codes"])
def test_bench_rccl_branch_runs_at_world_1(gather):
    """`bench.py --dist` takes the multi-rank code path on ONE GPU through bench.py itself: init_process_group("nccl",
    device_id=...), barrier(device_ids=...), all_gather_into_tensor on the communication stream (double-buffered outputs),
    the device-side all_reduce(MAX) of the elapsed time -- every distributed call `bench.py --gpus 8` makes.  The line says
    n_gpus 1 and its value is the plain run's (the collective of one rank is a copy, overlapped with the next simulate)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    common = ["--sets", "300000", "--steps", "6", "--warmup", "2", "--no-ceiling", "--no-ks", "--no-cpu-baseline", "--no-legs"]
    d = _one_line(_bench("--dist", "--backend", "nccl", "--gather", gather, *common, env=env))
    assert d["n_gpus"] == 1 and "distributed code path forced at world 1 (nccl)" in d["config"]["parallelism"]
    assert d["dist"]["backend"] == "nccl" and d["dist"]["rccl_version"] and d["dist"]["world"] == 1 and d["dist"]["ranks"][0]["pci_bus_id"]
    assert f"gather={gather}" in d["config"]["parallelism"] and "communication stream" in d["config"]["parallelism"]
    # how the gathered pass compares with the plain one is a NUMBER OF THE LINE (both passes interleaved in bench.py's own process),
    # not an assertion between two separately launched processes: a throughput comparison across processes inside a correctness suite
    # fired once on a hiccup of the box (gpurun_out/r5/gpu_suite1.log) and would again
    cmp_ = d["dist"]["gathered_over_plain_same_process"]
    assert cmp_["ratio"] > 0 and len(cmp_["gathered_trials_per_s"]) == len(cmp_["plain_trials_per_s"]) == 3
    # what the pass allocated is what `bench.py --plan` says it allocates
    sys.path.insert(0, ROOT)
    import bench
    want = sum(v for k, v in bench.pass_buffers(1, 300000, 300, 5, gather).items() if not k.startswith(("params", "library")))
    assert d["pass_allocated_bytes"] == want, (d["pass_allocated_bytes"], want)
    # and serialised on the simulate stream (the round-2 form) it still runs
    e = _one_line(_bench("--dist", "--backend", "nccl", "--gather", gather, "--no-overlap", *common, env=env))
    assert e["n_gpus"] == 1 and "communication stream" not in e["config"]["parallelism"]
    if gather == "summary":
        # the driver's PLAIN command over RCCL (gather none): the line carries the all-gather and strong-scaling side legs, and -- at
        # world 1 -- the N = 1 legs too, whose RCCL training form joins the process group that is already up
        f = _one_line(_bench("--dist", "--backend", "nccl", "--sets", "30000", "--steps", "2", "--warmup", "1", "--no-ceiling", "--no-ks",
                             "--no-cpu-baseline", "--leg-launches", "1", "--leg-train-iters", "30", env=env))
        assert f["dist"]["backend"] == "nccl" and set(f["side_legs"]) == {"gather_summary", "gather_codes", "strong"}
        assert all(v["value"] > 0 and v["kernel_ms"] > 0 for v in f["side_legs"].values()) and f["side_legs"]["strong"]["sets_total"] == 30000
        assert f["legs"]["train"]["gather_rccl_world1"]["dt.01_max400"]["iterations_per_s"] > 300


def test_bench_gather_summary_on_two_ranks_reports_its_cost_in_one_job():
    """Two ranks sharing cuda:0 (gloo) with the all-gather of the summaries on the communication stream: ONE job interleaves passes with
    and without the gather (`--compare-plain`) and the line reports gathered / plain as a number (measured on four boxes: 0.89-0.93 --
    gloo stages 6 MB per rank and step through the host while two processes time-share one card; serialised, or with oversubscribed host
    threads, 0.04-0.7).  The suite asserts the line's structure only: a throughput bar between separately launched processes does not
    belong in a correctness suite (it fired once on a hiccup of the box in round 5)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    d = _one_line(_bench("--gpus", "2", "--share-device", "--backend", "gloo", "--sets", "150000", "--steps", "20", "--warmup", "2",
                         "--gather", "summary", "--compare-plain", env=env))
    assert d["n_gpus"] == 2 and "gather=summary" in d["config"]["parallelism"] and "communication stream" in d["config"]["parallelism"]
    cmp_ = d["dist"]["gathered_over_plain_same_process"]
    assert cmp_["passes"] == 3 and len(cmp_["gathered_trials_per_s"]) == len(cmp_["plain_trials_per_s"]) == 3
    assert all(v > 0 for v in cmp_["gathered_trials_per_s"] + cmp_["plain_trials_per_s"]) and cmp_["ratio"] > 0
    print("two ranks on one card over gloo: gathered / plain =", round(cmp_["ratio"], 3))


def test_bench_train_two_ranks_sharded_feed():
    """BASELINE configs[4]: the online-training feed with the simulation sharded over two ranks (fresh processes sharing
    cuda:0 over gloo): one JSON line, n_gpus 2, minibatch 64, and the loss goes down -- for the eager loop and for the
    hipGraph loop (two graphs with the all-gather between them)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    d = _one_line(_bench("--train", "--gpus", "2", "--share-device", "--backend", "gloo", "--train-iters", "30", env=env, timeout=1200))
    assert d["n_gpus"] == 2 and "minibatch 64" in d["config"]["workload"] and "REPLICATED" in d["config"]["parallelism"]
    assert d["loss_last10"] < d["loss_first10"]
    for tag in ("dt.01_max400", "dt.001_max4000"):
        leg = d["train"][tag]
        for k in ("eager_prefetch_off", "eager_prefetch_on", "graph"):
            assert leg[k]["loss_last10"] < leg[k]["loss_first10"], (tag, k, leg[k])
        assert leg["graph"]["two_graphs_with_collective_between"] is True and leg["graph"]["pipelined_feed"] is True
        # the eager comparator LEARNS (it used to run its timed iterations at learning rate 0): same schedule, same number of
        # iterations as the graph leg -> the same neighbourhood of the loss
        for k in ("eager_prefetch_off", "eager_prefetch_on"):
            assert abs(leg[k]["loss_last10"] - leg["graph"]["loss_last10"]) < 0.25 * abs(leg["graph"]["loss_last10"]) + 0.5, (tag, k, leg)


def test_bench_train_ddp_at_world_1_over_rccl():
    """`--train --dist --train-parallel ddp`: sharded training (flat-gradient all-reduce between the two graphs of an
    iteration) over RCCL on one GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    d = _one_line(_bench("--train", "--dist", "--backend", "nccl", "--train-mode", "graph", "--train-parallel", "ddp",
                         "--train-iters", "30", env=env, timeout=1200))
    assert d["n_gpus"] == 1 and d["loss_last10"] < d["loss_first10"]
    assert d["train"]["dt.01_max400"]["graph"]["two_graphs_with_collective_between"] is True
    assert d["config"]["training_kernels"] == "libnddm_train.so"      # (a silent fall-back to PyTorch would be a 4 x slower line)


def test_bench_refuses_more_ranks_than_gpus_in_the_parent():
    """--gpus N > visible GPUs without --share-device: the parent exits non-zero with one line and starts nothing."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    import torch
    n = torch.cuda.device_count() + 1
    r = _bench("--gpus", str(n), "--sets", "1000", "--steps", "1", "--warmup", "0", env=env)
    assert r.returncode != 0 and f"--gpus {n} but this node shows {n - 1} GPU(s); nothing was started" in r.stderr
    assert len(r.stderr.strip().splitlines()) == 1 and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_under_torch_distributed_run():
    """The driver's launch form for N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` -- bench.py joins the ranks it is given (here two, sharing cuda:0 over gloo) and
    rank 0 prints the one line."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-device",
                        "--backend", "gloo", "--sets", "50000", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    d = _one_line(r)
    assert d["n_gpus"] == 2 and abs(d["value"] - 2 * 50000 * 300 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6


def test_bench_refuses_a_world_size_it_was_not_asked_for():
    """--gpus N must describe the job that runs: under a launcher with another WORLD_SIZE the bench exits non-zero instead
    of silently measuring something else."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = _bench("--gpus", "2", "--sets", "1000", "--steps", "1", "--warmup", "0", env=env)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_two_fresh_ranks_union_equals_unsharded(tmp_path):
    """Two fresh child ranks (gloo, both on cuda:0) simulate their shards with the HIP engine: what every rank holds after
    the all-gather -- and the concatenation of the ungathered shards -- equals the unsharded launch bit for bit."""
    import socket
    import torch
    import prior_util
    from bayesflow_nddms_amd import engine
    B, N = 1001, 77
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "two_rank_worker.py"), str(tmp_path), str(B), str(N)],
                                      env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-2000:]
    p_dev = torch.as_tensor(prior_util.basic_prior(B, 21)).cuda()
    full = engine.simulate(engine.BASIC_DDM_DC, p_dev, N, seed=31, set_offset=12345, dt=0.001, max_steps=4000, fast=True)
    ft, fs = full["trials"].cpu(), torch.nan_to_num(full["summary"].cpu())
    shards = [torch.load(os.path.join(str(tmp_path), f"rank{r}.pt")) for r in range(2)]
    for sh in shards:
        assert torch.equal(sh["both"]["trials"], ft) and torch.equal(torch.nan_to_num(sh["both"]["summary"]), fs)
        assert torch.equal(sh["codes"]["trials"], ft)               # exchanged as 2-byte codes, decoded on arrival: the same floats
    assert shards[0]["none"]["rows"] == (0, 501) and shards[1]["none"]["rows"] == (501, 1001)
    assert torch.equal(torch.cat([sh["none"]["trials"] for sh in shards]), ft)
    assert torch.equal(torch.nan_to_num(torch.cat([sh["none"]["summary"] for sh in shards])), fs)

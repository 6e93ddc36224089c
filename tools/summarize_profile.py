#!/usr/bin/env python3
"""Condense a tools/gpu_profile.sh output directory (gpurun_out/prof_<tag>) into profiles/<tag>_{kernel_stats.csv,pmc.json,summary.md}.

Usage: python tools/summarize_profile.py <tag>
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
command = open(os.path.join(src, "command.txt")).read().strip()
rows = list(csv.DictReader(open(os.path.join(src, "trace", "bench_kernel_stats.csv"))))
with open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in rows:
        name = r["Name"] if len(r["Name"]) < 120 else r["Name"][:117] + "..."
        w.writerow([name] + [r[k] for k in ("Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")])
bench = json.loads(open(os.path.join(src, "bench_trace.json")).read().strip().splitlines()[-1])
B, N = bench["config"]["sets_per_gpu"], bench["config"]["n_trials"]
sim = max((r for r in rows if "sim_kernel" in r["Name"]), key=lambda r: float(r["TotalDurationNs"]))
# per-dispatch durations of the simulator kernel: the timed launches (grid = resident waves, B sets) and the lockstep
# ceiling launches of bench.py (same kernel, other workload) are told apart by their duration
trace = list(csv.DictReader(open(os.path.join(src, "trace", "bench_kernel_trace.csv")))) if os.path.exists(os.path.join(src, "trace", "bench_kernel_trace.csv")) else []
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6 for r in trace if r["Kernel_Name"] == sim["Name"]]
km = bench["roofline"]["kernel_ms"]
timed = [d for d in durs if abs(d - km) / km < 0.15]
ceil_ms = bench.get("roofline_valu", {}).get("ceiling_kernel_ms")
lock = [d for d in durs if ceil_ms and abs(d - ceil_ms) / ceil_ms < 0.10 and d not in timed]
pmc = collections.OrderedDict()
for fn in sorted(glob.glob(os.path.join(src, "pmc*", "bench_counter_collection.csv"))):
    per_disp = collections.defaultdict(dict)
    for r in csv.DictReader(open(fn)):
        if r["Kernel_Name"] == sim["Name"]:
            per_disp[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    for disp in per_disp.values():
        for k, v in disp.items():
            pmc.setdefault(k, []).append(v)
m = {k: sum(v) / len(v) for k, v in pmc.items()}
table = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "resource_table.py"), "--md"], capture_output=True, text=True).stdout
model = [k for k in ("alpha_ns_bridge", "alpha_ns", "single", "basic") if (k if k != "basic" else "basic_ddm_dc") in bench["metric"]][0]
with open(os.path.join(ROOT, "profiles", f"{tag}_summary.md"), "w") as f:
    f.write(f"# rocprofv3 summary `{tag}` -- `{command}`\n\n")
    f.write("Collected by tools/gpu_profile.sh on one MI355X: pass 1 `--kernel-trace --stats`, then one `--kernel-trace --pmc` pass per counter group.\n\n")
    f.write(f"Dominant kernel: `{sim['Name']}` -- {sim['Calls']} calls, {sim['Percentage']} % of GPU time.\n\n")
    f.write("| launches of that kernel (rocprofv3 kernel trace) | n | average ms |\n|---|---|---|\n")
    if timed:
        f.write(f"| the bench's timed and warm-up steps ({B} sets x {N} trials) | {len(timed)} | {sum(timed)/len(timed):.3f} |\n")
    if lock:
        f.write(f"| bench.py's lockstep ceiling run ({bench['roofline_valu']['ceiling_workload']}) | {len(lock)} | {sum(lock)/len(lock):.3f} |\n")
    f.write(f"\nbench.py's own HIP-event figures in the same (profiled) run: {km:.3f} ms per timed step")
    if ceil_ms:
        f.write(f", {ceil_ms:.3f} ms for the lockstep launch")
    f.write(".\n\n")
    if m:
        f.write(f"| counter (per launch of the timed workload, mean of {len(next(iter(pmc.values())))} launches) | value |\n|---|---|\n")
        for k, v in m.items():
            f.write(f"| {k} | {v:.6g} |\n")
        dur = (sum(timed) / len(timed) if timed else km) * 1e-3
        if "GRBM_GUI_ACTIVE" in m and "SQ_INSTS_VALU" in m:
            clock = m["GRBM_GUI_ACTIVE"] / 8 / dur            # rocprofv3 sums the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back)
            useful_blocks = bench["em_steps_per_trial"] * B * N / 256.0
            f.write("\nDerived (per launch):\n\n")
            f.write(f"* effective clock = GRBM_GUI_ACTIVE / 8 / kernel time = {clock/1e9:.3f} GHz\n")
            f.write(f"* VALU wave-instructions per useful wave-block (64 lanes x 4 E-M steps) = {m['SQ_INSTS_VALU']/useful_blocks:.1f} "
                    f"(step-loop body: tools/isa_mix.py); SALU = {m['SQ_INSTS_SALU']/useful_blocks:.1f}; LDS = {m.get('SQ_INSTS_LDS', 0)/useful_blocks:.1f}\n")
            f.write(f"* VALU issue rate = {m['SQ_INSTS_VALU']/dur/1024/clock:.3f} wave-instructions per SIMD-cycle "
                    f"= {1024*clock*dur/m['SQ_INSTS_VALU']:.2f} SIMD-cycles per VALU instruction\n")
            if "SQ_THREAD_CYCLES_VALU" in m and "SQ_ACTIVE_INST_VALU" in m:
                f.write(f"* exec-mask utilisation of VALU instructions = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) = "
                        f"{m['SQ_THREAD_CYCLES_VALU']/(64*m['SQ_ACTIVE_INST_VALU']):.3f}\n")
            if "SQ_WAVE_CYCLES" in m:
                f.write(f"* mean resident waves per SIMD = 4 x SQ_WAVE_CYCLES / (1024 x kernel cycles) = "
                        f"{4*m['SQ_WAVE_CYCLES']/(1024*clock*dur):.2f}\n")
        if "WRITE_SIZE" in m:
            alg = bench["roofline"]["algorithmic_bytes_per_launch"]
            f.write(f"* HBM traffic = (2 x FETCH_SIZE + WRITE_SIZE) KiB = {(2*m.get('FETCH_SIZE',0)+m['WRITE_SIZE'])*1024/1e9:.3f} GB "
                    f"vs algorithmic {alg/1e9:.3f} GB; WRITE_SIZE alone / algorithmic output bytes = "
                    f"{m['WRITE_SIZE']*1024/(B*N*8 + B*40):.3f}\n")
    f.write("\n## Compiler resource usage (hipcc -Rpass-analysis=kernel-resource-usage; tools/resource_table.py)\n\n" + table)
    f.write("\n## bench line of the traced run\n\n```json\n" + json.dumps(bench) + "\n```\n")
with open(os.path.join(ROOT, "profiles", f"{tag}_pmc.json"), "w") as f:
    json.dump({"command": command, "model": model, "kernel": sim["Name"],
               # what the counters belong to: bench.py quotes this file only for the same step size / transform AND the same library
               "dt": bench["config"]["dt"], "max_steps": bench["config"]["max_steps"], "gauss": bench["config"]["gauss"],
               "source_hash": bench.get("library", {}).get("source_hash"),
               "kernel_avg_ms_rocprof_timed_steps": sum(timed) / len(timed) if timed else None,
               "kernel_avg_ms_rocprof_lockstep": sum(lock) / len(lock) if lock else None,
               "kernel_avg_ms_bench_events": km, "lockstep_ms_bench_events": ceil_ms,
               "sets_per_gpu": B, "n_trials": N, "pmc_per_launch": m}, f, indent=1)
print(open(os.path.join(ROOT, "profiles", f"{tag}_summary.md")).read()[:3000])

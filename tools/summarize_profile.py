#!/usr/bin/env python3
"""Condense a tools/gpu_profile.sh output directory (gpurun_out/prof_<tag>) into profiles/<tag>_*.{csv,md}."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
src = os.path.join("gpurun_out", f"prof_{tag}")
os.makedirs("profiles", exist_ok=True)
rows = list(csv.DictReader(open(os.path.join(src, "trace", "bench_kernel_stats.csv"))))
with open(os.path.join("profiles", f"{tag}_kernel_stats.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in rows:
        name = r["Name"] if len(r["Name"]) < 120 else r["Name"][:117] + "..."
        w.writerow([name] + [r[k] for k in ("Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")])
pmc = collections.OrderedDict()
meta = {}
for fn in sorted(glob.glob(os.path.join(src, "pmc*", "bench_counter_collection.csv"))):
    for r in csv.DictReader(open(fn)):
        if "sim_kernel" in r["Kernel_Name"]:
            pmc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            meta = {k: r[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count")}
bench = json.loads(open(os.path.join(src, "bench_trace.json")).read().strip().splitlines()[-1])
if "roofline_valu" in bench:      # re-price against the CURRENT ceiling in bench.py (the traced run may predate a recalibration)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import VALU_MODEL
    rv = bench["roofline_valu"]
    cpb = VALU_MODEL["cycles_per_block_fast"]
    peak = VALU_MODEL["simds"] * VALU_MODEL["clock_ghz"] / cpb * 256
    rv.update({"peak": peak, "frac": rv["achieved"] / peak, "issue_cycles_per_block": cpb,
               "sum_of_isolated_issue_costs": VALU_MODEL["sum_of_issue_costs_fast"]})
sim = [r for r in rows if "sim_kernel" in r["Name"]][0]
with open(os.path.join("profiles", f"{tag}_summary.md"), "w") as f:
    f.write(f"# rocprofv3 summary `{tag}` -- `python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-ks`\n\n")
    f.write("Collected by tools/gpu_profile.sh on one MI355X: pass 1 `--kernel-trace --stats`, then one `--kernel-trace --pmc` pass per counter group.\n\n")
    f.write(f"Dominant kernel: `{sim['Name']}` -- {sim['Calls']} calls, average {float(sim['AverageNs'])/1e6:.3f} ms "
            f"({sim['Percentage']} % of GPU time); bench.py's own HIP-event average in the same run: {bench['roofline']['kernel_ms']:.3f} ms.\n\n")
    f.write(f"Dispatch: {meta}\n\n| counter (per launch, mean of {len(next(iter(pmc.values())))} launches) | value |\n|---|---|\n")
    for k, v in pmc.items():
        f.write(f"| {k} | {sum(v)/len(v):.6g} |\n")
    m = {k: sum(v) / len(v) for k, v in pmc.items()}
    dur = float(sim["AverageNs"]) * 1e-9
    if "GRBM_GUI_ACTIVE" in m and "SQ_INSTS_VALU" in m:
        clock = m["GRBM_GUI_ACTIVE"] / 8 / dur            # rocprofv3 sums the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back)
        useful_blocks = bench["em_steps_per_trial"] * bench["config"]["sets_per_gpu"] * bench["config"]["n_trials"] / 256.0
        f.write("\nDerived (per launch):\n\n")
        f.write(f"* effective clock = GRBM_GUI_ACTIVE / 8 / kernel time = {clock/1e9:.3f} GHz\n")
        f.write(f"* VALU wave-instructions per useful wave-block (64 lanes x 4 E-M steps) = {m['SQ_INSTS_VALU']/useful_blocks:.1f} "
                f"(step-loop body: see tools/isa_mix.py); SALU = {m['SQ_INSTS_SALU']/useful_blocks:.1f}\n")
        f.write(f"* VALU issue rate = {m['SQ_INSTS_VALU']/dur/1024/clock:.3f} wave-instructions per SIMD-cycle "
                f"= {1024*clock*dur/m['SQ_INSTS_VALU']:.2f} SIMD-cycles per VALU instruction (a full-rate VGPR-only op takes ~2.3; "
                f"tools/isa_mix.py gives the loop's mix average)\n")
        if "SQ_THREAD_CYCLES_VALU" in m and "SQ_ACTIVE_INST_VALU" in m:
            f.write(f"* exec-mask utilisation of VALU instructions = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) = "
                    f"{m['SQ_THREAD_CYCLES_VALU']/(64*m['SQ_ACTIVE_INST_VALU']):.3f}\n")
        if "SQ_WAVE_CYCLES" in m:
            f.write(f"* mean resident waves per SIMD = 4 x SQ_WAVE_CYCLES / (1024 x kernel cycles) = "
                    f"{4*m['SQ_WAVE_CYCLES']/(1024*clock*dur):.2f}\n")
        if "WRITE_SIZE" in m:
            f.write(f"* HBM traffic = (2 x FETCH_SIZE + WRITE_SIZE) KiB = {(2*m.get('FETCH_SIZE',0)+m['WRITE_SIZE'])*1024/1e9:.3f} GB "
                    f"vs algorithmic {bench['roofline']['algorithmic_bytes_per_launch']/1e9:.3f} GB\n")
    f.write("\nbench line of the traced run:\n\n```json\n" + json.dumps(bench) + "\n```\n")
with open(os.path.join("profiles", f"{tag}_pmc.json"), "w") as f:
    json.dump({"command": "python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-ks", "kernel": sim["Name"],
               "kernel_avg_ms_rocprof": float(sim["AverageNs"]) / 1e6, "kernel_avg_ms_bench_events": bench["roofline"]["kernel_ms"],
               "sets_per_gpu": bench["config"]["sets_per_gpu"], "n_trials": bench["config"]["n_trials"],
               "pmc_per_launch": {k: sum(v) / len(v) for k, v in pmc.items()}}, f, indent=1)
print(open(os.path.join("profiles", f"{tag}_summary.md")).read())

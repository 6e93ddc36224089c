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
sim = [r for r in rows if "sim_kernel" in r["Name"]][0]
with open(os.path.join("profiles", f"{tag}_summary.md"), "w") as f:
    f.write(f"# rocprofv3 summary `{tag}` -- `python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-ks`\n\n")
    f.write("Collected by tools/gpu_profile.sh on one MI355X: pass 1 `--kernel-trace --stats`, then one `--kernel-trace --pmc` pass per counter group.\n\n")
    f.write(f"Dominant kernel: `{sim['Name']}` -- {sim['Calls']} calls, average {float(sim['AverageNs'])/1e6:.3f} ms "
            f"({sim['Percentage']} % of GPU time); bench.py's own HIP-event average in the same run: {bench['roofline']['kernel_ms']:.3f} ms.\n\n")
    f.write(f"Dispatch: {meta}\n\n| counter (per launch, mean of {len(next(iter(pmc.values())))} launches) | value |\n|---|---|\n")
    for k, v in pmc.items():
        f.write(f"| {k} | {sum(v)/len(v):.6g} |\n")
    f.write("\nbench line of the traced run:\n\n```json\n" + json.dumps(bench) + "\n```\n")
with open(os.path.join("profiles", f"{tag}_pmc.json"), "w") as f:
    json.dump({"command": "python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-ks", "kernel": sim["Name"],
               "kernel_avg_ms_rocprof": float(sim["AverageNs"]) / 1e6, "kernel_avg_ms_bench_events": bench["roofline"]["kernel_ms"],
               "sets_per_gpu": bench["config"]["sets_per_gpu"], "n_trials": bench["config"]["n_trials"],
               "pmc_per_launch": {k: sum(v) / len(v) for k, v in pmc.items()}}, f, indent=1)
print(open(os.path.join("profiles", f"{tag}_summary.md")).read())

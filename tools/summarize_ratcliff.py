#!/usr/bin/env python3
"""Condense a tools/gpu_profile_ratcliff.sh output directory into profiles/<tag>_summary.md + <tag>_pmc.json.
Usage: python tools/summarize_ratcliff.py [tag=r6_ratcliff]"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r6_ratcliff"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
B, N = 1_000_000, 300
trace = list(csv.DictReader(open(os.path.join(src, "trace", "r_kernel_trace.csv"))))
durs = collections.defaultdict(list)
for r in trace:
    if "ratcliff_kernel" in r["Kernel_Name"]:
        durs[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in sorted(glob.glob(os.path.join(src, "pmc*", "r_counter_collection.csv"))):
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(fn)):
        if "ratcliff_kernel" in r["Kernel_Name"]:
            per[(r["Kernel_Name"], r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    for (name, _), d in per.items():
        for k, v in d.items():
            pmc[name][k].append(v)
sys.path.insert(0, ROOT)
from bayesflow_nddms_amd.build import source_hash  # noqa: E402
import re  # noqa: E402
m_hash = re.search(r"# library source hash ([0-9a-f]{64})", open(os.path.join(src, "run_trace.txt")).read())
out = {"command": open(os.path.join(src, "command.txt")).read().strip(), "sets": B, "n_trials": N,
       "source_hash": m_hash.group(1) if m_hash else None,          # of the library that was PROFILED (printed by the profiled program)
       "source_hash_of_the_tree_at_summary_time": source_hash(), "kernels": {}}
lines = [f"# rocprofv3 summary `{tag}` -- `{out['command']}` (nddm::ratcliff_kernel, {B} sets x {N} trials per launch)\n",
         "Collected by tools/gpu_profile_ratcliff.sh on one MI355X: pass 1 `--kernel-trace --stats`, then one `--kernel-trace --pmc` pass per counter group.\n",
         open(os.path.join(src, "run_trace.txt")).read().strip() and "bench-side event timing of the traced run:\n\n```\n" + open(os.path.join(src, "run_trace.txt")).read().strip() + "\n```\n"]
for name, d in durs.items():
    m = {k: sum(v) / len(v) for k, v in pmc[name].items()}
    mean_ms = sum(d) / len(d)
    out["kernels"][name] = {"launches": len(d), "mean_ms_rocprof": mean_ms, "pmc_per_launch": m}
    lines.append(f"\n## `{name}` -- {len(d)} launches, mean {mean_ms:.3f} ms = {B * N / mean_ms / 1e6:.2f} G trials/s\n")
    if m:
        lines.append("| counter (per launch) | value |\n|---|---|")
        lines += [f"| {k} | {v:.6g} |" for k, v in m.items()]
        dur = mean_ms * 1e-3
        if "GRBM_GUI_ACTIVE" in m and "SQ_INSTS_VALU" in m:
            clock = m["GRBM_GUI_ACTIVE"] / 8 / dur
            lines.append("\nDerived:\n")
            lines.append(f"* effective clock = GRBM_GUI_ACTIVE / 8 / kernel time = {clock / 1e9:.3f} GHz")
            lines.append(f"* VALU wave-instructions per trial = {m['SQ_INSTS_VALU'] * 64 / (B * N):.0f} lane-instructions issued per trial (64 lanes per wave-instruction, all lanes counted)")
            lines.append(f"* VALU issue rate = {m['SQ_INSTS_VALU'] / dur / 1024 / clock:.3f} wave-instructions per SIMD-cycle = {1024 * clock * dur / m['SQ_INSTS_VALU']:.2f} SIMD-cycles per VALU instruction")
            if "SQ_THREAD_CYCLES_VALU" in m and "SQ_ACTIVE_INST_VALU" in m:
                lines.append(f"* exec-mask utilisation of VALU instructions (lane efficiency of the flattened loop) = {m['SQ_THREAD_CYCLES_VALU'] / (64 * m['SQ_ACTIVE_INST_VALU']):.3f}")
            if "SQ_WAVE_CYCLES" in m:
                lines.append(f"* mean resident waves per SIMD = {4 * m['SQ_WAVE_CYCLES'] / (1024 * clock * dur):.2f}")
        if "WRITE_SIZE" in m:
            alg = B * N * 8 + B * 64
            lines.append(f"* HBM traffic = (2 x FETCH_SIZE + WRITE_SIZE) KiB = {(2 * m.get('FETCH_SIZE', 0) + m['WRITE_SIZE']) * 1024 / 1e9:.3f} GB vs algorithmic {alg / 1e9:.3f} GB")
open(os.path.join(ROOT, "profiles", f"{tag}_summary.md"), "w").write("\n".join(l for l in lines if l) + "\n")
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc.json"), "w"), indent=1)
print(open(os.path.join(ROOT, "profiles", f"{tag}_summary.md")).read()[:3000])

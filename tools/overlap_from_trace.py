#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace of `bench.py --dist --gather ...` (one rank): for every timed step, when the simulator
kernel ran (the simulate stream), when the all-gather of that step's output ran (RCCL at one rank is a device copy kernel, on
the communication stream) and what the simulate stream was doing meanwhile -- the NEXT step's kernels.

usage: python tools/overlap_from_trace.py <dir with *_kernel_trace.csv> > profiles/rN_dist_overlap.md"""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    path = glob.glob(os.path.join(d, "*kernel_trace.csv"))[0]
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    sims = [r for r in rows if "sim_kernel" in r["Kernel_Name"]]
    sim_stream = sims[0]["Stream_Id"]
    t0 = sims[0]["s"]
    on_sim_stream = sorted((r for r in rows if r["Stream_Id"] == sim_stream), key=lambda r: r["s"])
    copies = sorted((r for r in rows if r["Stream_Id"] != sim_stream and "copyBuffer" in r["Kernel_Name"]
                     and r["e"] - r["s"] > 50_000), key=lambda r: r["s"])
    ms = lambda t: (t - t0) / 1e6
    print("# all-gather on the communication stream vs the simulate stream (rocprofv3 --kernel-trace, one rank, RCCL)\n")
    print(f"`{os.path.basename(path)}`: simulate stream = rocprof stream {sim_stream}; the one-rank all-gather is a device copy kernel "
          f"(`__amd_rocclr_copyBuffer`) on stream {copies[0]['Stream_Id'] if copies else '?'}.  Times in ms from the first simulator kernel.\n")
    print("| step | simulator kernel | all-gather of this step's output | simulate-stream kernels running during the gather |")
    print("|---|---|---|---|")
    for i, s in enumerate(sims):
        g = next((c for c in copies if c["s"] >= s["e"] - 1000), None)
        if g is None or (i + 1 < len(sims) and g["s"] > sims[i + 1]["e"]):
            print(f"| {i} | {ms(s['s']):.3f} – {ms(s['e']):.3f} | (none) | |")
            continue
        during = [r for r in on_sim_stream if r["s"] < g["e"] and r["e"] > g["s"]]
        names = ", ".join(sorted({r["Kernel_Name"].split("(")[0].replace("nddm::", "").replace("void ", "")[:40] for r in during})) or "—"
        ov = sum(min(r["e"], g["e"]) - max(r["s"], g["s"]) for r in during)
        print(f"| {i} | {ms(s['s']):.3f} – {ms(s['e']):.3f} | {ms(g['s']):.3f} – {ms(g['e']):.3f} (starts {((g['s'] - s['e']) / 1e3):.0f} µs after it) | "
              f"{names} ({ov / 1e3:.0f} µs of the gather's {((g['e'] - g['s']) / 1e3):.0f} µs) |")
    print("\nThe gather of step i is issued on its own stream and waits only for step i's simulate; step i+1's kernels (hand-out "
          "records, then the simulator's persistent grid) are already running beside it.  With one rank the gather is a 0.3 ms copy; "
          "what a 30 ms collective does beside the persistent grid is measured by `tools/overlap_probe.py` (`profiles/r3_overlap_probe.txt`).")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Is the graph-replayed training iteration GPU-bound?  From a rocprofv3 kernel trace of `bench.py --train --train-mode graph`:
takes the steady-state replays (the longest run of regularly spaced optimizer updates, at most `--iters` iterations of it) and reports kernels per iteration, the GPU time they sum to, the wall span they cover and the busy
fraction; plus the ten kernel names with the largest share.

usage: python tools/train_trace_summary.py <dir with *_kernel_trace.csv> [--iters 150]"""
import collections
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 150
    path = glob.glob(os.path.join(d, "*kernel_trace.csv"))[0]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Stream_Id"]))
    rows.sort()
    # one marker kernel per training iteration: the optimizer's update (older builds: the simulator launch); the steady-state block
    # is the longest run of markers with a regular spacing (captures, warm-up passes and the bench's other legs break the rhythm)
    marker = "adam_kernel" if any("adam_kernel" in r[2] for r in rows) else "sim_kernel"
    sims = [i for i, r in enumerate(rows) if marker in r[2]]
    gaps = [rows[sims[j + 1]][0] - rows[sims[j]][0] for j in range(len(sims) - 1)]
    train_gaps = [g for g in gaps if g > 200_000]
    med = sorted(train_gaps)[len(train_gaps) // 2]
    best, cur, start = (0, 0), 0, 0
    for j, g in enumerate(gaps):
        if 0.5 * med < g < 2 * med:
            cur += 1
            if cur > best[0]:
                best = (cur, start)
        else:
            cur, start = 0, j + 1
    n, s0 = best
    n = min(n, iters)
    first, last = sims[s0 + best[0] - n], sims[s0 + best[0]]
    seg = rows[first:last]
    busy = sum(e - s for s, e, _, _ in seg)
    span = seg[-1][1] - seg[0][0]
    # union of intervals (kernels on different streams may overlap)
    cov, end = 0, 0
    for s, e, _, _ in seg:
        if e > end:
            cov += e - max(s, end)
            end = e
    per = collections.Counter()
    for s, e, name, _ in seg:
        per[name.split("(")[0][:70]] += e - s
    print(f"# graph-replayed training iterations in `{os.path.basename(path)}`: {n} consecutive steady-state iterations\n")
    print(f"* kernels per iteration: {len(seg) / n:.0f}")
    print(f"* wall time per iteration (first kernel start to last kernel end / n): {span / n / 1e6:.3f} ms")
    print(f"* GPU time per iteration, sum of kernel durations: {busy / n / 1e6:.3f} ms; union of kernel intervals: {cov / n / 1e6:.3f} ms")
    print(f"* **GPU busy fraction = {cov / span:.3f}** (the remainder is the gap between consecutive 3-4 microsecond kernels)")
    print(f"* mean kernel duration {busy / len(seg) / 1e3:.2f} us\n")
    print("| kernel | share of GPU time |\n|---|---|")
    for name, t in per.most_common(10):
        print(f"| `{name}` | {t / busy:.3f} |")


if __name__ == "__main__":
    main()

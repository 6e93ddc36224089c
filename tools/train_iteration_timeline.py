#!/usr/bin/env python3
"""One steady-state training iteration as a timeline: from a rocprofv3 kernel trace of `bench.py --train --train-mode graph ...`
prints, for the iteration in the middle of the longest regular run of optimizer updates, every kernel with its start offset,
duration, hardware queue and the idle gap in front of it on its queue -- what sits on the critical path of the pipelined feed
(simulate / all-gather of batch i + 1 beside the training graph of batch i) shows as kernels that START late.

usage: python tools/train_iteration_timeline.py <dir with *_kernel_trace.csv> [--skip N]"""
import csv
import glob
import os
import re
import sys


def main():
    d = sys.argv[1]
    skip = int(sys.argv[sys.argv.index("--skip") + 1]) if "--skip" in sys.argv else 0
    path = glob.glob(os.path.join(d, "*kernel_trace.csv"))[0]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
    gaps = [rows[marks[j + 1]][0] - rows[marks[j]][0] for j in range(len(marks) - 1)]
    med = sorted(g for g in gaps if g > 100_000)[len(gaps) // 2]
    runs, cur, start = [], 0, 0                                   # every regular run of optimizer updates: (length, first marker)
    for j, g in enumerate(gaps + [10 ** 12]):
        if 0.6 * med < g < 1.6 * med:
            cur += 1
        else:
            if cur >= 40:
                runs.append((cur, start))
            cur, start = 0, j + 1
    if "--all" not in sys.argv:
        runs = [max(runs)]
    for n, s0 in runs:
        timeline(rows, marks, n, s0, skip, os.path.basename(path), med)


def timeline(rows, marks, n, s0, skip, label, med):
    print(f"# {label}: a regular run of {n} iterations (median spacing of the optimizer updates over the trace {med / 1e3:.1f} us)")
    j = s0 + n // 2 + skip
    a, b = marks[j] + 1, marks[j + 1] + 1                       # (after one update) ... (through the next update)
    t0 = rows[a][0]
    last_end = {}
    for s, e, _, q, _ in rows[:a]:
        last_end[q] = max(last_end.get(q, 0), e)
    print(f"# iteration window {(rows[b - 1][1] - t0) / 1e3:.1f} us; columns: start offset (us), duration (us), hardware queue, stream, gap before it on its queue (us), kernel")
    for s, e, name, q, st in rows[a:b]:
        name = re.sub(r"at::native::|\(anonymous namespace\)::|void ", "", name)
        gap = (s - last_end[q]) / 1e3 if q in last_end else float("nan")
        last_end[q] = max(last_end.get(q, 0), e)
        print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f}  q{q:>3s} s{st:>3s} {gap:8.1f}  {name[:100]}")
    print()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""rocprofv3 kernel trace of the DEFAULT bench command (legs included) -> per BASELINE config, the simulator kernel's launch durations
by the profiler next to the HIP-event figure the bench line carries for that leg, and the training legs' kernels per iteration.

usage: python tools/legs_trace_summary.py <dir with bench_kernel_trace.csv> <the traced run's bench JSON> > profiles/r5_legs_trace.md"""
import collections
import csv
import json
import os
import sys


def main():
    d, bench = sys.argv[1], json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    trace = list(csv.DictReader(open(os.path.join(d, "bench_kernel_trace.csv"))))
    by_name = collections.defaultdict(list)
    for r in trace:
        by_name[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    print(f"# rocprofv3 kernel trace of the default bench command with its side legs -- `{open(os.path.join(os.path.dirname(d.rstrip('/')), 'command_legs.txt')).read().strip() if os.path.exists(os.path.join(os.path.dirname(d.rstrip('/')), 'command_legs.txt')) else 'python3 bench.py ...'}`\n")
    print("Launch durations of each BASELINE config's simulator kernel: the profiler's (launches within 15 % of the bench's figure: the "
          "workload launches; the others are the lockstep-ceiling and KS launches of the same kernel) beside bench.py's HIP events.\n")
    print("| config | kernel (rocprofv3 name) | workload launches | rocprofv3 mean ms | bench.py HIP events ms | trials/s in the line |\n|---|---|---|---|---|---|")
    legs = bench.get("legs", {})
    g = lambda k, f: legs.get(k, {}).get(f)
    # (label, substring of the kernel's name, second substring or None, the leg's HIP-event ms, its trials/s).  The template arguments
    # after the model are FAST, CAP4, BRIDGE, SMALL, PACKED, VKEYS, CODES, F64.
    rows = [("configs[1] basic_ddm_dc (headline)", "sim_kernel<0, true,", "false, false>", bench["roofline"]["kernel_ms"], bench["value"]),
            ("configs[3] single_trial + fused summaries", "sim_kernel<1, true,", None, g("single", "kernel_ms"), g("single", "value")),
            ("configs[2] alpha_not_scaled + bridge", "sim_kernel<3,", "true, true, true,", g("alpha_ns_bridge", "kernel_ms"), g("alpha_ns_bridge", "value")),
            ("configs[2] with the reference's own generator (nddm_simulratcliff)", "ratcliff_kernel<true>", None, g("alpha_ns_exact_sampler", "kernel_ms"), g("alpha_ns_exact_sampler", "value")),
            ("basic, the reference's default dt=.01 / 400", "sim_kernel<0, true,", "false, false>", g("basic_dt01", "kernel_ms"), g("basic_dt01", "value")),
            ("basic with NDDM_GAUSS_EXACT (the bit-pinned transform)", "sim_kernel<0, false,", "false, false>", g("exact_gauss", "kernel_ms"), g("exact_gauss", "value")),
            ("basic with NDDM_STATE_F64, exact transform", "sim_kernel<0, false,", "false, true>", g("state_f64", "kernel_ms"), g("state_f64", "value")),
            ("basic with NDDM_STATE_F64, fast transform", "sim_kernel<0, true,", "false, true>", (legs.get("state_f64", {}).get("fast_transform") or {}).get("kernel_ms"),
             (legs.get("state_f64", {}).get("fast_transform") or {}).get("value"))]
    for label, key, extra, km, val in rows:
        if km is None:
            continue
        for name, durs in by_name.items():
            if key in name and (extra is None or extra in name):
                near = [x for x in durs if abs(x - km) / km < 0.15]
                if near:
                    print(f"| {label} | `{name[:78]}` | {len(near)} of {len(durs)} | {sum(near) / len(near):.3f} | {km:.3f} | {val:.3e} |")
    tr = legs.get("train")
    if tr:
        print("\nTraining legs (configs[4]): kernels of `libnddm_train.so` and the simulator's small-launch kernels in the trace:\n")
        print("| kernel | calls | mean us |\n|---|---|---|")
        for name, durs in sorted(by_name.items(), key=lambda kv: -sum(kv[1])):
            if any(k in name for k in ("flow_", "mlp", "adam", "deepset", "reduce", "nll_kernel", "stage", "set2")) and len(durs) > 50:
                print(f"| `{name[:90]}` | {len(durs)} | {1e3 * sum(durs) / len(durs):.1f} |")
        for form in ("one_rank", "gather_rccl_world1"):
            for tag, v in tr[form].items():
                print(f"\n* {form} {tag}: {v['iterations_per_s']:.0f} it/s ({v['ms_per_iteration']:.4f} ms per iteration) in the traced run")
    print("\n## bench line of the traced run\n\n```json\n" + json.dumps(bench) + "\n```")


if __name__ == "__main__":
    main()

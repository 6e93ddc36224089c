#!/usr/bin/env python3
"""The flattened loop of nddm::ratcliff_kernel<fast> in the SHIPPED library, priced like the simulator's step loop (tools/isa_mix.py:
the code object's disassembly x the per-instruction issue costs of tools/ubench_valu): VALU instructions of one trip, their mean issue
cost, and -- against the measured SIMD-cycles per VALU instruction of profiles/<tag>_ratcliff_pmc.json -- how busy the VALU pipe is.
The trip counts are data-dependent, so this is a statement about the instruction stream, not a time model.  CPU only.
usage: python tools/ratcliff_isa_mix.py [--ubench profiles/r6_ubench_valu.txt] [--pmc profiles/r6_ratcliff_pmc.json]
                                         [--json-into profiles/r6_issue_model.json]   (adds kernels.ratcliff_fast to the tracked model file
                                          bench.py reads, if that file is of the same code object; tools/refresh_issue_model.sh does)"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_mix  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SALU_CYCLES = 4.23          # SIMD-cycles per SALU instruction at 8 waves per SIMD, tools/ubench_salu.hip (profiles/r2_ubench_salu.txt)


def _tree_hash():
    sys.path.insert(0, ROOT)
    from bayesflow_nddms_amd.build import source_hash
    return source_hash()


def loop_of(txt, sym_re):
    lines = txt.splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <" + sym_re + r">:", l))
    insts = []
    for l in lines[start + 1:]:
        if re.match(r"^[0-9a-f]+ <", l):
            break
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*// ([0-9A-F]+):", l)
        if not m:
            continue
        tgt = re.search(r"<[^>]*\+0x([0-9a-f]+)>", l)
        insts.append((int(m.group(3), 16), m.group(1), m.group(2), None if not tgt else int(tgt.group(1), 16)))
    base = insts[0][0]
    insts = [(a - base, op, args, t) for a, op, args, t in insts]
    # the flattened loop: the backward branches whose region holds both Philox bodies (drift FIFO + uniform refill: >= 32 v_mad_u64_u32)
    # and no more of them than the smallest such region -- the loop has several latches (its `continue`s and the compiler's block
    # placement), the group loop around it holds further blocks (the set's external datum) -- from the first header to the last latch
    cands = []
    for a, op, _, t in insts:
        if op.startswith(("s_cbranch", "s_branch")) and t is not None and t < a:
            n_mad = sum(1 for x in insts if t <= x[0] <= a and x[1].startswith("v_mad_u64_u32"))
            if n_mad >= 32:
                cands.append((n_mad, t, a))
    n_min = min(c[0] for c in cands)
    lo = min(t for n, t, a in cands if n == n_min)
    hi = max(a for n, t, a in cands if n == n_min)
    return [x for x in insts if lo <= x[0] <= hi]


def main():
    args = sys.argv[1:]
    ub = os.path.join(ROOT, "profiles", "r6_ubench_valu.txt")
    pmc = os.path.join(ROOT, "profiles", "r6_ratcliff_pmc.json")
    if "--ubench" in args:
        ub = args[args.index("--ubench") + 1]
    if "--pmc" in args:
        pmc = args[args.index("--pmc") + 1]
    cost, sgpr, src = isa_mix.load_ubench(ub)
    txt, digest = isa_mix.disassemble()
    body = loop_of(txt, r"_ZN4nddm15ratcliff_kernelILb1EEEvNS_7RatArgsE")
    t = isa_mix.tally(body, cost, sgpr)
    # the drift FIFO's section (the first Philox body + its Box-Muller pair) runs once per 64 hand-outs, not once per trip
    print(f"# nddm::ratcliff_kernel<fast>, code object {digest}; issue costs from {src}")
    print(f"flattened loop, static: {t['valu']} VALU / {t['salu']} SALU / {t['lds']} LDS instructions; "
          f"sum of VALU issue costs {t['cycles_per_block']:.0f} SIMD-cycles = {t['cycles_per_block'] / t['valu']:.2f} per VALU instruction")
    print("  (every section of a trip -- hand-out, uniform refill, sphere set-up, attempt, acceptance -- runs once per trip; the drift FIFO's "
          "78 instructions once per 64 hand-outs)")
    if "--json-into" in args:
        path = args[args.index("--json-into") + 1]
        d = json.load(open(path))
        if d.get("library_sha256_16") != digest:
            sys.exit(f"{path} is of code object {d.get('library_sha256_16')}, the library is {digest}: run tools/refresh_issue_model.sh")
        d["kernels"]["ratcliff_fast"] = {"valu": t["valu"], "salu": t["salu"], "lds": t["lds"], "vmem": t["vmem"],
                                         "cycles_per_trip": t["cycles_per_block"], "salu_cycles_per_trip": t["salu"] * SALU_CYCLES,
                                         "unit_note": "one trip of the flattened loop (one rejection attempt of every lane that holds a trial), static; "
                                                      "the drift FIFO's instructions included although they run once per 64 hand-outs",
                                         "mix": t["mix"]}
        json.dump(d, open(path, "w"), indent=1)
        print("added kernels.ratcliff_fast to", path)
    for r in t["mix"][:12]:
        print(f"  {r['op']:28s} x{r['n']:3d}  {r['cycles_each']:.2f} cycles each{'' if r['costed'] else '  (default cost)'}")
    if os.path.exists(pmc):
        d = json.load(open(pmc))
        c = next(v["pmc_per_launch"] for k, v in d["kernels"].items() if "ratcliff_kernel<true>" in k)
        meas = 1024.0 * (c["GRBM_GUI_ACTIVE"] / 8.0) / c["SQ_INSTS_VALU"]
        mean = t["cycles_per_block"] / t["valu"]
        stale = "" if d.get("code_object", digest) == digest and d.get("source_hash") == _tree_hash() else "  [PMC file is from another build]"
        print(f"measured ({os.path.basename(pmc)}){stale}: one VALU instruction per {meas:.2f} SIMD-cycles = {meas * t['valu']:.0f} SIMD-cycles per trip "
              f"-> the VALU pipe is {mean / meas:.2f} busy with this mix; exec-mask utilisation "
              f"{c['SQ_THREAD_CYCLES_VALU'] / (64.0 * c['SQ_ACTIVE_INST_VALU']):.3f} -> "
              f"{mean / meas * c['SQ_THREAD_CYCLES_VALU'] / (64.0 * c['SQ_ACTIVE_INST_VALU']):.2f} of the pipe's lane-cycles do the sampler's arithmetic")
        print(f"the trip's {t['salu']} SALU instructions (exec-mask bookkeeping of the nested divergence) at {SALU_CYCLES} SIMD-cycles each "
              f"(profiles/r2_ubench_salu.txt: a CU's scalar unit serves its four SIMDs in turn) = {t['salu'] * SALU_CYCLES:.0f} SIMD-cycles: the "
              f"scalar unit is {t['salu'] * SALU_CYCLES / (meas * t['valu']):.2f} busy")


if __name__ == "__main__":
    main()

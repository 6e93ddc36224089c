#!/usr/bin/env python3
"""Compiler resource table of every kernel in nddm_kernels.hip -- and, with --train, of the amortizer's kernels
(train_kernels.hip, train_deepset.hip, train_update.hip) -- from hipcc -Rpass-analysis=kernel-resource-usage:
VGPRs, SGPRs, scratch, LDS, and the waves/SIMD the register files allow on gfx950.

The SGPR limit is the one measured with tools/ubench_residency.hip (profiles/r1_ubench_residency.txt): 800 SGPRs per
SIMD, a wave is charged its SGPRs + 22 rounded up to 16 -- the occupancy the compiler prints does not know it.

Usage: python tools/resource_table.py [--md] [--train]     (prints a table; --md = markdown; --train = the training kernels)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "bayesflow_nddms_amd", "csrc", "nddm_kernels.hip")
MODELS = {0: "basic", 1: "single", 2: "single_alt", 3: "alpha_ns", 4: "explicit"}


def pretty(name):
    m = re.match(r"_ZN4nddm10sim_kernelILi(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)E", name)
    if m:
        mod, fast, cap4, bridge, small, packed, vkeys, codes, f64 = (int(x) for x in m.groups())
        return (f"sim_kernel<{MODELS[mod]}, {'fast' if fast else 'exact'}, cap4={cap4}, bridge={bridge}, small={small}, "
                f"packed={packed}, vkeys={vkeys}{', codes' if codes else ''}{', state_f64' if f64 else ''}>")
    m = re.match(r"_ZN(?:4nddm|10nddm_train|12nddm_deepset|11nddm_update)(\d+)", name)
    if m:
        n = int(m.group(1))
        base, rest = name[len(m.group(0)):len(m.group(0)) + n], name[len(m.group(0)) + n:]
        t = re.match(r"ILb(\d)E", rest)                       # (one boolean template argument: the DeepSet kernels' BIG)
        return base + (f"<{'true' if t.group(1) == '1' else 'false'}>" if t else "")
    return name


def waves_by_sgpr(s):
    charged = ((s + 22 + 15) // 16) * 16
    return min(8, 800 // charged)


def waves_by_vgpr(v):
    return min(8, 512 // (((v + 7) // 8) * 8))


TRAIN_SRCS = [os.path.join(ROOT, "bayesflow_nddms_amd", "csrc", f) for f in ("train_kernels.hip", "train_deepset.hip", "train_update.hip")]


def collect(train=False):
    """One dict per kernel.  train=True: the amortizer's kernels (compiled as build.build_train does: -O3, contraction on)."""
    text = ""
    with tempfile.TemporaryDirectory() as td:
        sys.path.insert(0, ROOT)
        from bayesflow_nddms_amd.build import _hipcc
        for src in (TRAIN_SRCS if train else [SRC]):
            flags = ["-O3", "--offload-arch=gfx950", "-std=c++17"] + ([] if train else ["-ffp-contract=off"])
            r = subprocess.run([_hipcc()] + flags + ["-c", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o",
                                os.path.join(td, "k.o"), src], capture_output=True, text=True, check=True)
            text += r.stderr
    rows, cur = [], None
    for line in text.splitlines():
        m = re.search(r"remark: (?:\S+ )?\s*(Function Name|Name): (\S+)", line)
        if m:
            cur = {"name": m.group(2)}
            rows.append(cur)
            continue
        m = re.search(r"remark: (?:\S+ )?\s*([A-Za-z \[\]/]+): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return rows


def main():
    md = "--md" in sys.argv
    rows = collect(train="--train" in sys.argv)
    hdr = ["kernel", "VGPRs", "SGPRs", "scratch B/lane", "static LDS B", "waves/SIMD (VGPR)", "waves/SIMD (SGPR, measured rule)"]
    out = []
    for r in rows:
        v, s = r.get("VGPRs", 0), r.get("TotalSGPRs", r.get("SGPRs", 0))
        out.append([pretty(r["name"]), v, s, r.get("ScratchSize [bytes/lane]", r.get("ScratchSize", 0)),
                    r.get("LDS Size [bytes/block]", r.get("LDS Size", 0)), waves_by_vgpr(v), waves_by_sgpr(s)])
    if md:
        print("| " + " | ".join(hdr) + " |\n|" + "---|" * len(hdr))
        for o in out:
            print("| " + " | ".join(str(x) for x in o) + " |")
    else:
        for o in out:
            print(f"{o[0]:79s} VGPR {o[1]:3d}  SGPR {o[2]:3d}  scratch {o[3]:3d}  LDS {o[4]:5d}  waves/SIMD vgpr {o[5]} sgpr {o[6]}")


if __name__ == "__main__":
    main()

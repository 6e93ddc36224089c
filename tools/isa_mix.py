#!/usr/bin/env python3
"""First-principles VALU issue model of the step loop, from the SHIPPED library.

Carves the gfx950 code object out of bayesflow_nddms_amd/libnddm_hip.so (the clang offload bundle inside the ELF),
disassembles it with llvm-objdump, finds the step loop of a sim_kernel instantiation -- the smallest backward-branch
region holding the 16 Philox multiplies -- tallies its instructions and weights them with the per-instruction issue
costs measured by tools/ubench_valu on the MI355X (profiles/*_ubench_valu.txt, 8 waves/SIMD column).  The result is SIMD
cycles per wave64 Philox block (= 4 Euler-Maruyama steps x 64 lanes): an ISA-level ceiling that does not come from timing
the kernel itself.  bench.py reads profiles/<tag>_issue_model.json and reports `roofline_valu.frac_vs_issue_model`.

Usage: python tools/isa_mix.py [--json profiles/r2_issue_model.json] [--ubench profiles/r2_ubench_valu.txt] [kernel ...]
       kernel = basic | single | single_alt | alpha_ns | alpha_ns_bridge | explicit (fast transform), or *_exact
"""
import collections
import glob
import hashlib
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "bayesflow_nddms_amd", "libnddm_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
# template arguments <MODEL, FAST, CAP4, BRIDGE, SMALL, PACKED, VKEYS> of the instantiations the named workloads launch
# (max_steps a multiple of 4 and below 2^14, tiles of <= 512 trials, a full grid)
KERNELS = {"basic": (0, 1, 1, 0, 1, 0, 0), "single": (1, 1, 1, 0, 1, 0, 0), "single_alt": (2, 1, 1, 0, 1, 0, 0),
           "alpha_ns": (3, 1, 1, 0, 1, 0, 0), "alpha_ns_bridge": (3, 1, 1, 1, 0, 0, 0), "explicit": (4, 1, 1, 0, 1, 0, 0)}
KERNELS.update({k + "_exact": (m, 0, c, b, sm, pk, vk) for k, (m, f, c, b, sm, pk, vk) in list(KERNELS.items())})
KERNELS.update({k + "_packed": (m, f, c, b, sm, 1, vk) for k, (m, f, c, b, sm, pk, vk) in list(KERNELS.items()) if not b})  # NDDM_GAUSS_PACKED
KERNELS["basic_vkeys"] = (0, 1, 1, 0, 1, 0, 1)               # the small-launch variant: round keys in VGPRs
KERNELS = {k: v + (0,) for k, v in KERNELS.items()}          # (..., F64 = false)
for _k in ("basic", "single", "basic_exact", "single_exact"):
    KERNELS[_k + "_f64"] = KERNELS[_k][:7] + (1,)            # NDDM_STATE_F64: the reference's float64 recurrence

# fallback costs (profiles/r2_ubench_valu.txt, 8 waves/SIMD): cycles per wave64 instruction per SIMD
COST = {"v_mad_u64_u32": 4.61, "v_xor_b32": 2.34, "v_cvt_f32_u32": 4.13, "v_cvt_f32_i32": 4.13, "v_log_f32": 8.21,
        "v_sqrt_f32": 8.15, "v_sin_f32": 8.11, "v_cos_f32": 8.23, "v_exp_f32": 8.21, "v_rcp_f32": 8.21,
        "v_fma_f32": 2.22, "v_fmamk_f32": 2.31, "v_fmac_f32": 2.34, "v_fmaak_f32": 2.31, "v_add_f32": 2.27,
        "v_sub_f32": 2.27, "v_mul_f32": 2.27, "v_add_u32": 2.35, "v_sub_u32": 2.35, "v_subrev_u32": 2.35,
        "v_lshrrev_b32": 2.34, "v_lshlrev_b32": 2.34, "v_and_b32": 2.34, "v_or_b32": 2.34, "v_mov_b32": 2.34,
        "v_cmp": 4.05, "v_mul_lo_u32": 4.47, "v_mul_hi_u32": 4.22, "v_bfe_u32": 4.19, "v_and_or_b32": 4.19,
        "v_cvt_i32_f32": 4.13, "v_lshl_add_u32": 4.19, "v_addc_co_u32": 4.22, "v_max_f32": 2.27, "v_min_f32": 2.27,
        "v_med3_f32": 2.22, "v_add3_u32": 4.49, "v_alignbit_b32": 4.19, "v_cndmask_b32": 4.23, "v_pk_fma_f32": 4.16,
        "v_pk_mul_f32": 4.23, "v_pk_add_f32": 4.21, "v_bitop3_b32": 2.31, "v_add_co_u32": 4.22, "v_lshl_or_b32": 4.19}
SGPR_OPERAND_COST = 4.16           # a full-rate VOP2 op that reads an SGPR / VCC operand ("v_xor_b32 (sgpr)" row)
FULL_RATE = {"v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_add_f32", "v_sub_f32",
             "v_mul_f32", "v_lshrrev_b32", "v_lshlrev_b32", "v_mov_b32", "v_fmac_f32", "v_max_f32", "v_min_f32"}
DEFAULT = 2.4
# (three-operand instructions are costed from the rows measured with three DIFFERENT source registers, the form the kernels
# use: naming one register twice costs v_bitop3_b32 / v_fma_f32 ~1.5 cycles more -- the rows without a suffix)
UBENCH_ROWS = {"v_fma_f32 (3 regs)": ["v_fma_f32", "v_med3_f32"], "v_fmamk_f32": ["v_fmamk_f32", "v_fmaak_f32"], "v_add_f32": ["v_add_f32", "v_sub_f32", "v_max_f32", "v_min_f32"],
               "v_mul_f32": ["v_mul_f32"], "v_xor_b32": ["v_xor_b32", "v_and_b32", "v_or_b32", "v_lshrrev_b32", "v_lshlrev_b32", "v_mov_b32"],
               "v_add_u32": ["v_add_u32", "v_sub_u32", "v_subrev_u32"], "v_mul_lo_u32": ["v_mul_lo_u32"], "v_mul_hi_u32": ["v_mul_hi_u32"],
               "v_log_f32": ["v_log_f32", "v_exp_f32"], "v_sqrt_f32": ["v_sqrt_f32"],
               "v_sin_f32": ["v_sin_f32"], "v_cos_f32": ["v_cos_f32"], "v_rcp_f32": ["v_rcp_f32"],
               "v_cvt_f32_u32": ["v_cvt_f32_u32", "v_cvt_f32_i32", "v_cvt_i32_f32"], "v_cmp_lt_f32": ["v_cmp"],
               "v_add3_u32": ["v_add3_u32"], "v_alignbit_b32": ["v_alignbit_b32", "v_bfe_u32", "v_and_or_b32", "v_lshl_add_u32", "v_lshl_or_b32"],
               "v_cndmask_e64 (s)": ["v_cndmask_b32"], "v_pk_fma_f32": ["v_pk_fma_f32"], "v_pk_mul_f32": ["v_pk_mul_f32"],
               "v_pk_add_f32": ["v_pk_add_f32"], "v_bitop3_b32 (3 regs)": ["v_bitop3_b32"], "v_fmac_f32": ["v_fmac_f32"],
               "v_mad_u64_u32 (sgpr)": ["v_mad_u64_u32"],
               "v_add_co_u32_e64": ["v_add_co_u32", "v_addc_co_u32"],
               "v_add_f64": ["v_add_f64"], "v_mul_f64": ["v_mul_f64"], "v_fma_f64 (3 regs)": ["v_fma_f64"], "v_cvt_f64_f32": ["v_cvt_f64_f32"], "v_cmp_lt_f64": ["v_cmp_f64"]}


def load_ubench(path):
    """Per-instruction issue costs from a tools/ubench_valu run (last numeric column = 8 waves/SIMD)."""
    cost, sgpr = dict(COST), SGPR_OPERAND_COST
    if not path or not os.path.exists(path):
        return cost, sgpr, None
    for line in open(path):
        m = re.match(r"^(.+?)\s{2,}([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+clock", line)
        if not m:
            continue
        name, c8 = m.group(1).strip(), float(m.group(5))
        if name == "v_xor_b32 (sgpr)":
            sgpr = c8
        for op in UBENCH_ROWS.get(name, []):
            cost[op] = c8
    return cost, sgpr, os.path.basename(path)


def text_section(elf):
    """Bytes of the .text section of an ELF64 object (the kernels' machine code: independent of build paths, which the
    rest of the file is not -- hipcc derives a compile-unit id from the source path)."""
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
    sec = lambda i: struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize)
    stroff = sec(shstrndx)[4]
    for i in range(shnum):
        name_off, _, _, _, off, size = sec(i)[:6]
        end = elf.index(b"\0", stroff + name_off)
        if elf[stroff + name_off:end] == b".text":
            return elf[off:off + size]
    raise RuntimeError("no .text section")


def code_object(so_path):
    """The gfx950 code object inside the library's clang offload bundle."""
    data = open(so_path, "rb").read()
    i = data.find(b"__CLANG_OFFLOAD_BUNDLE__")
    if i < 0:
        raise RuntimeError("no offload bundle in " + so_path)
    n = struct.unpack_from("<Q", data, i + 24)[0]
    off = i + 32
    for _ in range(n):
        o, s, tl = struct.unpack_from("<QQQ", data, off)
        off += 24
        triple = data[off:off + tl].decode()
        off += tl
        if "gfx950" in triple:
            co = data[i + o:i + o + s]
            return co, hashlib.sha256(text_section(co)).hexdigest()[:16]     # the machine code identifies the library
    raise RuntimeError("no gfx950 code object in " + so_path)


def disassemble(so_path=SO):
    co, digest = code_object(so_path)
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "k.co")
        open(p, "wb").write(co)
        txt = subprocess.run([OBJDUMP, "-d", p], capture_output=True, text=True, check=True).stdout
    return txt, digest


def kernel_insts(txt, targs):
    """[(address, mnemonic, operand text, branch target or None)] of one sim_kernel instantiation."""
    sym = "_ZN4nddm10sim_kernelILi%dELb%dELb%dELb%dELb%dELb%dELb%dELb0ELb%dEEEvNS_7SimArgsE" % targs      # (..., CODES = false, F64)
    lines = txt.splitlines()
    start = next(i for i, l in enumerate(lines) if l.endswith(f"<{sym}>:"))
    insts = []
    for l in lines[start + 1:]:
        if re.match(r"^[0-9a-f]+ <", l):
            break
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*// ([0-9A-F]+):", l)
        if not m:
            continue
        tgt = re.search(r"<" + re.escape(sym) + r"\+0x([0-9a-f]+)>", l)
        insts.append((int(m.group(3), 16), m.group(1), m.group(2), None if not tgt else int(tgt.group(1), 16)))
    base = insts[0][0]
    return [(a - base, op, args, t) for a, op, args, t in insts]


def step_loop(insts):
    """Smallest backward-branch region with >= 16 v_mad_u64_u32 and two Box-Muller pairs (2 v_sqrt_f32): the step loop
    (the rejection loop of the per-trial latent also holds a Philox block, but one pair)."""
    best = None
    for a, op, args, t in insts:
        if op.startswith(("s_cbranch", "s_branch")) and t is not None and t <= a:
            body = [x for x in insts if t <= x[0] <= a]
            if (sum(x[1].startswith("v_mad_u64_u32") for x in body) >= 16 and sum(x[1].startswith("v_sqrt_f32") for x in body) >= 2
                    and (best is None or len(body) < len(best))):
                best = body
    if best is None:
        raise RuntimeError("step loop not found")
    # a loop may have several latches (the compiler rotates / splits the exit tests): the body runs from the header to the
    # LAST backward branch that targets it
    # ... and what the compiler makes of `continue` / several exit tests is a nest of latches whose headers lie a few
    # instructions before one another: grow the region over every backward branch that starts at most 64 bytes before the
    # current header and ends after the current end (never into the enclosing refill loop, whose header is far away)
    header, last = best[0][0], best[-1][0]
    grown = True
    while grown:
        grown = False
        for a, op, args, t in insts:
            if op.startswith(("s_cbranch", "s_branch")) and t is not None and header - 64 <= t <= header and a >= last and (t, a) != (header, last):
                header, last, grown = t, a, True
    return [x for x in insts if header <= x[0] <= last]


def tally(body, cost, sgpr_cost):
    t = collections.Counter()
    for _, op, args, _ in body:
        op = re.sub(r"_e(32|64)$", "", op)
        if op.startswith("v_cmp"):
            op = "v_cmp_f64" if op.endswith("_f64") else "v_cmp"
        if op in FULL_RATE and re.search(r"\bs\d+\b|s\[|vcc|exec", args.split(",", 1)[1] if "," in args else ""):
            op += " (sgpr)"
        t[op] += 1
    valu = {k: v for k, v in t.items() if k.startswith("v_")}
    c = lambda k: sgpr_cost if k.endswith(" (sgpr)") else cost.get(k, DEFAULT)
    rows = sorted(((k, v, c(k), k in cost or k.endswith(" (sgpr)")) for k, v in valu.items()), key=lambda r: -r[1] * r[2])
    return {"valu": sum(valu.values()), "salu": sum(v for k, v in t.items() if k.startswith("s_")),
            "lds": sum(v for k, v in t.items() if k.startswith("ds_")),
            "vmem": sum(v for k, v in t.items() if k.startswith(("global_", "flat_", "buffer_", "scratch_"))),
            "cycles_per_block": sum(v * cc for _, v, cc, _ in rows),
            "mix": [{"op": k, "n": v, "cycles_each": cc, "costed": known} for k, v, cc, known in rows]}


def main():
    args = sys.argv[1:]
    out_json = ubench = None
    if "--json" in args:
        i = args.index("--json"); out_json = args[i + 1]; del args[i:i + 2]
    if "--ubench" in args:
        i = args.index("--ubench"); ubench = args[i + 1]; del args[i:i + 2]
    if ubench is None:
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_ubench_valu.txt")))
        ubench = cands[-1] if cands else None
    names = args or ["basic"]
    cost, sgpr_cost, src = load_ubench(ubench)
    txt, digest = disassemble()
    result = {"library_sha256_16": digest, "issue_costs_from": src or "built-in table (profiles/r1_ubench_valu.txt)",
              "unit": "SIMD cycles per wave64 Philox block (steps_per_block E-M steps x 64 lanes), sum of isolated issue costs", "kernels": {}}
    for name in names:
        r = tally(step_loop(kernel_insts(txt, KERNELS[name])), cost, sgpr_cost)
        r["steps_per_block"] = 8 if (KERNELS[name][5] or KERNELS[name][3]) else 4      # packed, bridge: 8 steps per pass
        result["kernels"][name] = r
        print(f"{name}: step loop = {r['valu']} VALU / {r['salu']} SALU / {r['lds']} LDS / {r['vmem']} VMEM instructions per block")
        for m in r["mix"]:
            print(f"  {m['op']:24s} x{m['n']:3d}  {m['cycles_each']:5.2f} cyc  = {m['n'] * m['cycles_each']:7.1f}" + ("" if m["costed"] else "   (default cost)"))
        print(f"  VALU issue cycles per block: {r['cycles_per_block']:.0f}")
    if out_json:
        with open(out_json, "w") as f:
            json.dump(result, f, indent=1)
        print("wrote", out_json)


if __name__ == "__main__":
    main()

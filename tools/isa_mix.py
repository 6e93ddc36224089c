#!/usr/bin/env python3
"""VALU issue-cost model of the step loop: tally the instructions of the innermost loop of a sim_kernel
instantiation (from hipcc -S) and weight them with the per-instruction issue costs measured by tools/ubench_valu
(profiles/*_ubench_valu.txt, 8 waves/SIMD column).  Prints SIMD cycles per wave64 Philox block (= 4 E-M steps x 64 lanes).

Usage: python tools/isa_mix.py [mangled-kernel-substring]   (default: basic_ddm_dc, fast, CAP4, no bridge)
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# measured on MI355X (profiles/r1_ubench_valu.txt), cycles per wave64 instruction per SIMD at 8 waves/SIMD
COST = {"v_mad_u64_u32": 4.67, "v_xor_b32": 2.32, "v_cvt_f32_u32": 4.12, "v_cvt_f32_i32": 4.12, "v_log_f32": 8.17,
        "v_sqrt_f32": 8.19, "v_sin_f32": 8.14, "v_cos_f32": 8.15, "v_exp_f32": 8.15, "v_rcp_f32": 8.14,
        "v_fma_f32": 3.77, "v_fmamk_f32": 3.77, "v_fmac_f32": 2.34, "v_fmaak_f32": 3.77, "v_add_f32": 2.26,
        "v_sub_f32": 2.26, "v_mul_f32": 2.24, "v_add_u32": 2.35, "v_sub_u32": 2.35, "v_subrev_u32": 2.35,
        "v_lshrrev_b32": 2.32, "v_lshlrev_b32": 2.32, "v_and_b32": 2.32, "v_or_b32": 2.32, "v_alignbit_b32": 2.32,
        "v_cndmask_b32": 2.32, "v_add3_u32": 2.35, "v_mov_b32": 2.32, "v_cmp": 4.13, "v_mul_lo_u32": 4.23,
        "v_mul_hi_u32": 4.26, "v_bfe_u32": 2.32, "v_and_or_b32": 2.32, "v_floor_f32": 2.3, "v_cvt_i32_f32": 4.12,
        "v_lshl_add_u32": 4.2, "v_addc_co_u32": 4.2, "v_max_f32": 2.26, "v_min_f32": 2.26, "v_med3_f32": 3.77}
# measured: a VOP2 integer/float op that reads an SGPR (or VCC) operand, and the 3-operand integer VOP3 forms, issue at
# ~4.2 cycles instead of ~2.3 (profiles/r1_ubench_valu.txt rows "v_xor_b32 (sgpr)", v_add3_u32, v_alignbit_b32,
# "v_cndmask_e64 (s)")
COST.update({"v_add3_u32": 4.22, "v_alignbit_b32": 4.15, "v_cndmask_b32": 4.22, "v_pk_fma_f32": 4.19, "v_pk_mul_f32": 4.19,
             "v_pk_add_f32": 4.20, "v_bitop3_b32": 3.84, "v_add_co_u32": 4.23})
SGPR_OPERAND_COST = 4.16
FULL_RATE = {"v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_add_f32", "v_sub_f32",
             "v_mul_f32", "v_lshrrev_b32", "v_lshlrev_b32", "v_mov_b32"}
DEFAULT = 2.4


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else "sim_kernelILi0ELb1ELb1ELb0EE"
    src = os.path.join(ROOT, "bayesflow_nddms_amd", "csrc", "nddm_kernels.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["hipcc", "-O3", "-ffp-contract=off", "--offload-arch=gfx950", "-std=c++17", "-S",
                               "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL)
        lines = open(out).read().splitlines()
    # kernel body
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN4nddm") and want in l and l.rstrip().endswith(":") or (want in l and ": " in l and l.startswith("_ZN4nddm")))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    # group basic blocks by the loop LLVM's asm comments assign them to ("in Loop: Header=BBx_y" / "Inner Loop Header");
    # the step loop is the INNERMOST-loop group holding the Philox multiplies
    groups, cur, headers = collections.defaultdict(list), None, set()
    for i, l in enumerate(body):
        if re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", l):
            nxt = body[i + 1] if i + 1 < len(body) else ""
            m = re.search(r"Header=(BB\d+_\d+)", l)
            if l.startswith(".LBB") and "Inner Loop Header" in (l + nxt):
                cur = l.split(":")[0][2:]
                headers.add(cur)
            elif m:
                cur = m.group(1)
            else:
                cur = None
        if cur is not None:
            groups[cur].append(l)
    inner = {h: seg for h, seg in groups.items() if h in headers}
    best = max(inner.values(), key=lambda seg: sum("v_mad_u64_u32" in x for x in seg), default=None)
    if best is None or sum("v_mad_u64_u32" in x for x in best) < 16:
        sys.exit("step loop not found")
    tally = collections.Counter()
    for l in best:
        t = l.strip().split()
        if not t or t[0].startswith((";", ".")) or t[0].endswith(":"):
            continue
        op = t[0]
        if op.startswith("v_cmp"):
            op = "v_cmp"
        op = re.sub(r"_e(32|64)$", "", op)
        if op in FULL_RATE and re.search(r"\bs\d+\b|s\[|vcc|exec", " ".join(t[2:])):
            op = op + " (sgpr)"
        tally[op] += 1
    valu = {k: v for k, v in tally.items() if k.startswith("v_")}
    def cost(k):
        return SGPR_OPERAND_COST if k.endswith(" (sgpr)") else COST.get(k, DEFAULT)
    cyc = sum(cost(k) * v for k, v in valu.items())
    print(f"kernel {want}: innermost step loop = {len(best)} lines")
    for k, v in sorted(valu.items(), key=lambda kv: -cost(kv[0]) * kv[1]):
        print(f"  {k:22s} x{v:3d}  {cost(k):5.2f} cyc  = {cost(k) * v:7.1f}" + ("" if (k in COST or k.endswith(" (sgpr)")) else "   (default cost)"))
    print(f"VALU instructions per block: {sum(valu.values())}; SALU: {sum(v for k, v in tally.items() if k.startswith('s_'))}")
    print(f"VALU issue cycles per wave64 block (4 steps x 64 lanes): {cyc:.0f}")
    print(f"ceiling at 2.4 GHz x 1024 SIMDs: {1024 * 2.4e9 / cyc * 256 / 1e12:.3f} T E-M steps/s")


if __name__ == "__main__":
    main()

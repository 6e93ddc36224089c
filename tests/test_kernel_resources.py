"""CPU test (hipcc cross-compiles without a GPU): the register budget that keeps 8 waves per SIMD resident is part of
the design (the SGPR file limits these kernels -- <= 74 SGPRs, <= 64 VGPRs, no scratch), so a
change that silently pushes a fast kernel over it fails here instead of showing up as a 3-10 % throughput loss."""
import os
import shutil
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


HAVE_HIPCC = shutil.which("hipcc") is not None or os.path.exists("/opt/rocm/bin/hipcc")
HAVE_OBJDUMP = os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump") or shutil.which("llvm-objdump") is not None


@pytest.mark.skipif(not HAVE_HIPCC, reason="hipcc missing")
def test_fast_kernels_fit_eight_waves_per_simd():
    import resource_table as rt
    rows = {rt.pretty(r["name"]): r for r in rt.collect()}
    sims = {k: v for k, v in rows.items() if k.startswith("sim_kernel<")}
    assert len(sims) >= 80                                   # 5 models x {fast, exact} x {cap4} x {small, packed, vkeys, ...} + bridge
    for name, r in sims.items():
        assert r.get("ScratchSize [bytes/lane]", 0) == 0, name               # no private scratch anywhere
        assert r.get("VGPRs Spill", 0) == 0 and r.get("SGPRs Spill", 0) == 0, name
    fast_small = [k for k in sims if ", fast," in k and "small=1" in k and "state_f64" not in k]
    assert len(fast_small) == 36                             # 5 models x cap4 {0,1} x packed {0,1} + 3 models with a VGPR-keys variant
                                                             # + the wire-format (codes) variant of basic / alpha_ns x cap4 {0,1}
    for name in fast_small:
        v, s = sims[name]["VGPRs"], sims[name]["TotalSGPRs"]
        assert rt.waves_by_vgpr(v) == 8 and rt.waves_by_sgpr(s) == 8, (name, v, s)
    # NDDM_STATE_F64 (the reference's float64 recurrence: four more doubles per lane): the basic kernels still keep 8 waves, the
    # single-trial ones 7 or more
    f64 = [k for k in sims if "state_f64" in k and "small=1" in k]
    assert len(f64) == 8
    for name in f64:
        v, s = sims[name]["VGPRs"], sims[name]["TotalSGPRs"]
        assert min(rt.waves_by_vgpr(v), rt.waves_by_sgpr(s)) >= (8 if "<basic" in name else 7), (name, v, s)
    # the bridge kernel (alpha_not_scaled, 32-bit staging; three Philox blocks per pass): SGPRs for 8 waves, VGPRs for 7 -- at 64
    # VGPRs (a scheduling barrier between its generators) it measured no faster than at 66-68 (A/B on one box, HISTORY.md section C.5.1)
    for name in sims:
        if ", fast," in name and "bridge=1" in name:
            assert rt.waves_by_sgpr(sims[name]["TotalSGPRs"]) == 8 and rt.waves_by_vgpr(sims[name]["VGPRs"]) >= 7, name


@pytest.mark.skipif(not HAVE_HIPCC, reason="hipcc missing")
def test_training_kernels_have_no_scratch_and_fit_their_lds():
    """The amortizer's kernels (tools/resource_table.py --train): no private scratch, no spills, and the long chains' LDS
    footprints leave room for exactly the residency their launch shapes assume (one workgroup per CU: <= 160 KB)."""
    import resource_table as rt
    rows = {rt.pretty(r["name"]): r for r in rt.collect(train=True)}
    for k in ("flow_fwd_kernel", "flow_dgrad_kernel", "flow_wgrad_kernel", "mlp_fwd_kernel<true>", "mlp_fwd_kernel<false>",
              "mlp_bwd_kernel<true>", "mlp_bwd_kernel<false>", "reduce_partials_kernel", "sqnorm_partial_kernel", "adam_kernel"):
        assert k in rows, (k, sorted(rows))
    for name, r in rows.items():
        assert r.get("ScratchSize [bytes/lane]", 0) == 0, name
        assert r.get("VGPRs Spill", 0) == 0, name               # (flow_fwd_kernel keeps 16 of its 106 SGPRs in VGPR lanes: v_writelane /
        assert r.get("SGPRs Spill", 0) <= 16, name              #  v_readlane, not memory -- its struct of 96 weight pointers)
        assert r.get("LDS Size [bytes/block]", 0) <= 160 * 1024, name
        assert r["VGPRs"] <= 256, name                          # (wave64: 512 per SIMD lane; 256 keeps two waves per SIMD possible)


@pytest.mark.skipif(not (HAVE_HIPCC and HAVE_OBJDUMP), reason="needs the ROCm toolchain (hipcc, llvm-objdump)")
def test_issue_model_reads_the_shipped_library():
    """tools/isa_mix.py finds the step loop in the shipped library's code object: 65 VALU instructions per 4-step block
    for every fast non-bridge model, 16 of them the Philox multiplies; 8 steps per block with the packed layout."""
    import isa_mix as im
    from bayesflow_nddms_amd.build import SO_PATH, build_hip
    build_hip()
    assert os.path.exists(SO_PATH)
    txt, digest = im.disassemble(SO_PATH)
    cost, sgpr_cost, _ = im.load_ubench(None)
    for name in ("basic", "single", "alpha_ns", "explicit"):
        body = im.step_loop(im.kernel_insts(txt, im.KERNELS[name]))
        t = im.tally(body, cost, sgpr_cost)
        assert 63 <= t["valu"] <= 67 and t["vmem"] == 0, (name, t["valu"])       # 65 with ROCm 7.2's compiler (profiles/*_issue_model.json)
        assert sum(m["n"] for m in t["mix"] if m["op"] == "v_mad_u64_u32") == 16
    t = im.tally(im.step_loop(im.kernel_insts(txt, im.KERNELS["basic_packed"])), cost, sgpr_cost)
    assert 95 <= t["valu"] <= 105 and sum(m["n"] for m in t["mix"] if m["op"] == "v_mad_u64_u32") == 16
    # NDDM_STATE_F64: the same generator + per step two double multiplies, two double adds, two double compares, a conversion
    t = im.tally(im.step_loop(im.kernel_insts(txt, im.KERNELS["basic_f64"])), cost, sgpr_cost)
    n = {m["op"]: m["n"] for m in t["mix"]}
    assert n["v_mad_u64_u32"] == 16 and n["v_mul_f64"] == 8 and n["v_add_f64"] == 8 and 75 <= t["valu"] <= 95, t["valu"]
    # the TRACKED issue model bench.py quotes (roofline_valu.frac) is the one of THIS library: regenerate it with
    # tools/refresh_issue_model.sh whenever the kernels change
    import glob
    import json
    import re
    newest = max(glob.glob(os.path.join(ROOT, "profiles", "*_issue_model.json")), key=lambda p: int(re.match(r"r(\d+)_", os.path.basename(p)).group(1)))
    d = json.load(open(newest))
    assert d["library_sha256_16"] == digest, (newest, "stale: run tools/refresh_issue_model.sh")
    rf = d["kernels"]["ratcliff_fast"]                                     # the exact sampler's loop (tools/ratcliff_isa_mix.py), same file
    assert 150 <= rf["valu"] <= 300 and rf["vmem"] == 0 and sum(m["n"] for m in rf["mix"] if m["op"].startswith("v_mad_u64_u32")) >= 32
    assert {"basic", "single", "alpha_ns_bridge", "basic_exact", "basic_exact_f64", "basic_f64"} <= set(d["kernels"])


@pytest.mark.skipif(not HAVE_HIPCC, reason="hipcc missing")
def test_toolchain_probes_compile(tmp_path):
    """The workarounds that blame the toolchain name a TRACKED probe each (tools/probe_buffer_load_b128.hip, tools/probe_graph_memset_node.py,
    tools/probe_stream_validation.py; their MI355X outputs are under profiles/).  The b128 claim is checkable without a GPU: this
    hipcc compiles __builtin_amdgcn_raw_buffer_load_b128 to buffer_load_dword -- one dword, not four.  If a toolchain update makes
    this test fail, the builtin has been fixed and RowTile (csrc/train_deepset.hip) may take 16-byte loads again."""
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(ROOT, "tools", "probe_buffer_load_b128.hip")
    asm = tmp_path / "probe.s"
    r = subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", str(asm), src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    text = asm.read_text()
    b128 = text[text.index("copy_b128"):text.index("copy_b32")] if text.index("copy_b128") < text.index("copy_b32") else text[text.index("copy_b128"):]
    assert "buffer_load_dwordx4" not in b128 and "buffer_load_dword " in b128, "the b128 builtin now loads 16 bytes: drop the workaround"
    exe = tmp_path / "probe"
    r = subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-o", str(exe), src], capture_output=True, text=True)
    assert r.returncode == 0 and exe.exists(), r.stderr[-2000:]
    for py in ("probe_graph_memset_node.py", "probe_stream_validation.py"):
        compile(open(os.path.join(ROOT, "tools", py)).read(), py, "exec")
    # every workaround comment names its probe
    for rel, needle in (("bayesflow_nddms_amd/csrc/train_deepset.hip", "tools/probe_buffer_load_b128.hip"),
                        ("bayesflow_nddms_amd/csrc/nddm_kernels.hip", "tools/probe_graph_memset_node.py"),
                        ("bayesflow_nddms_amd/csrc/nddm_kernels.hip", "tools/probe_stream_validation.py")):
        assert needle in open(os.path.join(ROOT, rel)).read(), (rel, needle)

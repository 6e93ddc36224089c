#!/bin/bash
# The ISA issue model of the library as it is built NOW (CPU only: disassembles bayesflow_nddms_amd/libnddm_hip.so): every kernel
# bench.py prices, with the newest tracked issue costs (profiles/*_ubench_valu.txt).  bench.py quotes the file only while its
# library hash matches the library that runs (roofline_valu.issue_model.library_matches; tests/test_host_logic.py checks it here).
# Usage: bash tools/refresh_issue_model.sh [round tag, default r6]
TAG=${1:-r6}
cd "$(dirname "$0")/.."
python3 tools/isa_mix.py basic single single_alt alpha_ns alpha_ns_bridge explicit basic_exact single_exact alpha_ns_exact \
    alpha_ns_bridge_exact basic_packed single_packed alpha_ns_packed basic_vkeys basic_f64 single_f64 basic_exact_f64 single_exact_f64 \
    --json profiles/${TAG}_issue_model.json | grep -v "^  "
python3 tools/ratcliff_isa_mix.py --json-into profiles/${TAG}_issue_model.json | grep "^added\|^flattened"

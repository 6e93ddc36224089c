#!/bin/bash
# rocprofv3 kernel trace + PMC passes of nddm::ratcliff_kernel at the bench leg's shape (1M sets x 300 trials, fast and exact transform;
# tools/ratcliff_shapes.py --one).  Run on the GPU box through gpurun; writes gpurun_out/prof_<tag>/ ; tools/summarize_ratcliff.py <tag>
# condenses it into profiles/<tag>_summary.md.
set -o pipefail
TAG=${1:-r6_ratcliff}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
PY=$(readlink -f "$(command -v python3)")
echo "$PY tools/ratcliff_shapes.py --one" > "$OUT/command.txt"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o r --output-format csv -- "$PY" tools/ratcliff_shapes.py --one > "$OUT/run_trace.txt" 2> "$OUT/trace.err" || { tail -5 "$OUT/trace.err"; exit 1; }
i=0
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC -d "$OUT/pmc$i" -o r --output-format csv -- "$PY" tools/ratcliff_shapes.py --one > "$OUT/run_pmc$i.txt" 2> "$OUT/pmc$i.err" || { echo "pmc pass $i failed"; tail -3 "$OUT/pmc$i.err"; }
  echo "pmc pass $i done"
done

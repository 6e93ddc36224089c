#!/bin/bash
# Profiling recipe for the bench command (run on the GPU box through gpurun).  Writes under gpurun_out/.
# Usage: bash tools/gpu_profile.sh <tag>
set -o pipefail
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
BENCH="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-ks"
# pass 1: kernel trace + stats (durations)
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench --output-format csv -- $BENCH > "$OUT/bench_trace.json" 2> "$OUT/trace.err" || { tail -5 "$OUT/trace.err"; exit 1; }
# passes 2..: hardware counters, each in its own run (no tracing domains besides kernel-trace)
i=0
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC -d "$OUT/pmc$i" -o bench --output-format csv -- $BENCH > "$OUT/bench_pmc$i.json" 2> "$OUT/pmc$i.err" || { echo "pmc pass $i failed"; tail -3 "$OUT/pmc$i.err"; }
done
find "$OUT" -name "*.csv" | head -50

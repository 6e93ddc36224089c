#!/bin/bash
# Profiling recipe for a bench command (run on the GPU box through gpurun).  Writes under gpurun_out/prof_<tag>/.
# Usage: bash tools/gpu_profile.sh <tag> [bench.py arguments, e.g. --model single]
#   pass 1: rocprofv3 --kernel-trace --stats (per-kernel durations; includes bench.py's lockstep ceiling run)
#   passes 2..: hardware counters, one `--pmc` group per run, no tracing domain besides kernel-trace
# The program after `--` is the interpreter BINARY by absolute path, resolved once here: rocprofv3 (a Python script) replaces itself with
# the program via os.execvpe, and with a bare name that is a PATH search -- one execvp attempt per PATH entry, from a process that (with
# --pmc) has already loaded the profiler's libraries.  bench.py itself starts no program while it is being profiled (the compiler
# version comes from the library's build record, nddm_build_info()).
set -o pipefail
TAG=${1:-r2}
shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
PY=$(readlink -f "$(command -v python3)")
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --no-ks --no-legs $*"       # (the headline kernel alone; the legs have a trace pass of their own below)
echo "$PY bench.py $ARGS" > "$OUT/command.txt"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench --output-format csv -- "$PY" bench.py $ARGS > "$OUT/bench_trace.json" 2> "$OUT/trace.err" || { tail -5 "$OUT/trace.err"; exit 1; }
i=0
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC -d "$OUT/pmc$i" -o bench --output-format csv -- "$PY" bench.py $ARGS --no-ceiling > "$OUT/bench_pmc$i.json" 2> "$OUT/pmc$i.err" || { echo "pmc pass $i failed"; tail -3 "$OUT/pmc$i.err"; }
  echo "pmc pass $i done"
done
# the default command WITH its side legs (every BASELINE config in one trace): tools/legs_trace_summary.py
if [ -z "$*" ]; then
  LEGS="--steps 5 --warmup 1 --no-cpu-baseline --no-ks --no-ceiling"
  echo "$PY bench.py $LEGS" > "$OUT/command_legs.txt"
  rocprofv3 --kernel-trace --stats -d "$OUT/trace_legs" -o bench --output-format csv -- "$PY" bench.py $LEGS > "$OUT/bench_trace_legs.json" 2> "$OUT/trace_legs.err" || { tail -5 "$OUT/trace_legs.err"; }
fi
find "$OUT" -name "*.csv" | head -50

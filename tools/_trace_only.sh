#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/art
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/train_trace -o train --output-format csv -- python3 $ROOT/bench.py --train --train-mode graph > $OUT/train_trace_bench.json 2> $OUT/train_trace.err && echo ok trace
cd $ROOT
python3 tools/train_trace_summary.py $(dirname $(find gpurun_out/train_trace -name "*kernel_trace.csv" | head -1)) > $OUT/r3_train_graph_trace.md 2>&1 && echo ok summary
cp $(find gpurun_out/train_trace -name "*kernel_stats.csv" | head -1) $OUT/r3_train_graph_kernel_stats.csv
rm -rf gpurun_out/train_trace
cat $OUT/r3_train_graph_trace.md

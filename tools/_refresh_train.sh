#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/art
mkdir -p "$OUT"
cd "$ROOT"
python3 bench.py --train > $OUT/r3_train.json 2> $OUT/r3_train.err && echo ok train
python3 bench.py --train --dist --backend nccl --train-mode graph --train-parallel ddp > $OUT/r3_train_ddp_world1.json 2> $OUT/r3_train_ddp.err && echo ok ddp
python3 bench.py --gpus 2 --share-device --backend gloo --train --train-iters 60 > "$OUT/r3_two_rank_train.log" 2>&1 && echo "ok two_rank_train"
(python3 tools/recovery_probe.py 3000; python3 tools/recovery_probe.py 20000) > $OUT/r3_recovery.txt 2>&1 && echo ok recovery
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/train_trace -o train --output-format csv -- python3 $ROOT/bench.py --train --train-mode graph > $OUT/train_trace_bench.json 2> $OUT/train_trace.err && echo ok trace
cd $ROOT
python3 tools/train_trace_summary.py $(dirname $(find gpurun_out/train_trace -name "*kernel_trace.csv" | head -1)) > $OUT/r3_train_graph_trace.md 2>&1 && echo ok summary
cp $(find gpurun_out/train_trace -name "*kernel_stats.csv" | head -1) $OUT/r3_train_graph_kernel_stats.csv
rm -rf gpurun_out/train_trace
run() { name=$1; shift; python3 bench.py "$@" > "$OUT/$name" 2> "$OUT/$name.err" && echo "ok $name" || { echo "FAILED $name"; tail -3 "$OUT/$name.err"; }; }
run r3_bench_dist_summary_world1.json --dist --backend nccl --gather summary --no-ceiling --no-ks --no-cpu-baseline
run r3_bench_dist_trials_world1.json --dist --backend nccl --gather trials --no-ceiling --no-ks --no-cpu-baseline
run r3_bench_dist_codes_world1.json --dist --backend nccl --gather codes --no-ceiling --no-ks --no-cpu-baseline
(python3 tools/train_stamps.py flow 0; python3 tools/train_stamps.py deepset 0) 2>&1 | grep -v "amdgpu.ids" > $OUT/r3_train_stamps.txt && echo ok stamps

set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/art
mkdir -p "$OUT"
cd "$ROOT"
run() { name=$1; shift; python3 bench.py "$@" > "$OUT/$name" 2> "$OUT/$name.err" && echo "ok $name" || { echo "FAILED $name"; tail -3 "$OUT/$name.err"; }; }
run r6_train.json --train
run r6_train_gather_world1.json --train --dist --backend nccl --train-mode graph
run r6_train_ddp_world1.json --train --dist --backend nccl --train-mode graph --train-parallel ddp
python3 bench.py --gpus 2 --share-device --backend gloo --train --train-iters 60 > "$OUT/r6_two_rank_train.log" 2>&1 && echo "ok two_rank_train"
(python3 tools/recovery_probe.py 3000; python3 tools/recovery_probe.py 20000) 2>&1 | grep -v "amdgpu.ids" > "$OUT/r6_recovery.txt" && echo "ok recovery"
(python3 tools/train_stamps.py flow 0; python3 tools/train_stamps.py deepset 0) 2>&1 | grep -v "amdgpu.ids" > "$OUT/r6_train_stamps.txt" && echo "ok train_stamps"
(for f in plain gather ddp; do for dt in 0.01 0.001; do python3 tools/train_stage_times.py $f $dt 2>/dev/null | grep "it/s"; done; done) > "$OUT/r6_train_stage_times.txt" && echo "ok train_stage_times"
python3 tools/resource_table.py --train --md > "$OUT/r6_train_resource_table.md" 2>/dev/null && echo "ok train_resource_table"
for form in plain gather; do
  extra=""; [ $form = gather ] && extra="--dist --backend nccl"
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/train_trace_$form" -o train --output-format csv -- "$(readlink -f "$(command -v python3)")" "$ROOT/bench.py" --train --train-mode graph $extra > "$OUT/train_trace_bench_$form.json" 2> "$OUT/train_trace_$form.err") && echo "ok train_trace $form"
  D="$(dirname "$(find gpurun_out/train_trace_$form -name '*kernel_trace.csv' | head -1)")"
  if [ $form = plain ]; then
    python3 tools/train_trace_summary.py "$D" > "$OUT/r6_train_graph_trace.md" 2>&1 && echo "ok train_trace_summary"
    cp "$(find gpurun_out/train_trace_$form -name '*kernel_stats.csv' | head -1)" "$OUT/r6_train_graph_kernel_stats.csv"
    python3 tools/train_iteration_timeline.py "$D" --all > "$OUT/r6_train_timeline.txt" 2>&1 && echo "ok train_timeline"
  else
    python3 tools/train_iteration_timeline.py "$D" --all > "$OUT/r6_train_timeline_gather.txt" 2>&1 && echo "ok train_timeline gather"
  fi
  rm -rf gpurun_out/train_trace_$form
done

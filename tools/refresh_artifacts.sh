#!/bin/bash
# Re-collect every bench artifact that profiles/README.md lists (run on the GPU box through gpurun; ~12 min; the default bench
# line now includes BASELINE configs[0] at full size, ~35 s of CPU time per model that has a NumPy port).
# Writes gpurun_out/art/*; copy what you want judged into profiles/ (tools/refresh_artifacts.sh does not touch profiles/).
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/art
mkdir -p "$OUT"
cd "$ROOT"
run() { name=$1; shift; python3 bench.py "$@" > "$OUT/$name" 2> "$OUT/$name.err" && echo "ok $name" || { echo "FAILED $name"; tail -3 "$OUT/$name.err"; }; }
run r6_bench_basic.json
run r6_bench_single.json --model single
run r6_bench_alpha_ns.json --model alpha_ns
run r6_bench_alpha_ns_bridge.json --model alpha_ns_bridge
run r6_bench_basic_dt01.json --dt 0.01 --max-steps 400
run r6_bench_basic_packed.json --gauss packed
run r6_train.json --train
run r6_bench_dist_summary_world1.json --dist --backend nccl --gather summary --no-ceiling --no-ks --no-cpu-baseline
run r6_bench_dist_trials_world1.json --dist --backend nccl --gather trials --no-ceiling --no-ks --no-cpu-baseline
run r6_bench_dist_codes_world1.json --dist --backend nccl --gather codes --no-ceiling --no-ks --no-cpu-baseline
run r6_train_gather_world1.json --train --dist --backend nccl --train-mode graph
run r6_train_ddp_world1.json --train --dist --backend nccl --train-mode graph --train-parallel ddp
python3 bench.py --gpus 5 --share-device --backend gloo --sets 100000 --steps 3 --warmup 1 > "$OUT/r6_five_rank_none.log" 2>&1 && echo "ok five_rank_none"
python3 bench.py --gpus 2 --share-device --backend gloo --sets 100000 --steps 3 --warmup 1 --no-cpu-baseline --no-ks --no-ceiling > "$OUT/r6_two_rank_none.log" 2>&1 && echo "ok two_rank_none"
python3 bench.py --gpus 2 --share-device --backend gloo --sets 100000 --steps 3 --warmup 1 --no-cpu-baseline --no-ks --no-ceiling --gather summary > "$OUT/r6_two_rank_summary.log" 2>&1 && echo "ok two_rank_summary"
python3 bench.py --gpus 2 --share-device --backend gloo --sets 100000 --steps 3 --warmup 1 --no-cpu-baseline --no-ks --no-ceiling --gather trials > "$OUT/r6_two_rank_trials.log" 2>&1 && echo "ok two_rank_trials"
python3 bench.py --gpus 2 --share-device --backend gloo --train --train-iters 60 > "$OUT/r6_two_rank_train.log" 2>&1 && echo "ok two_rank_train"
python3 tools/overlap_probe.py > "$OUT/r6_overlap_probe.txt" 2>&1 && echo "ok overlap_probe"
python3 tools/quick_time.py > "$OUT/r6_quick_time.txt" 2>&1 && echo "ok quick_time"
python3 tools/wave_timeline.py 0:1000000:300:0.001:4000 0:1000000:300:0.01:400 0:1000000:60:0.01:400 0:30000:300:0.001:4000 0:10000:300:0.001:4000 > "$OUT/r6_wave_timeline.txt" 2>&1 && echo "ok wave_timeline"
(python3 tools/recovery_probe.py 3000; python3 tools/recovery_probe.py 20000) 2>&1 | grep -v "amdgpu.ids" > "$OUT/r6_recovery.txt" && echo "ok recovery"
(python3 tools/train_stamps.py flow 0; python3 tools/train_stamps.py deepset 0) 2>&1 | grep -v "amdgpu.ids" > "$OUT/r6_train_stamps.txt" && echo "ok train_stamps"
(for f in plain gather ddp; do for dt in 0.01 0.001; do python3 tools/train_stage_times.py $f $dt 2>/dev/null | grep "it/s"; done; done) > "$OUT/r6_train_stage_times.txt" && echo "ok train_stage_times"
(./tools/ubench_rocrand && ./tools/ubench_rocrand_fast) > "$OUT/r6_ubench_rocrand.txt" 2>&1 && echo "ok ubench_rocrand"
python3 tools/probe_stream_validation.py > "$OUT/r6_stream_validation.txt" 2>/dev/null && echo "ok stream_validation"
python3 tools/probe_graph_memset_node.py 2>&1 | grep -v "amdgpu.ids" > "$OUT/r6_probe_graph_memset_node.txt" && echo "ok probe_graph_memset_node"
(hipcc -O3 --offload-arch=gfx950 -o /tmp/probe_b128 tools/probe_buffer_load_b128.hip 2>/dev/null && /tmp/probe_b128) > "$OUT/r6_probe_buffer_load_b128.txt" 2>&1 && echo "ok probe_buffer_load_b128"
python3 tools/fused_accuracy.py 2>&1 | grep -v "amdgpu.ids" > "$OUT/r6_fused_accuracy.txt" && echo "ok fused_accuracy"
python3 bench.py --dist --backend nccl > "$OUT/r6_bench_dist_none_world1.json" 2> "$OUT/r6_bench_dist_none_world1.err" && echo "ok dist_none_world1"
# (the whole training runs + recovery on the reference's statistic + the tail-draw traces: ~6 min)
# python3 tools/full_training_run.py 500 basic gpurun_out/basic_500.pt > "$OUT/r6_full_training_run.txt"; python3 tools/locate_tail_draws.py gpurun_out/basic_500.pt basic 500 100000 > "$OUT/r6_tail_draw_incidence.txt"
python3 tools/resource_table.py --train --md > "$OUT/r6_train_resource_table.md" 2>/dev/null && echo "ok train_resource_table"
# rocprofv3 kernel trace of the graph-replayed training loop -> kernels per iteration, GPU busy fraction
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/train_trace" -o train --output-format csv -- "$(readlink -f "$(command -v python3)")" "$ROOT/bench.py" --train --train-mode graph > "$OUT/train_trace_bench.json" 2> "$OUT/train_trace.err") && echo "ok train_trace"
python3 tools/train_trace_summary.py "$(dirname "$(find gpurun_out/train_trace -name '*kernel_trace.csv' | head -1)")" > "$OUT/r6_train_graph_trace.md" 2>&1 && echo "ok train_trace_summary"
python3 tools/train_iteration_timeline.py "$(dirname "$(find gpurun_out/train_trace -name '*kernel_trace.csv' | head -1)")" --all > "$OUT/r6_train_timeline.txt" 2>&1 && echo "ok train_timeline"
cp "$(find gpurun_out/train_trace -name '*kernel_stats.csv' | head -1)" "$OUT/r6_train_graph_kernel_stats.csv"
rm -rf gpurun_out/train_trace

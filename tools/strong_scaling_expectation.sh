#!/bin/bash
# What strong scaling of the headline batch (1M sets x 300 trials) can be at N GPUs, from ONE GPU: the per-GPU share of the batch timed on
# this card (the path has no data-path collective, so a rank's step is its own launch; what a node adds is load imbalance between cards).
# efficiency(N) = t(1M) / (N * t(1M / N)).   usage: bash tools/strong_scaling_expectation.sh > profiles/r5_strong_scaling_expectation.txt
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
echo "# sets per GPU | ms per step (20 timed steps, bench.py --sets S) | trials/s | expected strong-scaling efficiency at N = 1M / S GPUs"
base=""
for S in 1000000 500000 250000 125000; do
  line=$(python3 bench.py --sets $S --steps 20 --warmup 3 --no-legs --no-cpu-baseline --no-ks --no-ceiling 2>/dev/null | tail -1)
  python3 - "$S" "$line" "$base" <<'PY'
import json, sys
S, d, base = int(sys.argv[1]), json.loads(sys.argv[2]), sys.argv[3]
ms = d["ms_per_step"]
n = 1000000 // S
eff = (float(base) / (n * ms)) if base else 1.0
print(f"{S:8d} | {ms:8.3f} | {d['value']:.3e} | N = {n}: {eff:.3f}  (launch: {d['launch']['grid_waves']} waves, kernel {d['roofline']['kernel_ms']:.3f} ms)")
PY
  if [ -z "$base" ]; then base=$(python3 -c "import json,sys; print(json.loads(sys.argv[1])['ms_per_step'])" "$line"); fi
done

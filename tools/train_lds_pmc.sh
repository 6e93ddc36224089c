#!/bin/bash
# LDS bank conflicts of the training kernels, per launch (run on the GPU box through gpurun): one rocprofv3 counter pass over the
# graph-replayed training loop (counters only with --kernel-trace: the pool refuses --pmc next to the hip/hsa trace domains).
# usage: bash tools/train_lds_pmc.sh > profiles/rN_train_lds_pmc.txt
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_train
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES \
    -d "$OUT" -o t --output-format csv -- "$(readlink -f "$(command -v python3)")" "$ROOT/bench.py" --train --train-mode graph --train-iters 20 \
    > "$ROOT/gpurun_out/pmc_train.json" 2> "$ROOT/gpurun_out/pmc_train.err" || { tail -3 "$ROOT/gpurun_out/pmc_train.err"; exit 1; }
python3 - "$OUT" <<'PY'
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0][:48]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_INSTS_LDS":
        cnt[k] += 1
print("# graph-replayed training loop (bench.py --train --train-mode graph), rocprofv3 --pmc, per launch (sums over the launch's waves):")
print("# LDS instructions | cycles the LDS index unit is active | of them bank-conflict cycles | conflict / active")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_LDS_BANK_CONFLICT", 0)):
    n = max(cnt[k], 1)
    if v.get("SQ_INSTS_LDS", 0) == 0:
        continue
    print(f"{k:50s} launches {n:5d}  insts {v.get('SQ_INSTS_LDS', 0) / n:9.0f}  active {v.get('SQ_LDS_IDX_ACTIVE', 0) / n:9.0f}  "
          f"conflict {v.get('SQ_LDS_BANK_CONFLICT', 0) / n:9.0f}  ratio {v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 1), 1):.3f}")
PY
rm -rf "$OUT"

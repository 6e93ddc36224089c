#!/bin/bash
# A/B two builds of the HIP library on ONE GPU box (boxes differ by ~1-2 %, alternating runs on one box resolve ~0.3 %).
#   1. here (build container):  bash tools/ab_compare.sh build <git-rev-A> [<git-rev-B>|WORKTREE]
#        compiles bayesflow_nddms_amd/csrc at the two revisions into tools/ab/libnddm_{a,b}.so (git-ignored, they travel
#        to the GPU box with the snapshot)
#   2. on the GPU box (through gpurun):  bash tools/ab_compare.sh run [rounds] -- <quick_time case> ...
#        runs tools/quick_time.py on every case with library a, then b, `rounds` times (default 2); NDDM_HIP_LIB selects
#        the library (bayesflow_nddms_amd/build.py)
set -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
FLAGS="-O3 -ffp-contract=off --offload-arch=gfx950 -fPIC -shared -std=c++17"
case "$1" in
build)
    mkdir -p "$ROOT/tools/ab"
    for side in a b; do
        rev=$2; [ $side = b ] && rev=${3:-WORKTREE}
        if [ "$rev" = WORKTREE ]; then src=$ROOT
        else src=$(mktemp -d); git -C "$ROOT" archive "$rev" bayesflow_nddms_amd/csrc include | tar -x -C "$src"; fi
        hipcc $FLAGS -o "$ROOT/tools/ab/libnddm_$side.so" "$src/bayesflow_nddms_amd/csrc/nddm_kernels.hip" 2>/dev/null || { echo "build of $rev failed"; exit 1; }
        echo "$side = $rev"
    done > "$ROOT/tools/ab/sides.txt"; cat "$ROOT/tools/ab/sides.txt" ;;
run)
    shift; rounds=2; if [ "$1" != "--" ]; then rounds=$1; shift; fi; shift
    cat "$ROOT/tools/ab/sides.txt"
    for r in $(seq "$rounds"); do for side in a b; do
        echo "== $side (round $r)"
        NDDM_HIP_LIB=$ROOT/tools/ab/libnddm_$side.so python3 "$ROOT/tools/quick_time.py" "$@" 2>&1 | grep "^model" |
            sed 's/fast=True //; s/trials_out=True //; s/lockstep=False //; s/bridge=False //; s/packed=False//'
    done; done ;;
*) sed -n 2,9p "$0" ;;
esac

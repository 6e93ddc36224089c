#!/bin/bash
# What is a VALU instruction of the refill path worth?  Builds the library three times -- as is, and with 16 / 32 extra full-rate
# VALU instructions in every hand-out (-DNDDM_EXTRA_REFILL_VALU=N) -- and times them alternately on ONE box.
# The slope (ms per added instruction) times the hand-out's ~22 instructions bounds what ANY cheaper hand-out can buy.
#   here:        bash tools/refill_sensitivity.sh build
#   on the box:  bash tools/refill_sensitivity.sh run [rounds] -- <quick_time case> ...
set -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
FLAGS="-O3 -ffp-contract=off --offload-arch=gfx950 -fPIC -shared -std=c++17"
case "$1" in
build)
    mkdir -p "$ROOT/tools/ab"
    for n in 0 16 32; do
        extra=""; [ $n != 0 ] && extra="-DNDDM_EXTRA_REFILL_VALU=$n"
        hipcc $FLAGS $extra -o "$ROOT/tools/ab/libnddm_extra$n.so" "$ROOT/bayesflow_nddms_amd/csrc/nddm_kernels.hip" || exit 1
        echo "built extra$n"
    done ;;
run)
    shift; rounds=2; if [ "$1" != "--" ]; then rounds=$1; shift; fi; shift
    for r in $(seq "$rounds"); do for n in 0 16 32; do
        echo "== +$n VALU per hand-out (round $r)"
        NDDM_HIP_LIB=$ROOT/tools/ab/libnddm_extra$n.so python3 "$ROOT/tools/quick_time.py" "$@" 2>&1 | grep "^model" |
            sed 's/fast=True //; s/tune=None //; s/trials_out=True //; s/lockstep=False //; s/bridge=False //; s/packed=False//'
    done; done ;;
*) sed -n 2,7p "$0" ;;
esac

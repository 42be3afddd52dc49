#!/bin/bash
# Diagnostic A/B of kernel variants on ONE GPU box (device-to-device variance is several %).
#   tools/ablate.sh build   (in the build container: hipcc cross-compiles)  -> abl/libmcgp_<variant>.so
#   tools/ablate.sh run     (on the GPU box, via gpurun)                    -> gpurun_out/ablate.txt   (DEV=53: the reference-width kernel)
# VARIANTS are -DMCGP_<name>=<value> switches of race_kernel_reg.hip.h:
#   DUP=k   run section k twice (idempotent, results unchanged): its cost shows as a time difference
#           (1 sorting network after the lap step, 4 _update_positions, 8 re-sort after an overtake pass, 16 event Philox block)
#   SKIP=k  leave section k out (results wrong): timing only
#           (1 overtake passes, 2 event handling, 4 grid sampling, 8 laps 2..L, 16 sorting network, 32 _update_positions, 64 lap step,
#            256 the retirement pre-draw -- nobody retires after lap 1 --, 512 the per-lap retirement handler -- likewise)
set -e
cd "$(dirname "$0")/.."
VARIANTS=${VARIANTS:-"DUP=0 DUP=1 DUP=4 DUP=8 DUP=16 SKIP=1 SKIP=2 SKIP=4 SKIP=8 SKIP=16 SKIP=32 SKIP=64 SKIP=256 SKIP=512"}
case "$1" in
build)
  mkdir -p abl
  root=$PWD
  for v in $VARIANTS; do
    name=${v//=/_}          # no '=' in a make goal: make would read it as a variable assignment
    ( cd monte_carlo_gp_amd/csrc && make -s -j8 BUILD=/tmp/abl_$name OUT=$root/abl/libmcgp_$name.so REG_SIZES=20 \
        HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -DMCGP_$v -DMCGP_ONLY_N20" \
        $root/abl/libmcgp_$name.so )
  done ;;
run)
  mkdir -p gpurun_out
  out=gpurun_out/ablate.txt
  : > $out
  for rep in 1 2; do
    for v in $VARIANTS; do
      ms=$(MCGP_LIB=$PWD/abl/libmcgp_${v//=/_}.so MCGP_BENCH_NOCHECK=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --deviates ${DEV:-32} \
           --sims-per-step ${SIMS:-4000000} 2>/dev/null | python -c "import json,sys; print(json.load(sys.stdin)['roofline']['kernel_ms_avg'])")
      echo "$v kernel_ms=$ms" | tee -a $out
    done
  done ;;
*) echo "usage: $0 build|run"; exit 2 ;;
esac

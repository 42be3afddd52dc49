#!/bin/bash
# Diagnostic: build libmcgp_hip variants with one section duplicated (MCGP_DUP bit) and time each.
# Run on the GPU box:  bash tools/ablate.sh   -> gpurun_out/ablate.txt
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/ablate.txt
: > $out
VARIANTS=${VARIANTS:-"DUP=0 DUP=1 DUP=2 DUP=4 DUP=8 SKIP=1 SKIP=2 SKIP=4 SKIP=8"}
for v in $VARIANTS; do
  ( mkdir -p /tmp/abl_$v && cd monte_carlo_gp_amd/csrc && \
    make -s -j8 BUILD=/tmp/abl_$v OUT=/tmp/libmcgp_$v.so REG_SIZES=20 \
      HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -DMCGP_$v -DMCGP_ONLY_N20" /tmp/libmcgp_$v.so ) &
done
wait
for v in $VARIANTS; do
  ms=$(MCGP_LIB=/tmp/libmcgp_$v.so MCGP_BENCH_NOCHECK=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --sims-per-step 4000000 2>/dev/null | python -c "import json,sys; print(json.load(sys.stdin)['roofline']['kernel_ms_avg'])")
  echo "$v kernel_ms=$ms" | tee -a $out
done

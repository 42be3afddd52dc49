#!/bin/bash
# Diagnostic: build libmcgp_hip variants with one section duplicated (MCGP_DUP bit) and time each.
# Run on the GPU box:  bash tools/ablate.sh   -> gpurun_out/ablate.txt
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/ablate.txt
: > $out
for dup in 0 1 2 4 8 16; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -DMCGP_DUP=$dup \
      -shared -o /tmp/libmcgp_dup$dup.so monte_carlo_gp_amd/csrc/mcgp_hip.hip
  ms=$(MCGP_LIB=/tmp/libmcgp_dup$dup.so python bench.py --steps 3 --warmup 1 --no-cpu-baseline --sims-per-step 4000000 2>/dev/null | python -c "import json,sys; print(json.load(sys.stdin)['roofline']['kernel_ms_avg'])")
  echo "dup=$dup kernel_ms=$ms" | tee -a $out
done

#!/bin/bash
cd "$(dirname "$0")/.."
export VARIANTS="DUP=0 SKIP=1 SKIP=2 SKIP=8 SKIP=16 SKIP=32 SKIP=64 SKIP=256 DUP=1 DUP=4 DUP=8"
tools/ablate.sh run > /dev/null 2>&1; cp gpurun_out/ablate.txt gpurun_out/r5_ablate.txt
DEV=53 tools/ablate.sh run > /dev/null 2>&1; cp gpurun_out/ablate.txt gpurun_out/r5_ablate_wide.txt
echo "== default"; cat gpurun_out/r5_ablate.txt; echo "== wide"; cat gpurun_out/r5_ablate_wide.txt

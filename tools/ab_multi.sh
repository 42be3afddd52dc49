#!/bin/bash
# tools/ab_multi.sh "WL lib1 lib2 ..." "WL2 libA libB ..."  -- tools/ab.sh for several workloads in one go -> gpurun_out/ab_multi.txt
cd "$(dirname "$0")/.."
: > gpurun_out/ab_multi.txt
for spec in "$@"; do
  tools/ab.sh $spec > /dev/null 2>&1
  cat gpurun_out/ab.txt >> gpurun_out/ab_multi.txt
done
cat gpurun_out/ab_multi.txt

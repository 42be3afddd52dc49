#!/bin/bash
# round 5, final measurement A: rocprofv3 passes of every profiled workload on the frozen source (raw output: gpurun_out/r5_*)
cd "$(dirname "$0")/.."
set -e
for wl in S60 S78 HET N10 N25; do tools/profile.sh r5 $wl; done
SFX=_orders tools/profile.sh r5 S60 --orders
SFX=_wide tools/profile.sh r5 S60 --deviates 53
SFX=_wide tools/profile.sh r5 S78 --deviates 53
echo profiles done

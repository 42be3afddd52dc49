#!/bin/bash
# round 5, final measurement A: rocprofv3 passes of every profiled workload on the frozen source (raw output: gpurun_out/r5_*)
cd "$(dirname "$0")/.."
set -e
for wl in S60 S78 HET N10 N25; do tools/profile.sh r5 $wl; done
SFX=_orders tools/profile.sh r5 S60 --orders
SFX=_wide tools/profile.sh r5 S60 --deviates 53
SFX=_wide tools/profile.sh r5 S78 --deviates 53
echo profiles done
# dynamic count of vector-memory instructions of the reference-width kernel (scratch spills in the race loop would show here)
args="bench.py --workload S60 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --deviates 53"
d=gpurun_out/r5_S60_wide_vmem; rm -rf $d
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU --output-format csv -d $d -- python3 $args > $d.json 2> $d.log && echo "pass vmem done"

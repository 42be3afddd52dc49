#!/bin/bash
# round 5, final measurement B: all workloads on one box (both deviate widths), issue-cost microbenchmark, batch launch
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
tools/workloads.sh > gpurun_out/r5_workloads.jsonl
for wl in S60 S78 S50 EVT HET DMP WET N10 N22 N25; do
  python bench.py --workload $wl --deviates 53 --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json, sys
d = json.load(sys.stdin); r = d['roofline']
print(json.dumps({'workload': '$wl', 'deviates': 53, 'value': d['value'], 'kernel_ms_avg': r['kernel_ms_avg'], 'kernel': r['kernel'],
                  'launch': d.get('valu', {}).get('launch'), 'steps': d['steps']}))" >> gpurun_out/r5_workloads.jsonl
done
cat gpurun_out/r5_workloads.jsonl | cut -c1-150
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/valu_peak tools/valu_peak.hip && tools/valu_peak > gpurun_out/r5_valu_peak.json && echo valu_peak done
python tools/batch_time.py 10000 100000 > gpurun_out/r5_batch_time.txt 2>&1; cat gpurun_out/r5_batch_time.txt

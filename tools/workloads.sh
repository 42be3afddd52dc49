#!/bin/bash
# Every workload of bench.py on one box (kernel ms from the library's hipEvents):  tools/workloads.sh > profiles/r4_workloads.jsonl
cd "$(dirname "$0")/.."
for wl in ${WORKLOADS:-S60 S78 S50 EVT HET DMP WET N10 N22 N25}; do
  python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json, sys
d = json.load(sys.stdin); r = d['roofline']
print(json.dumps({'workload': '$wl', 'value': d['value'], 'kernel_ms_avg': r['kernel_ms_avg'], 'kernel': r['kernel'],
                  'launch': d.get('valu', {}).get('launch'), 'steps': d['steps']}))"
done

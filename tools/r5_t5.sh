#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python bench.py > gpurun_out/r5_bench_full.json 2> gpurun_out/r5_bench_full.err || { tail -20 gpurun_out/r5_bench_full.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5_bench_full.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'wide', d['deviates53']['kernel_ms_avg'], d.get('value_at_reference_width'))
print('sweep24', {k:v for k,v in d['workloads']['sweep24'].items() if k not in ('note','metric')})
print('gpu_seconds', d['gpu_seconds'], 'cpu', d.get('cpu_baseline_seconds'))
PY
MCGP_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r5_bench_self2.json 2> gpurun_out/r5_bench_self2.err || { tail -20 gpurun_out/r5_bench_self2.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_bench_self2.json').read().strip().splitlines()[-1])
print('self-launched n_gpus', d['n_gpus'], 'pg', d['process_group'], 'value', d['value'], 'sweep', d['workloads']['sweep24']['wall_seconds'], d['workloads']['sweep24']['launches_per_rank'], 'devices', d['distinct_devices'])
PY

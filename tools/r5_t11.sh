#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python bench.py > gpurun_out/r5_bench_line.json 2> gpurun_out/r5_bench_line.err || { tail -20 gpurun_out/r5_bench_line.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5_bench_line.json'))
r=d['roofline']; w=d['deviates53']
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', r['frac'], 'useful', r.get('frac_useful_lanes'), 'mix', r.get('frac_of_mix_ceiling_at_occupancy'), 'note', r['counters_note'])
print('wide', w['kernel_ms_avg'], w['slowdown_vs_32bit_deviates'], w['roofline']['frac'], w['roofline'].get('traffic'), w['roofline']['counters_note'])
print('sweep24', d['workloads']['sweep24']['wall_seconds'], d['workloads']['sweep24']['value'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['all_cores']['value'])
PY
python -m monte_carlo_gp_amd.cli predict --race Bahrain --season 2024 --simulations 10000 --offline --seed 42 2>&1 | tail -12

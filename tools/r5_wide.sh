#!/bin/bash
# round 5: the reference-width kernel after the LDS table / lazy companions -- parity, then time
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -k "reference_width or deviates" > gpurun_out/r5_wide_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r5_wide_tests.log
[ $rc -ne 0 ] && exit $rc
for d in 53 32; do
  python3 bench.py --workload S60 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --deviates $d > gpurun_out/r5_wide_bench_$d.json 2> gpurun_out/r5_wide_bench_$d.log
  python3 -c "import json; d=json.load(open('gpurun_out/r5_wide_bench_$d.json')); print('deviates $d kernel ms', d['roofline']['kernel_ms_avg'], d['roofline']['kernel'], d['valu']['launch'])"
done
for wl in S78 HET N10 N22; do
  python3 bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-extras --deviates 53 > gpurun_out/r5_wide_bench_$wl.json 2> gpurun_out/r5_wide_bench_$wl.log
  python3 -c "import json; d=json.load(open('gpurun_out/r5_wide_bench_$wl.json')); print('$wl deviates 53 kernel ms', d['roofline']['kernel_ms_avg'], d['roofline']['kernel'], d['valu']['launch'])"
done

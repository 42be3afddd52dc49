#!/bin/bash
# Dynamic instruction counts per simulated lap and wave for kernel variants (run on the GPU box):
#   tools/valu_count.sh abl/libmcgp_A.so abl/libmcgp_B.so ...   -> gpurun_out/valu_count.txt
# One rocprofv3 --pmc pass per library (counters only, no tracing).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
out=gpurun_out/valu_count.txt
: > $out
SIMS=${SIMS:-2000000}
for lib in "$@"; do
  tag=$(basename $lib .so)
  rm -rf gpurun_out/vc_$tag
  export MCGP_LIB=$PWD/$lib MCGP_BENCH_NOCHECK=1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d gpurun_out/vc_$tag -- \
      python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --sims-per-step $SIMS > /dev/null 2>&1
  python3 - "$tag" "$SIMS" >> $out <<'PY'
import csv, glob, sys, collections
tag, sims = sys.argv[1], float(sys.argv[2])
acc = collections.defaultdict(list)
for f in glob.glob(f'gpurun_out/vc_{tag}/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'race_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
w = sims / 64 * 60
print(tag, ' '.join(f"{k}={sum(v)/len(v)/w:.1f}" for k, v in sorted(acc.items())))
PY
done
cat $out

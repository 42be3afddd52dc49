#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for v in we1 we0; do
  d=gpurun_out/vm_$v; rm -rf $d
  MCGP_LIB=$PWD/abl/libmcgp_$v.so MCGP_BENCH_NOCHECK=1 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU --output-format csv -d $d -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --deviates 53 > /dev/null 2>&1
  python3 - $v <<'PY'
import csv,glob,collections,sys
acc=collections.defaultdict(list)
for f in glob.glob(f'gpurun_out/vm_{sys.argv[1]}/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'race_kernel' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
wl=156250*59
print(sys.argv[1], {k: round(sum(v)/len(v)/wl,1) for k,v in acc.items()})
PY
done
DEV=53 tools/ab.sh S60 abl/libmcgp_we1.so abl/libmcgp_we0.so abl/libmcgp_wcno.so

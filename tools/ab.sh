#!/bin/bash
# Same-box A/B of kernel builds (device-to-device variance is several %):  tools/ab.sh [workload] lib1.so lib2.so ...
# -> gpurun_out/ab.txt, one line per (library, repetition): kernel ms from the library's hipEvents.
# DEV=53 times the reference-width kernel; SIMS = simulations per step.  Each run's stderr is kept
# (gpurun_out/ab_<library>.<rep>.err): a variant that prints nothing has said why there.
cd "$(dirname "$0")/.."
wl=S60
case "$1" in *.so) ;; *) wl=$1; shift ;; esac
out=gpurun_out/ab.txt
mkdir -p gpurun_out
: > $out
for rep in 1 2; do
  for lib in "$@"; do
    err=gpurun_out/ab_$(basename $lib .so).$rep.err
    ms=$(MCGP_LIB=$PWD/$lib MCGP_BENCH_NOCHECK=1 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-extras \
         --deviates ${DEV:-32} --sims-per-step ${SIMS:-10000000} 2>$err | python -c "import json,sys; print(json.load(sys.stdin)['roofline']['kernel_ms_avg'])" 2>>$err)
    [ -n "$ms" ] && rm -f $err || tail -3 $err
    echo "$wl dev${DEV:-32} $(basename $lib) kernel_ms=$ms" | tee -a $out
  done
done

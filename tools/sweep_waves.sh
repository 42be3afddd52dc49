#!/bin/bash
# Waves per SIMD by measurement, per field size:  tools/sweep_waves.sh build   (here: abl/libmcgp_sw_n<N>w<W>.so for the
# candidate wave counts of every N)   then   tools/sweep_waves.sh run   (on the GPU box: N<k> workload of bench.py per
# library -> gpurun_out/sweep_waves.txt).  reg_min_waves() in race_kernel_reg.hip.h is the table read off the result.
cd "$(dirname "$0")/.."
cands() {
  local n=$1
  if [ $n -le 8 ]; then echo "4 5 6 8"; elif [ $n -le 14 ]; then echo "3 4 5"; elif [ $n -le 20 ]; then echo "3 4"; else echo "2 3"; fi
}
case "$1" in
build)
  for n in $(seq ${FROM:-2} ${TO:-32}); do
    for w in $(cands $n); do
      SIZES=$n tools/variant.sh sw_n${n}w${w} "-DMCGP_ONLY_N=$n -DMCGP_MIN_WAVES=$w" > /dev/null 2>&1 || echo "build failed n=$n w=$w"
    done
    echo "built n=$n"
  done ;;
run)
  out=gpurun_out/sweep_waves.txt
  mkdir -p gpurun_out; : > $out
  for n in $(seq ${FROM:-2} ${TO:-32}); do
    for w in $(cands $n); do
      lib=abl/libmcgp_sw_n${n}w${w}.so
      [ -f $lib ] || continue
      ms=$(MCGP_LIB=$PWD/$lib MCGP_BENCH_NOCHECK=1 python bench.py --workload N$n --steps 2 --warmup 1 --no-cpu-baseline --no-extras \
           --sims-per-step ${SIMS:-4000000} 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print(r['kernel_ms_avg'], r['kernel'], r.get('waves_per_simd'), r.get('scratch_bytes_per_lane'))")
      echo "N=$n w=$w $ms" | tee -a $out
    done
  done ;;
esac

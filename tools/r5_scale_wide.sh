#!/bin/bash
# parity at scale of the reference-width kernel on the final source (GPU box): finishing orders and a whole histogram against the oracle
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
DEVIATES=53 python tools/deep_parity.py 60000 1000000 > gpurun_out/r5_deep_parity_wide.txt 2>&1; rc=$?
tail -2 gpurun_out/r5_deep_parity_wide.txt
[ $rc -ne 0 ] && exit $rc
DEVIATES=53 python tools/exact_hist.py S60 30000000 7 5000000000 > gpurun_out/r5_exact_hist_wide.txt 2>&1; rc=$?
tail -2 gpurun_out/r5_exact_hist_wide.txt
[ $rc -ne 0 ] && exit $rc
DEVIATES=53 python tools/exact_hist.py HET 10000000 13 200000000 >> gpurun_out/r5_exact_hist_wide.txt 2>&1; rc=$?
tail -1 gpurun_out/r5_exact_hist_wide.txt
exit $rc

#!/bin/bash
# parity at scale of the default kernel on the final source (GPU box)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python tools/deep_parity.py 100000 1500000 > gpurun_out/r5_deep_parity.txt 2>&1; rc=$?
tail -2 gpurun_out/r5_deep_parity.txt
[ $rc -ne 0 ] && exit $rc
python tools/exact_hist.py S60 40000000 11 6000000000 > gpurun_out/r5_exact_hist.txt 2>&1; rc=$?
tail -1 gpurun_out/r5_exact_hist.txt
[ $rc -ne 0 ] && exit $rc
python tools/exact_hist.py S78 20000000 12 900000000 >> gpurun_out/r5_exact_hist.txt 2>&1; rc=$?
tail -1 gpurun_out/r5_exact_hist.txt
exit $rc

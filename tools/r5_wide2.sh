#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -k "reference_width or deviates" > gpurun_out/r5_wide_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r5_wide_tests.log
[ $rc -ne 0 ] && exit $rc
DEV=53 tools/ab.sh S60 monte_carlo_gp_amd/libmcgp_hip.so abl/libmcgp_wbase.so abl/libmcgp_wstep8.so abl/libmcgp_wpace10.so abl/libmcgp_wboth.so
cp gpurun_out/ab.txt gpurun_out/r5_ab_wide_batches.txt

#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -k "reference_width or deviates" > gpurun_out/r5_wide_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r5_wide_tests.log
[ $rc -ne 0 ] && exit $rc
python tools/dbg_wide_n.py $(seq 23 32) 2>&1 | grep -v amdgpu.ids
echo "== every pass through the exact path (MCGP_WIDE_EXACT=1 build), all sizes"
MCGP_LIB=$PWD/abl/libmcgp_wexact.so python tools/dbg_wide_n.py $(seq 1 32) 2>&1 | grep -v amdgpu.ids | awk '{print $1, $4, $5}' | tr '\n' ';'
echo
MCGP_LIB=$PWD/abl/libmcgp_wexact.so DEVIATES=53 python tools/deep_parity.py 20000 200000 > gpurun_out/r5_deep_parity_wide_exact.txt 2>&1; rc=$?
tail -2 gpurun_out/r5_deep_parity_wide_exact.txt
exit $rc

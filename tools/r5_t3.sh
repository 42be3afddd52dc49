#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -x -q -k "batch or lds or season or reference_width or other_field" > gpurun_out/r5_t3.log 2>&1
rc=$?
tail -15 gpurun_out/r5_t3.log
[ $rc -ne 0 ] && exit $rc
DEV=53 tools/ab.sh S60 monte_carlo_gp_amd/libmcgp_hip.so abl/libmcgp_wbase.so

#!/bin/bash
# the whole GPU suite on the current source (what the driver runs at round end)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r5_gpu_suite.log 2>&1
rc=$?
tail -8 gpurun_out/r5_gpu_suite.log
exit $rc

#!/bin/bash
# build-container helper: gpurun with a retry when no GPU slot / box is free (exit code 3: nothing was charged, nothing ran).
#   tools/gpu_retry.sh <timeout-seconds> <logfile> <command...>
t=$1; log=$2; shift 2
for attempt in 1 2 3 4 5 6 7 8 9 10 11 12; do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@" > $log 2>&1
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 120
done
exit 3

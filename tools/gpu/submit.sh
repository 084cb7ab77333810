#!/bin/bash
# tools/gpu/submit.sh <timeout-seconds> '<command>': gpurun, waiting for a free slot (exit code 3 = nothing free, nothing charged)
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3

#!/bin/bash
# usage: scratch/gpu_retry.sh <timeout-seconds> '<command>' : retries while gpurun reports no free slot (exit code 3)
t=$1; shift
for k in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3

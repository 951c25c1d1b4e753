#!/bin/bash
# usage: tools/gpurun_retry.sh <timeout_s> '<command>'  -- retries only while the pool has no free box (exit code 3: nothing ran, nothing charged)
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3

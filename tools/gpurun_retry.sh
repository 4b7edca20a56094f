#!/bin/bash
# usage: tools_gpurun_retry.sh <timeout> '<command>'   - retries while the pod's GPU slots are busy (exit 3)
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3

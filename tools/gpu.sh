#!/bin/bash
# gpurun with a retry ONLY when no box / slot was free (exit 3: nothing ran, nothing was charged).  usage: tools/gpu.sh <timeout_s> '<command>'
t=$1; shift
for i in 1 2 3 4 5 6 7 8 9 10; do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 75
done
exit 3

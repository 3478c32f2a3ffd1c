#!/bin/bash
# usage: tools/gpurun_retry.sh <logfile> <timeout> <command...>   -- retries while the pod has no free GPU slot (exit 3: nothing charged)
log=$1; to=$2; shift 2
for i in 1 2 3 4 5 6 7 8; do
  /usr/local/graft/bin/gpurun --timeout $to -- "$@" > $log 2>&1
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3

#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit 3 = nothing charged).  usage: tools/gpu_retry.sh <timeout> <log> <command...>
T=$1; LOG=$2; shift 2
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@" > $LOG 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then echo "rc=$rc" >> $LOG; echo done >> $LOG; exit $rc; fi
  sleep 45
done
echo "gave up" >> $LOG; echo done >> $LOG

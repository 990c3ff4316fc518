#!/bin/bash
# gpu_retry.sh <timeout_s> '<command>': gpurun with retries while no slot / box is free (exit code 3: nothing charged)
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "mkdir -p gpurun_out/r05; $*"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3

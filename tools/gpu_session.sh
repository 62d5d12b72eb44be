#!/bin/bash
# Runs a list of GPU steps on the gpurun box.  Ordinary failures (exit 1..123, e.g. a failing
# assertion) do not stop the session; a timeout / kill (exit >= 124) does: after a GPU step hangs
# or is killed no further GPU step is started in the same call.
# usage: tools/gpu_session.sh <name> <timeout_s> '<command>' [<name> <timeout_s> '<command>' ...]
mkdir -p gpurun_out
overall=0
while [ $# -ge 3 ]; do
  name=$1; tmo=$2; cmd=$3; shift 3
  echo "=== [$name] $(date +%T) : $cmd" | tee -a gpurun_out/session.log
  timeout -k 10 "$tmo" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== [$name] exit $rc" | tee -a gpurun_out/session.log
  tail -n 15 "gpurun_out/$name.log"
  if [ $rc -ge 124 ]; then echo "step $name timed out or was killed: stopping"; exit $rc; fi
  [ $rc -ne 0 ] && overall=$rc
done
exit $overall

#!/bin/bash
# submit one session script to gpurun; resubmit (at most 5 times, a minute apart) ONLY when no slot was free (exit 3: nothing ran, nothing charged)
# usage: tools/sessions/submit.sh <timeout_s> <script> [args]
t=$1; shift
for attempt in 1 2 3 4 5; do
  /usr/local/graft/bin/gpurun --timeout $t -- "bash $*"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  echo "[submit] no GPU slot free (attempt $attempt); waiting 60 s"; sleep 60
done
exit 3

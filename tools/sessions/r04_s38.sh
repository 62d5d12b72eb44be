#!/bin/bash
# round 4 session 38: after the rules from set 4 (the copy from 2.2 / 1.9 entries per piece, V = 2 on short f64 rows; V = 1 / 2 tiles for f64 up to 60 % jumps): set 4 and set 1 again, the whole suite
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s38; mkdir -p $O
timeout -k 10 400 python3 tools/auto_regret.py --set 4 --log $O/auto_regret_set4.jsonl > $O/auto_regret_set4.txt 2>&1; echo "set 4 exit $?"
grep -A26 "== regret" $O/auto_regret_set4.txt | cut -c1-215
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -8 $O/pytest_gpu.txt | cut -c1-250

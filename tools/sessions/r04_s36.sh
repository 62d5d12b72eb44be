#!/bin/bash
# round 4 session 36: after the rules from sets 2 / 3 (band matrices inside the cache, mean >= 2, all-long rows keep the row-tile kernel): both sets again, the whole suite
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s36; mkdir -p $O
timeout -k 10 500 python3 tools/auto_regret.py --set 2 --log $O/auto_regret_set2.jsonl > $O/auto_regret_set2.txt 2>&1; echo "set 2 exit $?"
grep -A30 "== regret" $O/auto_regret_set2.txt | cut -c1-215
timeout -k 10 500 python3 tools/auto_regret.py --set 3 --log $O/auto_regret_set3.jsonl > $O/auto_regret_set3.txt 2>&1; echo "set 3 exit $?"
grep -A30 "== regret" $O/auto_regret_set3.txt | cut -c1-215
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -8 $O/pytest_gpu.txt | cut -c1-250

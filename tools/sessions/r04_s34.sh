#!/bin/bash
# round 4 session 34: the regret table's second set -- sizes inside the cache, very short rows, rows of 10..60 without / with runs, long rows, rectangular
set -o pipefail
mkdir -p gpurun_out/r4s34
cd /root/repo
timeout -k 10 1000 python3 tools/auto_regret.py --set 2 --log gpurun_out/r4s34/auto_regret_set2.jsonl > gpurun_out/r4s34/auto_regret_set2.txt 2>&1; echo "regret exit $?"
grep -A32 "== regret" gpurun_out/r4s34/auto_regret_set2.txt | cut -c1-250

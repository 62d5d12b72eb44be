#!/bin/bash
# round 4 session 39: the regret table's first set once more, on the tree with every rule of sessions 19-38
set -o pipefail
mkdir -p gpurun_out/r4s39
cd /root/repo
timeout -k 10 1130 python3 tools/auto_regret.py --log gpurun_out/r4s39/auto_regret.jsonl > gpurun_out/r4s39/auto_regret.txt 2>&1; echo "regret exit $?"
grep -A45 "== regret" gpurun_out/r4s39/auto_regret.txt | cut -c1-215

#!/bin/bash
# round 4 session 37: the regret table's fourth set -- column runs of other lengths, 50..90 per row, an arrow matrix, a stencil with holes
set -o pipefail
mkdir -p gpurun_out/r4s37
cd /root/repo
timeout -k 10 1000 python3 tools/auto_regret.py --set 4 --log gpurun_out/r4s37/auto_regret_set4.jsonl > gpurun_out/r4s37/auto_regret_set4.txt 2>&1; echo "regret exit $?"
grep -A26 "== regret" gpurun_out/r4s37/auto_regret_set4.txt | cut -c1-215

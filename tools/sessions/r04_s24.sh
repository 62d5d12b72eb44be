#!/bin/bash
# round 4 session 24: the regret table again, on the tree with the rules it led to
set -o pipefail
mkdir -p gpurun_out/r4s24
cd /root/repo
timeout -k 10 1100 python3 tools/auto_regret.py --log gpurun_out/r4s24/auto_regret.jsonl > gpurun_out/r4s24/auto_regret.txt 2>&1; echo "regret exit $?"
grep -v amdgpu.ids gpurun_out/r4s24/auto_regret.txt | tail -48 | cut -c1-260

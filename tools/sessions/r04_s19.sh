#!/bin/bash
# round 4 session 19: what the offline selection gives away against a per-matrix search (tools/auto_regret.py)
set -o pipefail
mkdir -p gpurun_out/r4s19
cd /root/repo
timeout -k 10 1050 python3 tools/auto_regret.py --log gpurun_out/r4s19/auto_regret.jsonl > gpurun_out/r4s19/auto_regret.txt 2>&1; echo "regret exit $?"
grep -v amdgpu.ids gpurun_out/r4s19/auto_regret.txt | tail -50

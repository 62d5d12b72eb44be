#!/bin/bash
# round 4 session 40: does the per-bucket ELL / DIA table generalise?  (tools/format_regret.py)
set -o pipefail
mkdir -p gpurun_out/r4s40
cd /root/repo
timeout -k 10 1100 python3 tools/format_regret.py --log gpurun_out/r4s40/format_regret.jsonl 2>&1 | tee gpurun_out/r4s40/format_regret.txt | grep -v amdgpu.ids | cut -c1-330

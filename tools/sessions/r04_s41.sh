#!/bin/bash
# round 4 session 41: what the plans cost to make, on the final tree (the stencil plans now build a partition and, with the columns, look at the columns first)
set -o pipefail
mkdir -p gpurun_out/r4s41
cd /root/repo
timeout -k 10 600 python3 tools/plan_cost_probe.py 2>&1 | tee gpurun_out/r4s41/plan_cost.txt | grep -v amdgpu.ids | cut -c1-220

#!/bin/bash
# round 4 session 32: the regret table's rows for equal row lengths WITHOUT a stencil's columns (the stencil rules see only the lengths)
set -o pipefail
mkdir -p gpurun_out/r4s32
cd /root/repo
timeout -k 10 800 python3 tools/auto_regret.py --only "per row exactly" --log gpurun_out/r4s32/auto_regret_equal_lengths.jsonl 2>&1 | tee gpurun_out/r4s32/auto_regret_equal_lengths.txt | grep -v amdgpu.ids | cut -c1-250

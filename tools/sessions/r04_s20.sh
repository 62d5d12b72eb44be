#!/bin/bash
# round 4 session 20: stencil rows -- csr_wave against wave tiles V = 1 / 2 and the run-compressed copy (A/B for a rule change); the regret run's last three matrices
set -o pipefail
mkdir -p gpurun_out/r4s20
cd /root/repo
timeout -k 10 600 python3 tools/stencil_tiles_probe.py > gpurun_out/r4s20/stencil_tiles.txt 2>&1; echo "probe exit $?"
grep -v amdgpu.ids gpurun_out/r4s20/stencil_tiles.txt | cut -c1-330
timeout -k 10 400 python3 tools/auto_regret.py --only "8 rows,scattered,cache-resident" --log gpurun_out/r4s20/auto_regret_rest.jsonl > gpurun_out/r4s20/auto_regret_rest.txt 2>&1; echo "regret exit $?"
grep -v amdgpu.ids gpurun_out/r4s20/auto_regret_rest.txt | tail -12 | cut -c1-300

#!/bin/bash
# round 4 session 28: XCD dealing of the wave tiles on configs[4]'s per-rank block (rows reach 10000 columns to either side); the stencil rule's size gate
set -o pipefail
mkdir -p gpurun_out/r4s28
cd /root/repo
timeout -k 10 400 python3 tools/rank_block_dealing_probe.py > gpurun_out/r4s28/rank_block_dealing.txt 2>&1; echo "dealing probe exit $?"
grep -v amdgpu.ids gpurun_out/r4s28/rank_block_dealing.txt | cut -c1-220
CMI_CSR_WAVE_VEC=0 timeout -k 10 400 python3 tools/stencil_tiles_probe.py --matrices 5pt_1500,5pt_2000,5pt_2400,5pt_2800 --rounds 3 --cg-iterations 50 > gpurun_out/r4s28/stencil_size_gate.txt 2>&1; echo "gate probe exit $?"
grep -v amdgpu.ids gpurun_out/r4s28/stencil_size_gate.txt | cut -c1-300

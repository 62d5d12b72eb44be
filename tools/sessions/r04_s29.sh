#!/bin/bash
# round 4 session 29: the XCD-dealing sweeps AGAIN -- an asked-for dealing of a csr_wavev plan was silently replaced by the table's until now (plan.hip), so
# sessions 22 / 28 measured the same launch ten times
set -o pipefail
mkdir -p gpurun_out/r4s29
cd /root/repo
timeout -k 10 400 python3 tools/rank_block_dealing_probe.py > gpurun_out/r4s29/rank_block_dealing.txt 2>&1; echo "dealing probe exit $?"
grep -v amdgpu.ids gpurun_out/r4s29/rank_block_dealing.txt | cut -c1-220
timeout -k 10 420 python3 tools/stencil_tiles_probe.py --matrices 5pt --sweep --rounds 3 --cg-iterations 10 > gpurun_out/r4s29/headline_wave_tiles_sweep.txt 2>&1; echo "sweep exit $?"
grep -v amdgpu.ids gpurun_out/r4s29/headline_wave_tiles_sweep.txt | cut -c1-200

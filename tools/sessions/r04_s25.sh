#!/bin/bash
# round 4 session 25: f32 stencil rows -- csr_wave against wave tiles V = 1 / 2 (the regret table after the rule changes still shows V = 2 ahead by 2-5 % there)
set -o pipefail
mkdir -p gpurun_out/r4s25
cd /root/repo
CMI_CSR_WAVE_VEC=0 timeout -k 10 420 python3 tools/stencil_tiles_probe.py --matrices 7pt32,5pt32 > gpurun_out/r4s25/stencil_tiles_f32.txt 2>&1; echo "probe exit $?"
grep -v amdgpu.ids gpurun_out/r4s25/stencil_tiles_f32.txt | cut -c1-330

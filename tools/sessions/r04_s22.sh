#!/bin/bash
# round 4 session 22: the stencil A/B again on the tree WITH the new rules (AUTO = wave tiles V = 1 on the f64 5- / 7-point matrices; CMI_CSR_WAVE_VEC=0
# in a second process gives csr_wave back as "AUTO plan") with the CG column repaired, and a cache-policy x XCD-dealing sweep of the new headline kernel
set -o pipefail
mkdir -p gpurun_out/r4s22
cd /root/repo
CMI_CSR_WAVE_VEC=0 timeout -k 10 420 python3 tools/stencil_tiles_probe.py --matrices 5pt,7pt,3pt,5pt32 > gpurun_out/r4s22/stencil_tiles_ab.txt 2>&1; echo "probe exit $?"
grep -v amdgpu.ids gpurun_out/r4s22/stencil_tiles_ab.txt | cut -c1-330
timeout -k 10 420 python3 tools/stencil_tiles_probe.py --matrices 5pt --sweep --rounds 3 --cg-iterations 10 > gpurun_out/r4s22/headline_wave_tiles_sweep.txt 2>&1; echo "sweep exit $?"
grep -v amdgpu.ids gpurun_out/r4s22/headline_wave_tiles_sweep.txt | cut -c1-250

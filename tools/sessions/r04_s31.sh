#!/bin/bash
# round 4 session 31: the plan-less probe in three settings (table / + csr_wave rule / + 16-byte-vector wave tiles on fixed row ranges)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s31; mkdir -p $O
timeout -k 10 900 python3 tools/planless_wave_probe.py 2>&1 | tee $O/planless_wave_rule.txt | grep -v amdgpu.ids | cut -c1-240

#!/bin/bash
# round 4 session 30: csr_wavef -- the 16-byte-vector wave tiles on fixed row ranges, WITHOUT a plan: its tests, and the plan-less probe in three settings
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s30; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_round4_gpu.py tests/test_round3_gpu.py -m gpu -x -q -k "fixed_row_ranges or plan_less_call or refusals or overflow or wavev" > $O/pytest.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -12 $O/pytest.txt | cut -c1-250
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python3 tools/planless_wave_probe.py > $O/planless_wave_rule.txt 2>&1; echo "probe exit $?"; grep -v amdgpu.ids $O/planless_wave_rule.txt | cut -c1-230

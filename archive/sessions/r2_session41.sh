#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s41; mkdir -p $O
timeout -k 10 300 python tools/plan_cost_probe.py > $O/plan_cost.txt 2>&1; echo "exit $?"; grep -v amdgpu.ids $O/plan_cost.txt | cut -c1-200

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s36; mkdir -p $O
timeout -k 10 600 python tools/cg_dot_shape_probe.py --wide > $O/cg_dot_shape_wide.txt 2>&1; echo "exit $?"; grep -v amdgpu.ids $O/cg_dot_shape_wide.txt | cut -c1-200

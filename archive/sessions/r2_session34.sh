#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s34; mkdir -p $O
timeout -k 10 600 python tools/cg_dot_shape_probe.py > $O/cg_dot_shape.txt 2>&1; echo "exit $?"; grep -v amdgpu.ids $O/cg_dot_shape.txt | cut -c1-200

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s43; mkdir -p $O
timeout -k 10 400 python tools/hyb_shape_probe.py > $O/hyb_shape.txt 2>&1; echo "exit $?"; grep -v amdgpu.ids $O/hyb_shape.txt | cut -c1-120

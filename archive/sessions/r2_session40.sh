#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s40; mkdir -p $O
timeout -k 10 800 python tools/soak.py > $O/soak.txt 2>&1; echo "soak exit $?"; grep -v amdgpu.ids $O/soak.txt | cut -c1-200

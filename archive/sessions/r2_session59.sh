#!/bin/bash
# session 59: the wave-tile kernel against the csr_stream entry it replaces, per stencil and value type (tools/wave_ab.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s59; mkdir -p $O
timeout -k 10 500 python tools/wave_ab.py > $O/wave_ab.txt 2>&1; echo "exit $?"; grep -v amdgpu.ids $O/wave_ab.txt

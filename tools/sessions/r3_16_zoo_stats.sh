#!/bin/bash
# r3 session 16: the A/B zoo again with the column statistics printed and a large unstructured FEM matrix added -> auto rule for csr_wavex
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s16; mkdir -p $O
timeout -k 10 900 python3 tools/wavev_ab.py > $O/wavev_ab.txt 2>&1; echo "ab exit $?"; grep -v amdgpu.ids $O/wavev_ab.txt | cut -c1-140

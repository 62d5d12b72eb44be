#!/bin/bash
# r3 session 10: (1) the C++ sharded layer with 2 / 3 ranks SHARING the GPU: staged collectives, real IPC pulls, one-sided fused CG;
# (2) tools/wavev_ab.py: csr_wavev against today's auto plan across the matrix zoo -> the plan's auto rule
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s10; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_cpp_layer.py -m gpu -x -q -k "sharded" > $O/pytest_sharded.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -n 25 $O/pytest_sharded.txt
[ $rc -ge 124 ] && exit $rc
timeout -k 10 700 python3 tools/wavev_ab.py > $O/wavev_ab.txt 2>&1; echo "ab exit $?"; grep -v amdgpu.ids $O/wavev_ab.txt | cut -c1-420

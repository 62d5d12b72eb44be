#!/bin/bash
# r3 session 14: csr_wavex (x window in LDS): parity, then the matrix zoo A/B beside csr_wavev and the auto plan
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s14; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_round3_gpu.py -m gpu -x -q -k "wavex or wavev or table" > $O/pytest.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -n 12 $O/pytest.txt | cut -c1-250
[ $rc -eq 0 ] || exit 1
timeout -k 10 800 python3 tools/wavev_ab.py > $O/wavev_ab.txt 2>&1; echo "ab exit $?"; grep -v amdgpu.ids $O/wavev_ab.txt | cut -c1-560

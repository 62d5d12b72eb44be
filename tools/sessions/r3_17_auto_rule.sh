#!/bin/bash
# r3 session 17: the auto rule with the column profile (csr_wavex / V choice): round-3 tests, then the zoo with the table's csr_stream beside everything
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s17; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_round3_gpu.py tests/test_plan_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -n 12 $O/pytest.txt | cut -c1-250
[ $rc -ge 124 ] && exit $rc
timeout -k 10 900 python3 tools/wavev_ab.py > $O/wavev_ab.txt 2>&1; echo "ab exit $?"; grep -v amdgpu.ids $O/wavev_ab.txt | cut -c1-330

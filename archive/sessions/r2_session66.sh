#!/bin/bash
# session 66: csr_wave on a plan-built partition (irregular short rows): plan / csr16 / spmv tests, then tools/wavep_ab.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s66; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_plan_gpu.py tests/test_csr16_gpu.py -m gpu -x -q > $O/pytest_subset.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -n 8 $O/pytest_subset.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 500 python tools/wavep_ab.py > $O/wavep_ab.txt 2>&1; echo "ab exit $?"; grep -v amdgpu.ids $O/wavep_ab.txt | cut -c1-300

#!/bin/bash
# r3 session 32: the tests added since the last full run (cusp/sort.h and cusp/format_utils.h on device arrays through the C++ device build, cmi_csr_interior_rows,
# the explicit partition request) -- then the whole -m gpu suite
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s32; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?
echo "pytest exit $rc"; tail -n 12 $O/pytest_gpu.txt | cut -c1-300

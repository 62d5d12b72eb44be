#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s42; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -n 8 $O/pytest_gpu.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python tools/suitesparse_sweep.py > $O/suitesparse_like_sweep.txt 2>&1; echo "sweep exit $?"; grep -E "^==|csr_vector|table|plan" $O/suitesparse_like_sweep.txt | cut -c1-120

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s10; mkdir -p $O
tools/bin/r2_probe --only "lib csr table|lib csr_dot|lib ell table|lib dia table" > $O/probe.txt 2>&1; cat $O/probe.txt
timeout -k 10 900 python tools/suitesparse_sweep.py > $O/suitesparse_like_sweep.txt 2>&1; rc=$?; echo "sweep exit $rc"; grep -E "^==|table|plan|vectors/lane=2|\*" $O/suitesparse_like_sweep.txt | cut -c1-150
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -n 8 $O/pytest_gpu.txt

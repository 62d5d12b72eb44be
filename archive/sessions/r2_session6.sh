#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s6; mkdir -p $O
timeout -k 10 180 tools/bin/r2_probe --only "lib coo|lib csr table|lib csr_dot|csrx flags 1 rpb 176|lib ell table|lib dia table" > $O/probe_timing.txt 2>&1 || { echo probe failed; tail -5 $O/probe_timing.txt; exit 1; }
cat $O/probe_timing.txt
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -n 25 $O/pytest_gpu.txt

#!/bin/bash
# r3 session 18: the FULL -m gpu suite on the tree with the final auto rule and table
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s18; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; rc=$?; echo "full pytest exit $rc"; tail -n 30 $O/pytest_gpu.txt | cut -c1-250

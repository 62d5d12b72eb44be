#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s39; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_plan_gpu.py -m gpu -x -q > $O/pytest_plan.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -n 8 $O/pytest_plan.txt

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s27; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_plan_gpu.py tests/test_cg_gpu.py tests/test_cpp_layer.py -m gpu -x -q > $O/pytest_hybdot.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -n 12 $O/pytest_hybdot.txt
[ $rc -eq 0 ] || exit 1
tools/bin/cg_bench --iterations=100 --format=hyb 2>&1 | grep fused | tail -1

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s26; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_spmv_gpu.py tests/test_cg_gpu.py tests/test_cpp_layer.py -m gpu -x -q -k "dot or cg or cpp" > $O/pytest_dot.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -n 8 $O/pytest_dot.txt

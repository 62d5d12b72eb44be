#!/bin/bash
# r3 session 38: the reference's example programs (16 now) on the device through tests/test_cpp_layer.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s38; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_cpp_layer.py -m gpu -x -q -k "examples" > $O/pytest_examples.txt 2>&1; rc=$?
echo "pytest exit $rc"; tail -n 12 $O/pytest_examples.txt | cut -c1-300

#!/bin/bash
# r3 session 26: the property-based parity tests (tests/test_fuzz_gpu.py: hypothesis-drawn shapes x every multiply path against the oracle)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s26; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_fuzz_gpu.py -m gpu -x -q > $O/pytest_fuzz.txt 2>&1; rc=$?
echo "pytest exit $rc"; tail -n 40 $O/pytest_fuzz.txt | cut -c1-400

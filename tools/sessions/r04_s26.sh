#!/bin/bash
# round 4 session 26: the whole -m gpu suite on the tree with the f32 stencil rule
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s26; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -15 $O/pytest_gpu.txt | cut -c1-250

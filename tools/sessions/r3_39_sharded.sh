#!/bin/bash
# r3 session 39: the sharded C++ layer on the device with the sharded bicgstab: 1 rank through RCCL, 2 / 3 ranks sharing the GPU
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s39; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_cpp_layer.py -m gpu -x -q -k "sharded" > $O/pytest_sharded.txt 2>&1; rc=$?
echo "sharded pytest exit $rc"; tail -n 8 $O/pytest_sharded.txt | cut -c1-300

#!/bin/bash
# r3 session 12: sharded C++ layer with 2 / 3 ranks sharing the GPU (after the barrier fix), then the FULL -m gpu suite on the tree
# with the csr_wavev auto rule, the re-set table entry and the Python layer rebased on binding.Comm
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s12; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_cpp_layer.py -m gpu -x -q -k "sharded" > $O/pytest_sharded.txt 2>&1; rc=$?; echo "sharded pytest exit $rc"; tail -n 30 $O/pytest_sharded.txt | cut -c1-250
[ $rc -ge 124 ] && exit $rc
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; rc=$?; echo "full pytest exit $rc"; tail -n 30 $O/pytest_gpu.txt | cut -c1-250

#!/bin/bash
# r3 session 34: the multiply's other callers on the device: the C++ device build (BLAS-1 rest of the set, precond::diagonal, preconditioned cg, cr, bicgstab,
# sort / format_utils / verify) -- then the whole -m gpu suite
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s34; mkdir -p $O
timeout -k 10 600 tests/cpp/bin/test_device > $O/test_device.txt 2>&1; rc=$?
echo "test_device exit $rc"; grep -v "^    ok\|passed" $O/test_device.txt | tail -n 25 | cut -c1-300
[ $rc -ne 0 ] && exit $rc
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?
echo "pytest exit $rc"; tail -n 8 $O/pytest_gpu.txt | cut -c1-300

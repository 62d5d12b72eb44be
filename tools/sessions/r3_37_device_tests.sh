#!/bin/bash
# r3 session 37: the C++ device build with transpose / bicg / the cross-solver test
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s37; mkdir -p $O
timeout -k 10 600 tests/cpp/bin/test_device > $O/test_device.txt 2>&1; rc=$?
echo "test_device exit $rc"; tail -n 2 $O/test_device.txt
[ $rc -ne 0 ] && grep -i -B2 -A8 "fail" $O/test_device.txt | head -60
exit $rc

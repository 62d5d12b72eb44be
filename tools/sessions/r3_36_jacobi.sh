#!/bin/bash
# r3 session 36: the fused Jacobi-preconditioned cg on the device: tests (fused vs operation-by-operation: iteration counts, solutions), then cg_bench --solvers
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s36; mkdir -p $O
timeout -k 10 600 tests/cpp/bin/test_device > $O/test_device.txt 2>&1; rc=$?
echo "test_device exit $rc"; tail -n 2 $O/test_device.txt
[ $rc -ne 0 ] && { grep -i -B2 -A6 "fail" $O/test_device.txt | head -40; exit $rc; }
timeout -k 10 600 tools/bin/cg_bench --solvers --iterations=200 > $O/cg_bench_solvers.txt 2>&1; echo "solvers exit $?"; cat $O/cg_bench_solvers.txt

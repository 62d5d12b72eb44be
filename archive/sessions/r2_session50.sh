#!/bin/bash
# session 50: the 16-byte-vector single-pass bodies (csr_stream IPT 1/2/4, csr_stream16) made branch-free: tests, then the matrix
# kinds of stream_shape_ab (the plan's time is the number: ldoor-like 89.0, nlpkkt-like 206.5, 27-point 82.0 / 61.9 us before)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s50; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_spmv_gpu.py tests/test_csr16_gpu.py tests/test_plan_gpu.py tests/test_cg_gpu.py -m gpu -x -q > $O/pytest_gpu_subset.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -n 6 $O/pytest_gpu_subset.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 500 python tools/stream_shape_ab.py > $O/stream_shape_ab.txt 2>&1; echo "ab exit $?"; grep -v amdgpu.ids $O/stream_shape_ab.txt | grep -v "policy [367] "
timeout -k 10 300 python tools/c16_sweep.py --matrix ldoor --quick > $O/c16_ldoor.txt 2>&1; echo "c16 exit $?"; grep -v amdgpu.ids $O/c16_ldoor.txt | tail -8

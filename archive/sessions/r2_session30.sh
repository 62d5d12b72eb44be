#!/bin/bash
# HYB as two launches now runs its COO half through a COO plan (row offsets + CSR kernels): tests, the one-vs-two probe again, the width rule re-fit
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s30; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_plan_gpu.py tests/test_spmv_gpu.py tests/test_cpp_layer.py -m gpu -x -q -k "hyb or plan or cpp or suitesparse or golden" > $O/pytest_hyb.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -n 10 $O/pytest_hyb.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 900 python tools/hyb_fuse_probe.py > $O/hyb_one_vs_two.txt 2>&1; echo "probe exit $?"; grep -v amdgpu.ids $O/hyb_one_vs_two.txt | cut -c1-200

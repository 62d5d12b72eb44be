#!/bin/bash
# r3 session 20: the COO container's device sort (cmi_coo_sort_by_row_*): its tests (Python + the reference's coo_matrix.cu cases through the C++ device
# build + the new C++ test), then what it costs on the headline matrix's 50 M entries in a random order
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s20; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_round3_gpu.py tests/test_spmv_gpu.py tests/test_plan_gpu.py tests/test_cpp_layer.py -m gpu -x -q -k "sort or coo or Coo or device_tests" > $O/pytest_sort.txt 2>&1; rc=$?
echo "pytest exit $rc"; tail -n 15 $O/pytest_sort.txt | cut -c1-300
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python3 tools/coo_sort_time.py > $O/coo_sort_time.txt 2> $O/coo_sort_time.err; echo "time exit $?"; cat $O/coo_sort_time.txt | cut -c1-400; tail -3 $O/coo_sort_time.err

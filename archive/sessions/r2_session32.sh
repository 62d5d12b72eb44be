#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s32; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -n 12 $O/pytest_gpu.txt
[ $rc -eq 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -n 1 $O/smoke.txt
timeout -k 10 300 tools/bin/spmv_bench --grid=3162 > $O/spmv_bench_cpp.txt 2>&1; echo "spmv_bench exit $?"; grep "coo\|csr\|hyb\|ell\|dia" $O/spmv_bench_cpp.txt | cut -c1-160

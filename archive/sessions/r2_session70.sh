#!/bin/bash
# session 70: DIA two-rows-per-lane kernel with ONE request for the row pair's two x values (this tree) against two (HEAD, tmp_prev/):
# the DIA tests, then r2_probe "lib dia" from both trees, interleaved
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s70; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_spmv_gpu.py tests/test_cg_gpu.py -m gpu -x -q -k "dia or golden or reference or formats or 1e8 or full_size or cg" > $O/pytest_dia.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -n 5 $O/pytest_dia.txt
[ $rc -eq 0 ] || exit 1
for r in 1 2 3; do
  echo "prev: $(tmp_prev/tools/bin/r2_probe --only 'lib dia table' | grep median)"
  echo "new : $(tools/bin/r2_probe --only 'lib dia table' | grep median)"
done 2>&1 | tee $O/dia_ab.txt

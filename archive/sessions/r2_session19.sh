#!/bin/bash
# fold-ahead CG steps: parity first (bounded spins: a wrong hand-off shows as a NaN, not a hang), then timing on / off
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s19; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_cg_gpu.py -m gpu -x -q -k "fold_ahead or quickstart or fused_equals or float32" > $O/pytest_fold.txt 2>&1; rc=$?; echo "pytest(fold) exit $rc"; tail -n 15 $O/pytest_fold.txt
[ $rc -eq 0 ] || exit 1
for fa in 1 0 1 0; do
  echo "== CMI_CG_FOLD_AHEAD=$fa plain"; CMI_CG_FOLD_AHEAD=$fa timeout -k 10 120 tools/bin/cg_bench --iterations=200 2>&1 | grep fused
  echo "== CMI_CG_FOLD_AHEAD=$fa 16-bit columns"; CMI_CG_FOLD_AHEAD=$fa CMI_COMPRESS_INDICES=1 timeout -k 10 120 tools/bin/cg_bench --iterations=200 2>&1 | grep fused
done > $O/cg_fold_ahead.txt 2>&1
cat $O/cg_fold_ahead.txt
timeout -k 10 300 python -m pytest tests/test_cpp_layer.py tests/test_cg_gpu.py -m gpu -x -q > $O/pytest_cg.txt 2>&1; rc=$?; echo "pytest(cg+cpp) exit $rc"; tail -n 6 $O/pytest_cg.txt

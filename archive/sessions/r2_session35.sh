#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s35; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_cg_gpu.py tests/test_csr16_gpu.py tests/test_cpp_layer.py -m gpu -x -q > $O/pytest_cg.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -n 6 $O/pytest_cg.txt
[ $rc -eq 0 ] || exit 1
for round in 1 2 3; do
  for sw in 0 32; do
    for c16 in 0 1; do
      echo "== round $round dot swizzle $sw c16 $c16: $(CMI_DOT_SWIZZLE=$sw CMI_COMPRESS_INDICES=$c16 timeout -k 10 120 tools/bin/cg_bench --iterations=300 2>&1 | grep fused | tail -1)"
    done
  done
  echo "== round $round fold-ahead + swizzle 0: $(CMI_CG_FOLD_AHEAD=1 timeout -k 10 120 tools/bin/cg_bench --iterations=300 2>&1 | grep fused | tail -1)"
  echo "== round $round fold-ahead + swizzle 0 + c16: $(CMI_CG_FOLD_AHEAD=1 CMI_COMPRESS_INDICES=1 timeout -k 10 120 tools/bin/cg_bench --iterations=300 2>&1 | grep fused | tail -1)"
done > $O/cg_dot_swizzle.txt 2>&1
cat $O/cg_dot_swizzle.txt | cut -c1-160

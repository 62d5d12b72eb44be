#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s38; mkdir -p $O
for round in 1 2 3; do
  for f in ell dia; do
    for sw in table 0 16 64; do
      if [ $sw = table ]; then unset CMI_DOT_SWIZZLE; else export CMI_DOT_SWIZZLE=$sw; fi
      echo "== round $round $f dot swizzle $sw: $(timeout -k 10 120 tools/bin/cg_bench --iterations=200 --format=$f 2>&1 | grep fused | tail -1 | cut -c1-150)"
    done
  done
done > $O/cg_dot_swizzle_ell_dia.txt 2>&1
cat $O/cg_dot_swizzle_ell_dia.txt

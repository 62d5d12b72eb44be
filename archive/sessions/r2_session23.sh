#!/bin/bash
# cache policy (bit 0: nt loads of the matrix streams, bit 1: nt y stores) of the fused SpMV+dot instance inside CG, interleaved
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s23; mkdir -p $O
for round in 1 2 3; do
  for pol in 2 0 1 3; do
    for c16 in 0 1; do
      echo "== round $round policy $pol c16 $c16: $(CMI_DOT_POLICY=$pol CMI_COMPRESS_INDICES=$c16 timeout -k 10 120 tools/bin/cg_bench --iterations=200 2>&1 | grep fused | tail -1)"
    done
  done
done > $O/cg_dot_policy.txt 2>&1
cat $O/cg_dot_policy.txt | cut -c1-150

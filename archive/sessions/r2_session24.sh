#!/bin/bash
# load hints of the CG vector kernels (bit 3: x loads nt, bit 4: y loads nt in update, bit 5: r loads nt in direction; bit 0: x stores nt)
# with the SpMV+dot instance on nt matrix loads (policy 3), interleaved
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s24; mkdir -p $O
for round in 1 2 3; do
  for sp in 1 9 17 25 57 24 0; do
    for c16 in 0 1; do
      echo "== round $round store/load policy $sp c16 $c16: $(CMI_CG_STORE_POLICY=$sp CMI_DOT_POLICY=3 CMI_COMPRESS_INDICES=$c16 timeout -k 10 120 tools/bin/cg_bench --iterations=200 2>&1 | grep fused | tail -1)"
    done
  done
done > $O/cg_vector_load_policy.txt 2>&1
cat $O/cg_vector_load_policy.txt | cut -c1-150

#!/bin/bash
# y-store hint of the fused SpMV+dot instance inside CG: table's (nt) vs plain vs nt forced, plain CSR and 16-bit columns, interleaved
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s22; mkdir -p $O
for round in 1 2 3; do
  for nt in table 0 1; do
    for c16 in 0 1; do
      if [ $nt = table ]; then unset CMI_DOT_STORE_NT; else export CMI_DOT_STORE_NT=$nt; fi
      echo "== round $round y-store $nt c16 $c16: $(CMI_COMPRESS_INDICES=$c16 timeout -k 10 120 tools/bin/cg_bench --iterations=200 2>&1 | grep fused | tail -1)"
    done
  done
done > $O/cg_y_store_policy.txt 2>&1
cat $O/cg_y_store_policy.txt | cut -c1-150

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s9; mkdir -p $O
for rep in 1 2; do for pad in 0 0.5 1 3 7 16 33 100; do echo -n "pad $pad MB: "; CG_BENCH_PAD_MB=$pad tools/bin/cg_bench --iterations=300 | grep fused | tail -1; done; done | tee $O/cg_pad.txt

#!/bin/bash
# session 57: the CG iteration with the wave-tile kernel's fused-dot instance against csr_stream's, and its XCD dealing / cache policy
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s57; mkdir -p $O
run() { echo -n "$1: "; env $2 tools/bin/cg_bench --iterations=300 2>&1 | grep fused | tail -1 | sed 's/.*iterations in//' | cut -c1-150; }
for rep in 1 2; do
  run "wave, launch order      " "CMI_X=0"
  run "csr_stream (CMI_CSR_WAVE=0)" "CMI_CSR_WAVE=0"
  run "wave, chunks of 16      " "CMI_DOT_SWIZZLE=16"
  run "wave, chunks of 64      " "CMI_DOT_SWIZZLE=64"
  run "wave, chunks of 128     " "CMI_DOT_SWIZZLE=128"
  run "wave, policy 2          " "CMI_DOT_POLICY=2"
  run "wave, policy 1          " "CMI_DOT_POLICY=1"
done 2>&1 | tee $O/cg_wave_dot.txt

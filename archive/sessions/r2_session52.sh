#!/bin/bash
# session 52: where are the 8 us between the lane-strided multiply (124) and its dependence-free twin (116)?  csrd ablations
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s52; mkdir -p $O
timeout -k 10 200 tools/bin/r2_probe --only "lib csr table|csrd ablation|shape 2 nt-load 1 rows 176 swz 64" > $O/ablation.txt 2>&1; echo "probe exit $?"; cat $O/ablation.txt

#!/bin/bash
# r3 session 31: tools/soak.py -- every kernel thousands of times, results must never change (round 3's kernels, fenced folds and device sort included)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s31; mkdir -p $O
timeout -k 10 1000 python3 tools/soak.py > $O/soak.txt 2> $O/soak.err; echo "soak exit $?"; tail -3 $O/soak.err; cat $O/soak.txt

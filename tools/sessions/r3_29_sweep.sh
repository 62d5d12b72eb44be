#!/bin/bash
# r3 session 29: the configs[3] sweep table of round 2 (tools/suitesparse_sweep.py: every CSR variant on the three full-size stand-ins) with round 3's kernels and table
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s29; mkdir -p $O
timeout -k 10 900 python3 tools/suitesparse_sweep.py --shapes > $O/suitesparse_like_sweep.txt 2> $O/sweep.err; echo "sweep exit $?"; tail -3 $O/sweep.err
grep -n "==\|\*\|plan:\|wavev\|wavex\|table" $O/suitesparse_like_sweep.txt | cut -c1-200

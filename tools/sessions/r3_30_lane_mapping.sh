#!/bin/bash
# r3 session 30: does the lane -> entry mapping of the gathers matter?  csr_wave's LANE-STRIDED body on a plan-built partition (csr_wavep, k = 8 / 10 entries per lane: a
# gather instruction covers 64 CONSECUTIVE entries) against csr_wavev (pairs: lane l holds entries 2l, 2l+1 of a 128-entry span) at similar tile sizes, nlpkkt120-like
# (longest row 28: csr_wavep admits it) -- the cheap test before building a lane-strided csr_wavev
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s30; mkdir -p $O
PMC_WAVEP=6,8,10 PMC_WAVEP_POL=2,3 PMC_WAVEV=2,4 PMC_WAVEV_POL=2,3 timeout -k 10 400 python3 tools/pmc_matrix_probe.py nlpkkt120 --time > $O/lane_mapping.txt 2> $O/lane_mapping.err; echo "exit $?"; tail -2 $O/lane_mapping.err
grep "^TIME\|^#" $O/lane_mapping.txt | cut -c1-260

#!/bin/bash
# r3 session 7: csr_wavev WITHOUT nt loads (session 6: the nt hint costs csr_stream's 16-byte-vector body 13-17 % on these matrices --
# its two value vectors per lane touch every line twice) beside csr_stream, same box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s07; mkdir -p $O
PMC_WAVEV=2,4 PMC_WAVEV_POL=2,3 PMC_WAVEV_SWZ=0,16 timeout -k 10 500 python3 tools/pmc_matrix_probe.py ldoor,nlpkkt120 --time > $O/time.txt 2>&1; grep -E "^TIME" $O/time.txt | cut -c1-110
PMC_WAVEV=1,2 PMC_WAVEV_POL=2,6 PMC_WAVEV_SWZ=0,64 timeout -k 10 300 python3 tools/pmc_matrix_probe.py thermal2 --time > $O/time_thermal2.txt 2>&1; grep -E "^TIME" $O/time_thermal2.txt | cut -c1-110

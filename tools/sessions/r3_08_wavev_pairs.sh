#!/bin/bash
# r3 session 8: csr_wavev with the one-line-per-instruction request shape (f64: int2 + double2 pairs), with and without nt loads
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s08; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_round3_gpu.py -m gpu -x -q -k "wavev or validate or replans or fold" > $O/pytest.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -n 6 $O/pytest.txt
[ $rc -eq 0 ] || exit 1
PMC_WAVEV=2,4 PMC_WAVEV_POL=2,3 timeout -k 10 500 python3 tools/pmc_matrix_probe.py ldoor,nlpkkt120 --time > $O/time.txt 2>&1; grep -E "^TIME" $O/time.txt | cut -c1-110
PMC_WAVEV=1,2 PMC_WAVEV_POL=2,3 PMC_WAVEV_SWZ=64 timeout -k 10 300 python3 tools/pmc_matrix_probe.py thermal2 --time > $O/time_thermal2.txt 2>&1; grep -E "^TIME" $O/time_thermal2.txt | cut -c1-110

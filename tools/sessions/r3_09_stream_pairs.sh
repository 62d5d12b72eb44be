#!/bin/bash
# r3 session 9: csr_stream's f64 streams as (int2, double2) pairs (policy bit 8) against 16-byte vectors, with / without nt loads, at the
# table's shape and the sweep's best shapes; csr_wavev (pairs) beside them
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s09; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_round3_gpu.py tests/test_spmv_gpu.py -m gpu -x -q -k "not configs3" > $O/pytest.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -n 4 $O/pytest.txt
[ $rc -eq 0 ] || exit 1
PMC_WAVEV=4 PMC_WAVEV_POL=3 PMC_STREAM=512:80:2:2,512:80:2:10,512:80:2:11,256:32:2:2,256:32:2:10,256:32:2:11,256:40:2:11,512:64:2:11 timeout -k 10 400 python3 tools/pmc_matrix_probe.py ldoor --time > $O/time_ldoor.txt 2>&1; grep -E "^TIME" $O/time_ldoor.txt | cut -c1-110
PMC_WAVEV=2,4 PMC_WAVEV_POL=3 PMC_STREAM=512:128:2:2,512:128:2:10,512:128:2:11,256:72:2:2,256:72:2:10,256:72:2:11,256:64:2:11 timeout -k 10 400 python3 tools/pmc_matrix_probe.py nlpkkt120 --time > $O/time_nlpkkt.txt 2>&1; grep -E "^TIME" $O/time_nlpkkt.txt | cut -c1-110
PMC_WAVEV=1 PMC_WAVEV_POL=2 PMC_STREAM=256:144:1:6,256:144:1:14,256:144:1:10,256:128:1:10 timeout -k 10 300 python3 tools/pmc_matrix_probe.py thermal2 --time > $O/time_thermal2.txt 2>&1; grep -E "^TIME" $O/time_thermal2.txt | cut -c1-110

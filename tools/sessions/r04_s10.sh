#!/bin/bash
# r4 session 10: csr_waver for f32 (tests; timing against the f32 kernels the plans choose today)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s10; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_round4_gpu.py -q -m gpu -x > $O/tests.txt 2>&1; echo "pytest exit $?"; tail -6 $O/tests.txt | cut -c1-250
PMC_DTYPE=f32 PMC_WAVEV=4 PMC_WAVEV_POL=3 PMC_WAVER=4,2 PMC_PLAN_AGAIN=1 timeout -k 10 500 python3 tools/pmc_matrix_probe.py ldoor,nlpkkt120,thermal2 --time > $O/f32_time.txt 2>&1; grep -E "^#|TIME" $O/f32_time.txt | cut -c1-175

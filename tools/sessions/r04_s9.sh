#!/bin/bash
# r4 session 9: the new tests (sharded formats through Python, csr_wave's overflow passes, the sharded C++ tests with the other formats), XCD dealing of csr_waver
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s9; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_round4_gpu.py tests/test_cpp_layer.py -q -m gpu > $O/tests.txt 2>&1; echo "pytest exit $?"; tail -6 $O/tests.txt | cut -c1-250
PMC_WAVEV= PMC_WAVER=4 PMC_PACKED=0 PMC_WAVER_SWZ=0,-1,4,16,64,0,-1,4,16,64 PMC_PLAN_AGAIN=1 timeout -k 10 500 python3 tools/pmc_matrix_probe.py ldoor,nlpkkt120 --time > $O/swizzle_time.txt 2>&1; grep TIME $O/swizzle_time.txt | cut -c1-175

#!/bin/bash
# r4 session 14: the tile-staged plan build of csr_waver (tests, cost), then the -m gpu suite
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s14; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_round4_gpu.py -q -m gpu -x > $O/tests.txt 2>&1; echo "pytest exit $?"; tail -4 $O/tests.txt | cut -c1-250
timeout -k 10 400 python3 tools/plan_cost_probe.py > $O/plan_cost.txt 2>&1; grep -v amdgpu.ids $O/plan_cost.txt | cut -c1-200
timeout -k 10 1150 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "full pytest exit $?"; tail -n 5 $O/pytest_gpu.txt | cut -c1-250

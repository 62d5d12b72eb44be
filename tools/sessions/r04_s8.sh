#!/bin/bash
# r4 session 8: the whole -m gpu suite on the tree with the sharded ELL / DIA / COO / HYB operators, the new AUTO size gates and thermal2-like V = 1; smoke()
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s8; mkdir -p $O
timeout -k 10 1150 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "full pytest exit $?"; tail -n 8 $O/pytest_gpu.txt | cut -c1-250
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt | cut -c1-250
PMC_WAVEV=1 PMC_WAVER= PMC_PLAN_AGAIN=1 timeout -k 10 200 python3 tools/pmc_matrix_probe.py thermal2 --time > $O/thermal2_time.txt 2>&1; grep TIME $O/thermal2_time.txt | cut -c1-140

#!/bin/bash
# r4 session 15: the -m gpu suite on the tree with cmi_plan_create_coo and csr_waver / packed in the property-based tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s15; mkdir -p $O
timeout -k 10 1150 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "full pytest exit $?"; tail -n 12 $O/pytest_gpu.txt | cut -c1-250

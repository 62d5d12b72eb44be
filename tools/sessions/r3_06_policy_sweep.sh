#!/bin/bash
# r3 session 6: cache policy x XCD dealing of csr_stream on the long-row matrices (the table's entries for these buckets were tuned on
# other matrices), plus the explicit tile shapes again -- everything validated against csr_scalar before it is timed
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s06; mkdir -p $O
timeout -k 10 800 python3 tools/suitesparse_sweep.py --only ldoor,nlpkkt120 --policies --shapes --rounds 3 > $O/sweep_policies.txt 2>&1; echo "exit $?"
grep -E "policy|table|plan|==|\*" $O/sweep_policies.txt | cut -c1-200

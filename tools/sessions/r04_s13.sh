#!/bin/bash
# r4 session 13: where does the time of making a run-compressed plan go?  kernel stats of the plan-cost probe
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s13; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o plancost -- python3 tools/plan_cost_probe.py > $O/plan_cost.txt 2> $O/rocprof.err || { tail -3 $O/rocprof.err; exit 3; }
rm -f $O/stats/plancost_kernel_trace.csv
grep -E "runs_|exclusive|scan|fingerprint|max_row|column_locality|partition|csr16" $O/stats/plancost_kernel_stats.csv | cut -c1-60,120-220

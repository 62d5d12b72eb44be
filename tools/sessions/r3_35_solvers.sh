#!/bin/bash
# r3 session 35: what an iteration of each solver costs on the headline matrix (tools/cg_bench --solvers), beside the plain cg_bench run on the same box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s35; mkdir -p $O
timeout -k 10 600 tools/bin/cg_bench --iterations=200 > $O/cg_bench_csr.txt 2>&1; echo "cg_bench exit $?"; grep fused $O/cg_bench_csr.txt | cut -c1-200
timeout -k 10 600 tools/bin/cg_bench --solvers --iterations=200 > $O/cg_bench_solvers.txt 2>&1; echo "solvers exit $?"; cat $O/cg_bench_solvers.txt
timeout -k 10 600 tools/bin/cg_bench --solvers --iterations=200 > $O/cg_bench_solvers_again.txt 2>&1; echo "solvers exit $?"; cat $O/cg_bench_solvers_again.txt

#!/bin/bash
# round 4 session 27: configs[4]'s per-rank shape and the soak again, on the final tree (the per-rank block is an f64 5-point stencil: its AUTO plan now runs wave tiles V = 1)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s27; mkdir -p $O
timeout -k 10 500 python3 tools/configs4_rank_shape_probe.py > $O/configs4_rank_shape.txt 2>&1; echo "rank shape exit $?"; grep -v amdgpu.ids $O/configs4_rank_shape.txt | tail -25 | cut -c1-220
timeout -k 10 500 python3 tools/soak.py > $O/soak.txt 2>&1; echo "soak exit $?"; grep -v amdgpu.ids $O/soak.txt | head -3; tail -5 $O/soak.txt | cut -c1-200

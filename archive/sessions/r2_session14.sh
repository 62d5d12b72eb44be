#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s14; mkdir -p $O
timeout -k 10 400 python tools/c16_sweep.py > $O/c16_sweep_poisson.txt 2>&1; echo "c16 sweep exit $?"; grep -v amdgpu.ids $O/c16_sweep_poisson.txt | cut -c1-200
timeout -k 10 400 python tools/c16_sweep.py --matrix ldoor --quick > $O/c16_sweep_ldoor.txt 2>&1; echo "c16 ldoor exit $?"; grep -v amdgpu.ids $O/c16_sweep_ldoor.txt | cut -c1-200
tools/bin/cg_bench --iterations=200 > $O/cg_plain.txt 2>&1; grep fused $O/cg_plain.txt
CMI_COMPRESS_INDICES=1 tools/bin/cg_bench --iterations=200 > $O/cg_c16.txt 2>&1; grep "fused\|format" $O/cg_c16.txt

#!/bin/bash
# session 61: launch shape of the wave-tile kernel on the headline matrix (tools/wave_shape_sweep.py), f64 and f32
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s61; mkdir -p $O
for t in f64 f32; do timeout -k 10 300 python tools/wave_shape_sweep.py $t > $O/wave_shape_$t.txt 2>&1; echo "$t exit $?"; grep -v amdgpu.ids $O/wave_shape_$t.txt; done

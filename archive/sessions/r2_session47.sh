#!/bin/bash
# session 47: csr_stream's lane-strided request shape in the library: the headline matrix over policy x rows per tile x dealing
# (r2_probe `lib shapes`), then across matrix kinds (tools/stream_shape_ab.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s47; mkdir -p $O
timeout -k 10 240 tools/bin/r2_probe --only "lib csr table|lib shapes|lib ell table" > $O/lib_shapes.txt 2>&1; rc=$?; echo "probe exit $rc"; sort -t'n' -k3 $O/lib_shapes.txt | awk '{print}' | sort -k9 -n | head -24
timeout -k 10 500 python tools/stream_shape_ab.py > $O/stream_shape_ab.txt 2>&1; rc=$?; echo "ab exit $rc"; grep -v amdgpu.ids $O/stream_shape_ab.txt

#!/bin/bash
# r4 session 16: soak (incl. csr_waver / packed tiles in f64 and f32, plans made and destroyed in a loop), the C++ device tests with the FEM-block plan test
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s16; mkdir -p $O
timeout -k 10 600 python3 tools/soak.py > $O/soak.txt 2>&1; echo "soak exit $?"; grep -v amdgpu.ids $O/soak.txt | tail -16 | cut -c1-200
timeout -k 10 600 python -m pytest tests/test_cpp_layer.py -q -m gpu > $O/cpp.txt 2>&1; echo "cpp layer exit $?"; tail -4 $O/cpp.txt | cut -c1-200

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s17; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_plan_gpu.py tests/test_cpp_layer.py -m gpu -x -q > $O/pytest_new.txt 2>&1; rc=$?; echo "pytest(new) exit $rc"; tail -n 12 $O/pytest_new.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 900 python tools/hyb_fuse_probe.py > $O/hyb_one_vs_two.txt 2>&1; echo "probe exit $?"; grep -v amdgpu.ids $O/hyb_one_vs_two.txt | cut -c1-200

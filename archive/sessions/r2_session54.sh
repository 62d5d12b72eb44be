#!/bin/bash
# session 54: csrw -- wave-private tiles, no barrier (tools/r2_probe.hip)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s54; mkdir -p $O
timeout -k 10 200 tools/bin/r2_probe --only "lib csr table|csrw |csrd ablation 0 rpb 192" > $O/csrp.txt 2>&1; echo "probe exit $?"; cat $O/csrp.txt

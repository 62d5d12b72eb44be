#!/bin/bash
# session 55 (one row-offset load per lane:csrw1): csrw -- wave-private tiles, no barrier (tools/r2_probe.hip)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s55; mkdir -p $O
timeout -k 10 200 tools/bin/r2_probe --only "lib csr table|csrw k 5|csrw1 |csrd ablation 0 rpb 192" > $O/csrp.txt 2>&1; echo "probe exit $?"; cat $O/csrp.txt

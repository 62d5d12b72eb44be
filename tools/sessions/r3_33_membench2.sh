#!/bin/bash
# r3 session 33: tools/membench2 -- does ANY read pattern (width, nt, wave-contiguous spans, XCD dealing, loads in flight) read a 0.5 / 1.2 / 3 GiB buffer faster
# than the ~6.6 TB/s every SpMV kernel here sits on?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s33; mkdir -p $O
timeout -k 10 300 tools/bin/membench2 > $O/membench2.txt 2>&1; echo "exit $?"; cat $O/membench2.txt

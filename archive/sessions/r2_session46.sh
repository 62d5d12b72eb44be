#!/bin/bash
# session 46 (csrd branch-free): csr_stream's single-pass path with dword-shaped entry streams (tools/r2_probe.hip `csrd`) against the library kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s46; mkdir -p $O
timeout -k 10 240 tools/bin/r2_probe --only "lib csr table|lib ell table|csrd " > $O/csrd.txt 2>&1; rc=$?; echo "probe exit $rc"; cat $O/csrd.txt

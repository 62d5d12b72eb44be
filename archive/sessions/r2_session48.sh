#!/bin/bash
# session 48: re-tune csr_stream's short-row table keys with the lane-strided request shape in the space (policy bits 6 / 7),
# then the driver's bench command on the re-tuned table
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s48; mkdir -p $O
cp cusp-autotuned_amd/tuned/gfx950.json $O/gfx950.json
timeout -k 10 900 python tools/autotune.py --formats csr --merge --csr-stream-only --csr-max-mean 10 --out $O/gfx950.json --log $O/autotune_csr_short_rows.jsonl > $O/autotune.txt 2>&1; rc=$?; echo "autotune exit $rc"; grep -v amdgpu.ids $O/autotune.txt | tail -20
[ $rc -eq 0 ] || exit 1
cp $O/gfx950.json cusp-autotuned_amd/tuned/gfx950.json
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/s48/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"], d.get("cg"), d.get("compressed_index_plan"))
PY
gzip -f $O/autotune_csr_short_rows.jsonl

#!/bin/bash
# session 49: short-row CSR keys re-tuned on matrices whose streams do not fit the Infinity Cache (8e6-row synthetic ones), with
# the residency rule for the nt-load bit in the library; then the GPU tests touched by the lane-strided shape, the A/B across
# matrix kinds on the new table, and the driver's bench command
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s49; mkdir -p $O
cp cusp-autotuned_amd/tuned/gfx950.json $O/gfx950.json
timeout -k 10 1000 python tools/autotune.py --formats csr --merge --csr-stream-only --csr-max-mean 10 --synthetic-rows 8000000 --out $O/gfx950.json --log $O/autotune_csr_short_rows.jsonl > $O/autotune.txt 2>&1; rc=$?; echo "autotune exit $rc"; grep -v amdgpu.ids $O/autotune.txt | grep "^csr/\|wrote"
gzip -f $O/autotune_csr_short_rows.jsonl
[ $rc -eq 0 ] || exit 1
cp $O/gfx950.json cusp-autotuned_amd/tuned/gfx950.json
timeout -k 10 900 python -m pytest tests/test_spmv_gpu.py tests/test_csr16_gpu.py tests/test_plan_gpu.py tests/test_cg_gpu.py -m gpu -x -q > $O/pytest_gpu_subset.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -n 6 $O/pytest_gpu_subset.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 500 python tools/stream_shape_ab.py > $O/stream_shape_ab.txt 2>&1; echo "ab exit $?"; grep -v amdgpu.ids $O/stream_shape_ab.txt | grep -v "policy [23] "
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/s49/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, {k:d["roofline"][k] for k in ("frac","kernel_avg_ms","kernel_avg_over_ms_per_step")}, d.get("cg"), {k:d["compressed_index_plan"].get(k) for k in ("kernel_config","kernel_avg_ms","speedup_over_the_headline_kernel","cg_us_per_iteration")})
PY

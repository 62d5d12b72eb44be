#!/bin/bash
# 4-rank rehearsal of the driver's multi-GPU launch line on the one GPU of the box (CMI_BENCH_REHEARSAL=1: all ranks on GPU 0, gloo)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s21; mkdir -p $O
CMI_BENCH_REHEARSAL=1 MASTER_ADDR=127.0.0.1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 4 --steps 10 --warmup 3 --cg-iterations 50 --configs4 on --configs4-grid 2000 > $O/bench_rehearsal_4.json 2> $O/bench_rehearsal_4.err; echo "exit $?"
tail -n 3 $O/bench_rehearsal_4.err | cut -c1-300
python - <<PY
import json
l=[x for x in open("$O/bench_rehearsal_4.json") if x.startswith("{")]
d=json.loads(l[-1]); print(d["n_gpus"], d["value"], d["config"]["x_exchange"]); print(json.dumps(d.get("exchanges"))[:900]); print(json.dumps(d.get("configs4"))[:900]); print(d.get("cg"))
PY

cd $GRAFT_REPO_ROOT
export CMI_BENCH_REHEARSAL=1 MASTER_ADDR=127.0.0.1
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 5 --warmup 2 --configs4 on --configs4-grid 400 > gpurun_out/dbg_bench2.json 2> gpurun_out/dbg_bench2.err
echo "exit $?"; grep -v "^W\|^\[W\|UserWarning\|warnings.warn" gpurun_out/dbg_bench2.err | tail -20; python3 -c "
import json
l=[x for x in open('gpurun_out/dbg_bench2.json') if x.startswith('{')]
d=json.loads(l[0]); print(json.dumps(d['config']['x_exchange'])); print(json.dumps(d.get('exchanges'))); print(json.dumps(d.get('configs4'))[:1500])"

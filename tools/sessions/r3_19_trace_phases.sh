#!/bin/bash
# r3 session 19: the driver's bench command under rocprofv3 --kernel-trace --stats, cut into bench.py's own launch phases (replay / cold launch the
# same kernel); the same without the cold leg (CMI_BENCH_COLD=0); the sharded C++ tests again (float instances added)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s19; mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 2; }
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err || { echo rocprof bench failed; tail -3 $O/rocprof.err; exit 3; }
python3 tools/trace_phases.py $O/stats/bench_kernel_trace.csv $O/bench_under_rocprof.json > $O/trace_phases.txt 2>&1; cat $O/trace_phases.txt
rm -f $O/stats/bench_kernel_trace.csv
mkdir -p $O/nocold
CMI_BENCH_COLD=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/nocold -o bench -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof_no_cold.json 2> $O/rocprof2.err || { echo rocprof bench 2 failed; tail -3 $O/rocprof2.err; exit 3; }
rm -f $O/nocold/bench_kernel_trace.csv
head -3 $O/nocold/bench_kernel_stats.csv | cut -c1-260
python3 -c "import json; d=json.load(open('$O/bench_under_rocprof_no_cold.json')); print('no-cold run: kernel_avg_ms', d['roofline']['kernel_avg_ms'], 'frac', d['roofline']['frac'])"
timeout -k 10 600 python -m pytest tests/test_cpp_layer.py -m gpu -x -q -k "sharded" > $O/pytest_sharded.txt 2>&1; echo "sharded pytest exit $?"; tail -n 4 $O/pytest_sharded.txt | cut -c1-250

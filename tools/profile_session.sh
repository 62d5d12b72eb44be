# Collects the evidence kept under profiles/: bench line, the same bench under rocprofv3 --kernel-trace --stats, PMC passes
# (FETCH_SIZE / WRITE_SIZE / TCC hit+miss, one counter set per pass), CG kernel stats, the other formats' bench lines.
# usage (on the GPU box): bash tools/profile_session.sh ; results under gpurun_out/prof/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
python bench.py > gpurun_out/prof/bench_n1.json 2> gpurun_out/prof/bench_n1.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats -o bench -- python3 bench.py --steps 200 --warmup 5 --no-cpu-baseline > gpurun_out/prof/bench_under_rocprof.json 2> gpurun_out/prof/rocprof.err || exit 2
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/pmc -o fetch -- python3 tools/pmc_probe.py > gpurun_out/prof/probe.json 2> gpurun_out/prof/pmc1.err || exit 3
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/pmc -o write -- python3 tools/pmc_probe.py > gpurun_out/prof/probe2.json 2> gpurun_out/prof/pmc2.err || exit 4
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/prof/pmc -o tcc -- python3 tools/pmc_probe.py > gpurun_out/prof/probe3.json 2> gpurun_out/prof/pmc3.err || exit 5
python tools/pmc_summary.py gpurun_out/prof/pmc gpurun_out/prof/probe.json gpurun_out/prof/pmc.json || exit 6
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/cgstats -o cg -- tools/bin/cg_bench --iterations=100 > gpurun_out/prof/cg_bench.txt 2> gpurun_out/prof/cg.err || exit 8
for f in ell dia coo hyb; do python bench.py --format $f --no-cpu-baseline > gpurun_out/prof/bench_n1_$f.json 2>/dev/null || exit 7; done
ls gpurun_out/prof gpurun_out/prof/stats/* | head -30

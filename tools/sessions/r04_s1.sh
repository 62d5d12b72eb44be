set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s1
timeout -k 10 900 python -m pytest tests/test_round4_gpu.py -x -q -m gpu 2>&1 | tail -30 > gpurun_out/s1/tests.txt
cat gpurun_out/s1/tests.txt | tail -15
PMC_WAVEV=4 PMC_WAVEV_POL=3 PMC_WAVER=4,2 PMC_C16=1 timeout -k 10 600 python tools/pmc_matrix_probe.py ldoor,nlpkkt120,thermal2 --time > gpurun_out/s1/time.txt 2>&1
grep TIME gpurun_out/s1/time.txt

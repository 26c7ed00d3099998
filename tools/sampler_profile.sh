cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "sampler or running" 2>&1 | tail -2
rm -rf gpurun_out/sp
L2_USE_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sp -o p -- python3 tools/sampler_bench.py stories15M 48 > /dev/null 2>&1
find gpurun_out/sp -name "*kernel_stats.csv" -exec head -4 {} \; | cut -c1-110
python tools/sampler_bench.py stories110M

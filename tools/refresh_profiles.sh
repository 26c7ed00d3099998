set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/final/pytest_gpu.txt
python bench.py > gpurun_out/final/bench_default_llama2_7b.json 2> gpurun_out/final/bench_default.err
python bench.py --config stories110M > gpurun_out/final/stories110M_bench.json 2>/dev/null
python bench.py --config stories15M > gpurun_out/final/stories15M_bench.json 2>/dev/null
for cfg in llama2_7b stories110M; do
  L2_USE_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/prof_$cfg -o p -- python3 bench.py --config $cfg --no-cpu-baseline --no-extra --no-dropin --steps 64 --warmup 8 > gpurun_out/final/${cfg}_bench_under_rocprof.json 2> gpurun_out/final/${cfg}_rocprof.err
  find gpurun_out/final/prof_$cfg -name "*kernel_stats.csv" -exec cp {} gpurun_out/final/${cfg}_kernel_stats.csv \;
  find gpurun_out/final/prof_$cfg -name "*kernel_trace.csv" -exec python tools/trace_gaps.py {} \; > gpurun_out/final/${cfg}_durations_and_gaps_eager.txt 2>&1
  rm -rf gpurun_out/final/prof_$cfg
done
cat gpurun_out/final/pytest_gpu.txt

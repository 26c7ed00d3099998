# The one GPU-box command that regenerates profiles/r03/ (copy gpurun_out/r03/* there afterwards):
#   bash tools/refresh_profiles.sh
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r03
rm -rf $out; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $out/pytest_gpu.txt
# bench lines (default = what the driver runs; roofline.traffic from the PMC passes inside the run, cpu_baseline on this box)
python bench.py > $out/bench_default_llama2_7b.json 2> $out/bench_default.err
python bench.py --config stories110M > $out/stories110M_bench.json 2>/dev/null
python bench.py --config stories15M > $out/stories15M_bench.json 2>/dev/null
# what `python bench.py --gpus 2` prints when both ranks land on this one GPU (the ranks meet through files; proof tokens in the line)
L2_BENCH_FORCE_DEVICE=0 python bench.py --gpus 2 --config llama2_7b_L2 --steps 64 --warmup 8 --no-cpu-baseline > $out/bench_gpus2_two_ranks_one_gpu.json 2> $out/bench_gpus2.err
# rocprofv3 kernel trace of the same command (eager launches: rocprofv3 crashes on long graph replays, profiles/README.md)
for cfg in llama2_7b stories110M stories15M; do
  L2_USE_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$cfg -o p -- python3 bench.py --config $cfg --no-cpu-baseline --no-extra --no-dropin --no-pmc --steps 64 --warmup 8 > $out/${cfg}_bench_under_rocprof.json 2> $out/${cfg}_rocprof.err
  cp $out/prof_$cfg/p_kernel_stats.csv $out/${cfg}_kernel_stats.csv
  python tools/trace_gaps.py $out/prof_$cfg/p_kernel_trace.csv > $out/${cfg}_durations_and_gaps_eager.txt 2>&1
  rm -rf $out/prof_$cfg
done
# prompt ingestion: per-kernel times old / register-blocked (1, 2, 4 chunks per launch), SQ counters, end to end
bash tools/prefill_variants.sh > $out/prefill_kernels_old_vs_register_blocked.txt 2>&1
python tools/prefill_bench.py llama2_7b > $out/prefill_bench_llama2_7b.txt 2>&1
python tools/prefill_bench.py stories110M > $out/prefill_bench_stories110M.txt 2>&1
bash tools/prefill_pmc.sh > /dev/null 2>&1; cp gpurun_out/pfpmc/summary.txt $out/prefill_pmc_sq_counters_7b_width_64tok.txt
# device sampler
for c in stories110M stories110M stories15M llama2_7b_L2; do python tools/sampler_bench.py $c; done > $out/sampler_bench_run.txt 2>&1
bash tools/sampler_profile.sh > $out/sampler_kernels.txt 2>&1
# the experiments of this round that were measured and not kept
hipcc --offload-arch=gfx950 -O3 -o /tmp/mbo tools/microbench_overlap.hip 2>/dev/null && timeout 300 /tmp/mbo > $out/microbench_overlap.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -o /tmp/mbr tools/microbench_rows.hip 2>/dev/null && timeout 400 /tmp/mbr > $out/microbench_rows.txt 2>&1
cat $out/pytest_gpu.txt

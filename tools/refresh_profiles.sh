# The one GPU-box command that regenerates profiles/r06/ (copy gpurun_out/r06prof/* there afterwards):
#   bash tools/refresh_profiles.sh
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r06prof
rm -rf $out; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -q --durations=25 2>&1 | tail -40 > $out/pytest_gpu.txt
# bench lines (default = what the driver runs; parity of the timed tokens against the reference goldens, roofline.traffic from the PMC
# passes inside the run, kernel_trace_us from a rocprofv3 --kernel-trace child, cpu_baseline on this box, tp_predicted from shard-timing contexts)
python bench.py > $out/bench_default_llama2_7b.json 2> $out/bench_default.err
L2_BENCH_SKIP_JS_7B=1 python bench.py --steps 20 --warmup 5 > $out/bench_driver_style_llama2_7b.json 2>/dev/null      # (the in-process JS baseline of the 7B shape: once is enough)
python bench.py --config stories110M > $out/stories110M_bench.json 2>/dev/null
python bench.py --config stories15M > $out/stories15M_bench.json 2>/dev/null
# what `python bench.py --gpus N` prints when all ranks land on this one GPU (supervised stages; the ranks end up meeting through files)
L2_BENCH_FORCE_DEVICE=0 python bench.py --gpus 2 --config llama2_7b_L2 --steps 64 --warmup 8 --no-cpu-baseline > $out/bench_gpus2_two_ranks_one_gpu.json 2> $out/bench_gpus2.err
L2_BENCH_FORCE_DEVICE=0 L2_TP_WAIT_S=20 python bench.py --gpus 8 --config llama2_7b_L2 --steps 64 --warmup 8 --no-cpu-baseline > $out/bench_gpus8_eight_ranks_one_gpu.json 2> $out/bench_gpus8.err
# rocprofv3 kernel trace of the same command (eager launches: the library's queue stands down under a tool library, profiles/README.md)
for cfg in llama2_7b stories110M stories15M; do
  L2_USE_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$cfg -o p -- python3 bench.py --config $cfg --no-cpu-baseline --no-extra --no-dropin --no-pmc --steps 64 --warmup 8 > $out/${cfg}_bench_under_rocprof.json 2> $out/${cfg}_rocprof.err
  cp $out/prof_$cfg/p_kernel_stats.csv $out/${cfg}_kernel_stats.csv
  python tools/trace_gaps.py $out/prof_$cfg/p_kernel_trace.csv > $out/${cfg}_durations_and_gaps_eager.txt 2>&1
  rm -rf $out/prof_$cfg
done
# the decode path's MFMA counters (north_star: "evidenced by rocprof HBM GB/s and MFMA utilisation"): one counter per pass
for cfg in llama2_7b stories110M; do python3 tools/decode_mfma_pmc.py $cfg $out/decode_mfma_pmc_$cfg.json > $out/decode_mfma_pmc_$cfg.txt 2>&1; done
python tools/prefill_f32_eval.py > $out/prefill_f32_eval.txt 2>&1; cp gpurun_out/r06/prefill_f32_eval.json $out/ 2>/dev/null
python tools/prefill_bench.py llama2_7b > $out/prefill_bench_llama2_7b.txt 2>&1
python tools/prefill_bench.py stories110M > $out/prefill_bench_stories110M.txt 2>&1
for c in stories110M stories15M llama2_7b_L2; do python tools/sampler_bench.py $c; done > $out/sampler_bench_run.txt 2>&1
# one rank's shard of the tensor-parallel step alone on this GPU (l2_tp_mode 5): ms per token at 8 / 4 / 2 ranks, then per kernel under rocprofv3
( export L2_TEST_HOOKS=1
  for G in 8 4 2; do python3 tools/tp_solo_step.py $G llama2_7b 64; done
  for G in 8 2; do
    rm -rf $out/tps$G
    L2_USE_GRAPH=0 L2_PROFILE_SYNC=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/tps$G -o p -- python3 tools/tp_solo_step.py $G llama2_7b 16 > /dev/null 2>&1
    echo "== G=$G (eager, kernel trace)"
    python3 tools/kernel_stats_table.py $out/tps$G
    rm -rf $out/tps$G
  done ) > $out/tp_shard_step_kernels.txt 2>&1
set +x
cat $out/pytest_gpu.txt | tail -5

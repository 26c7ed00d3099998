# The one GPU-box command that regenerates profiles/r02/ (copy gpurun_out/r02/* there afterwards):
#   bash tools/refresh_profiles.sh
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02
rm -rf $out; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $out/pytest_gpu.txt
# bench lines (default = what the driver runs; roofline.traffic from the PMC passes inside the run, cpu_baseline on this box)
python bench.py > $out/bench_default_llama2_7b.json 2> $out/bench_default.err
python bench.py --config stories110M > $out/stories110M_bench.json 2>/dev/null
python bench.py --config stories15M > $out/stories15M_bench.json 2>/dev/null
# rocprofv3 kernel trace of the same command (eager launches: rocprofv3 crashes on long graph replays, profiles/README.md)
for cfg in llama2_7b stories110M stories15M; do
  L2_USE_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$cfg -o p -- python3 bench.py --config $cfg --no-cpu-baseline --no-extra --no-dropin --no-pmc --steps 64 --warmup 8 > $out/${cfg}_bench_under_rocprof.json 2> $out/${cfg}_rocprof.err
  cp $out/prof_$cfg/p_kernel_stats.csv $out/${cfg}_kernel_stats.csv
  python tools/trace_gaps.py $out/prof_$cfg/p_kernel_trace.csv > $out/${cfg}_durations_and_gaps_eager.txt 2>&1
  rm -rf $out/prof_$cfg
done
# in-kernel anatomy (diagnostic build) and the floors
python tools/stamps.py stories110M 100 > $out/stamps_phase_110M.txt 2>&1
python tools/stamps.py stories15M 100 > $out/stamps_phase_15M.txt 2>&1
python tools/stamps.py llama2_7b_L2 20 > $out/stamps_phase_7b_width.txt 2>&1
for c in stories15M stories110M llama2_7b_L2; do python tools/stamps_attn.py $c 100; python tools/stamps_attn.py $c 250; done > $out/stamps_attention.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -o /tmp/mbp tools/microbench_phase.hip 2>/dev/null && /tmp/mbp > $out/microbench_phase_floor.txt
hipcc --offload-arch=gfx950 -O3 -o /tmp/mbl tools/microbench_launch.hip 2>/dev/null && /tmp/mbl > $out/microbench_launch.txt
hipcc --offload-arch=gfx950 -O3 -o /tmp/mbi tools/microbench_icache.hip 2>/dev/null && /tmp/mbi > $out/microbench_icache.txt
hipcc --offload-arch=gfx950 -O3 -o /tmp/mbr tools/microbench_rows.hip 2>/dev/null && timeout 300 /tmp/mbr > $out/microbench_rows_run.txt
STAMPS_WG=2 python tools/stamps.py llama2_7b_L2 20 > $out/stamps_workgroups_7b_width_run.txt 2>&1
for r in 0 5 0 5; do bash tools/kernel_times.sh llama2_7b L2_TUNE_ROT=$r; done > $out/kernel_times_rot_ab.txt 2>&1
# context curves (attention split levels) and the prefill / sampler extras
for c in llama2_7b stories110M; do for s in 1 8; do L2_ATTN_SPLITS=$s python tools/ctx_curve.py $c | tail -1; done; python tools/ctx_curve.py $c | tail -1; done > $out/ctx_curve.txt 2>&1
bash tools/prefill_pmc.sh > /dev/null 2>&1; cp gpurun_out/pfpmc_r02/summary.json $out/prefill_mfma_pmc_7b_width_64tok.json
for c in stories110M stories110M stories15M llama2_7b_L2; do python tools/sampler_bench.py $c; done > $out/sampler_bench_run.txt 2>&1
bash tools/sampler_profile.sh > $out/sampler_kernels.txt 2>&1
cat $out/pytest_gpu.txt

#!/usr/bin/env python3
"""bench.py -- decode tokens/s + achieved HBM GB/s of the llama2.ts forward pass on MI355X.

A "step" is ONE transformer() call (llama2.ts:205-303) = one decoded token, weights resident in HBM,
greedy feed (-t 0, llama2.ts:476-478) from BOS at pos 0, synthetic seeded weights of the named shape.
`value` times the device-resident loop (forward + on-device argmax, no host round trip); the same K
steps through the blocking drop-in call l2_forward (logits handed to the host every token, as
llama2.ts:468-478 consumes them) are reported next to it as `dropin_tok_s`.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config llama2_7b|stories110M|stories15M]

Everything in the JSON line is measured by this run:
  * `roofline`: the dominant kernel (rmsnorm + w1/w3 GEMV + SwiGLU) timed in situ with HIP events on the
    library's stream; `traffic` = HBM bytes per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE,
    gfx950 corrections of the MI355X guide) that this script runs over itself as child processes (--pmc-child)
    BEFORE it touches the GPU; null when rocprofv3 is unavailable;
  * `cpu_baseline`: the C restatement of the reference (oracle/, one thread like the single-threaded reference)
    timed on this box's host cores on a bounded sample of the same workload;
  * `stories110M`: BASELINE.json's other named shape, same fields.

N > 1 (launched by torch.distributed.run, one rank per GPU, or by this script itself): Llama-2-7B is tensor-parallel (heads / FFN
rows sharded, fp64 all-reduce of d partials twice per layer -- llama2.ts:270, 292) => strong scaling; shapes that do
not shard run as independent replicas => weak scaling.  Every rank is a CPU-only SUPERVISOR that starts a fresh worker process per
way of forming the group and bounds every phase of it (benchparts/ranks.py, benchparts/worker.py): a hang costs a deadline, not
the line.

The legs live in benchparts/: profiler.py (rocprofv3 children), dropin.py (the drop-in boundary: ctypes, N-API), baselines.py (CPU),
single.py (roofline, per-kernel, shard-step prediction), ranks.py + worker.py (N > 1).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from llama2_ts_amd import configs, runtime  # noqa: E402
from benchparts import profiler  # noqa: E402
from benchparts.baselines import cpu_baseline, reference_js_figure  # noqa: E402,F401  (reference_js_figure: tests/test_bench_cpu.py)
from benchparts.common import HBM_PEAK_GBS, avg_bytes_per_token, parity_block, under_profiler  # noqa: E402
from benchparts.dropin import dropin_child, dropin_direct_dispatch_off, dropin_loop, napi_dropin  # noqa: E402
from benchparts.single import contract_keys, dispatch_note, per_kernel_block, prefill_rates, roofline_block, secondary_config, tp_prediction  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--config", default="llama2_7b", choices=sorted(configs.CONFIGS))
    ap.add_argument("--seed", type=int, default=configs.DEFAULT_SEED)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary measurements (stories110M block, long context, sampler, prefill)")
    ap.add_argument("--no-dropin", action="store_true", help="skip the l2_forward (host round trip per token) loop")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes (roofline.traffic = null)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--trace-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dropin-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--worker-stage", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    # the host driver of this pool only supports dmabuf IPC: without this, RCCL and hipIpcGetMemHandle fail in ranks a launcher
    # other than ours started (set before the first HIP call of the process)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.pmc_child:
        return profiler.pmc_child(args.config, args.seed)
    if args.trace_child:
        return profiler.trace_child(args.config, args.seed)
    if args.dropin_child:
        return dropin_child(args.config, args.seed)

    # ---- N > 1: this process supervises (it never touches the GPU) or is one stage's worker (benchparts/ranks.py, worker.py)
    argv = [a for a in sys.argv[1:]]
    if args.worker_stage:
        from benchparts.worker import worker
        return worker(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        from benchparts.ranks import spawn_ranks
        sys.exit(spawn_ranks(args, argv))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        from benchparts.ranks import supervise
        sys.exit(supervise(args, argv))

    if under_profiler():
        # rocprofv3 (ROCm 7.2) segfaults once a process has replayed a hipGraph more than ~128 times with kernel tracing on
        # (profiles/README.md): under a profiler the library launches the same kernels eagerly -- per-kernel durations carry over,
        # the tokens/s of such a run is host-bound and says nothing
        os.environ.setdefault("L2_USE_GRAPH", "0")
    if args.gpus != 1:
        print("bench.py: --gpus %d: reporting the one GPU that runs" % args.gpus, file=sys.stderr)
    hdr = configs.header(args.config)
    K = min(args.steps, hdr[6])
    W = min(args.warmup, hdr[6])
    device = int(os.environ.get("L2_BENCH_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0")))

    # counter passes first: they are child processes, and nothing in THIS process has touched the GPU yet
    traffic = {args.config: (None, "skipped (--no-pmc)")}
    trace_us, mfma = {}, {}
    if not args.no_pmc:
        traffic[args.config] = profiler.pmc_traffic(args.config, args.seed)
        trace_us[args.config] = profiler.kernel_trace_us(args.config, args.seed)
        mfma[args.config] = profiler.mfma_instructions(args.config, args.seed)
        if not args.no_extra and args.config == "llama2_7b":
            traffic["stories110M"] = profiler.pmc_traffic("stories110M", args.seed)
            trace_us["stories110M"] = profiler.kernel_trace_us("stories110M", args.seed)
            mfma["stories110M"] = profiler.mfma_instructions("stories110M", args.seed)

    cfg = runtime.Config(hdr)
    ctx = runtime.Context(hdr, device=device)
    ctx.synth_fill(args.seed)

    # untimed: small models finish their K steps in tens of milliseconds, less than the clock ramp of an idle GPU, so
    # they first decode for ~0.3 s; then the W warm-up steps, the last of them at the deepest position of the timed run
    # (it records the step of the split-attention level, which would otherwise be recorded inside the timed region)
    if configs.checkpoint_bytes(hdr) < (1 << 30):
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < 0.3:
            ctx.bench_decode(1, 0, min(64, hdr[6]))
    else:
        # untimed: the first step repacks the matrices and gives their row-major tensors back to the driver (one copy of the weights);
        # the driver then scrubs the freed memory in the background -- 25 GB at 7B, 2.5 - 3 s of extra HBM traffic that costs a decode
        # running beside it ~2 % (profiles/r04/one_copy_release_transient.txt).  The steady state is what a serving process sees.
        ctx.bench_decode(1, 0, min(8, hdr[6]))
        if ctx.get_option(runtime.OPT_PACKED_MIB) > 0:
            t_spin = time.perf_counter()
            while time.perf_counter() - t_spin < 3.5:
                ctx.bench_decode(1, 0, min(64, hdr[6]))
    if W > 1:
        ctx.bench_decode(1, 0, W - 1)
    if W > 0:
        ctx.bench_decode(1, K - 1, 1)
    t0 = time.perf_counter()
    dev_ms = ctx.bench_decode(1, 0, K)         # EXACTLY K timed steps (the call returns when the last one is done)
    wall = time.perf_counter() - t0
    timed_tokens = ctx.bench_tokens(K)         # what the timed run decoded (checked against the reference's golden below)

    value = K / wall
    bpt = avg_bytes_per_token(hdr, 0, K)
    out = {
        "metric": "decode tokens/sec + achieved HBM GB/s (% peak), 1 GPU",
        "value": round(value, 3), "unit": "tokens/s", "n_gpus": 1, "steps": K, "warmup": W,
        "ms_per_step": round(1e3 * wall / K, 5), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "storage_dtype": "f32",
        "data": "synthetic (seeded hash generator, llama2.c-v0 layout)",
        "config": {"workload": "%s batch-1 greedy decode, %d tokens from BOS (-t 0 -s 1 -n %d)" % (args.config, K, K),
                   "header": list(hdr), "parallelism": "single",
                   "loop": ("device-resident (forward + argmax on GPU, %s) -- `value` times THIS loop, a SURVEY.md 8(f1) extra; the contract's own call "
                            "(one blocking transformer() per token, logits to the host) is `contract_tok_s`" % dispatch_note(ctx)),
                   "weights": "fp32, ONE copy on the device (%d MiB): the matrices of the streaming phases repacked in the order the chip consumes them "
                              "(%d MiB, DESIGN.md section 3), everything else row-major as the checkpoint stores it"
                              % (ctx.get_option(runtime.OPT_WEIGHT_MIB), ctx.get_option(runtime.OPT_PACKED_MIB)),
                   "weights_mib": {"on_device": ctx.get_option(runtime.OPT_WEIGHT_MIB), "repacked": ctx.get_option(runtime.OPT_PACKED_MIB),
                                   "checkpoint": configs.checkpoint_bytes(hdr) >> 20}},
        "device_ms_per_step": round(dev_ms / K, 5),
        "algorithmic_bytes_per_token": int(bpt),
        "hbm_gbs_end_to_end": round(bpt * value / 1e9, 2),
        "hbm_frac_end_to_end": round(bpt * value / 1e9 / HBM_PEAK_GBS, 4),
    }

    # the tokens of the TIMED run against the real reference's tokens for this checkpoint (fixtures are data: they travel)
    out["parity"] = parity_block(args.config, args.seed, timed_tokens)
    if under_profiler() and os.environ.get("L2_USE_GRAPH") == "0":
        out["profiled"] = "this run was started under a profiler: eager launches instead of a recorded step per token (host-bound; read the kernel durations, not `value`)"
    if not args.no_dropin:   # the same K steps through the blocking drop-in boundary (logits to the host every token)
        rate, dropin_tokens = dropin_loop(ctx, K)
        out["dropin_tok_s"] = round(rate, 3)
        out["parity"]["dropin_equal_to_reference_golden"] = parity_block(args.config, args.seed, dropin_tokens)["equal_to_reference_golden"]
        out["napi_dropin_tok_s"] = napi_dropin(ctx, args.config, args.seed, K)      # the N-API addon under Node (small models: the 7B file is not written)
        if configs.checkpoint_bytes(hdr) < (2 << 30):
            out["dropin_tok_s_direct_dispatch_off"] = dropin_direct_dispatch_off(args.config, args.seed)
        contract_keys(out, bpt)
    out["roofline"] = roofline_block(ctx, cfg, K, *traffic[args.config], trace_us.get(args.config), mfma.get(args.config))
    out["per_kernel"] = per_kernel_block(ctx, cfg)
    S = hdr[6]
    if not args.no_extra:
        # the whole context window, position 0 .. S-1 (attention reads up to S rows per layer; split form beyond 144 rows)
        ms = ctx.bench_decode(1, 0, S)
        out["whole_context_tok_s"] = round(S / (ms * 1e-3), 3)
        if S >= 1024:   # and its last 128 positions on their own
            n_long = 128
            p0 = S - n_long
            ctx.bench_decode(1, p0, 8)
            ms = ctx.bench_decode(1, p0, n_long)
            b_long = avg_bytes_per_token(hdr, p0, S)
            out["long_context"] = {"positions": [p0, S - 1], "value": round(n_long / (ms * 1e-3), 3), "unit": "tokens/s",
                                   "ms_per_step": round(ms / n_long, 5), "algorithmic_bytes_per_token": int(b_long),
                                   "hbm_frac_end_to_end": round(b_long * n_long / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        # the sampled branch of the loop on the device (l2_decode_sample): reference semantics, same token ids per seed
        n_s = min(64, S)
        samp = {}
        for nm, t, p in (("sample_t0.9", 0.9, 1.0), ("topp0.9_t0.9", 0.9, 0.9)):
            ctx.decode_sample(1, 0, 8, t, p, 42)
            t0 = time.perf_counter()
            ctx.decode_sample(1, 0, n_s, t, p, 42)
            samp[nm] = round(n_s / (time.perf_counter() - t0), 3)
        out["sampled_decode_tok_s"] = samp
        # prompt ingestion (l2_prefill: 64-token chunks on the fp64 MFMA path, up to four chunks per launch) next to the token-by-token loop it replaces
        out.update(prefill_rates(ctx, cfg, S))
    ctx.close()

    if not args.no_extra and args.config.startswith("llama2_7b"):
        out["tp_predicted"] = tp_prediction(hdr, args.seed, device, 1e3 * wall / K)
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.config, hdr, args.seed)
    if not args.no_extra and args.config == "llama2_7b":
        out["stories110M"] = secondary_config("stories110M", args.seed, device, not args.no_cpu_baseline,
                                              traffic.get("stories110M", (None, "skipped")), trace_us.get("stories110M"), mfma.get("stories110M"))

    # a timed run that decoded other tokens than the reference is not a measurement: the line is still printed (it says where
    # the first mismatch is), the exit code says no
    bad = [nm for nm, blk in (("main", out), ("stories110M", out.get("stories110M") or {}))
           if (blk.get("parity") or {}).get("equal_to_reference_golden") is False
           or (blk.get("parity") or {}).get("dropin_equal_to_reference_golden") is False]
    print(json.dumps(out))
    if bad:
        print("bench.py: PARITY FAILURE in %s: the timed decode does not reproduce the reference's golden tokens" % ", ".join(bad), file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()

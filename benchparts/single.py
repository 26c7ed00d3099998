"""One configuration on one GPU: the device-resident loop, the drop-in loops, the roofline of the dominant kernel, the shard-step prediction."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from llama2_ts_amd import configs, runtime  # noqa: E402

from .baselines import cpu_baseline  # noqa: E402
from .common import DOMINANT, HBM_PEAK_GBS, avg_bytes_per_token, dominant_kernel_bytes, parity_block  # noqa: E402
from .dropin import dropin_direct_dispatch_off, dropin_loop, napi_dropin  # noqa: E402


def contract_keys(out, bpt):
    """`value` times the device-resident loop -- a SURVEY.md 8(f1) extra.  The contract's own call is one blocking transformer() per
    token with the logits handed to the host (llama2.ts:468 -> 478): through the N-API addon under Node where a checkpoint file can be
    written, else through ctypes.  Both are in the line already; these keys put the contract's figure next to `value`."""
    napi = (out.get("napi_dropin_tok_s") or {}).get("value")
    rate = napi or out.get("dropin_tok_s")
    if rate:
        out["contract_tok_s"] = rate
        out["contract_hbm_frac"] = round(bpt * rate / 1e9 / HBM_PEAK_GBS, 4)
        out["contract_how"] = ("one blocking l2_forward per token through the N-API addon under Node (host/l2_run.mjs --loop host), logits to the host, first maximum there"
                               if napi else "one blocking l2_forward per token through ctypes, logits to the host, argmax there (no checkpoint file of this size is written for the N-API run)")


def dispatch_note(ctx):
    """How the device-resident loop's launches reached the chip in the run just made (L2_OPT_AQL_QUEUE, include/llama2_hip.h)."""
    if ctx.get_option(runtime.OPT_AQL_QUEUE):
        return ("a token's launches written as AQL packets on the library's own HSA queue: barrier bit, agent-scope release, "
                "no acquire fence between the launches of a token (csrc/aql_queue.h)")
    why = ctx.dispatch_reason()
    if not ctx.get_option(runtime.OPT_USE_GRAPH):
        return "eager launches (%s)" % (why or "L2_USE_GRAPH=0: the step is not recorded")
    return "one hipGraph replay per token (the library's AQL queue is not in use: %s)" % (why or "not taken")


def roofline_block(ctx, cfg, K, traffic, traffic_how, trace_us=None, mfma=None):
    iters = 200 if cfg.dim * cfg.hidden_dim < (1 << 24) else 50
    kms_isolated = ctx.bench_gemv(runtime.T_W1, cfg.n_layers // 2, iters)
    kus, nlaunch = ctx.bench_dominant_in_situ(1, 0, min(K, 128))    # HIP events on every dispatch of the kernel, eager launches of the same kernels
    kb = dominant_kernel_bytes(cfg)
    ach = kb / (kus * 1e-6) / 1e9
    out = {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_how": traffic_how,
           "kernel": DOMINANT, "bytes_per_launch": kb, "avg_launch_us": round(kus, 3), "launches_timed": nlaunch,
           "duration_used": "avg_launch_us: HIP start/stop events on every dispatch, in situ (an event pair adds about 1 us to a dispatch: "
                            "an upper bound on the kernel's own time, within 2 % at 50 us, tens of percent at 5 us)",
           "how": "HIP start/stop events attached to every dispatch of this kernel inside a decode run on the library's stream (hipExtLaunchKernelGGL)",
           "isolated_back_to_back_us": round(kms_isolated * 1e3, 3)}
    if mfma is not None:      # (benchparts/profiler.py: mfma_instructions)
        out["mfma_insts"], out["mfma_how"] = mfma
    if trace_us:
        out["kernel_trace_us"] = trace_us
        out["frac_kernel_trace"] = round(kb / (trace_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
        out["kernel_trace_how"] = "rocprofv3 --kernel-trace over a child of this run decoding 24 tokens with eager launches: End - Start of every dispatch of this kernel, mean of the last three quarters"
    return out


def per_kernel_block(ctx, cfg):
    iters = 200 if cfg.dim * cfg.hidden_dim < (1 << 24) else 50
    out = {}
    d, h, V = cfg.dim, cfg.hidden_dim, cfg.vocab_size
    for nm, kind in (("qkv", runtime.T_WQ), ("wo", runtime.T_WO), ("w13", runtime.T_W1), ("w2", runtime.T_W2), ("wcls", runtime.T_WCLS)):
        ms = ctx.bench_gemv(kind, 0, iters)
        nb = {"qkv": 3 * d * d, "wo": d * d, "w13": 2 * d * h, "w2": d * h, "wcls": V * d}[nm] * 4
        out[nm] = {"us": round(ms * 1e3, 3), "GBs": round(nb / (ms * 1e-3) / 1e9, 1)}
    return out


def secondary_config(name, seed, device, with_cpu, traffic, trace_us=None, mfma=None):
    """BASELINE.json's metric names stories110M next to 7B: the same measurement as a block of the same JSON line."""
    hdr = configs.header(name)
    ctx = runtime.Context(hdr, device=device)
    ctx.synth_fill(seed)
    cfg = ctx.cfg
    K = min(256, hdr[6])
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.3:      # untimed: clocks up (see main)
        ctx.bench_decode(1, 0, 64)
    ctx.bench_decode(1, 0, 31)
    ctx.bench_decode(1, K - 1, 1)                  # captures the split-attention graph outside the timed region
    t0 = time.perf_counter()
    ctx.bench_decode(1, 0, K)
    wall = time.perf_counter() - t0
    parity = parity_block(name, seed, ctx.bench_tokens(K))
    bpt = avg_bytes_per_token(hdr, 0, K)
    out = {"value": round(K / wall, 2), "unit": "tokens/s", "steps": K, "ms_per_step": round(1e3 * wall / K, 5),
           "algorithmic_bytes_per_token": int(bpt),
           "hbm_gbs_end_to_end": round(bpt * K / wall / 1e9, 2),
           "hbm_frac_end_to_end": round(bpt * K / wall / 1e9 / HBM_PEAK_GBS, 4),
           "loop": "device-resident (forward + argmax on GPU, %s) -- `value` times THIS loop (a SURVEY.md 8(f1) extra); the contract's own call is `contract_tok_s`" % dispatch_note(ctx),
           "parity": parity}
    # the same K steps through the blocking drop-in boundary (llama2.ts:468 -> 478: logits to the host every token, argmax there)
    rate, dropin_tokens = dropin_loop(ctx, K)
    out["dropin_tok_s"] = round(rate, 2)
    out["parity"]["dropin_equal_to_reference_golden"] = parity_block(name, seed, dropin_tokens)["equal_to_reference_golden"]
    out["napi_dropin_tok_s"] = napi_dropin(ctx, name, seed, K)               # ... and through the N-API addon under Node (the contract's binding)
    out["dropin_tok_s_direct_dispatch_off"] = dropin_direct_dispatch_off(name, seed)
    contract_keys(out, bpt)
    out["roofline"] = roofline_block(ctx, cfg, K, traffic[0], traffic[1], trace_us, mfma)
    out["per_kernel"] = per_kernel_block(ctx, cfg)      # back-to-back launches of each GEMV phase: us and GB/s of its matrix bytes
    S = hdr[6]
    ms = ctx.bench_decode(1, 0, S)
    out["whole_context_tok_s"] = round(S / (ms * 1e-3), 2)
    out.update(prefill_rates(ctx, cfg, S))     # prompt ingestion, as in the main block
    ctx.close()
    if with_cpu:
        out["cpu_baseline"] = cpu_baseline(name, hdr, seed)
    return out


def prefill_rates(ctx, cfg, S):
    """Prompt ingestion (l2_prefill; llama2.ts:471-473 feeds a prompt one transformer() call per token): 128 and 256 tokens
    in one call, in the reference's arithmetic (fp64 MFMA: the default) and -- reported beside it, never instead of it -- with the opt-in
    fp32-accumulate GEMMs (L2_OPT_PREFILL_F32_MFMA; DESIGN.md section 6 has their measured exactness)."""
    out, f32 = {}, {}
    for key, n_p in (("prefill_tok_s", min(128, S)), ("prefill_256_tok_s", min(256, S))):
        ptoks = (np.arange(n_p, dtype=np.int32) * 7919 + 2) % cfg.vocab_size
        for opt in (0, 1):
            ctx.set_option(runtime.OPT_PREFILL_F32_MFMA, opt)
            ctx.prefill(ptoks, 0)
            t0 = time.perf_counter()
            ctx.prefill(ptoks, 0)
            r = round(n_p / (time.perf_counter() - t0), 1)
            if opt:
                f32[str(n_p)] = r
            else:
                out[key] = r
        ctx.set_option(runtime.OPT_PREFILL_F32_MFMA, 0)
    out["prefill_f32_mfma_tok_s"] = dict(f32, what="opt-in (L2_OPT_PREFILL_F32_MFMA): the same GEMMs on v_mfma_f32_16x16x4_f32, fp32 accumulate -- not the reference's arithmetic; "
                                                     "prefill_tok_s / prefill_256_tok_s are the default fp64 form")
    return out


def tp_prediction(hdr, seed, device, single_ms):
    """What the scaling curve should look like, measured on ONE GPU (no multi-GPU node is reachable in development): for G = 2, 4, 8
    one rank's shard of the step ALONE on this GPU -- 1/G of every matrix, the 2L + 1 exchange kernels of the step running against the
    rank's own inbox, so every launch, store and flag of the product step is there and every wait is satisfied at once (l2_create_tp
    with L2_TP_SOLO_ID).  That is the step with a zero-latency exchange: an UPPER bound on tok/s.  What a node adds per exchange is the
    xGMI hop (remote uncached stores + the flag's way back) and the ranks' skew; the table prices it at 2 and 5 us per exchange."""
    out = {"how": "one rank's shard step alone on this GPU (exchange kernels against its own inbox: l2_tp_mode 5), 64 tokens from BOS; "
                  "tok_s_zero_latency = 1 / that; the other columns add 2 us / 5 us per exchange (2L + 1 per token) for the xGMI hop and rank skew",
           "exchanges_per_token": 2 * hdr[2] + 1, "1": {"shard_step_ms": round(single_ms, 4), "tok_s": round(1e3 / single_ms, 2)}}
    for G in (2, 4, 8):
        try:
            c = runtime.Context(hdr, device=device, tp_rank=0, tp_size=G, nccl_id=runtime.TP_SOLO_ID)
            c.synth_fill(seed)
            n = min(64, hdr[6])
            t_spin = time.perf_counter()
            while time.perf_counter() - t_spin < 2.5:      # untimed: pack, release, and the driver's scrub of what was released (see main)
                c.bench_decode(1, 0, n)
            ms = c.bench_decode(1, 0, n) / n
            c.close()
            nx = 2 * hdr[2] + 1
            out[str(G)] = {"shard_step_ms": round(ms, 4), "tok_s_zero_latency": round(1e3 / ms, 2),
                           "tok_s_2us_per_exchange": round(1e3 / (ms + nx * 2e-3), 2), "tok_s_5us_per_exchange": round(1e3 / (ms + nx * 5e-3), 2)}
        except Exception as e:   # noqa: BLE001 -- a prediction that cannot be made must not fail the benchmark
            out[str(G)] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def committed_prediction(name, world):
    """The prediction for this group size from the last single-GPU run whose line was committed (profiles/tp_predicted.json): a
    multi-GPU run cannot make it itself (every GPU is busy being a rank), so the first real curve is compared with this."""
    try:
        p = json.load(open(os.path.join(ROOT, "profiles", "tp_predicted.json")))[name]
        return {"from": "profiles/tp_predicted.json (single-GPU run, shard step alone)", "how": p.get("how"), str(world): p.get(str(world)), "1": p.get("1")}
    except (OSError, ValueError, KeyError):
        return {"from": None}



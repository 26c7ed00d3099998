#!/usr/bin/env python3
"""bench.py -- decode tokens/s + achieved HBM GB/s of the llama2.ts forward pass on MI355X.

A "step" is ONE transformer() call (llama2.ts:205-303) = one decoded token, weights resident in HBM,
greedy feed (-t 0, llama2.ts:476-478) from BOS at pos 0, synthetic seeded weights of the named shape.
`value` times the device-resident loop (forward + on-device argmax, no host round trip); the same K
steps through the blocking drop-in call l2_forward (logits handed to the host every token, as
llama2.ts:468-478 consumes them) are reported next to it as `dropin_tok_s`.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config llama2_7b|stories110M|stories15M]

N > 1 (launched by torch.distributed.run, one rank per GPU): Llama-2-7B is tensor-parallel (heads / FFN
rows sharded, RCCL all-reduce of d fp64 partials twice per layer) => strong scaling;
shapes that do not shard run as independent replicas => weak scaling.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from llama2_ts_amd import configs, runtime  # noqa: E402

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def avg_bytes_per_token(hdr, steps):
    return sum(configs.algorithmic_bytes_per_token(hdr, p) for p in range(steps)) / float(steps)


def dominant_kernel_bytes(cfg):
    """Algorithmic bytes of one launch of the dominant kernel: the fused rmsnorm + w1/w3 GEMV + SwiGLU
    phase (llama2.ts:276-289): both matrices once, x and the norm weight in, hb out."""
    d, h = cfg.dim, cfg.hidden_dim
    return 4 * (2 * h * d + 2 * d + h)


def cpu_baseline(name, hdr, seed):
    """The CPU oracle (C restatement of llama2.ts, one thread like the single-threaded reference) timed on
    this box's host cores on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    d, h, L, H, kv, V, S = hdr
    if L * d * h > 4 * 768 * 2048 * 12:
        # too big to materialise on the host in seconds: time 1-layer and 3-layer models of the same width
        # and extrapolate linearly in the layer count (time per token is linear in bytes streamed)
        t = {}
        for layers in (1, 3):
            o = O.Oracle((d, h, layers, H, kv, V, S), seed)
            o.forward(1, 0)
            t0 = time.perf_counter()
            tok = 1
            for pos in range(1, 7):
                tok = O.argmax(o.forward(tok, pos))
            t[layers] = (time.perf_counter() - t0) / 6.0
            o.close()
        per_layer = (t[3] - t[1]) / 2.0
        sec = t[1] + (L - 1) * per_layer
        sample = "oracle on 1- and 3-layer models of this width, 6 tokens each, extrapolated to %d layers" % L
    else:
        o = O.Oracle(hdr, seed)
        steps = 8
        sec0, _ = o.time_forward(steps)
        steps = int(max(8, min(S, 12.0 / (sec0 / steps))))
        o2 = O.Oracle(hdr, seed)
        sec_total, _ = o2.time_forward(steps)
        sec = sec_total / steps
        sample = "oracle, %d greedy tokens from BOS on the full %s shape" % (steps, name)
        o.close(); o2.close()
    return {"value": round(1.0 / sec, 4), "unit": "tokens/s", "cores": 1, "kind": "port", "sample": sample}


def run_single(args, hdr, device, tp=None):
    cfg = runtime.Config(hdr)
    if tp:
        ctx = runtime.Context(hdr, device=device, tp_rank=tp["rank"], tp_size=tp["size"], nccl_id=tp["id"])
    else:
        ctx = runtime.Context(hdr, device=device)
    ctx.synth_fill(args.seed)
    return cfg, ctx


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--config", default="llama2_7b", choices=sorted(configs.CONFIGS))
    ap.add_argument("--seed", type=int, default=configs.DEFAULT_SEED)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary stories110M line")
    ap.add_argument("--no-dropin", action="store_true", help="skip the l2_forward (host round trip per token) loop")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    hdr = configs.header(args.config)
    K = min(args.steps, hdr[6])
    W = min(args.warmup, hdr[6])
    dist = None
    tp = None
    shards = world > 1 and args.config.startswith("llama2_7b")
    if world > 1:
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo")   # rendezvous + barriers only; the data path is RCCL inside the library
        if shards:
            idbuf = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                import ctypes as C
                b = C.create_string_buffer(128)
                runtime._check(runtime.lib().l2_tp_unique_id(b))
                idbuf = torch.frombuffer(bytearray(b.raw), dtype=torch.uint8).clone()
            dist.broadcast(idbuf, 0)
            tp = {"rank": rank, "size": world, "id": bytes(idbuf.numpy().tobytes())}

    device = int(os.environ.get("L2_BENCH_FORCE_DEVICE", local_rank))   # test hook: several ranks on one GPU (replicas only)
    cfg, ctx = run_single(args, hdr, device, tp)

    def sync_all():
        if dist is not None:
            dist.barrier()
            import torch
            torch.cuda.synchronize(device)

    ctx.bench_decode(1, 0, W)                  # W untimed warm-up steps (captures the graph too)
    sync_all()
    t0 = time.perf_counter()
    dev_ms = ctx.bench_decode(1, 0, K)         # EXACTLY K timed steps, HIP events on the library's stream
    sync_all()
    wall = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([wall, dev_ms], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, dev_ms = float(t[0]), float(t[1])

    tokens_total = K if (shards or world == 1) else K * world
    value = tokens_total / wall
    bpt = avg_bytes_per_token(hdr, K)

    out = {
        "metric": "decode tokens/sec + achieved HBM GB/s (% peak), 1 GPU" if world == 1 else "decode tokens/sec (whole job)",
        "value": round(value, 3), "unit": "tokens/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": round(1e3 * wall / K, 5), "higher_is_better": True,
        "scaling": "strong" if shards else "weak", "vs_baseline": None, "dtype": "f64", "storage_dtype": "f32",
        "data": "synthetic (seeded hash generator, llama2.c-v0 layout)",
        "config": {"workload": "%s batch-1 greedy decode, %d tokens from BOS (-t 0 -s 1 -n %d)" % (args.config, K, K),
                   "header": list(hdr), "parallelism": ("tp%d" % world) if shards else ("replicas%d" % world if world > 1 else "single"),
                   "loop": ("device-resident (forward + argmax on GPU), eager launches with 2L fp64 RCCL all-reduces + 1 all-gather per token"
                            if shards else "device-resident (forward + argmax on GPU, one hipGraph replay per token)")},
        "device_ms_per_step": round(dev_ms / K, 5),
        "algorithmic_bytes_per_token": int(bpt),
        "hbm_gbs_end_to_end": round(bpt * value / 1e9 / (1 if shards or world == 1 else world), 2),
        "hbm_frac_end_to_end": round(bpt * value / 1e9 / (1 if shards or world == 1 else world) / HBM_PEAK_GBS / (world if shards else 1), 4),
    }

    if rank == 0 and world == 1:
        # the same K steps through the blocking drop-in boundary (logits to the host every token)
        if not args.no_dropin:
            tok = 1
            ctx.forward(1, 0)
            t0 = time.perf_counter()
            for pos in range(K):
                lg = ctx.forward(tok, pos)
                tok = int(np.argmax(lg))
            dt = time.perf_counter() - t0
            out["dropin_tok_s"] = round(K / dt, 3)
        # dominant kernel alone, HIP events on the library's stream
        iters = 200 if cfg.dim * cfg.hidden_dim < (1 << 24) else 50
        kms_isolated = ctx.bench_gemv(runtime.T_W1, cfg.n_layers // 2, iters)
        # the dominant kernel timed IN SITU: HIP event pairs around each of its launches inside K decode steps
        # (eager launches of the same kernels, on the library's stream)
        kus, nlaunch = ctx.bench_dominant_in_situ(1, 0, min(K, 128))
        kms = kus * 1e-3
        kb = dominant_kernel_bytes(cfg)
        ach = kb / (kms * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.config)
        if os.path.exists(tfile):
            traffic = json.load(open(tfile)).get("hbm_bytes_per_launch")
        out["roofline"] = {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                           "kernel": "phase_kernel<MODE_W13> (rmsnorm + w1/w3 GEMV + SwiGLU, llama2.ts:276-289)",
                           "bytes_per_launch": kb, "avg_launch_us": round(kms * 1e3, 3), "launches_timed": nlaunch,
                           "how": "HIP event pair around every launch inside a decode run on the library's stream",
                           "isolated_back_to_back_us": round(kms_isolated * 1e3, 3)}
        per_kernel = {}
        for nm, kind in (("qkv", runtime.T_WQ), ("wo", runtime.T_WO), ("w13", runtime.T_W1), ("w2", runtime.T_W2), ("wcls", runtime.T_WCLS)):
            ms = ctx.bench_gemv(kind, 0, iters)
            d, h, V = cfg.dim, cfg.hidden_dim, cfg.vocab_size
            nb = {"qkv": 3 * d * d, "wo": d * d, "w13": 2 * d * h, "w2": d * h, "wcls": V * d}[nm] * 4
            per_kernel[nm] = {"us": round(ms * 1e3, 3), "GBs": round(nb / (ms * 1e-3) / 1e9, 1)}
        out["per_kernel"] = per_kernel
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.config, hdr, args.seed)
        if not args.no_extra:
            # the same loop near the end of the context window (attention reads ~S rows per layer; split-attention form)
            S = hdr[6]
            if S >= 1024:
                n_long = 128
                p0 = S - n_long
                ctx.bench_decode(1, p0, 8)
                ms = ctx.bench_decode(1, p0, n_long)
                b_long = sum(configs.algorithmic_bytes_per_token(hdr, p) for p in range(p0, S)) / n_long
                out["long_context"] = {"positions": [p0, S - 1], "value": round(n_long / (ms * 1e-3), 3), "unit": "tokens/s",
                                       "ms_per_step": round(ms / n_long, 5), "algorithmic_bytes_per_token": int(b_long),
                                       "hbm_frac_end_to_end": round(b_long * n_long / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            # the sampled branch of the loop on the device (l2_decode_sample): reference semantics, same token ids per seed
            n_s = min(64, hdr[6])
            samp = {}
            for nm, t, p in (("sample_t0.9", 0.9, 1.0), ("topp0.9_t0.9", 0.9, 0.9)):
                ctx.decode_sample(1, 0, 8, t, p, 42)
                t0 = time.perf_counter()
                ctx.decode_sample(1, 0, n_s, t, p, 42)
                samp[nm] = round(n_s / (time.perf_counter() - t0), 3)
            out["sampled_decode_tok_s"] = samp
            # prompt ingestion (l2_prefill: chunks of up to 64 tokens on the fp64 MFMA path) next to the token-by-token loop it replaces
            n_p = min(128, hdr[6])
            ptoks = (np.arange(n_p, dtype=np.int32) * 7919 + 2) % cfg.vocab_size
            ctx.prefill(ptoks[:64], 0)
            t0 = time.perf_counter()
            ctx.prefill(ptoks, 0)
            out["prefill_tok_s"] = round(n_p / (time.perf_counter() - t0), 1)
    ctx.close()

    if rank == 0 and world == 1 and not args.no_extra and args.config == "llama2_7b":
        # BASELINE.json's metric names stories110M too: same measurement, secondary line in the same JSON
        h2 = configs.header("stories110M")
        c2 = runtime.Context(h2, device=local_rank)
        c2.synth_fill(args.seed)
        K2 = min(256, h2[6])
        c2.bench_decode(1, 0, 8)
        t0 = time.perf_counter()
        c2.bench_decode(1, 0, K2)
        w2 = time.perf_counter() - t0
        b2 = avg_bytes_per_token(h2, K2)
        out["stories110M"] = {"value": round(K2 / w2, 2), "unit": "tokens/s", "steps": K2,
                              "hbm_gbs_end_to_end": round(b2 * K2 / w2 / 1e9, 2),
                              "hbm_frac_end_to_end": round(b2 * K2 / w2 / 1e9 / HBM_PEAK_GBS, 4)}
        c2.close()

    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

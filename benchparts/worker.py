"""`bench.py --worker-stage <stage>`: ONE rank's GPU work for ONE way of forming the group, as a fresh process started by the rank's
supervisor (benchparts/ranks.py).  It prints marks the supervisor times (`@@l2 started / created / proved`, `@@l2 failed <why>`) and, on
rank 0, the one JSON line.  Stages: p2p (RCCL communicator + the one-shot peer-to-peer exchange over xGMI), rccl (RCCL collectives only),
file (no RCCL: the ranks meet through files and exchange over IPC-mapped inboxes), replicas (no group: N independent contexts).

The group is PROVED before it is timed: created on every rank (all ranks report over gloo before anyone enters a collective of the
library's), filled, and sixteen tokens decoded that must be the same on every rank and equal the real reference's golden tokens where a
fixture exists.  A stage that cannot do that ends here with `@@l2 failed`; the supervisors then start the next stage from fresh processes
-- this process is never reused."""
import ctypes
import datetime
import json
import os
import sys
import threading
import time

from .common import HBM_PEAK_GBS, ROOT, avg_bytes_per_token, parity_block, under_profiler
from .ranks import MARK, deadlines
from .single import committed_prediction, dispatch_note

from llama2_ts_amd import configs, runtime


def mark(what):
    print(MARK + what, flush=True)


def die_with_parent():
    """The worker leads a session of its own (so that its supervisor can kill its whole group): it must not outlive a supervisor that the
    launcher killed.  PR_SET_PDEATHSIG = 1."""
    try:
        ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, 9, 0, 0, 0)
        if os.getppid() == 1:
            os._exit(6)
    except OSError:
        pass


def committed_cpu_baseline(name):
    """cpu_baseline is taken on rank 0 at N = 1 only (the host cores are busy being ranks here): the record of the last single-GPU run
    whose line was committed, with where it came from."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "cpu_baseline_n1.json")))[name]
        return dict(rec, copied_from="profiles/cpu_baseline_n1.json (the N = 1 run of `%s`: rank 0, one host core)" % rec.get("run", "bench.py"))
    except (OSError, ValueError, KeyError):
        return {"value": None, "why": "no committed N = 1 record for this configuration (profiles/cpu_baseline_n1.json)"}


def shard_dominant_bytes(cfg, G):
    """Algorithmic bytes of one launch of the dominant kernel on ONE rank of G: its h / G rows of w1 and of w3, x and the norm weight in, its
    slice of hb out (llama2.ts:276-289; SURVEY.md 8(e): w1 / w3 are sharded by rows)."""
    d, hl = cfg.dim, cfg.hidden_dim // G
    return 4 * (2 * hl * d + 2 * d + hl)


def worker(args):
    die_with_parent()
    import torch
    import torch.distributed as dist
    dl = deadlines()
    stage = args.worker_stage
    rank, world, local_rank = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    port = os.environ["L2_BENCH_WORKER_PORT"]
    # (a store of the workers' own, hosted by rank 0's worker: the launcher's agent store -- TORCHELASTIC_USE_AGENT_STORE -- belongs to the supervisors)
    store = dist.TCPStore("127.0.0.1", int(port), world, is_master=(rank == 0), timeout=datetime.timedelta(seconds=dl["start"]))
    dist.init_process_group("gloo", store=store, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=dl["create"] + dl["prove"]))
    mark("started")
    if under_profiler():
        os.environ.setdefault("L2_USE_GRAPH", "0")
    hdr = configs.header(args.config)
    K, W = min(args.steps, hdr[6]), min(args.warmup, hdr[6])
    shards = stage != "replicas"
    device = int(os.environ.get("L2_BENCH_FORCE_DEVICE", local_rank))   # test hook: several ranks on one GPU
    cfg = runtime.Config(hdr)
    ctx = None

    def fail(why):
        mark("failed " + " ".join(str(why).split()))
        if ctx is not None:
            try:
                ctx.close()
            except Exception:      # noqa: BLE001
                pass
        sys.stdout.flush()
        os._exit(4)

    def golden_tokens(n):
        try:
            g = json.load(open(os.path.join(ROOT, "tests", "golden", args.config + ".json")))
            return g["argmax"][:n] if g.get("seed") == args.seed and g.get("prompt") is None and len(g["argmax"]) >= n else None
        except (OSError, ValueError, KeyError):
            return None

    # ---- phase: create.  Inside the rank a watchdog bounds it too (RCCL has no timeout: an ncclCommInitRank that hangs would sit here
    # until the supervisor's deadline; the watchdog says which call it was)
    dog = threading.Timer(0.9 * dl["create"], lambda: (mark("failed l2_create%s did not return within %.0f s (watchdog inside the rank)" % ("_tp" if shards else "", 0.9 * dl["create"])), os._exit(5)))
    dog.daemon = True
    dog.start()
    err = ""
    try:
        if shards:
            nid = b"\x01" * 128      # the file rendezvous ignores it
            if stage in ("p2p", "rccl") and not os.environ.get("L2_TP_IPC_DIR"):
                idb = torch.zeros(128, dtype=torch.uint8)      # a communicator id is good for one ncclCommInitRank round: a fresh one per stage
                if rank == 0:
                    b = ctypes.create_string_buffer(128)
                    if runtime.lib().l2_tp_unique_id(b) == 0:
                        idb = torch.frombuffer(bytearray(b.raw), dtype=torch.uint8).clone()
                dist.broadcast(idb, 0)
                nid = bytes(idb.numpy().tobytes())
                if not any(nid):
                    raise RuntimeError("rank 0 could not create an RCCL id (%s)" % runtime.lib().l2_last_error().decode("utf8", "replace"))
            ctx = runtime.Context(hdr, device=device, tp_rank=rank, tp_size=world, nccl_id=nid)
        else:
            ctx = runtime.Context(hdr, device=device)
    except Exception as e:      # noqa: BLE001 -- whatever it is, the other ranks have to hear about it
        err = "%s: %s" % (type(e).__name__, e)
    dog.cancel()
    created = [None] * world
    dist.all_gather_object(created, {"rank": rank, "err": err})
    errs = [r["err"] for r in created if r["err"]]
    if errs:
        fail(errs[0])
    mark("created")

    # ---- phase: prove.  Fill, decode sixteen tokens (at 32 layers: 1 040 exchanges, every one of them part of the proof), compare
    toks, err = [], ""
    try:
        ctx.synth_fill(args.seed)
        toks = ctx.decode_greedy(1, 0, min(16, K)).tolist()
    except Exception as e:      # noqa: BLE001
        err = "%s: %s" % (type(e).__name__, e)
    every = [None] * world
    dist.all_gather_object(every, {"rank": rank, "err": err, "tokens": toks, "mode": ctx.tp_mode_id() if not err else -1})
    errs = [r["err"] for r in every if r["err"]]
    same = all(r["tokens"] == every[0]["tokens"] for r in every)
    gold = golden_tokens(len(every[0]["tokens"]))
    if errs:
        fail(errs[0])
    if not same:
        fail("ranks decoded different tokens: %s" % [r["tokens"] for r in every])
    if gold is not None and every[0]["tokens"] != gold:
        fail("tokens %s differ from the reference golden %s" % (every[0]["tokens"], gold))
    proof = {"tokens": every[0]["tokens"], "same_on_every_rank": same, "equals_reference_golden": (None if gold is None else every[0]["tokens"] == gold)}
    mark("proved")

    def sync_all():
        dist.barrier()
        torch.cuda.synchronize(device)

    # ---- untimed: as at N = 1 (clock ramp for small models; for repacked ones the driver's scrub of the released row-major tensors)
    if configs.checkpoint_bytes(hdr) < (1 << 30):
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < 0.3:
            ctx.bench_decode(1, 0, min(64, hdr[6]))
    else:
        ctx.bench_decode(1, 0, min(8, hdr[6]))
        if ctx.get_option(runtime.OPT_PACKED_MIB) > 0:
            t_spin = time.perf_counter()
            while time.perf_counter() - t_spin < 3.5:
                ctx.bench_decode(1, 0, min(64, hdr[6]))
    if W > 1:
        ctx.bench_decode(1, 0, W - 1)
    if W > 0:
        ctx.bench_decode(1, K - 1, 1)
    sync_all()
    t0 = time.perf_counter()
    dev_ms = ctx.bench_decode(1, 0, K)         # EXACTLY K timed steps
    sync_all()
    wall = time.perf_counter() - t0
    timed_tokens = ctx.bench_tokens(K)
    t = torch.tensor([wall, dev_ms], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall, dev_ms = float(t[0]), float(t[1])

    tokens_total = K if shards else K * world
    value = tokens_total / wall
    bpt = avg_bytes_per_token(hdr, 0, K)
    per_gpu_streams = 1 if shards else world
    out = {
        "metric": "decode tokens/sec (whole job)", "value": round(value, 3), "unit": "tokens/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": round(1e3 * wall / K, 5), "higher_is_better": True, "scaling": "strong" if shards else "weak", "vs_baseline": None,
        "dtype": "f64", "storage_dtype": "f32", "data": "synthetic (seeded hash generator, llama2.c-v0 layout)",
        "config": {"workload": "%s batch-1 greedy decode, %d tokens from BOS (-t 0 -s 1 -n %d)" % (args.config, K, K), "header": list(hdr),
                   "parallelism": ("tp%d" % world) if shards else "replicas%d" % world,
                   "loop": ("device-resident (forward + argmax on GPU); tensor-parallel step: %s; dispatch: %s" % (ctx.tp_mode(), dispatch_note(ctx))
                            if shards else "device-resident (forward + argmax on GPU, %s)" % dispatch_note(ctx)),
                   "weights_mib": {"on_device": ctx.get_option(runtime.OPT_WEIGHT_MIB), "repacked": ctx.get_option(runtime.OPT_PACKED_MIB),
                                   "checkpoint": configs.checkpoint_bytes(hdr) >> 20}},
        "device_ms_per_step": round(dev_ms / K, 5), "algorithmic_bytes_per_token": int(bpt),
        "hbm_gbs_end_to_end": round(bpt * value / 1e9 / per_gpu_streams, 2),
        "hbm_frac_end_to_end": round(bpt * value / 1e9 / per_gpu_streams / HBM_PEAK_GBS / (world if shards else 1), 4),
    }
    out["parity"] = parity_block(args.config, args.seed, timed_tokens)
    ranks = [None] * world
    dist.all_gather_object(ranks, {"rank": rank, "device": device, "tp_mode": ctx.tp_mode_id()})
    out["tp"] = {"ranks": world, "sharded": bool(shards), "stage": stage, "l2_tp_mode": sorted({r["tp_mode"] for r in ranks}),
                 "devices": [r["device"] for r in ranks], "step": ctx.tp_mode(), "proved_before_timing": proof}
    if shards:
        out["tp_predicted"] = committed_prediction(args.config, world)
    if under_profiler() and os.environ.get("L2_USE_GRAPH") == "0":
        out["profiled"] = "this run was started under a profiler: eager launches (host-bound; read the kernel durations, not `value`)"
    # ---- roofline of the dominant kernel on THIS rank's shard, in situ (every rank decodes: the step's exchanges are collective)
    try:
        kus, nlaunch = ctx.bench_dominant_in_situ(1, 0, min(K, 64))
        kb = shard_dominant_bytes(cfg, world if shards else 1)
        ach = kb / (kus * 1e-6) / 1e9
        out["roofline"] = {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                           "traffic_how": "not collected in a multi-rank run (the counter passes are single-process children: N = 1)",
                           "kernel": "rmsnorm + w1/w3 GEMV + SwiGLU (llama2.ts:276-289), rank 0's shard: %d of %d rows of each matrix" % (cfg.hidden_dim // (world if shards else 1), cfg.hidden_dim),
                           "bytes_per_launch": kb, "avg_launch_us": round(kus, 3), "launches_timed": nlaunch,
                           "how": "HIP start/stop events attached to every dispatch of this kernel inside a decode run of the whole group (hipExtLaunchKernelGGL), rank 0"}
    except Exception as e:      # noqa: BLE001 -- a side measurement must not cost the line
        out["roofline"] = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None, "why": "%s: %s" % (type(e).__name__, e)}
    out["cpu_baseline"] = committed_cpu_baseline(args.config)
    ctx.close()
    bad = out["parity"].get("equal_to_reference_golden") is False
    if rank == 0:
        print(json.dumps(out), flush=True)
        if bad:
            print("bench.py: PARITY FAILURE: the timed decode does not reproduce the reference's golden tokens", file=sys.stderr)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(3 if bad else 0)

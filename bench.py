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

N > 1 (launched by torch.distributed.run, one rank per GPU): Llama-2-7B is tensor-parallel (heads / FFN
rows sharded, fp64 all-reduce of d partials twice per layer) => strong scaling;
shapes that do not shard run as independent replicas => weak scaling.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from llama2_ts_amd import configs, runtime  # noqa: E402

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
DOMINANT = "rmsnorm + w1/w3 GEMV + SwiGLU (llama2.ts:276-289)"


def avg_bytes_per_token(hdr, p0, p1):
    return sum(configs.algorithmic_bytes_per_token(hdr, p) for p in range(p0, p1)) / float(p1 - p0)


def dominant_kernel_bytes(cfg):
    """Algorithmic bytes of one launch of the dominant kernel: the fused rmsnorm + w1/w3 GEMV + SwiGLU
    phase (llama2.ts:276-289): both matrices once, x and the norm weight in, hb out."""
    d, h = cfg.dim, cfg.hidden_dim
    return 4 * (2 * h * d + 2 * d + h)


# ---- HBM traffic of the dominant kernel: rocprofv3 --pmc over a child of this script ------------------------------
def pmc_child(name, seed):
    """Target of the counter passes: the model of this config, one forward, a few launches of the dominant kernel."""
    ctx = runtime.Context(configs.header(name))
    ctx.synth_fill(seed)
    ctx.forward(1, 0)
    ctx.bench_gemv(runtime.T_W1, ctx.cfg.n_layers // 2, 6)
    ctx.close()


PROFILER_ENV = ("ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "ROCPROFILER_REGISTER_FORCE_LOAD", "ROCP_TOOL_LIBRARY")


def under_profiler():
    """This process was itself started by rocprofv3 (its tool library is preloaded): it must not start profiled children."""
    if any(os.environ.get(k) for k in PROFILER_ENV):
        return True
    return "rocprof" in os.environ.get("LD_PRELOAD", "")


def clean_child_env(**extra):
    """Environment for a child process: nothing of a profiler that may wrap THIS process leaks into it."""
    env = {k: v for k, v in os.environ.items() if k not in PROFILER_ENV and not k.startswith("ROCPROF")}
    if "rocprof" in env.get("LD_PRELOAD", ""):
        del env["LD_PRELOAD"]
    env.update(extra)
    return env


def pmc_traffic(name, seed):
    """FETCH_SIZE and WRITE_SIZE in SEPARATE passes (kernel trace only), corrected as the MI355X guide prescribes:
    both are in KiB and FETCH_SIZE reports exactly half of a 16-byte-per-lane coalesced stream on gfx950."""
    if under_profiler():
        return None, "skipped: this run is itself being profiled"
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not found"
    out = {}
    work = tempfile.mkdtemp(prefix="l2_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(work, ctr)
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "p", "--",
                   sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child", "--config", name, "--seed", str(seed)]
            env = clean_child_env(TMPDIR="/tmp", L2_USE_GRAPH="0")
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=600)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, "rocprofv3 --pmc %s failed (rc %d)" % (ctr, r.returncode)
            vals = []
            for row in csv.DictReader(open(files[0])):
                kn = row["Kernel_Name"]
                if row["Counter_Name"] == ctr and ("phase_kernel<2," in kn or "phase_small_kernel<2," in kn):
                    vals.append(float(row["Counter_Value"]))
            if len(vals) < 3:
                return None, "no launches of the dominant kernel in the %s pass" % ctr
            vals = vals[2:]   # the first launches follow a forward: drop them like warm-up
            out[ctr] = sum(vals) / len(vals)
    except Exception as e:   # noqa: BLE001 -- a missing profiler must not fail the benchmark
        return None, "pmc pass: %r" % (e,)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    total = out["FETCH_SIZE"] * 1024.0 * 2.0 + out["WRITE_SIZE"] * 1024.0
    return int(total), ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of this run (FETCH_SIZE KiB x 1024 x 2 + "
                        "WRITE_SIZE KiB x 1024; per launch, mean of %d)" % len(vals))


def kernel_trace_us(name, seed):
    """Average duration of the dominant kernel as a kernel trace reports it (rocprofv3 --kernel-trace, no counters): a child of
    this script decodes 24 tokens with eager launches.  Quoted next to the HIP-event figure: on a 5 us kernel the event pair
    itself costs about 1 us."""
    if under_profiler():
        return None
    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    work = tempfile.mkdtemp(prefix="l2_kt_", dir="/tmp")
    try:
        cmd = [exe, "--kernel-trace", "--output-format", "csv", "-d", work, "-o", "k", "--",
               sys.executable, os.path.join(ROOT, "bench.py"), "--trace-child", "--config", name, "--seed", str(seed)]
        env = clean_child_env(TMPDIR="/tmp", L2_USE_GRAPH="0", L2_PROFILE_SYNC="1", L2_TEST_HOOKS="1")
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
        files = glob.glob(os.path.join(work, "**", "*kernel_trace.csv"), recursive=True)
        if r.returncode != 0 or not files:
            return None
        durs = []
        for row in csv.DictReader(open(files[0])):
            kn = row["Kernel_Name"]
            if "phase_kernel<2," in kn or "phase_small_kernel<2," in kn:
                durs.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)
        durs = durs[len(durs) // 4:]     # the first quarter is warm-up (clocks, caches)
        return round(sum(durs) / len(durs), 3) if durs else None
    except Exception:   # noqa: BLE001 -- a missing profiler must not fail the benchmark
        return None
    finally:
        shutil.rmtree(work, ignore_errors=True)


def trace_child(name, seed):
    ctx = runtime.Context(configs.header(name))
    ctx.synth_fill(seed)
    ctx.decode_greedy(1, 0, min(24, configs.header(name)[6]))
    ctx.close()


def dropin_loop(ctx, K):
    """K steps through the blocking drop-in call (llama2.ts:468 -> 478: logits land on the host every token, argmax there)."""
    tok = 1
    ctx.forward(1, 0)
    t0 = time.perf_counter()
    toks = []
    for pos in range(K):
        tok = int(np.argmax(ctx.forward(tok, pos, view=True)))
        toks.append(tok)
    return K / (time.perf_counter() - t0), toks


def dropin_child(name, seed):
    """`--dropin-child`: the drop-in loop alone, as a process of its own, so that it can be timed under another runtime setting
    (AMD_DIRECT_DISPATCH=0: the HIP runtime submits from a thread of its own) than the parent was started with.  Prints one JSON line."""
    hdr = configs.header(name)
    ctx = runtime.Context(hdr)
    ctx.synth_fill(seed)
    K = min(256, hdr[6])
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.3:
        ctx.bench_decode(1, 0, 64)
    dropin_loop(ctx, K)
    rate, toks = dropin_loop(ctx, K)
    ctx.close()
    print(json.dumps({"dropin_tok_s": round(rate, 2), "tokens": toks}))


def dropin_direct_dispatch_off(name, seed):
    """dropin_tok_s with the one runtime knob that moves the replayed-graph floor (profiles/r04/direct_dispatch_ab.txt): measured in a
    child, reported beside the default -- a deployment may set it, the library does not change the host's runtime configuration."""
    try:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--dropin-child", "--config", name, "--seed", str(seed)]
        r = subprocess.run(cmd, env=clean_child_env(AMD_DIRECT_DISPATCH="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        j = json.loads(r.stdout.decode().strip().splitlines()[-1])
        gold = golden_argmax(name, seed)
        same = None if gold is None else j["tokens"] == gold[:len(j["tokens"])]
        out = {"value": j["dropin_tok_s"], "env": "AMD_DIRECT_DISPATCH=0", "equal_to_reference_golden": same}
        if same is False:
            # seen on ROCm 7.2 (round 5): under AMD_DIRECT_DISPATCH=0 the blocking call hands back logits of the wrong step from the
            # second token on (with and without the zero-copy logits) -- the rate above is then not a measurement of this path
            out["first_mismatch"] = next(i for i, (a, b) in enumerate(zip(j["tokens"], gold)) if a != b)
            out["note"] = "tokens differ from the reference under this runtime setting: not a configuration to deploy; the rate is not comparable"
        return out
    except Exception as e:   # noqa: BLE001 -- a side measurement must not fail the benchmark
        return {"value": None, "why": "%s: %s" % (type(e).__name__, e)}


def write_checkpoint(ctx, path):
    """The context's weights as a llama2.c-v0 file (header + tensors in checkpoint order, llama2.ts:80-93, 112-129), read back from
    the DEVICE through l2_read_tensor: what the Node host loads below is what the decode above ran on."""
    cfg = ctx.cfg
    with open(path, "wb") as f:
        f.write(np.asarray(cfg.header, dtype="<i4").tobytes())
        for kind, layers, count in runtime.tensor_shapes(cfg):
            for layer in range(max(layers, 1)):
                ctx.read_tensor(kind, layer if layers else -1, 0, count).tofile(f)


def napi_dropin(ctx, name, seed, K):
    """The boundary the contract names: the SAME K steps through the real N-API addon under Node -- host/l2_run.mjs --loop host is the
    reference's loop (one transformer() per position, llama2.ts:468; first maximum on the host, :478) over l2_backend's readWeights /
    transformer, clock started after the first iteration like llama2.ts:507 -- on a checkpoint file written from this context's weights."""
    node = shutil.which("node")
    addon = os.path.join(ROOT, "llama2.ts_amd", "host", "l2_napi.node")
    if not node or not os.path.exists(addon):
        return {"value": None, "why": "no node / no built addon on this box"}
    if configs.checkpoint_bytes(ctx.cfg.header) > (2 << 30):
        return {"value": None, "why": "checkpoint of %.0f GB: not written to a file inside a benchmark run (ctypes dropin_tok_s is the figure for this shape)"
                % (configs.checkpoint_bytes(ctx.cfg.header) / 2.0 ** 30)}
    path = os.path.join(tempfile.gettempdir(), "l2_napi_%s_%d_%d.bin" % (name, seed, os.getpid()))
    try:
        write_checkpoint(ctx, path)
        cmd = [node, os.path.join(ROOT, "llama2.ts_amd", "host", "l2_run.mjs"), path, "--steps", str(K), "--loop", "host", "--metrics"]
        best = None
        for _ in range(2):      # the first run also pays the page cache and the addon's first dlopen
            r = subprocess.run(cmd, env=clean_child_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            if r.returncode != 0:
                return {"value": None, "why": "node failed: %s" % r.stderr.decode("utf8", "replace")[-200:]}
            m = [json.loads(ln)["metrics"] for ln in r.stderr.decode().splitlines() if ln.startswith("{") and '"metrics"' in ln][-1]
            toks = json.loads(r.stdout.decode().strip().splitlines()[-1])["tokens"]
            if best is None or m["tok_s"] > best[0]["tok_s"]:
                best = (m, toks)
        m, toks = best
        gold = golden_argmax(name, seed)
        return {"value": round(m["tok_s"], 2), "unit": "tokens/s", "tokens_timed": m["tokens_timed"], "timer": m["timer"], "hbm_frac": round(m["hbm_frac"], 4),
                "how": "node host/l2_run.mjs --loop host --metrics (N-API addon -> C ABI), best of 2 runs",
                "equal_to_reference_golden": (None if gold is None else toks == gold[:len(toks)])}
    except Exception as e:   # noqa: BLE001
        return {"value": None, "why": "%s: %s" % (type(e).__name__, e)}
    finally:
        try:
            os.remove(path)
        except OSError:
            pass


# ---- parity of the run that was timed ---------------------------------------------------------------------------
def golden_argmax(name, seed):
    """The tokens the REAL reference chose on this synthetic checkpoint (tests/golden/<config>.json, written by
    oracle/make_goldens.py from a run of /root/reference/llama2.ts): data, so it travels to the GPU box."""
    try:
        g = json.load(open(os.path.join(ROOT, "tests", "golden", name + ".json")))
    except (OSError, ValueError):
        return None
    if g.get("seed") != seed or g.get("prompt") is not None or g.get("tokens_fed", [None])[0] != 1:
        return None
    return list(g["argmax"])


def parity_block(name, seed, tokens):
    """Compare the tokens of the TIMED decode with the reference's golden tokens, step by step."""
    gold = golden_argmax(name, seed)
    if gold is None:
        return {"steps_checked": 0, "equal_to_reference_golden": None, "why": "no reference golden for this config / seed"}
    n = min(len(gold), len(tokens))
    got = [int(t) for t in tokens[:n]]
    ok = got == gold[:n]
    out = {"steps_checked": n, "steps_timed": len(tokens), "equal_to_reference_golden": ok,
           "golden": "tests/golden/%s.json (%d steps of the real reference, -t 0 -s 1)" % (name, len(gold)),
           "what": "tokens of the timed device-resident decode (l2_bench_tokens) vs the reference's argmax per step"}
    if not ok:
        first = next(i for i in range(n) if got[i] != gold[i])
        out["first_mismatch"] = {"step": first, "got": got[first], "reference": gold[first]}
    return out


# ---- CPU baseline ------------------------------------------------------------------------------------------------
def mem_available_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                return int(line.split()[1]) / 1048576.0
    except OSError:
        pass
    return 0.0


def cpu_baseline(name, hdr, seed):
    """The CPU oracle (C restatement of llama2.ts, ONE thread like the single-threaded reference) timed on this box's
    host cores on a bounded sample of the same workload.  (Its synthetic-weight generator may use every core; the
    timed forward passes do not.)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    d, h, L, H, kv, V, S = hdr
    weights_gb = configs.checkpoint_bytes(hdr) / 2.0 ** 30
    extrapolated = False
    if weights_gb > 2.0 and mem_available_gb() > weights_gb + 8.0 and not os.environ.get("L2_BENCH_CPU_EXTRAPOLATE"):
        o = O.Oracle(hdr, seed)          # the full model in host memory
        tok = O.argmax(o.forward(1, 0))
        n = 2
        t0 = time.perf_counter()
        for pos in range(1, 1 + n):
            tok = O.argmax(o.forward(tok, pos))
        sec = (time.perf_counter() - t0) / n
        o.close()
        sample = "oracle, %d greedy tokens (after 1 untimed) on the full %d-layer %s shape, %.0f GB of weights in host memory" % (n, L, name, weights_gb)
    elif weights_gb > 2.0:
        # not enough host memory for the whole model: 1- and 3-layer models of the same width, linear in the layer count
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
        sec = t[1] + (L - 1) * (t[3] - t[1]) / 2.0
        extrapolated = True
        sample = "oracle on 1- and 3-layer models of this width, 6 tokens each, extrapolated to %d layers (host memory too small for the full model)" % L
    else:
        o = O.Oracle(hdr, seed)
        sec0, _ = o.time_forward(8)
        steps = int(max(8, min(S, 12.0 / (sec0 / 8))))
        o2 = O.Oracle(hdr, seed)
        sec_total, _ = o2.time_forward(steps)
        sec = sec_total / steps
        sample = "oracle, %d greedy tokens from BOS on the full %s shape" % (steps, name)
        o.close(); o2.close()
    out = {"value": round(1.0 / sec, 4), "unit": "tokens/s", "cores": 1, "kind": "port", "sample": sample, "extrapolated": extrapolated,
           "host_cpu": host_cpu_model()}
    out.update(reference_js_figure(name))
    out["js_port"] = js_port_baseline(name, hdr, seed)
    if (out["js_port"] or {}).get("value"):
        out["note"] = ("same box, one core each: the C port %.2f tok/s, the reference's arithmetic under this box's own Node (js_port) %.2f tok/s; "
                       "reference_js_tok_s is the reference ITSELF, but on the build container's slower CPU" % (out["value"], out["js_port"]["value"]))
    elif out.get("reference_js_tok_s"):
        out["port_vs_reference_js"] = round(out["value"] / out["reference_js_tok_s"], 2)
        out["note"] = ("the C port on this box's host core runs %.1fx what the reference itself did under Node in the build container (a slower CPU): "
                       "no JS figure from this box for this shape (js_port.why)" % out["port_vs_reference_js"])
    return out


def js_port_baseline(name, hdr, seed):
    """The reference's arithmetic in the reference's RUNTIME on this box: oracle/llama2_oracle.mjs (a JavaScript restatement of
    llama2.ts:168-303, bit-identical to the real reference on every golden fixture: tests/test_oracle_golden.py) under this box's
    Node, one thread like the reference, on the same synthetic checkpoint (written to /tmp by the C generator), tok/s as the
    reference counts them (llama2.ts:507, 511: the clock starts after the first token).  Bounded to ~10 s of JS time; checkpoints
    beyond 2 GB are skipped (a 27 GB file would have to be written and read back: minutes)."""
    node = shutil.which("node")
    if not node:
        return {"value": None, "why": "no node on this box"}
    if configs.checkpoint_bytes(hdr) > (2 << 30):
        return js_port_in_process(name, hdr, seed, node)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    ref = reference_js_figure(name).get("reference_js_tok_s") or 5.0
    steps = int(max(8, min(hdr[6], 10.0 * ref)))
    path = os.path.join(tempfile.gettempdir(), "l2_js_%s_%d_%d.bin" % (name, seed, os.getpid()))
    try:
        O.synth_write(hdr, seed, path)
        r = subprocess.run([node, os.path.join(ROOT, "oracle", "llama2_oracle.mjs"), path, str(steps)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        if r.returncode != 0:
            return {"value": None, "why": "node failed: %s" % r.stderr.decode("utf8", "replace")[-200:]}
        j = json.loads(r.stdout.decode())
        gold = golden_argmax(name, seed)
        return {"value": round(j["tok_s"], 4), "unit": "tokens/s", "cores": 1, "kind": "port", "runtime": "node %s on this box" % j.get("node"),
                "sample": "oracle/llama2_oracle.mjs, %d greedy tokens from BOS on the full %s shape" % (steps, name),
                "tokens_equal_reference_golden": (None if gold is None else j["tokens"] == gold[:steps])}
    except Exception as e:   # noqa: BLE001 -- a baseline that cannot be taken must not fail the benchmark
        return {"value": None, "why": "%s: %s" % (type(e).__name__, e)}
    finally:
        try:
            os.remove(path)
        except OSError:
            pass


def js_port_in_process(name, hdr, seed, node):
    """The same for a checkpoint too large to go through a file (Llama-2-7B: 27 GB): llama2_oracle.mjs --synth fills its typed arrays
    IN PROCESS with the repo's generator restated in JavaScript (pinned per tensor against oracle_cli's bytes by
    tests/test_oracle_golden.py), then times 3 tokens after the first like llama2.ts:507, 511.  Needs the model + KV caches in host
    memory; generating 6.7e9 values in one JS thread takes a couple of minutes, outside the timed region."""
    need_gb = configs.checkpoint_bytes(hdr) / 2.0 ** 30 + 2.0 * hdr[2] * hdr[6] * hdr[0] * 4 / 2.0 ** 30 + 4.0
    if mem_available_gb() < need_gb:
        return {"value": None, "why": "MemAvailable %.0f GB < the %.0f GB the full model needs in this process" % (mem_available_gb(), need_gb)}
    if os.environ.get("L2_BENCH_SKIP_JS_7B"):
        return {"value": None, "why": "skipped (L2_BENCH_SKIP_JS_7B)"}
    steps = 4
    try:
        r = subprocess.run([node, "--max-old-space-size=4096", os.path.join(ROOT, "oracle", "llama2_oracle.mjs"), "--synth", ",".join(str(v) for v in list(hdr) + [seed]), str(steps)],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
        if r.returncode != 0:
            return {"value": None, "why": "node failed: %s" % r.stderr.decode("utf8", "replace")[-200:]}
        j = json.loads(r.stdout.decode())
        gold = golden_argmax(name, seed)
        return {"value": round(j["tok_s"], 4), "unit": "tokens/s", "cores": 1, "kind": "port", "runtime": "node %s on this box" % j.get("node"),
                "sample": "oracle/llama2_oracle.mjs --synth (weights generated in process, %.0f s), %d greedy tokens from BOS on the full %s shape, clock started after the first"
                          % (j.get("load_s") or 0.0, steps, name),
                "tokens_equal_reference_golden": (None if gold is None else j["tokens"] == gold[:steps])}
    except Exception as e:   # noqa: BLE001
        return {"value": None, "why": "%s: %s" % (type(e).__name__, e)}


def host_cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def reference_js_figure(name):
    """The reference ITSELF (unmodified llama2.ts under Node, one JS thread) cannot run on the GPU box -- its source does not travel.
    oracle/make_goldens.py --speed timed it in the build container on this same synthetic checkpoint and stored the tok/s it prints
    (llama2.ts:511) in tests/golden/reference_speed.json; quoted here next to the C port's figure, with where it was measured."""
    try:
        ref = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_speed.json"))).get(name)
    except (OSError, ValueError):
        ref = None
    if not ref:
        return {"reference_js_tok_s": None}
    return {"reference_js_tok_s": round(ref["tok_s"], 4),
            "reference_js_measured": "build container (not this box): %s, node %s, %d thread, %s, %d steps" % (ref["cpu"], ref["node"], ref["threads"], ref["argv"], ref["steps"])}


# ---- one config on one GPU: decode loop + roofline + CPU baseline ------------------------------------------------
def dispatch_note(ctx):
    """How the device-resident loop's launches reached the chip in the run just made (L2_OPT_AQL_QUEUE, include/llama2_hip.h)."""
    try:
        if ctx.get_option(runtime.OPT_AQL_QUEUE):
            return ("a token's launches written as AQL packets on the library's own HSA queue: barrier bit, agent-scope release, "
                    "no acquire fence between the launches of a token (csrc/aql_queue.h)")
        why = runtime.lib().l2_last_error().decode("utf8", "replace")
        if not ctx.get_option(runtime.OPT_USE_GRAPH):
            return "eager launches (L2_USE_GRAPH=0: the step is not recorded, so neither the library's AQL queue nor a hipGraph replays it)"
        return "one hipGraph replay per token (%s)" % (why if "AQL" in why else "the library's AQL queue was not taken: L2_AQL=0, or a step with collectives of the runtime's")
    except Exception as e:      # an older library
        return "one hipGraph replay per token (%s)" % type(e).__name__


def roofline_block(ctx, cfg, K, traffic, traffic_how, trace_us=None):
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


def secondary_config(name, seed, device, with_cpu, traffic, trace_us=None):
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
           "loop": "device-resident (forward + argmax on GPU, %s)" % dispatch_note(ctx),
           "parity": parity}
    # the same K steps through the blocking drop-in boundary (llama2.ts:468 -> 478: logits to the host every token, argmax there)
    rate, dropin_tokens = dropin_loop(ctx, K)
    out["dropin_tok_s"] = round(rate, 2)
    out["parity"]["dropin_equal_to_reference_golden"] = parity_block(name, seed, dropin_tokens)["equal_to_reference_golden"]
    out["napi_dropin_tok_s"] = napi_dropin(ctx, name, seed, K)               # ... and through the N-API addon under Node (the contract's binding)
    out["dropin_tok_s_direct_dispatch_off"] = dropin_direct_dispatch_off(name, seed)
    out["roofline"] = roofline_block(ctx, cfg, K, traffic[0], traffic[1], trace_us)
    out["per_kernel"] = per_kernel_block(ctx, cfg)      # back-to-back launches of each GEMV phase: us and GB/s of its matrix bytes
    S = hdr[6]
    ms = ctx.bench_decode(1, 0, S)
    out["whole_context_tok_s"] = round(S / (ms * 1e-3), 2)
    for key, n_p in (("prefill_tok_s", min(128, S)), ("prefill_256_tok_s", min(256, S))):     # prompt ingestion, as in the main block
        ptoks = (np.arange(n_p, dtype=np.int32) * 7919 + 2) % cfg.vocab_size
        ctx.prefill(ptoks, 0)
        t0 = time.perf_counter()
        ctx.prefill(ptoks, 0)
        out[key] = round(n_p / (time.perf_counter() - t0), 1)
    ctx.close()
    if with_cpu:
        out["cpu_baseline"] = cpu_baseline(name, hdr, seed)
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


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: THIS process has not touched the GPU (nothing above imports torch or
    calls HIP) and starts `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a CHILD process,
    one rank per GPU; it relays the child's one JSON line and its exit code.  (Never an exec: the parent stays a plain
    supervisor.)"""
    import socket
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = clean_child_env(HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    line = None
    for ln in r.stdout.decode("utf8", "replace").splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        elif ln.strip():
            print(ln, file=sys.stderr)      # anything else a rank printed is not the result line
    if line is not None:
        print(line)
    elif r.returncode == 0:
        print("bench.py: %d ranks finished without a result line" % n, file=sys.stderr)
        return 1
    return r.returncode


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
    args = ap.parse_args()
    # the host driver of this pool only supports dmabuf IPC: without this, RCCL and hipIpcGetMemHandle fail in ranks a launcher
    # other than spawn_ranks() started (set before the first HIP call of the process)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.pmc_child:
        pmc_child(args.config, args.seed)
        return
    if args.trace_child:
        trace_child(args.config, args.seed)
        return
    if args.dropin_child:
        dropin_child(args.config, args.seed)
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    if under_profiler():
        # rocprofv3 (ROCm 7.2) segfaults once a process has replayed a hipGraph more than ~128 times with kernel tracing on
        # (profiles/README.md): under a profiler the library launches the same kernels eagerly -- per-kernel durations carry over,
        # the tokens/s of such a run is host-bound and says nothing
        os.environ.setdefault("L2_USE_GRAPH", "0")

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1) and rank == 0:
        print("bench.py: --gpus %d but the launcher started %d ranks; reporting the %d that run" % (args.gpus, world, world), file=sys.stderr)
    hdr = configs.header(args.config)
    K = min(args.steps, hdr[6])
    W = min(args.warmup, hdr[6])
    extras = rank == 0 and world == 1

    # counter passes first: they are child processes, and nothing in THIS process has touched the GPU yet
    traffic = {args.config: (None, "skipped (--no-pmc)")}
    trace_us = {}
    if extras and not args.no_pmc:
        traffic[args.config] = pmc_traffic(args.config, args.seed)
        trace_us[args.config] = kernel_trace_us(args.config, args.seed)
        if not args.no_extra and args.config == "llama2_7b":
            traffic["stories110M"] = pmc_traffic("stories110M", args.seed)
            trace_us["stories110M"] = kernel_trace_us("stories110M", args.seed)

    dist = None
    tp = None
    shards = world > 1 and args.config.startswith("llama2_7b")
    if world > 1:
        # (the second-chance rendezvous below -- ranks meeting through files -- has a switch of its own in the library,
        # L2_TP_FILE_RENDEZVOUS, set only around that attempt: the development gate L2_TEST_HOOKS stays closed in a measured run)
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo")   # rendezvous + barriers only; the data path is inside the library
        if shards:
            tp = {"rank": rank, "size": world}

    device = int(os.environ.get("L2_BENCH_FORCE_DEVICE", local_rank))   # test hook: several ranks on one GPU (replicas only)
    cfg = runtime.Config(hdr)
    tp_note = None
    ctx = None
    tp_proof = None
    if tp:
        # The tensor-parallel group has never run on more than one GPU (no multi-GPU box in development), so every way of forming
        # it is PROVED before it is timed: the group is created, filled, and decodes a few tokens; every rank reports over gloo
        # whether that worked and what it decoded (all ranks must agree, and agree with the real reference's golden tokens where
        # a fixture exists).  Order: (1) RCCL communicator + the one-shot peer-to-peer exchange if its self-test passes,
        # (2) RCCL collectives only, (3) no RCCL: the ranks meet through files and exchange over IPC-mapped inboxes,
        # (4) independent replicas.  The line says which one ran and why.
        import tempfile
        import torch

        def golden_tokens(n):
            try:
                g = json.load(open(os.path.join(ROOT, "tests", "golden", args.config + ".json")))
                return g["argmax"][:n] if g.get("seed") == args.seed and len(g["argmax"]) >= n else None
            except (OSError, ValueError, KeyError):
                return None

        attempts = [0]
        ipc_base = os.environ.get("L2_TP_IPC_DIR")      # set from outside (tests): every attempt then meets in a directory of its own

        def attempt(env, fresh_id=True):
            """Two phases, each ended by a report of every rank over gloo: (1) create the context (communicator, peer mappings,
            start-up self-test) -- if ANY rank failed, every rank closes and nobody enters a collective; (2) fill and decode sixteen
            tokens.  (RCCL has no timeout: a rank that entered an all-reduce its peer never reaches would hang the job instead of
            printing the fallback line.)"""
            err, c, toks = "", None, []
            attempts[0] += 1
            env = dict(env)
            if ipc_base and os.path.isdir(ipc_base) and "L2_TP_IPC_DIR" not in env:
                sub = os.path.join(ipc_base, "attempt%d" % attempts[0])     # files of an earlier, failed formation must not be read again
                os.makedirs(sub, exist_ok=True)
                env["L2_TP_IPC_DIR"] = sub
            saved = {k: os.environ.get(k) for k in env}      # whatever the caller had set is put back afterwards
            for k, v in env.items():
                os.environ[k] = v
            try:
                nid = b"\x01" * 128     # the file rendezvous ignores it
                if fresh_id:             # a communicator id is good for one ncclCommInitRank round
                    idb = torch.zeros(128, dtype=torch.uint8)
                    if rank == 0:
                        import ctypes as C
                        b = C.create_string_buffer(128)
                        if runtime.lib().l2_tp_unique_id(b) == 0:
                            idb = torch.frombuffer(bytearray(b.raw), dtype=torch.uint8).clone()
                    dist.broadcast(idb, 0)
                    nid = bytes(idb.numpy().tobytes())
                # ---- phase 1: create
                try:
                    if fresh_id and not any(nid):
                        raise RuntimeError("rank 0 could not create an RCCL id (%s)" % runtime.lib().l2_last_error().decode("utf8", "replace"))
                    c = runtime.Context(hdr, device=device, tp_rank=tp["rank"], tp_size=tp["size"], nccl_id=nid)
                except Exception as e:      # noqa: BLE001 -- whatever it is, the other ranks have to hear about it
                    err = "%s: %s" % (type(e).__name__, e)
                created = [None] * world
                dist.all_gather_object(created, {"rank": rank, "err": err})
                errs = [r["err"] for r in created if r["err"]]
                if errs:
                    if c is not None:
                        c.close()
                    return None, errs[0], None
                # ---- phase 2: every rank has a context: fill, decode, compare
                try:
                    c.synth_fill(args.seed)
                    toks = c.decode_greedy(1, 0, min(16, K)).tolist()      # (16 tokens: 1 040 exchanges at 32 layers, every one of them part of the proof)
                except Exception as e:      # noqa: BLE001
                    err = "%s: %s" % (type(e).__name__, e)
                mine = {"rank": rank, "err": err, "tokens": toks, "mode": c.tp_mode_id() if not err else -1}
                every = [None] * world
                dist.all_gather_object(every, mine)
                errs = [r["err"] for r in every if r["err"]]
                same = all(r["tokens"] == every[0]["tokens"] for r in every)
                gold = golden_tokens(len(every[0]["tokens"]))
                ok = not errs and same and (gold is None or every[0]["tokens"] == gold)
                why = "" if ok else (errs[0] if errs else ("ranks decoded different tokens: %s" % [r["tokens"] for r in every] if not same
                                                         else "tokens %s differ from the reference golden %s" % (every[0]["tokens"], gold)))
                if not ok:
                    c.close()
                    c = None
                proof = {"tokens": every[0]["tokens"], "same_on_every_rank": same, "equals_reference_golden": (None if gold is None else every[0]["tokens"] == gold)}
                return c, why, proof
            finally:
                for k, v in saved.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v

        notes = []
        ctx, why, tp_proof = attempt({})
        if ctx is None:
            notes.append("RCCL + peer-to-peer exchange: %s" % why)
            ctx, why, tp_proof = attempt({"L2_TP_ALLREDUCE": "rccl"})
        if ctx is None and "L2_TP_IPC_DIR" not in os.environ:
            notes.append("RCCL collectives only: %s" % why)
            meet = [tempfile.mkdtemp(prefix="l2_meet_") if rank == 0 else None]
            dist.broadcast_object_list(meet, 0)
            ctx, why, tp_proof = attempt({"L2_TP_IPC_DIR": meet[0], "L2_TP_FILE_RENDEZVOUS": "1"}, fresh_id=False)
            if ctx is not None:
                notes.append("the ranks met through files and exchange peer to peer (no RCCL)")
        if ctx is None:
            notes.append("file rendezvous + peer-to-peer exchange: %s" % why)
            tp, shards, tp_proof = None, False, None
            notes.append("measured %d independent replicas instead" % world)
        if notes:
            tp_note = "; ".join(notes)
    if ctx is None:
        ctx = runtime.Context(hdr, device=device)
        ctx.synth_fill(args.seed)

    def sync_all():
        if dist is not None:
            dist.barrier()
            import torch
            torch.cuda.synchronize(device)

    # untimed: small models finish their K steps in tens of milliseconds, less than the clock ramp of an idle GPU, so
    # they first decode for ~0.3 s; then the W warm-up steps, the last of them at the deepest position of the timed run
    # (it captures the hipGraph of the split-attention level, which would otherwise be captured inside the timed region)
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
    sync_all()
    t0 = time.perf_counter()
    dev_ms = ctx.bench_decode(1, 0, K)         # EXACTLY K timed steps, HIP events on the library's stream
    sync_all()
    wall = time.perf_counter() - t0
    timed_tokens = ctx.bench_tokens(K)         # what the timed run decoded (checked against the reference's golden below)
    if dist is not None:
        import torch
        t = torch.tensor([wall, dev_ms], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, dev_ms = float(t[0]), float(t[1])

    tokens_total = K if (shards or world == 1) else K * world
    value = tokens_total / wall
    bpt = avg_bytes_per_token(hdr, 0, K)
    per_gpu_streams = 1 if (shards or world == 1) else world

    out = {
        "metric": "decode tokens/sec + achieved HBM GB/s (% peak), 1 GPU" if world == 1 else "decode tokens/sec (whole job)",
        "value": round(value, 3), "unit": "tokens/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": round(1e3 * wall / K, 5), "higher_is_better": True,
        "scaling": "strong" if shards else "weak", "vs_baseline": None, "dtype": "f64", "storage_dtype": "f32",
        "data": "synthetic (seeded hash generator, llama2.c-v0 layout)",
        "config": {"workload": "%s batch-1 greedy decode, %d tokens from BOS (-t 0 -s 1 -n %d)" % (args.config, K, K),
                   "header": list(hdr), "parallelism": ("tp%d" % world) if shards else ("replicas%d" % world if world > 1 else "single"),
                   "loop": ("device-resident (forward + argmax on GPU); tensor-parallel step: %s; dispatch: %s" % (ctx.tp_mode(), dispatch_note(ctx))
                            if shards else "device-resident (forward + argmax on GPU, %s)" % dispatch_note(ctx)),
                   "weights": "fp32, ONE copy on the device (%d MiB): the matrices of the streaming phases repacked in the order the chip consumes them "
                              "(%d MiB, DESIGN.md section 3), everything else row-major as the checkpoint stores it"
                              % (ctx.get_option(runtime.OPT_WEIGHT_MIB), ctx.get_option(runtime.OPT_PACKED_MIB)),
                   "weights_mib": {"on_device": ctx.get_option(runtime.OPT_WEIGHT_MIB), "repacked": ctx.get_option(runtime.OPT_PACKED_MIB),
                                   "checkpoint": configs.checkpoint_bytes(hdr) >> 20}},
        "device_ms_per_step": round(dev_ms / K, 5),
        "algorithmic_bytes_per_token": int(bpt),
        "hbm_gbs_end_to_end": round(bpt * value / 1e9 / per_gpu_streams, 2),
        "hbm_frac_end_to_end": round(bpt * value / 1e9 / per_gpu_streams / HBM_PEAK_GBS / (world if shards else 1), 4),
    }

    # the tokens of the TIMED run against the real reference's tokens for this checkpoint (fixtures are data: they travel)
    out["parity"] = parity_block(args.config, args.seed, timed_tokens)
    if dist is not None:
        # what actually ran, rank by rank, so a reader of the line can see the group had N members: l2_tp_mode
        # (0 single GPU, 1 RCCL eager, 3 peer-to-peer exchange in one hipGraph per token) and the device of every rank
        mine = {"rank": rank, "device": device, "tp_mode": ctx.tp_mode_id()}
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
        out["tp"] = {"ranks": world, "sharded": bool(shards), "l2_tp_mode": sorted({r["tp_mode"] for r in ranks}),
                     "devices": [r["device"] for r in ranks], "step": ctx.tp_mode(), "proved_before_timing": tp_proof}
    if shards:
        out["tp_predicted"] = committed_prediction(args.config, world)
    if under_profiler() and os.environ.get("L2_USE_GRAPH") == "0":
        out["profiled"] = "this run was started under a profiler: eager launches instead of one hipGraph replay per token (host-bound; read the kernel durations, not `value`)"
    if tp_note:
        out["note"] = tp_note
    if extras:
        if not args.no_dropin:   # the same K steps through the blocking drop-in boundary (logits to the host every token)
            rate, dropin_tokens = dropin_loop(ctx, K)
            out["dropin_tok_s"] = round(rate, 3)
            out["parity"]["dropin_equal_to_reference_golden"] = parity_block(args.config, args.seed, dropin_tokens)["equal_to_reference_golden"]
            out["napi_dropin_tok_s"] = napi_dropin(ctx, args.config, args.seed, K)      # the N-API addon under Node (small models: the 7B file is not written)
            if configs.checkpoint_bytes(hdr) < (2 << 30):
                out["dropin_tok_s_direct_dispatch_off"] = dropin_direct_dispatch_off(args.config, args.seed)
        out["roofline"] = roofline_block(ctx, cfg, K, *traffic[args.config], trace_us.get(args.config))
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
            for key, n_p in (("prefill_tok_s", min(128, S)), ("prefill_256_tok_s", min(256, S))):
                ptoks = (np.arange(n_p, dtype=np.int32) * 7919 + 2) % cfg.vocab_size
                ctx.prefill(ptoks, 0)
                t0 = time.perf_counter()
                ctx.prefill(ptoks, 0)
                out[key] = round(n_p / (time.perf_counter() - t0), 1)
    ctx.close()

    if extras and not args.no_extra and args.config.startswith("llama2_7b"):
        out["tp_predicted"] = tp_prediction(hdr, args.seed, device, 1e3 * wall / K)
    if extras and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.config, hdr, args.seed)
    if extras and not args.no_extra and args.config == "llama2_7b":
        out["stories110M"] = secondary_config("stories110M", args.seed, local_rank, not args.no_cpu_baseline,
                                              traffic.get("stories110M", (None, "skipped")), trace_us.get("stories110M"))

    # a timed run that decoded other tokens than the reference is not a measurement: the line is still printed (it says where
    # the first mismatch is), the exit code says no
    bad = [nm for nm, blk in (("main", out), ("stories110M", out.get("stories110M") or {}))
           if (blk.get("parity") or {}).get("equal_to_reference_golden") is False
           or (blk.get("parity") or {}).get("dropin_equal_to_reference_golden") is False]
    if rank == 0:
        print(json.dumps(out))
        if bad:
            print("bench.py: PARITY FAILURE in %s: the timed decode does not reproduce the reference's golden tokens" % ", ".join(bad), file=sys.stderr)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if bad:
        sys.exit(3)


if __name__ == "__main__":
    main()

"""cpu_baseline of bench.py: the C oracle (one thread, like the single-threaded reference) and its JavaScript restatement under this box's Node, next to the figure the reference itself printed in the build container.  The oracle is imported HERE and nowhere else in the benchmark: it is the thing timed in this leg, never part of the GPU path."""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from llama2_ts_amd import configs  # noqa: E402

from .common import golden_argmax, host_cpu_model  # noqa: E402


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



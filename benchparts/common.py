"""Shared pieces of bench.py: the peak it prices against, child-process environments, the parity check of a timed run."""
import json
import os
import sys


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from llama2_ts_amd import configs  # noqa: E402

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
DOMINANT = "rmsnorm + w1/w3 GEMV + SwiGLU (llama2.ts:276-289)"


def avg_bytes_per_token(hdr, p0, p1):
    return sum(configs.algorithmic_bytes_per_token(hdr, p) for p in range(p0, p1)) / float(p1 - p0)


def dominant_kernel_bytes(cfg):
    """Algorithmic bytes of one launch of the dominant kernel: the fused rmsnorm + w1/w3 GEMV + SwiGLU
    phase (llama2.ts:276-289): both matrices once, x and the norm weight in, hb out."""
    d, h = cfg.dim, cfg.hidden_dim
    return 4 * (2 * h * d + 2 * d + h)


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


def host_cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"



"""Build both, choose by evidence (hard part 3) for prompt ingestion (llama2.ts:471-473): the register-blocked GEMMs on
v_mfma_f64_16x16x4_f64 (the reference's arithmetic: fp64 accumulate, one rounding per stored element -- the default) against the same
blocking on v_mfma_f32_16x16x4_f32 (L2_OPT_PREFILL_F32_MFMA: a k-ordered fp32 fmaf chain per element, no widening conversions, twice
the instruction rate).  Per form: prompt tok/s at Llama-2-7B (128 / 256 tokens) and stories110M, max |dlogit| against the REAL
reference's logits at the kept positions of the goldens, and the exactness of the greedy continuation behind every prompt length.

  python tools/prefill_f32_eval.py            (on the GPU box)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from llama2_ts_amd import configs, runtime  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def rate(name, n, f32):
    hdr = configs.header(name)
    ctx = runtime.Context(hdr); ctx.synth_fill(1)
    ctx.set_option(runtime.OPT_PREFILL_F32_MFMA, int(f32))
    toks = (np.arange(n, dtype=np.int32) * 7919 + 2) % abs(hdr[5])
    ctx.prefill(toks, 0)
    best = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        ctx.prefill(toks, 0)
        best = max(best, n / (time.perf_counter() - t0))
    ctx.close()
    return best


def exactness(name, f32):
    """For every kept position p of the golden (with p + 1 tokens of prompt): |dlogit| of the prefill's last logits against the reference's,
    then how many of the next `m` greedy tokens equal the reference's (decode is the same fp64 path in both forms: what differs is the
    KV cache and the logits the prefill left)."""
    meta = json.load(open(os.path.join(GOLD, name + ".json")))
    g = np.load(os.path.join(GOLD, name + ".npz"))
    keep = {p: i for i, p in enumerate(meta["logit_positions"])}
    ctx = runtime.Context(meta["header"]); ctx.synth_fill(meta["seed"])
    ctx.set_option(runtime.OPT_PREFILL_F32_MFMA, int(f32))
    rows = []
    S = meta["header"][6]
    for p in sorted(keep):
        n = p + 1
        if n < 33:
            continue      # (chunks of at most 32 tokens take the 16-row-tile kernels: fp64 in both forms)
        lg = np.array(ctx.prefill(meta["tokens_fed"][:n], 0), copy=True)
        err = float(np.abs(lg - g["logits"][keep[p]]).max())
        m = min(64, S - n)
        same = 0
        if m > 0:
            cont = ctx.decode_greedy(runtime.argmax(lg), n, m).tolist() if n < S else []
            want = meta["argmax"][n:n + m]
            same = next((i for i, (a, b) in enumerate(zip(cont, want)) if a != b), m)
        rows.append({"prompt_tokens": n, "max_dlogit": err, "argmax_equal": runtime.argmax(lg) == meta["argmax"][p], "continuation_checked": m, "continuation_equal_until": same})
    ctx.close()
    return rows


def main():
    out = {"what": __doc__.split("\n\n")[0], "rates_tok_s": {}, "exactness": {}}
    for name, ns in (("llama2_7b", (128, 256)), ("stories110M", (128, 256))):
        for n in ns:
            r64, r32 = rate(name, n, False), rate(name, n, True)
            out["rates_tok_s"]["%s_%d" % (name, n)] = {"fp64_mfma": round(r64, 1), "fp32_mfma": round(r32, 1), "speedup": round(r32 / r64, 3)}
            print("%-12s %4d prompt tokens: fp64 MFMA %8.1f tok/s   fp32 MFMA %8.1f tok/s   x%.2f" % (name, n, r64, r32, r32 / r64), flush=True)
    for name in ("stories110M", "llama2_7b_L2", "llama2_7b"):
        for f32 in (False, True):
            rows = exactness(name, f32)
            out["exactness"]["%s_%s" % (name, "fp32" if f32 else "fp64")] = rows
            worst = max(r["max_dlogit"] for r in rows)
            cont = sum(r["continuation_equal_until"] for r in rows), sum(r["continuation_checked"] for r in rows)
            print("%-13s %s: %d prompt lengths %s, max |dlogit| vs the reference %.3g, argmax equal %d / %d, greedy continuation %d of %d tokens before the first difference"
                  % (name, "fp32" if f32 else "fp64", len(rows), [r["prompt_tokens"] for r in rows], worst, sum(r["argmax_equal"] for r in rows), len(rows), cont[0], cont[1]), flush=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06", "prefill_f32_eval.json"), "w"), indent=1)


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "gpurun_out", "r06"), exist_ok=True)
    main()

#!/usr/bin/env python3
"""Golden-vector recipe: run the TRUE reference (/root/reference/llama2.ts) and record its outputs.

TEST INFRASTRUCTURE.  Runs only in the build container (needs /root/reference and `node`); the
GPU box never runs it -- it consumes the fixtures this script writes to tests/golden/.

What it does:
  1. in a scratch dir under /tmp, lift the self-contained sucrase bundle out of the reference's
     own t348.mjs (lines 898-9042) and use it to strip the TypeScript types from llama2.ts
     (line-preserving), so Node 12 can execute the reference unmodified;
  2. insert ONE statement after the `transformer(...)` call at llama2.ts:468 that appends
     `state.logits`, the fed `token` and (for small shapes) the RunState scratch buffers to files;
  3. generate synthetic checkpoints with the repo's deterministic generator (oracle_cli synth);
  4. run `node llama2.stripped.mjs <ckpt> -t 0 -s 1 -n <steps>` and reduce the dumps to small
     fixtures: fed tokens, per-step sha256 of the raw logits, full logits at a few positions,
     RunState buffers and KV cache for the tiny shapes.

Nothing derived from the reference's source text is written into the repo -- only numbers.
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from llama2_ts_amd import configs  # noqa: E402

REF = "/root/reference"
WORK = "/tmp/l2_goldens"
GOLD = os.path.join(ROOT, "tests", "golden")
CLI = os.path.join(ROOT, "oracle", "build", "oracle_cli")

# name -> (steps, positions whose full logits are kept, keep RunState dumps?, prompt[, extra argv, synthetic tokenizer?])
PLAN = {
    "tiny": (64, "all", True, None),
    "ragged": (33, "all", True, None),
    "tinylong": (1280, [0, 255, 256, 257, 1023, 1024, 1279], False, None),
    "stories15M": (256, [0, 1, 2, 127, 255], False, None),
    "stories15M_prompt": (24, [3, 4, 23], False, "Once upon a time"),
    # the whole context window: every attention split threshold of the HIP path at head_size 64 ...
    "stories110M": (1024, [0, 39, 127, 128, 129, 255, 256, 257, 299, 511, 512, 513, 1023], False, None),
    # ... and at head_size 128 (7B width, 2 layers; ~1 tok/s in the reference: about an hour)
    "llama2_7b_L2": (2048, [0, 5, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 559, 1023, 1024, 1919, 1920, 1984, 2047], False, None),
    # the real thing: all 32 layers -- the 256 steps bench.py times by default and on through every attention split level of the full model
    # (round 6: 1024 steps; 27 GB synthetic checkpoint in /tmp, ~10 s per token in the reference: about three hours)
    "llama2_7b": (1024, [0, 2, 19, 63, 127, 143, 144, 145, 255, 256, 257, 511, 512, 513, 1023], False, None),
    # whole-CLI goldens (stdout text): the reference runs with the repo's SYNTHETIC tokenizer.bin in its cwd
    "cli_greedy": (48, [], False, None, ["-t", "0", "-s", "1"], True),
    "cli_prompt": (40, [], False, "wetds oyn fra uynia", ["-t", "0", "-s", "1"], True),
    "cli_temp": (40, [], False, None, ["-t", "0.9", "-s", "42"], True),
    "cli_topp": (24, [], False, "once", ["-t", "1.0", "-p", "0.9", "-s", "7"], True),
    # the multi-head expansion of a grouped-query model is a checkpoint the reference CAN run (it reads wk / wv as
    # (d, d), llama2.ts:117-118); its outputs pin the L2_F_GQA path.  "_rope": freq_cis_* hold the tables llama2.c's run.c would
    # compute (what L2_F_GENERATE_ROPE generates, llama2.ts:125-126 reads them from the file) -- pins version-1 checkpoints.
    "wide_gqa_mha": (320, [0, 1, 2, 23, 143, 144, 145, 159, 160, 319], False, None),
    "tiny_gqa_rope_mha": (64, "all", False, None),
    # the reference's argmax (llama2.ts:364-366) on logits that hit its edges: models with patched classifier rows (tests/argmax_cases.py:
    # exact ties across the device's workgroups, -0 beside +0, +-inf, NaN, nothing but NaN, NaN at index 0) -- what is pinned is the
    # token the reference FED next (its own pick), not numpy's argmax of the dumped logits
    **{"argmax_%s_%s" % (c, sh): (16, [0, 3, 15], False, None) for sh in ("vec", "odd") for c in ("ties", "specials", "nan0", "allnan", "zeros")},
}

DUMP_STMT = (
    "if (process.env.L2_DUMP) {"
    " const __w = (n, a) => fs.appendFileSync(process.env.L2_DUMP + n, Buffer.from(a.buffer, a.byteOffset, a.byteLength));"
    " __w('.logits', state.logits); __w('.tokens', new Int32Array([token]));"
    " if (process.env.L2_DUMP_STATE) { for (const n of ['x','xb','xb2','hb','hb2','q','k','v','att']) __w('.' + n, state[n]);"
    " if (pos == steps - 1) { __w('.key_cache', state.key_cache); __w('.value_cache', state.value_cache); } } }\n"
)


def strip_types(src, out):
    """Type-strip a TypeScript file with the sucrase bundle the reference ships inside its own t348.mjs (lines 898-9042):
    line-preserving, so llama2.ts:468 is still line 468 of the output, and Node 12 can run it."""
    os.makedirs(WORK, exist_ok=True)
    with open(os.path.join(REF, "t348.mjs")) as f:
        lines = f.readlines()
    assert lines[897].startswith("const transform=(()=>{"), lines[897][:40]
    assert lines[9041].startswith("})();"), lines[9041][:40]
    strip = "".join(lines[897:9042]) + """
const fs=require('fs');
fs.writeFileSync(process.argv[3],
  transform(fs.readFileSync(process.argv[2],'utf8'),
            {transforms:["typescript"],filePath:'llama2.ts',disableESTransforms:true}).code);
"""
    with open(os.path.join(WORK, "strip.cjs"), "w") as f:
        f.write(strip)
    subprocess.run(["node", os.path.join(WORK, "strip.cjs"), src, out], check=True)


def build_reference():
    os.makedirs(WORK, exist_ok=True)
    for f in ("tokenizer.bin",):
        shutil.copy(os.path.join(REF, f), WORK)
    out = os.path.join(WORK, "llama2.stripped.mjs")
    strip_types(os.path.join(REF, "llama2.ts"), out)
    with open(out) as f:
        js = f.readlines()
    assert "transformer(token, pos, config, state, weights);" in js[467], js[467]
    js.insert(468, DUMP_STMT)
    inst = os.path.join(WORK, "llama2.instrumented.mjs")
    with open(inst, "w") as f:
        f.writelines(js)
    return inst


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_one(inst, name):
    plan = PLAN[name]
    steps, keep, keep_state, prompt = plan[:4]
    extra = plan[4] if len(plan) > 4 else ["-t", "0", "-s", "1"]
    synth_tok = plan[5] if len(plan) > 5 else False
    shape = "stories15M" if name.startswith("cli_") else name.replace("_prompt", "")
    gqa_hdr = None
    edge = None
    if name.startswith("argmax_"):     # classifier rows patched so that the logits hit the edges of the reference's argmax
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import argmax_cases
        _, edge, shape = name.split("_")
        hdr, seed = argmax_cases.SHAPES[shape], argmax_cases.SEED
        ckpt = os.path.join(WORK, name + ".bin")
        argmax_cases.write_v0(ckpt, edge, shape)
    elif name.endswith("_mha"):      # grouped-query model, expanded to the multi-head checkpoint the reference can read
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import gqa_cases
        shape = name[:-4]
        base = shape.replace("_rope", "")
        gqa_hdr = gqa_cases.GQA_SHAPES[base]
        seed = gqa_cases.GQA_SEEDS[base]
        d, h, L, H, KVH, V, S = gqa_hdr
        hdr = (d, h, L, H, H, V, S)
        ckpt = os.path.join(WORK, name + ".bin")
        gqa_cases.expanded_mha_file(gqa_hdr, seed, ckpt, runc_rope="_rope" in shape)
    else:
        hdr = configs.header(shape)
        seed = configs.DEFAULT_SEED
        ckpt = os.path.join(WORK, shape + ".bin")
        if not os.path.exists(ckpt):
            subprocess.run([CLI, "synth", *map(str, hdr), str(seed), ckpt], check=True)
    assert os.path.getsize(ckpt) == configs.checkpoint_bytes(hdr)
    dump = os.path.join(WORK, name + ".dump")
    for f in os.listdir(WORK):
        if f.startswith(name + ".dump"):
            os.remove(os.path.join(WORK, f))
    env = dict(os.environ, L2_DUMP=dump)
    if keep_state:
        env["L2_DUMP_STATE"] = "1"
    cwd = WORK
    if synth_tok:   # llama2.ts:444 reads "tokenizer.bin" from the working directory
        cwd = os.path.join(WORK, "synthtok")
        os.makedirs(cwd, exist_ok=True)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import synth_tokenizer
        synth_tokenizer.write(os.path.join(cwd, "tokenizer.bin"))
    cmd = ["node", inst, ckpt, *extra, "-n", str(steps)]
    if prompt is not None:
        cmd += ["-i", prompt]
    import time
    t_run = time.perf_counter()
    r = subprocess.run(cmd, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
    t_run = time.perf_counter() - t_run
    V = abs(hdr[5])
    logits = np.fromfile(dump + ".logits", dtype="<f4").reshape(-1, V)
    tokens = np.fromfile(dump + ".tokens", dtype="<i4")
    n = logits.shape[0]
    assert n == len(tokens)
    meta = {
        "config": shape, "header": list(hdr), "seed": seed, "steps_requested": steps, "steps_run": int(n),
        "prompt": prompt, "argv": cmd[3:], "node": subprocess.run(["node", "--version"], stdout=subprocess.PIPE).stdout.decode().strip(),
        "reference": "wizzard0/llama2.ts @ /root/reference (llama2.ts, types stripped by its bundled sucrase 3.21.0)",
        "checkpoint_sha256": hashlib.sha256(open(ckpt, "rb").read()).hexdigest() if os.path.getsize(ckpt) < (1 << 30) else None,
        "logits_sha256": [sha(logits[i]) for i in range(n)],
        "tokens_fed": tokens.tolist(),
        "argmax": [int(np.argmax(logits[i])) for i in range(n)],
        "stdout_tail": r.stdout.decode("utf8", "replace")[-80:],
        "tokenizer": "synthetic (tests/synth_tokenizer.py)" if synth_tok else "reference tokenizer.bin",
    }
    if edge is not None:
        # the reference's OWN picks: the token it fed at step i + 1 is what its argmax returned at step i (llama2.ts:478, 504)
        meta["picks"] = tokens.tolist()[1:]
        meta["case"] = edge
        meta["what"] = "tests/argmax_cases.py model (%s, %s): `picks` are the reference's own argmax results; `argmax` here is numpy's, which treats NaN differently" % (edge, shape)
        with np.errstate(invalid="ignore"):
            meta["logit_census"] = [{"nan": int(np.isnan(logits[i]).sum()), "inf": int(np.isinf(logits[i]).sum()),
                                     "neg_zero": int((np.signbit(logits[i]) & (logits[i] == 0)).sum())} for i in range(n)]
    if gqa_hdr is not None:
        meta["gqa_header"] = list(gqa_hdr)
        meta["what"] = ("the reference ran the MULTI-HEAD expansion of the grouped-query model (gqa_header, seed): wk / wv rows of each cache "
                        "head repeated for its query heads" + ("; freq_cis_* = llama2.c run.c's per-position tables" if "_rope" in shape else ""))
    # the reference's own throughput figure (llama2.ts:507-511 prints "achieved tok/s" to stderr), with where it was measured
    tail = r.stderr.decode("utf8", "replace")
    import re
    m = re.search(r"achieved tok/s:\s*([0-9.eE+-]+)", tail + r.stdout.decode("utf8", "replace"))
    meta["reference_run"] = {"tok_s_printed": float(m.group(1)) if m else None, "wall_s": round(t_run, 3), "steps": int(n),
                             "threads": 1, "cpu": cpu_model(), "node": meta["node"], "includes_dump_overhead": True}
    if synth_tok:
        meta["stdout"] = r.stdout.decode("utf8")
    arrays = {}
    if keep == "all":
        arrays["logits"] = logits
        meta["logit_positions"] = list(range(n))
    else:
        keep = [p for p in keep if p < n]
        arrays["logits"] = logits[keep]
        meta["logit_positions"] = keep
    if keep_state:
        d, h, L, H, _, _, S = hdr
        for nm, width in (("x", d), ("xb", d), ("xb2", d), ("hb", h), ("hb2", h), ("q", d), ("k", d), ("v", d), ("att", H * S)):
            arrays[nm] = np.fromfile(dump + "." + nm, dtype="<f4").reshape(n, width)
        arrays["key_cache"] = np.fromfile(dump + ".key_cache", dtype="<f4")
        arrays["value_cache"] = np.fromfile(dump + ".value_cache", dtype="<f4")
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **arrays)
    with open(os.path.join(GOLD, name + ".json"), "w") as f:
        json.dump(meta, f, indent=1)
    print(name, "steps", n, "tokens", tokens[:8].tolist(), "...", "npz KB",
          os.path.getsize(os.path.join(GOLD, name + ".npz")) // 1024)


# config -> steps for the clean timing runs of the reference (no dump statement): the figure the reference prints itself
SPEED_PLAN = {"stories15M": 256, "stories110M": 48, "llama2_7b_L2": 8, "llama2_7b": 4}


def time_reference(names):
    """the reference's OWN CPU figure -- the unmodified (type-stripped) llama2.ts timed in the build container,
    one JS thread, the tok/s it prints itself (llama2.ts:507-511: timer starts after the first token).  Stored in
    tests/golden/reference_speed.json; bench.py quotes it next to the C port's number as cpu_baseline.reference_js_tok_s."""
    import re
    import time
    build_reference()
    plain = os.path.join(WORK, "llama2.stripped.mjs")
    path = os.path.join(GOLD, "reference_speed.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    node = subprocess.run(["node", "--version"], stdout=subprocess.PIPE).stdout.decode().strip()
    for name in names:
        hdr = configs.header(name)
        ckpt = os.path.join(WORK, name + ".bin")
        if not os.path.exists(ckpt):
            subprocess.run([CLI, "synth", *map(str, hdr), str(configs.DEFAULT_SEED), ckpt], check=True)
        steps = SPEED_PLAN[name]
        t0 = time.perf_counter()
        r = subprocess.run(["node", plain, ckpt, "-t", "0", "-s", "1", "-n", str(steps)], cwd=WORK, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
        wall = time.perf_counter() - t0
        m = re.search(r"achieved tok/s:\s*([0-9.eE+-]+)", r.stdout.decode("utf8", "replace"))
        out[name] = {"tok_s": float(m.group(1)), "steps": steps, "wall_s_incl_load": round(wall, 2), "threads": 1, "cpu": cpu_model(), "node": node,
                     "argv": "-t 0 -s 1 -n %d" % steps, "what": "unmodified reference llama2.ts (types stripped by its bundled sucrase), figure printed by llama2.ts:511"}
        print(name, out[name])
        with open(path, "w") as f:
            json.dump(out, f, indent=1)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--speed":
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, stdout=subprocess.DEVNULL)
        return time_reference(sys.argv[2:] or ["stories15M", "stories110M", "llama2_7b_L2"])
    names = sys.argv[1:] or list(PLAN)
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, stdout=subprocess.DEVNULL)
    os.makedirs(GOLD, exist_ok=True)
    inst = build_reference()
    for name in names:
        run_one(inst, name)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Execute INTEGRATION.md section 1 once, against the REAL reference file (build container only; CPU only).

What it does, all of it in a scratch directory under /tmp (nothing derived from the reference is written into the repo):
  1. copies /root/reference/llama2.ts;
  2. applies INTEGRATION.md section 1's patch MECHANICALLY -- four anchored edits (PATCH below): load the addon, create the
     context from the 7 header ints after readConfig (llama2.ts:433), hand every Float32Array of `weights` to `upload` after
     readWeights (llama2.ts:435), replace the call at llama2.ts:468 by `forward`;
  3. strips the types with the recipe oracle/make_goldens.py uses (the reference's own bundled sucrase; Node 12 runs the result);
  4. runs the patched reference on a synthetic checkpoint with a RECORDING STUB in place of l2_napi.node: a CommonJS module
     with the addon's entry points (`open / create / upload / forward`) that writes down what it is called with and fills
     state.logits with a deterministic one-hot so that the reference's own loop picks known tokens.
It prints the recording as JSON; tests/test_integration_patch_cpu.py asserts on it (14 kinds in file order, layer indices, float
counts, contents == the file's bytes at the right offsets whatever the view's byteOffset, one forward per position).
"""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

REF = "/root/reference"
WORK = "/tmp/l2_integration"

# (anchor text that must occur exactly once in the reference, how the edit is placed, text added) -- INTEGRATION.md section 1
PATCH = [
    ('import * as fs from "fs";', "after",
     'import { createRequire } from "module";\n'
     'const l2 = createRequire(import.meta.url)(process.env.L2_ADDON || "./l2_napi.node"); l2.open(process.env.L2_LIB || "./libllama2hip.so");\n'),
    ("  let config = readConfig(new BufferReader(configBuffer));", "after",
     "  const l2ctx = l2.create(new Int32Array(configBuffer.buffer, configBuffer.byteOffset, 7), 0);\n"),
    ("  let weights = readWeights(config, new FileHandleReader(fileHandle, configSize),config.shared_weights);", "after",
     '  ["token_embedding_table","rms_att_weight","wq","wk","wv","wo","rms_ffn_weight","w1","w2","w3","rms_final_weight","freq_cis_real","freq_cis_imag","wcls"]\n'
     "    .forEach((name, kind) => { const t = (weights as any)[name]; if (kind == 13 && config.shared_weights) return;\n"
     "      if (Array.isArray(t)) t.forEach((a, layer) => l2.upload(l2ctx, kind, layer, a)); else l2.upload(l2ctx, kind, -1, t); });\n"),
    ("    transformer(token, pos, config, state, weights);", "replace",
     "    l2.forward(l2ctx, token, pos, state.logits);\n"),
]

STUB = r"""
// recording stand-in for l2_napi.node (tools/run_integration_patch.py): same entry points, no GPU
const fs = require('fs');
const rec = { open: [], create: [], upload: [], forward: [] };
let V = 0;
function fnv(a) { let h = 0x811c9dc5; const b = new Uint8Array(a.buffer, a.byteOffset, a.byteLength);
  for (let i = 0; i < b.length; ++i) { h ^= b[i]; h = Math.imul(h, 0x01000193) >>> 0; } return h; }
module.exports = {
  open(path) { rec.open.push(path); },
  create(hdr, device) { if (!(hdr instanceof Int32Array)) throw new Error('create: header must be an Int32Array');
    rec.create.push({ header: Array.from(hdr), device }); V = Math.abs(hdr[5]); return { handle: 1 }; },
  upload(ctx, kind, layer, a) { if (!(a instanceof Float32Array)) throw new Error('upload: not a Float32Array');
    rec.upload.push({ kind, layer, floats: a.length, byteOffset: a.byteOffset, fnv: fnv(a), first: a[0], last: a[a.length - 1] }); },
  forward(ctx, token, pos, logits) { if (!(logits instanceof Float32Array) || logits.length < V) throw new Error('forward: logits array too small');
    rec.forward.push({ token, pos, logits_len: logits.length });
    logits.fill(0); logits[(token * 7 + pos * 13 + 3) % V] = 1; },
};
process.on('exit', () => fs.writeFileSync(process.env.L2_RECORD, JSON.stringify(rec)));
"""


def apply_patch(src_text):
    out = src_text
    for anchor, how, text in PATCH:
        assert out.count(anchor + "\n") == 1, "anchor not found exactly once in the reference: %r" % anchor
        if how == "after":
            out = out.replace(anchor + "\n", anchor + "\n" + text, 1)
        else:
            out = out.replace(anchor + "\n", text, 1)
    return out


def run(hdr, seed, steps):
    import make_goldens
    import oracle_lib as O
    shutil.rmtree(WORK, ignore_errors=True)
    os.makedirs(WORK)
    with open(os.path.join(REF, "llama2.ts")) as f:
        src = f.read()
    patched_ts = os.path.join(WORK, "llama2.patched.ts")
    with open(patched_ts, "w") as f:
        f.write(apply_patch(src))
    patched = os.path.join(WORK, "llama2.patched.mjs")
    make_goldens.strip_types(patched_ts, patched)
    with open(os.path.join(WORK, "l2_stub.cjs"), "w") as f:
        f.write(STUB)
    ckpt = os.path.join(WORK, "model.bin")
    O.synth_write(hdr, seed, ckpt)
    import synth_tokenizer
    synth_tokenizer.write(os.path.join(WORK, "tokenizer.bin"))       # llama2.ts:444 reads it from the working directory
    rec = os.path.join(WORK, "record.json")
    env = dict(os.environ, L2_ADDON=os.path.join(WORK, "l2_stub.cjs"), L2_LIB="/nonexistent/libllama2hip.so", L2_RECORD=rec)
    r = subprocess.run(["node", patched, ckpt, "-t", "0", "-s", "1", "-n", str(steps)], cwd=WORK, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if r.returncode != 0:
        raise RuntimeError("patched reference failed: %s" % r.stderr.decode("utf8", "replace")[-2000:])
    out = json.load(open(rec))
    out["stdout"] = r.stdout.decode("utf8", "replace")
    out["checkpoint"] = ckpt
    return out


if __name__ == "__main__":
    hdr = tuple(int(v) for v in sys.argv[1:8]) if len(sys.argv) >= 8 else (64, 176, 2, 4, 4, -512, 64)
    print(json.dumps(run(hdr, 1, 16))[:4000])

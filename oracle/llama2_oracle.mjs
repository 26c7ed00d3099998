// llama2_oracle.mjs -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// The forward pass of wizzard0/llama2.ts restated for a JavaScript ENGINE, so that bench.py's cpu_baseline leg can time
// "the reference's arithmetic in the reference's runtime" on the GPU box's host cores, where the reference's own source is
// not present (BASELINE.json north_star: "alongside the reference TS CPU path timed on the same box's host cores").  Like
// oracle/llama2_oracle.c it follows /root/reference/llama2.ts:168-303 step by step -- every number is a JS double, every
// Float32Array store rounds to fp32 (SURVEY.md 8(a-N)) -- and it is pinned the same way: tests/test_oracle_golden.py runs
// it under Node on the synthetic checkpoints of the fixtures and requires the sha256 of the logits of EVERY step to equal
// what the real reference produced (tests/golden/*.json: logits_sha256).  Own structure: one ArrayBuffer image of the
// checkpoint, typed-array views per tensor, a model object -- not the reference's classes or reader.
//
// Only tests/ and bench.py's cpu_baseline leg run this.  The product (llama2.ts_amd/) never does.
//
//   node llama2_oracle.mjs <checkpoint.bin> <steps> [--sha] [--prompt id,id,...]
// prints ONE JSON line: {"tokens": [...], "tok_s": ..., "steps": n, "sha256": [...] (with --sha)}.
// Timing as the reference does it (llama2.ts:507, 511): the clock starts after the first token.
import * as fs from "fs";
import * as crypto from "crypto";

/** Views on a llama2.c-v0 checkpoint image (header llama2.ts:80-93, tensor order :112-129). */
function openModel(file) {
  const fd = fs.openSync(file, "r");
  const size = fs.fstatSync(fd).size;
  const head = Buffer.alloc(28);
  fs.readSync(fd, head, 0, 28, 0);
  const h = new Int32Array(head.buffer, head.byteOffset, 7);
  const cfg = { dim: h[0], hidden: h[1], layers: h[2], heads: h[3], kvHeads: h[4], vocab: Math.abs(h[5]), seqLen: h[6], shared: h[5] > 0 };
  cfg.headSize = cfg.dim / cfg.heads;                                      // :91
  // the float stream after the header, read in pieces below Node's 2 GiB Buffer limit into ONE Float32Array
  const nFloats = (size - 28) / 4;
  const all = new Float32Array(nFloats);
  const bytes = new Uint8Array(all.buffer);
  for (let done = 0; done < bytes.length;) {
    const n = fs.readSync(fd, bytes, done, Math.min(bytes.length - done, 1 << 30), 28 + done);
    if (n <= 0) throw new Error("checkpoint truncated");
    done += n;
  }
  fs.closeSync(fd);
  const { dim: d, hidden: hd, layers: L, vocab: V, seqLen: S, headSize: hs } = cfg;
  let at = 0;
  const take = (n) => { const v = all.subarray(at, at + n); at += n; return v; };
  const perLayer = (n) => { const out = []; for (let l = 0; l < L; ++l) out.push(take(n)); return out; };
  const w = {};
  w.emb = take(V * d);                                                     // :114
  w.rmsAtt = perLayer(d);                                                  // :115
  w.wq = perLayer(d * d); w.wk = perLayer(d * d); w.wv = perLayer(d * d); w.wo = perLayer(d * d);   // :116-119 (wk, wv always (d, d): :117-118)
  w.rmsFfn = perLayer(d);                                                  // :120
  w.w1 = perLayer(hd * d); w.w2 = perLayer(d * hd); w.w3 = perLayer(hd * d);                       // :121-123
  w.rmsFinal = take(d);                                                    // :124
  w.fcr = take(S * hs / 2); w.fci = take(S * hs / 2);                      // :125-126
  w.cls = cfg.shared ? w.emb : take(V * d);                                // :127
  if (at != nFloats) throw new Error("checkpoint size does not match its header");
  const st = {                                                             // newRunState, :147-163
    x: new Float32Array(d), xb: new Float32Array(d), xb2: new Float32Array(d), hb: new Float32Array(hd), hb2: new Float32Array(hd),
    q: new Float32Array(d), k: new Float32Array(d), v: new Float32Array(d), att: new Float32Array(cfg.heads * S),
    logits: new Float32Array(V), kc: new Float32Array(L * S * d), vc: new Float32Array(L * S * d),
  };
  return { cfg, w, st };
}

/** out[i] = sum_j m[i * n + j] * x[j]: a double sum, ONE rounding at the store (:196-203). */
function gemv(out, x, m, n, rows) {
  for (let i = 0; i < rows; ++i) {
    let acc = 0.0;
    const base = i * n;
    for (let j = 0; j < n; ++j) acc += m[base + j] * x[j];
    out[i] = acc;
  }
}

/** o = g * (x / sqrt(mean(x^2) + 1e-5)), associated as g * (s * x) (:172-179). */
function norm(o, x, g, n) {
  let ss = 0.0;
  for (let j = 0; j < n; ++j) ss += x[j] * x[j];
  ss /= n;
  ss = 1.0 / Math.sqrt(1e-5 + ss);
  for (let j = 0; j < n; ++j) o[j] = g[j] * (ss * x[j]);
}

/** In place over a[from .. from + n): exps stored (rounded), then summed, then the quotients stored (:181-194). */
function softmaxRange(a, from, n) {
  let mx = a[from];
  for (let i = 1; i < n; ++i) if (a[from + i] > mx) mx = a[from + i];
  for (let i = 0; i < n; ++i) a[from + i] = Math.exp(a[from + i] - mx);
  let sum = 0.0;
  for (let i = 0; i < n; ++i) sum += a[from + i];
  for (let i = 0; i < n; ++i) a[from + i] = a[from + i] / sum;
}

/** One transformer() call (:205-303): leaves the logits of `pos` in st.logits. */
function step(model, token, pos) {
  const { cfg, w, st } = model;
  const d = cfg.dim, hd = cfg.hidden, hs = cfg.headSize, S = cfg.seqLen, H = cfg.heads;
  st.x.set(w.emb.subarray(token * d, token * d + d));                      // :211
  const rope = pos * hs / 2;
  for (let l = 0; l < cfg.layers; ++l) {
    norm(st.xb, st.x, w.rmsAtt[l], d);                                     // :216
    gemv(st.q, st.xb, w.wq[l], d, d); gemv(st.k, st.xb, w.wk[l], d, d); gemv(st.v, st.xb, w.wv[l], d, d);   // :219-221
    for (let i = 0; i < d; i += 2) {                                       // RoPE on adjacent pairs, same angles for every head (:224-235)
      const c = w.fcr[rope + (i % hs) / 2], s = w.fci[rope + (i % hs) / 2];
      const q0 = st.q[i], q1 = st.q[i + 1], k0 = st.k[i], k1 = st.k[i + 1];
      st.q[i] = q0 * c - q1 * s; st.q[i + 1] = q0 * s + q1 * c;
      st.k[i] = k0 * c - k1 * s; st.k[i + 1] = k0 * s + k1 * c;
    }
    const slab = l * S * d;
    st.kc.set(st.k, slab + pos * d); st.vc.set(st.v, slab + pos * d);      // :238-240
    const scale = Math.sqrt(hs);
    for (let h = 0; h < H; ++h) {                                          // :244-267
      const qo = h * hs, ao = h * S;
      for (let t = 0; t <= pos; ++t) {
        const ko = slab + t * d + qo;
        let dot = 0.0;
        for (let i = 0; i < hs; ++i) dot += st.q[qo + i] * st.kc[ko + i];
        st.att[ao + t] = dot / scale;                                      // :253
      }
      softmaxRange(st.att, ao, pos + 1);                                   // :256
      for (let i = 0; i < hs; ++i) st.xb[qo + i] = 0;
      for (let t = 0; t <= pos; ++t) {                                     // the accumulator is a Float32Array element: rounded every t (:260-265)
        const vo = slab + t * d + qo, p = st.att[ao + t];
        for (let i = 0; i < hs; ++i) st.xb[qo + i] += p * st.vc[vo + i];
      }
    }
    gemv(st.xb2, st.xb, w.wo[l], d, d);                                    // :270
    for (let i = 0; i < d; ++i) st.x[i] += st.xb2[i];                      // :273
    norm(st.xb, st.x, w.rmsFfn[l], d);                                     // :276
    gemv(st.hb, st.xb, w.w1[l], d, hd); gemv(st.hb2, st.xb, w.w3[l], d, hd);   // :280-281
    for (let i = 0; i < hd; ++i) st.hb[i] = st.hb[i] * (1.0 / (1.0 + Math.exp(-st.hb[i])));   // silu, stored (:285)
    for (let i = 0; i < hd; ++i) st.hb[i] = st.hb[i] * st.hb2[i];          // then the product, stored (:289)
    gemv(st.xb, st.hb, w.w2[l], hd, d);                                    // :292
    for (let i = 0; i < d; ++i) st.x[i] += st.xb[i];                       // :295
  }
  norm(st.x, st.x, w.rmsFinal, d);                                         // :299
  gemv(st.logits, st.x, w.cls, d, cfg.vocab);                              // :302
}

function firstMax(a) {                                                     // :364-366
  let at = 0;
  for (let i = 1; i < a.length; ++i) if (a[i] > a[at]) at = i;
  return at;
}

const [, , file, stepsArg, ...flags] = process.argv;
if (!file || !stepsArg) { console.error("usage: node llama2_oracle.mjs <checkpoint.bin> <steps> [--sha] [--prompt id,id,...]"); process.exit(1); }
const wantSha = flags.includes("--sha");
const pi = flags.indexOf("--prompt");
const prompt = pi >= 0 ? flags[pi + 1].split(",").filter((t) => t.length).map((t) => parseInt(t)) : [];
const model = openModel(file);
const steps = Math.min(parseInt(stepsArg), model.cfg.seqLen);
const tokens = [], shas = [];
let token = 1, t0 = 0;
for (let pos = 0; pos < steps; ++pos) {
  step(model, token, pos);
  if (wantSha) shas.push(crypto.createHash("sha256").update(Buffer.from(model.st.logits.buffer, model.st.logits.byteOffset, model.st.logits.byteLength)).digest("hex"));
  token = pos < prompt.length ? prompt[pos] : firstMax(model.st.logits);   // teacher-forced prompt positions (:471-473), else greedy (:478)
  tokens.push(token);
  if (!t0) t0 = Date.now();
}
const ms = Date.now() - t0;
const out = { tokens, steps, tok_s: steps > 1 && ms > 0 ? (steps - 1) / ms * 1000 : null, node: process.version };
if (wantSha) out.sha256 = shas;
process.stdout.write(JSON.stringify(out) + "\n");

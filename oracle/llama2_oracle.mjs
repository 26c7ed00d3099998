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
//   node llama2_oracle.mjs --synth d,hidden,L,H,KV,V,S,seed <steps> [--sha] [--tensor-sha]
// prints ONE JSON line: {"tokens": [...], "tok_s": ..., "steps": n, "sha256": [...] (with --sha)}.
// Timing as the reference does it (llama2.ts:507, 511): the clock starts after the first token.
// --synth: no file -- the typed arrays are filled IN PROCESS by the repo's deterministic generator restated here (the integer hash
// of oracle/llama2_oracle.c: orc_synth_fill / orc_synth_freq; same header ints, same seed => the same bytes `oracle_cli synth`
// writes, pinned by tests/test_oracle_golden.py through --tensor-sha: sha256 of every tensor).  That is how bench.py times this
// restatement on the 27 GB Llama-2-7B shape, which cannot go through a file in a benchmark's time.
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

// ---- the synthetic generator (oracle/llama2_oracle.c:73-150): 32-bit integer hashing, one fp32 multiply and one fp32 add per element
function hash32(a) {
  a ^= a >>> 16; a = Math.imul(a, 0x7feb352d); a ^= a >>> 15; a = Math.imul(a, 0x846ca68b); a ^= a >>> 16;
  return a >>> 0;
}
/** out[i] = bias + fl(c(g0 + i) * scale), c = a centred sum of four 16-bit uniforms hashed from the element's index in the float stream.
 *  (One thread, like everything here: V8 ran the forward pass ~15 % slower in a process that had filled its arrays from worker threads,
 *  whatever the arrays were backed by -- a baseline must not carry that.  The three hashes are written out: ~7 ns per element.) */
function synthFill(out, g0, seed, scale, bias) {
  const sk = Math.imul(seed, 0x9E3779B9) ^ 0x85ebca6b;
  const n = out.length;
  let lo = g0 % 4294967296, hi = Math.floor(g0 / 4294967296), k = hash32(hi ^ sk) | 0;
  for (let i = 0; i < n; ++i) {
    let a = (lo ^ k) | 0;
    a ^= a >>> 16; a = Math.imul(a, 0x7feb352d); a ^= a >>> 15; a = Math.imul(a, 0x846ca68b); a ^= a >>> 16;      // h1 = hash32(lo ^ k)
    let b = (a + 0x9E3779B9) | 0;
    b ^= b >>> 16; b = Math.imul(b, 0x7feb352d); b ^= b >>> 15; b = Math.imul(b, 0x846ca68b); b ^= b >>> 16;      // h2 = hash32(h1 + golden)
    const c = (a & 0xffff) + (a >>> 16) + (b & 0xffff) + (b >>> 16) - 131070;
    out[i] = bias + Math.fround(c * scale);             // the product rounded to fp32, the sum rounded by the store
    if (++lo == 4294967296) { lo = 0; ++hi; k = hash32(hi ^ sk) | 0; }
  }
}
function detExp(x) {                                    // x in [-10, 0], IEEE basic operations only
  const y = x / 1024.0;
  let t = 1.0, s = 1.0;
  for (let k = 1; k <= 14; ++k) { t = (t * y) / k; s = s + t; }
  for (let i = 0; i < 10; ++i) s = s * s;
  return s;
}
function detSinCos(x) {                                 // |x| <= 1
  const x2 = x * x;
  let ts = x, tc = 1.0, ss = x, cc = 1.0;
  for (let k = 1; k <= 12; ++k) {
    tc = ((-tc) * x2) / ((2 * k - 1) * (2 * k)); cc = cc + tc;
    ts = ((-ts) * x2) / ((2 * k) * (2 * k + 1)); ss = ss + ts;
  }
  return [ss, cc];
}
/** The model of openModel(), every tensor its own Float32Array filled by the generator (checkpoint order = index in the float stream). */
function synthModel(hdr, seed) {
  const cfg = { dim: hdr[0], hidden: hdr[1], layers: hdr[2], heads: hdr[3], kvHeads: hdr[4], vocab: Math.abs(hdr[5]), seqLen: hdr[6], shared: hdr[5] > 0 };
  cfg.headSize = cfg.dim / cfg.heads;
  const { dim: d, hidden: hd, layers: L, vocab: V, seqLen: S, headSize: hs } = cfg;
  const STD = 37837.22723720648;                        // sqrt(4 * (65536^2 - 1) / 12)
  let at = 0;
  const gen = (n, sigma, bias) => { const a = new Float32Array(n); synthFill(a, at, seed, Math.fround(sigma / STD), bias); at += n; return a; };
  const perLayer = (n, sigma, bias) => { const out = []; for (let l = 0; l < L; ++l) out.push(gen(n, sigma, bias)); return out; };
  const sd = 1.0 / Math.sqrt(d), sh = 1.0 / Math.sqrt(hd);
  const w = {};
  w.emb = gen(V * d, 0.02, 0);
  w.rmsAtt = perLayer(d, 0.1, 1);
  w.wq = perLayer(d * d, sd, 0); w.wk = perLayer(d * d, sd, 0); w.wv = perLayer(d * d, sd, 0); w.wo = perLayer(d * d, sd, 0);
  w.rmsFfn = perLayer(d, 0.1, 1);
  w.w1 = perLayer(hd * d, sd, 0); w.w2 = perLayer(d * hd, sh, 0); w.w3 = perLayer(hd * d, sd, 0);
  w.rmsFinal = gen(d, 0.1, 1);
  const hs2 = hs / 2;
  w.fcr = new Float32Array(S * hs2); w.fci = new Float32Array(S * hs2);
  for (let j = 0; j < hs2; ++j) {                       // angle t * theta by complex rotation (orc_synth_freq)
    const theta = detExp(-(((2.0 * j) / hs) * 9.210340371976184));
    const [st, ct] = detSinCos(theta);
    let cr = 1.0, ci = 0.0;
    for (let t = 0; t < S; ++t) {
      w.fcr[t * hs2 + j] = cr; w.fci[t * hs2 + j] = ci;
      const nr = cr * ct - ci * st, ni = cr * st + ci * ct;
      cr = nr; ci = ni;
    }
  }
  at += 2 * S * hs2;
  w.cls = cfg.shared ? w.emb : gen(V * d, 0.02, 0);
  const st = {
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

let [, , file, stepsArg, ...flags] = process.argv;
let synth = null;
if (file == "--synth") { synth = (stepsArg || "").split(",").map((t) => parseInt(t)); [stepsArg, ...flags] = flags; }
if (!file || !stepsArg || (synth && synth.length != 8)) {
  console.error("usage: node llama2_oracle.mjs <checkpoint.bin> <steps> [--sha] [--prompt id,id,...]\n       node llama2_oracle.mjs --synth d,hidden,L,H,KV,V,S,seed <steps> [--sha] [--tensor-sha]");
  process.exit(1);
}
const wantSha = flags.includes("--sha");
const pi = flags.indexOf("--prompt");
const prompt = pi >= 0 ? flags[pi + 1].split(",").filter((t) => t.length).map((t) => parseInt(t)) : [];
const tGen = Date.now();
const model = synth ? synthModel(synth.slice(0, 7), synth[7] >>> 0) : openModel(file);
const genSeconds = (Date.now() - tGen) / 1000;
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
const out = { tokens, steps, tok_s: steps > 1 && ms > 0 ? (steps - 1) / ms * 1000 : null, node: process.version, load_s: genSeconds };
if (wantSha) out.sha256 = shas;
if (flags.includes("--tensor-sha")) {                   // sha256 of every tensor in checkpoint order (per-layer tensors: layers back to back)
  const sha = (arrays) => { const h = crypto.createHash("sha256"); for (const a of arrays) h.update(Buffer.from(a.buffer, a.byteOffset, a.byteLength)); return h.digest("hex"); };
  const w = model.w, one = (a) => sha([a]);
  out.tensor_sha256 = [one(w.emb), sha(w.rmsAtt), sha(w.wq), sha(w.wk), sha(w.wv), sha(w.wo), sha(w.rmsFfn), sha(w.w1), sha(w.w2), sha(w.w3),
    one(w.rmsFinal), one(w.fcr), one(w.fci)].concat(model.cfg.shared ? [] : [one(w.cls)]);
}
process.stdout.write(JSON.stringify(out) + "\n");

// @ts-check
// llama2.mjs -- the llama2.ts command line with the forward pass on an MI355X.
//
//   node llama2.mjs <checkpoint> [-t temp] [-p topp] [-s seed] [-n steps] [-i prompt]
//
// Same CLI, `.bin` reader, Config / TransformerWeights / RunState field names, tokenizer, sampler and
// token loop as the reference (wizzard0/llama2.ts, llama2.ts:399-526), but transformer(token, pos, ...)
// (llama2.ts:205-303) runs in libllama2hip.so through the N-API addon l2_napi.node: every weight array is
// handed to the GPU as soon as the reader produces it, activations and the KV cache live in HBM, and
// state.logits is a Float32Array over the pinned host buffer the classifier kernel writes.
//
// Plain ECMAScript module with JSDoc types (valid under `tsc --allowJs --checkJs`, runs unchanged on
// Node >= 12 and Bun): the build image has no TypeScript compiler, see DESIGN.md.
// There is no CPU fallback: without the addon, the library or a GPU the program exits with an error.
import * as fs from "fs";
import * as path from "path";
import { createRequire } from "module";
import { fileURLToPath } from "url";

const here = path.dirname(fileURLToPath(import.meta.url));
const require_ = createRequire(import.meta.url);

// tensor kinds / option keys of include/llama2_hip.h
const T = { token_embedding_table: 0, rms_att_weight: 1, wq: 2, wk: 3, wv: 4, wo: 5, rms_ffn_weight: 6, w1: 7, w2: 8, w3: 9,
  rms_final_weight: 10, freq_cis_real: 11, freq_cis_imag: 12, wcls: 13 };

/** @returns {any} the N-API addon with the HIP library opened (throws if either is missing) */
function loadBackend() {
  const addonPath = process.env.L2_NAPI_PATH || path.join(here, "l2_napi.node");
  let addon;
  try {
    addon = require_(addonPath);
  } catch (e) {
    throw new Error("cannot load the N-API addon " + addonPath + " (run __graft_entry__.build()): " + e.message);
  }
  addon.open(process.env.L2_LIB_PATH || path.join(here, "..", "lib", "libllama2hip.so"));
  return addon;
}

// ----------------------------------------------------------------------------
// binary readers (llama2.ts:15-68)

class BufferReader {
  /** @param {Buffer} buffer */
  constructor(buffer) {
    this.view = new DataView(buffer.buffer, buffer.byteOffset, buffer.byteLength);
    this.position = 0;
  }
  getInt32LE() { const v = this.view.getInt32(this.position, true); this.position += 4; return v; }
  getFloat32LE() { const v = this.view.getFloat32(this.position, true); this.position += 4; return v; }
  /** @param {Uint8Array} bytes */
  getBytesInto(bytes) {
    bytes.set(new Uint8Array(this.view.buffer, this.view.byteOffset + this.position, bytes.length));
    this.position += bytes.length;
    return bytes;
  }
}

class FileHandleReader {
  /** @param {number} handle @param {number} offset */
  constructor(handle, offset) { this.handle = handle; this.position = offset; }
  /** one tensor of the checkpoint as a Float32Array (llama2.ts:51-59) @param {...number} dims */
  getF32Array(...dims) {
    const totalFloats = dims.reduce((a, b) => a * b);
    const bytes = Buffer.alloc(totalFloats * 4);
    let got = 0;
    while (got < bytes.length) {   // positional reads; a single readSync may return short on huge tensors
      const n = fs.readSync(this.handle, bytes, got, Math.min(bytes.length - got, 1 << 30), this.position + got);
      if (n <= 0) throw new Error("checkpoint truncated");
      got += n;
    }
    this.position += bytes.length;
    return new Float32Array(bytes.buffer, bytes.byteOffset, totalFloats);
  }
}

// ----------------------------------------------------------------------------
// Config / TransformerWeights / RunState (llama2.ts:69-163)

/** @typedef {{dim:number, hidden_dim:number, n_layers:number, n_heads:number, n_kv_heads:number, vocab_size:number,
 *             seq_len:number, shared_weights:boolean, head_size:number, header:Int32Array}} Config */

/** readConfig (llama2.ts:80-93) @param {BufferReader} buffer @returns {Config} */
function readConfig(buffer) {
  const header = new Int32Array(7);
  for (let i = 0; i < 7; i++) header[i] = buffer.getInt32LE();
  const c = /** @type {Config} */ ({});
  c.header = header;
  c.dim = header[0];
  c.hidden_dim = header[1];
  c.n_layers = header[2];
  c.n_heads = header[3];
  c.n_kv_heads = header[4];
  c.vocab_size = Math.abs(header[5]);
  c.seq_len = header[6];
  c.shared_weights = header[5] > 0;
  c.head_size = c.dim / c.n_heads;
  return c;
}

/** TransformerWeights lives in HBM; the JS object only keeps the handle (and which arrays were sent).
 *  readWeights (llama2.ts:112-129): same order, same shapes; each array goes to the GPU the moment the
 *  reader returns it and is dropped, so host memory never holds more than one tensor.
 *  @param {Config} config @param {FileHandleReader} buffer @param {boolean} shared_weights @param {any} be @param {any} ctx */
function readWeights(config, buffer, shared_weights, be, ctx) {
  const w = { ctx, uploaded: /** @type {string[]} */ ([]) };
  const one = (/** @type {string} */ name, /** @type {number[]} */ dims) => {
    be.upload(ctx, T[name], -1, buffer.getF32Array(...dims));
    w.uploaded.push(name);
  };
  const perLayer = (/** @type {string} */ name, /** @type {number[]} */ dims) => {
    for (let l = 0; l < config.n_layers; ++l) be.upload(ctx, T[name], l, buffer.getF32Array(...dims));
    w.uploaded.push(name);
  };
  const d = config.dim, h = config.hidden_dim;
  one("token_embedding_table", [config.vocab_size, d]);
  perLayer("rms_att_weight", [d]);
  perLayer("wq", [d, d]);
  perLayer("wk", [d, d]);
  perLayer("wv", [d, d]);
  perLayer("wo", [d, d]);
  perLayer("rms_ffn_weight", [d]);
  perLayer("w1", [h, d]);
  perLayer("w2", [d, h]);
  perLayer("w3", [h, d]);
  one("rms_final_weight", [d]);
  one("freq_cis_real", [config.seq_len, config.head_size / 2]);
  one("freq_cis_imag", [config.seq_len, config.head_size / 2]);
  if (!shared_weights) one("wcls", [config.vocab_size, d]);   // shared: the library aliases the table (llama2.ts:127)
  return w;
}

/** newRunState (llama2.ts:147-163): only `logits` and `indices` are touched by the host loop.
 *  By default `logits` is an ordinary Float32Array that l2_forward fills from the pinned buffer the classifier
 *  kernel wrote (a 128 KB memcpy).  L2_ZERO_COPY_JS=1 wraps the pinned buffer itself instead; it is opt-in
 *  because Node 12 crashes at exit with external ArrayBuffers once the heap has been collected (see DESIGN.md).
 *  @param {Config} config @param {any} be @param {any} ctx */
function newRunState(config, be, ctx) {
  let logits = null, zeroCopyOk = false;
  if (process.env.L2_ZERO_COPY_JS == "1") {
    try {
      logits = new Float32Array(be.logitsBuffer(ctx, config.vocab_size), 0, config.vocab_size);
      zeroCopyOk = true;
    } catch (e) { logits = null; }
  }
  if (!logits) logits = new Float32Array(config.vocab_size);
  return { logits, zeroCopyOk, indices: new Array(config.vocab_size) };
}

/** transformer(token, pos, p, s, w) (llama2.ts:205-303): fills s.logits.
 *  @param {number} token @param {number} pos @param {Config} p @param {any} s @param {any} w @param {any} be */
function transformer(token, pos, p, s, w, be) {
  be.forward(w.ctx, token, pos, s.zeroCopyOk ? null : s.logits);
}

// ----------------------------------------------------------------------------
// softmax for the sampler (llama2.ts:181-194)
/** @param {Float32Array} x @param {number} xPtr @param {number} size */
function softmax(x, xPtr, size) {
  let max_val = x[xPtr];
  for (let i = 1; i < size; i++) if (x[i + xPtr] > max_val) max_val = x[i + xPtr];
  for (let i = 0; i < size; i++) x[i + xPtr] = Math.exp(x[i + xPtr] - max_val);
  let sum = 0;
  for (let i = 0; i < size; i++) sum += x[i + xPtr];
  for (let i = 0; i < size; i++) x[i + xPtr] /= sum;
}

// ----------------------------------------------------------------------------
// tokenizer: greedy best-score pair merging (llama2.ts:305-344)
/** @param {string} text @param {string[]} vocab @param {number[]} vocab_scores @param {number} vocab_size @param {Int32Array} tokens */
function bpe_encode(text, vocab, vocab_scores, vocab_size, tokens) {
  let n_tokens = 0;
  for (let i = 0; i < text.length; ++i) {
    const id = vocab.indexOf(text.charAt(i));
    if (id == -1) throw new Error("Error: character not found in vocab: " + text.charAt(i));
    tokens[n_tokens++] = id;
  }
  for (;;) {
    let best_score = -1e10, best_id = -1, best_idx = -1;
    for (let i = 0; i < n_tokens - 1; ++i) {
      const id = vocab.indexOf(vocab[tokens[i]] + vocab[tokens[i + 1]]);
      if (id != -1 && vocab_scores[id] > best_score) { best_score = vocab_scores[id]; best_id = id; best_idx = i; }
    }
    if (best_idx == -1) break;
    tokens[best_idx] = best_id;
    for (let i = best_idx + 1; i < n_tokens - 1; i++) tokens[i] = tokens[i + 1];
    n_tokens--;
  }
  return n_tokens;
}

// ----------------------------------------------------------------------------
// rng: xorshift* on a 64-bit BigInt state (llama2.ts:348-360)
let rng_seed = 0n;
function random_u32() {
  rng_seed ^= rng_seed >> 12n;
  rng_seed ^= (rng_seed << 25n) & 0xffffffffffffffffn;
  rng_seed ^= rng_seed >> 27n;
  return Number(((rng_seed * 0x2545F4914F6CDD1Dn) >> 32n) & 0xffffffffn);
}
const floatCaster = new Float32Array(1);
function random_f32() {
  floatCaster[0] = (random_u32() / 256) / 16777216.0;
  return floatCaster[0];
}

// ----------------------------------------------------------------------------
// samplers (llama2.ts:364-394), quirks kept: argmax = first maximum; sample_topp never picks the element at
// lastIdx and falls back to id 0
/** @param {Float32Array} arr */
function argmax(arr) {
  let best = 0;
  for (let i = 1; i < arr.length; i++) if (arr[i] > arr[best]) best = i;
  return best;
}
/** @param {Float32Array} logits @param {number} vocabSize */
function sample(logits, vocabSize) {
  let sum = 0;
  for (let i = 0; i < logits.length; i++) sum += logits[i];
  const randValue = random_f32() * sum;
  let cumProb = 0;
  for (let i = 0; i < vocabSize; i++) { cumProb += logits[i]; if (randValue < cumProb) return i; }
  return 0;
}
/** @param {Float32Array} logits @param {number} topp @param {{index:number, prob:number}[]} probindex */
function sample_topp(logits, topp, probindex) {
  for (let i = 0; i < probindex.length; i++) probindex[i] = { index: i, prob: logits[i] };
  probindex.sort((a, b) => b.prob - a.prob);
  let cumProb = 0, lastIdx = 0;
  for (let i = 0; i < probindex.length; i++) { cumProb += probindex[i].prob; if (cumProb > topp) { lastIdx = i; break; } }
  const randValue = random_f32() * cumProb;
  cumProb = 0;
  for (let i = 0; i < lastIdx; i++) { cumProb += probindex[i].prob; if (randValue < cumProb) return probindex[i].index; }
  return 0;
}

// ----------------------------------------------------------------------------
function error_usage() {
  console.error("Usage: ... llama2.ts <checkpoint> [options]");
  console.error("Example: llama2.ts model.bin -n 256 -i \"Once upon a time\"");
  console.error("Options:");
  console.error("  -t <float>  temperature, default 1.0");
  console.error("  -p <float>  p value in top-p (nucleus) sampling. default 0.9, 0 = off");
  console.error("  -s <int>    random seed, default time(NULL)");
  console.error("  -n <int>    number of steps to run for, default 256. 0 = max_seq_len");
  console.error("  -i <string> input prompt");
  process.exit(1);
}

function main() {
  const [, , checkpoint, ...args] = process.argv;
  let temperature = 1.0, topp = 1.0, steps = 256;
  /** @type {string|null} */
  let prompt = null;
  rng_seed = 0n;
  if (!checkpoint) return error_usage();
  for (let i = 0; i < args.length; i += 2) {        // "-x value" pairs only (llama2.ts:409-423)
    if (i + 1 >= args.length) return error_usage();
    const arg = args[i], val = args[i + 1];
    if (arg.charAt(0) != "-" || arg.length != 2) return error_usage();
    switch (arg[1]) {
      case "t": temperature = parseFloat(val); break;
      case "p": topp = parseFloat(val); break;
      case "s": rng_seed = BigInt(parseInt(val)); break;
      case "n": steps = parseInt(val); break;
      case "i": prompt = val; break;
      default: return error_usage();
    }
  }
  if (rng_seed == 0n) rng_seed = BigInt(Date.now());

  const be = loadBackend();
  const device = parseInt(process.env.L2_DEVICE || "0");

  // checkpoint: 7-int header, then the tensors in llama2.c-v0 order (llama2.ts:427-436)
  let config, ctx, weights;
  if (process.env.L2_NATIVE_LOADER == "1") {
    // opt-in extra (SURVEY.md 8(f2)): the library reads the file itself through pinned staging buffers
    const r = be.loadCheckpoint(checkpoint, device);
    config = readConfig(new BufferReader(Buffer.from(r.header.buffer, r.header.byteOffset, 28)));
    ctx = r.ctx;
    weights = { ctx, uploaded: ["(native loader)"] };
  } else {
    const fileHandle = fs.openSync(checkpoint, "r");
    const configBuffer = Buffer.alloc(28);
    fs.readSync(fileHandle, configBuffer, 0, 28, 0);
    config = readConfig(new BufferReader(configBuffer));
    ctx = be.create(config.header, device);
    weights = readWeights(config, new FileHandleReader(fileHandle, 28), config.shared_weights, be, ctx);
    fs.closeSync(fileHandle);
  }

  if (steps <= 0 || steps > config.seq_len) steps = config.seq_len;

  // tokenizer.bin from the working directory (llama2.ts:442-449)
  const vocab = new Array(config.vocab_size);
  const vocab_scores = new Array(config.vocab_size);
  const tokBuffer = new BufferReader(fs.readFileSync("tokenizer.bin"));
  tokBuffer.getInt32LE();   // max_token_length, unused
  const utf8 = new TextDecoder();
  for (let i = 0; i < config.vocab_size; i++) {
    vocab_scores[i] = tokBuffer.getFloat32LE();
    vocab[i] = utf8.decode(tokBuffer.getBytesInto(new Uint8Array(tokBuffer.getInt32LE())));
  }

  const state = newRunState(config, be, ctx);

  const prompt_tokens = new Int32Array(config.seq_len);
  let num_prompt_tokens = 0;
  if (prompt != null) num_prompt_tokens = bpe_encode(prompt, vocab, vocab_scores, config.vocab_size, prompt_tokens);

  // opt-in extra (SURVEY.md 8(f3)): feed the teacher-forced prompt positions in one batched call instead of one
  // transformer() each (the reference ignores their logits anyway, llama2.ts:471-473)
  let prefilled = 0;
  if (process.env.L2_PREFILL == "1" && num_prompt_tokens > 1) {
    const n = Math.min(num_prompt_tokens, steps);
    const fed = new Int32Array(n);
    fed[0] = 1;
    for (let i = 1; i < n; i++) fed[i] = prompt_tokens[i - 1];
    be.prefill(ctx, fed, 0, null);
    prefilled = n;
  }

  // opt-in extra (SURVEY.md 8(f1)): keep the greedy loop on the device, `chunk` tokens per call
  const deviceGreedy = process.env.L2_DEVICE_GREEDY == "1" && temperature == 0.0;
  // ... and the sampled branch too (llama2.ts:480-493): temperature, softmax, sample / sample_topp and the RNG run on
  // the device, which hands back token ids and the advanced rng_seed (needs 0 <= rng_seed < 2^64, as xorshift keeps it)
  const deviceSampler = process.env.L2_DEVICE_SAMPLER == "1" && temperature != 0.0 && rng_seed >= 0n && rng_seed < (1n << 64n);
  const rngHalves = new Uint32Array(2);

  let start = 0, next = 0, token = 1, pos = 0;   // token 1 = BOS (llama2.ts:463)
  /** @type {number[]} */
  let ahead = [];                                 // tokens the device already chose
  while (pos < steps) {
    if (deviceGreedy && pos >= num_prompt_tokens) {
      if (ahead.length == 0) ahead = Array.from(be.decodeGreedy(ctx, token, pos, Math.min(16, steps - pos)));
      next = /** @type {number} */ (ahead.shift());
    } else if (deviceSampler && pos >= num_prompt_tokens) {
      if (ahead.length == 0) {
        rngHalves[0] = Number(rng_seed & 0xffffffffn); rngHalves[1] = Number(rng_seed >> 32n);
        ahead = Array.from(be.decodeSample(ctx, token, pos, Math.min(16, steps - pos), temperature, topp, rngHalves));
        rng_seed = (BigInt(rngHalves[1]) << 32n) | BigInt(rngHalves[0]);
      }
      next = /** @type {number} */ (ahead.shift());
    } else {
      if (pos >= prefilled) transformer(token, pos, config, state, weights, be);
      if (pos < num_prompt_tokens) {
        next = prompt_tokens[pos];                // teacher-forced prompt (llama2.ts:471-473)
      } else if (temperature == 0.0) {
        next = argmax(state.logits);
      } else {
        for (let q = 0; q < config.vocab_size; q++) state.logits[q] /= temperature;
        softmax(state.logits, 0, config.vocab_size);
        next = (topp <= 0 || topp >= 1) ? sample(state.logits, config.vocab_size) : sample_topp(state.logits, topp, state.indices);
      }
    }
    pos++;
    if (next == 1) break;                         // BOS ends the sequence (llama2.ts:499)
    const piece = (token == 1 && vocab[next].charAt(0) == " ") ? vocab[next].substring(1) : vocab[next];
    process.stdout.write(piece);
    token = next;
    if (start == 0) start = Date.now();
  }
  const elapsed_ms = Date.now() - start;
  console.log("\n\nachieved tok/s: %f\n", (pos - 1) / elapsed_ms * 1000.0);
  if (process.env.L2_STATS == "1") {
    // opt-in, on stderr so that stdout stays what the reference prints: the same rate against the HBM roofline,
    // with SURVEY.md 8(d)'s algorithmic bytes per token averaged over the positions of this run
    const d = config.dim, h = config.hidden_dim, L = config.n_layers, V = config.vocab_size, hs = d / config.n_heads;
    let bytes = 0;
    for (let q = 1; q < pos; q++) bytes += 4 * (L * (4 * d * d + 3 * d * h + 2 * d) + d + V * d + d + L * (2 * (q + 1) * d + 2 * d) + hs) + 4 * V;
    const tok_s = (pos - 1) / elapsed_ms * 1000.0, bpt = pos > 1 ? bytes / (pos - 1) : 0;
    console.error(JSON.stringify({ tokens_timed: pos - 1, tok_s, algorithmic_bytes_per_token: Math.round(bpt),
      hbm_gbs: bpt * tok_s / 1e9, hbm_frac_of_8tbs: bpt * tok_s / 8e12 }));
  }
  be.destroy(ctx);
}

try {
  main();
} catch (e) {
  console.error(String(e && e.message ? e.message : e));   // the reference dies on an uncaught Error: exit code 1
  process.exit(1);
}

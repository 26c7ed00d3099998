// @ts-check
// l2_backend.mjs -- the MI355X forward pass as an ES module for a Node / Bun host.
//
// What a maintainer of wizzard0/llama2.ts imports to put the GPU behind the call at llama2.ts:468 (INTEGRATION.md
// shows the four changed lines of the reference's own file): the names it exports are the reference's --
// readWeights, newRunState, transformer -- with the same arguments plus the backend handle, so the sampling
// loop (llama2.ts:470-508) does not change.  Weights, activations and the KV cache live in HBM; `state.logits`
// is the only host-visible field.  Also exported: the device-resident loops (decodeGreedy / decodeSample), the
// batched prompt ingestion (prefill) and the native checkpoint loader -- SURVEY.md 8(f) rows, all opt-in.
//
// Plain ECMAScript with JSDoc types (runs unchanged on Node >= 12 and Bun; the image has no TypeScript compiler).
// There is NO CPU path here: without the addon, the library or a gfx950 device every call throws.
import * as fs from "fs";
import * as path from "path";
import { createRequire } from "module";
import { fileURLToPath } from "url";

const here = path.dirname(fileURLToPath(import.meta.url));

/** Checkpoint order of the llama2.c-v0 file = tensor kinds of include/llama2_hip.h.  [name, per layer?, shape(cfg)] */
const TENSORS = [
  ["token_embedding_table", false, (c) => [c.vocab_size, c.dim]],
  ["rms_att_weight", true, (c) => [c.dim]],
  ["wq", true, (c) => [c.dim, c.dim]],
  ["wk", true, (c) => [c.dim, c.dim]],
  ["wv", true, (c) => [c.dim, c.dim]],
  ["wo", true, (c) => [c.dim, c.dim]],
  ["rms_ffn_weight", true, (c) => [c.dim]],
  ["w1", true, (c) => [c.hidden_dim, c.dim]],
  ["w2", true, (c) => [c.dim, c.hidden_dim]],
  ["w3", true, (c) => [c.hidden_dim, c.dim]],
  ["rms_final_weight", false, (c) => [c.dim]],
  ["freq_cis_real", false, (c) => [c.seq_len, c.head_size / 2]],
  ["freq_cis_imag", false, (c) => [c.seq_len, c.head_size / 2]],
  ["wcls", false, (c) => [c.vocab_size, c.dim]],
];

/** Open the N-API addon and, through it, libllama2hip.so.  @returns {any} */
export function openBackend() {
  const addonPath = process.env.L2_NAPI_PATH || path.join(here, "l2_napi.node");
  let addon;
  try {
    addon = createRequire(import.meta.url)(addonPath);
  } catch (e) {
    throw new Error("cannot load the N-API addon " + addonPath + " (run __graft_entry__.build()): " + e.message);
  }
  addon.open(process.env.L2_LIB_PATH || path.join(here, "..", "lib", "libllama2hip.so"));
  return addon;
}

/** The 7 header ints as the reference's Config (llama2.ts:69-93).  @param {Int32Array} header */
export function configOf(header) {
  const [dim, hidden_dim, n_layers, n_heads, n_kv_heads, vocab, seq_len] = Array.from(header);
  return { header: Int32Array.from(header), dim, hidden_dim, n_layers, n_heads, n_kv_heads, vocab_size: Math.abs(vocab), seq_len,
    shared_weights: vocab > 0, head_size: dim / n_heads };
}

/** Read `count` floats at byte `offset` of an open file into a fresh Float32Array (whole reads, 1 GiB at a time). */
function floatsAt(fd, offset, count) {
  const bytes = Buffer.alloc(count * 4);
  for (let done = 0; done < bytes.length;) {
    const n = fs.readSync(fd, bytes, done, Math.min(bytes.length - done, 1 << 30), offset + done);
    if (n <= 0) throw new Error("checkpoint truncated at byte " + (offset + done));
    done += n;
  }
  return new Float32Array(bytes.buffer, bytes.byteOffset, count);   // byteOffset honoured by the addon
}

/** readWeights (llama2.ts:112-129) with the GPU as destination: every Float32Array is uploaded the moment it has been
 *  read and then dropped, so the host never holds more than one tensor of a 27 GB checkpoint.
 *  @returns {{ctx:any, uploaded:string[]}} */
export function readWeights(config, fd, be, ctx) {
  let offset = 28;
  const uploaded = [];
  TENSORS.forEach(([name, layered, shape], kind) => {
    if (name == "wcls" && config.shared_weights) return;          // the library aliases the embedding table (llama2.ts:127)
    const count = shape(config).reduce((a, b) => a * b, 1);
    for (let l = 0; l < (layered ? config.n_layers : 1); ++l, offset += count * 4) be.upload(ctx, kind, layered ? l : -1, floatsAt(fd, offset, count));
    uploaded.push(name);
  });
  return { ctx, uploaded };
}

/** newRunState (llama2.ts:147-163): `logits` is what the sampling loop reads; everything else stays in HBM. */
export function newRunState(config) {
  return { logits: new Float32Array(config.vocab_size), indices: new Array(config.vocab_size) };
}

/** transformer(token, pos, p, s, w) (llama2.ts:205-303, call site :468): fills s.logits. */
export function transformer(token, pos, p, s, w, be) {
  be.forward(w.ctx, token, pos, s.logits);
}

/** Open a checkpoint: header, context, weights (per-array upload, or the library's own pinned-buffer streaming loader).
 *  @param {string} file @param {any} be @param {{device?:number, nativeLoader?:boolean}} [opt] */
export function loadModel(file, be, opt = {}) {
  const device = opt.device || 0;
  if (opt.nativeLoader) {
    const r = be.loadCheckpoint(file, device);
    const config = configOf(r.header);
    return { config, weights: { ctx: r.ctx, uploaded: ["(native loader)"] }, state: newRunState(config) };
  }
  const fd = fs.openSync(file, "r");
  try {
    const hb = Buffer.alloc(28);
    if (fs.readSync(fd, hb, 0, 28, 0) != 28) throw new Error("checkpoint shorter than its header");
    const config = configOf(new Int32Array(hb.buffer, hb.byteOffset, 7));
    const ctx = be.create(config.header, device);
    return { config, weights: readWeights(config, fd, be, ctx), state: newRunState(config) };
  } finally {
    fs.closeSync(fd);
  }
}

// l2_backend.ts -- the MI355X forward pass as a TypeScript module for a Node (>= 22.6, --experimental-strip-types) / Bun host.
//
// The TypeScript twin of l2_backend.mjs: the same statements with type annotations in ERASABLE syntax only (no enums, no
// namespaces, no parameter properties), so stripping the types gives l2_backend.mjs back token for token --
// tests/test_host_cpu.py::test_typescript_twin_erases_to_the_module checks exactly that with the type stripper the reference
// itself ships.  wizzard0/llama2.ts is TypeScript under Node / Bun: this is the file its maintainer imports.
//
// What a maintainer of wizzard0/llama2.ts imports to put the GPU behind the call at llama2.ts:468 (INTEGRATION.md
// shows the four changed lines of the reference's own file): the names it exports are the reference's --
// readWeights, newRunState, transformer -- with the same arguments plus the backend handle, so the sampling
// loop (llama2.ts:470-508) does not change.  Weights, activations and the KV cache live in HBM; `state.logits`
// is the only host-visible field.  Also exported: the device-resident loops (decodeGreedy / decodeSample), the
// batched prompt ingestion (prefill) and the native checkpoint loader -- SURVEY.md 8(f) rows, all opt-in.
//
// (l2_backend.mjs is the same module for runtimes without type stripping: Node >= 12.)
// There is NO CPU path here: without the addon, the library or a gfx950 device every call throws.
import * as fs from "fs";
import * as path from "path";
import { createRequire } from "module";
import { fileURLToPath } from "url";

const here = path.dirname(fileURLToPath(import.meta.url));

/** Config of the reference (llama2.ts:69-79) plus the raw header. */
export interface Config { header: Int32Array; dim: number; hidden_dim: number; n_layers: number; n_heads: number; n_kv_heads: number;
  vocab_size: number; seq_len: number; shared_weights: boolean; head_size: number }
/** The addon's exports (l2_napi.d.ts). */
export type Backend = typeof import("./l2_napi");
export type L2Handle = object;
export interface Weights { ctx: L2Handle; uploaded: string[] }
export interface RunState { logits: Float32Array; indices: any[] }
type TensorRow = [string, boolean, (c: Config) => number[]];

/** Checkpoint order of the llama2.c-v0 file = tensor kinds of include/llama2_hip.h.  [name, per layer?, shape(cfg)] */
const TENSORS: TensorRow[] = [
  ["token_embedding_table", false, (c: Config) => [c.vocab_size, c.dim]],
  ["rms_att_weight", true, (c: Config) => [c.dim]],
  ["wq", true, (c: Config) => [c.dim, c.dim]],
  ["wk", true, (c: Config) => [c.dim, c.dim]],
  ["wv", true, (c: Config) => [c.dim, c.dim]],
  ["wo", true, (c: Config) => [c.dim, c.dim]],
  ["rms_ffn_weight", true, (c: Config) => [c.dim]],
  ["w1", true, (c: Config) => [c.hidden_dim, c.dim]],
  ["w2", true, (c: Config) => [c.dim, c.hidden_dim]],
  ["w3", true, (c: Config) => [c.hidden_dim, c.dim]],
  ["rms_final_weight", false, (c: Config) => [c.dim]],
  ["freq_cis_real", false, (c: Config) => [c.seq_len, c.head_size / 2]],
  ["freq_cis_imag", false, (c: Config) => [c.seq_len, c.head_size / 2]],
  ["wcls", false, (c: Config) => [c.vocab_size, c.dim]],
];

/** Open the N-API addon and, through it, libllama2hip.so. */
export function openBackend(): Backend {
  const addonPath = process.env.L2_NAPI_PATH || path.join(here, "l2_napi.node");
  let addon: Backend;
  try {
    addon = createRequire(import.meta.url)(addonPath);
  } catch (e: any) {
    throw new Error("cannot load the N-API addon " + addonPath + " (run __graft_entry__.build()): " + e.message);
  }
  addon.open(process.env.L2_LIB_PATH || path.join(here, "..", "lib", "libllama2hip.so"));
  return addon;
}

/** The 7 header ints as the reference's Config (llama2.ts:69-93). */
export function configOf(header: Int32Array): Config {
  const [dim, hidden_dim, n_layers, n_heads, n_kv_heads, vocab, seq_len] = Array.from(header);
  return { header: Int32Array.from(header), dim, hidden_dim, n_layers, n_heads, n_kv_heads, vocab_size: Math.abs(vocab), seq_len,
    shared_weights: vocab > 0, head_size: dim / n_heads };
}

/** Bytes one transformer() call at position `pos` has to move at the least (SURVEY.md 8(d)): every weight once, the norms, the
 *  embedding row, the KV rows read and written, the RoPE row, the logits. */
export function algorithmicBytesPerToken(c: Config, pos: number): number {
  const d = c.dim, h = c.hidden_dim, L = c.n_layers, V = c.vocab_size;
  return 4 * (L * (4 * d * d + 3 * d * h + 2 * d) + d + V * d + d + L * (2 * (pos + 1) * d + 2 * d) + c.head_size) + 4 * V;
}

/** Read `count` floats at byte `offset` of an open file into a fresh Float32Array (whole reads, 1 GiB at a time). */
function floatsAt(fd: number, offset: number, count: number): Float32Array {
  const bytes = Buffer.alloc(count * 4);
  for (let done = 0; done < bytes.length;) {
    const n = fs.readSync(fd, bytes, done, Math.min(bytes.length - done, 1 << 30), offset + done);
    if (n <= 0) throw new Error("checkpoint truncated at byte " + (offset + done));
    done += n;
  }
  return new Float32Array(bytes.buffer, bytes.byteOffset, count);   // byteOffset honoured by the addon
}

/** readWeights (llama2.ts:112-129) with the GPU as destination: every Float32Array is uploaded the moment it has been
 *  read and then dropped, so the host never holds more than one tensor of a 27 GB checkpoint. */
export function readWeights(config: Config, fd: number, be: Backend, ctx: L2Handle): Weights {
  let offset = 28;
  const uploaded: string[] = [];
  TENSORS.forEach(([name, layered, shape], kind) => {
    if (name == "wcls" && config.shared_weights) return;          // the library aliases the embedding table (llama2.ts:127)
    const count = shape(config).reduce((a: number, b: number) => a * b, 1);
    for (let l = 0; l < (layered ? config.n_layers : 1); ++l, offset += count * 4) be.upload(ctx, kind, layered ? l : -1, floatsAt(fd, offset, count));
    uploaded.push(name);
  });
  return { ctx, uploaded };
}

/** newRunState (llama2.ts:147-163): `logits` is what the sampling loop reads; everything else stays in HBM. */
export function newRunState(config: Config): RunState {
  return { logits: new Float32Array(config.vocab_size), indices: new Array(config.vocab_size) };
}

/** transformer(token, pos, p, s, w) (llama2.ts:205-303, call site :468): fills s.logits. */
export function transformer(token: number, pos: number, p: Config, s: RunState, w: Weights, be: Backend): void {
  be.forward(w.ctx, token, pos, s.logits);
}

/** Open a checkpoint: header, context, weights (per-array upload, or the library's own pinned-buffer streaming loader). */
export function loadModel(file: string, be: Backend, opt: { device?: number; nativeLoader?: boolean } = {}): { config: Config; weights: Weights; state: RunState } {
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

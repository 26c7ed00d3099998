// l2_napi.d.ts -- TypeScript declarations of the N-API addon (l2_napi.cc) over libllama2hip.so, for hosts that stay TypeScript
// (wizzard0/llama2.ts under Node / Bun; INTEGRATION.md section 1 shows the four edits to the reference's own file).
// Every function throws an Error carrying l2_last_error() when the library returns a negative code; there is no CPU fallback.

/** Opaque handle of one l2_ctx (weights + RunState of one model on one MI355X). */
export type L2Handle = object;

/** dlopen libllama2hip.so; throws if it is missing. */
export function open(libPath: string): true;
/** readConfig + newRunState (llama2.ts:80-93, 147-163): the 7 header ints verbatim (sign of vocab_size kept). */
export function create(header: Int32Array, device: number): L2Handle;
/** Native checkpoint reader (llama2.c v0 and version-1 files): streams the tensors to HBM, returns the context and its header. */
export function loadCheckpoint(path: string, device: number): { ctx: L2Handle; header: Int32Array };
export function destroy(ctx: L2Handle): void;
/** One Float32Array of readWeights (llama2.ts:112-129). kind = index in checkpoint order (0 token_embedding_table ... 13 wcls);
 *  layer = index into the Float32Array[] of a per-layer tensor, -1 otherwise.  The view's byteOffset is honoured; copied before return. */
export function upload(ctx: L2Handle, kind: number, layer: number, array: Float32Array): void;
/** Deterministic synthetic weights (tests, benchmarks). */
export function synthFill(ctx: L2Handle, seed: number): void;
/** transformer(token, pos, ...) at llama2.ts:468: blocking; fills `logits` (>= vocab_size floats) when given. */
export function forward(ctx: L2Handle, token: number, pos: number, logits: Float32Array | null): void;
/** Teacher-forced prompt positions pos0 .. pos0 + tokens.length - 1 in one batched call (llama2.ts:471-473); logits of the last one. */
export function prefill(ctx: L2Handle, tokens: Int32Array, pos0: number, logits: Float32Array | null): void;
/** The pinned host buffer the classifier kernel writes (zero-copy RunState.logits); valid until destroy(). */
export function logitsBuffer(ctx: L2Handle, vocabSize: number): ArrayBuffer;
/** The -t 0 loop (llama2.ts:465-508) kept on the device: `steps` tokens from (firstToken, pos0). */
export function decodeGreedy(ctx: L2Handle, firstToken: number, pos0: number, steps: number): Int32Array;
/** The sampled branch (llama2.ts:480-493) on the device; `rng` = [low, high] halves of the reference's 64-bit rng_seed, updated in place. */
export function decodeSample(ctx: L2Handle, firstToken: number, pos0: number, steps: number, temperature: number, topp: number, rng: Uint32Array): Int32Array;
/** Parity reads of RunState fields (L2_S_* of include/llama2_hip.h); fields only transformer() reads need setOption(ctx, 3, 1) first. */
export function readState(ctx: L2Handle, which: number, layer: number, out: Float32Array): void;
/** Read back `out.length` floats at float offset `offset` of a weight tensor as l2_upload stored it (tests). */
export function readTensor(ctx: L2Handle, kind: number, layer: number, offset: number, out: Float32Array): void;
/** L2_OPT_* of include/llama2_hip.h: 1 exact attention, 2 use hipGraph, 3 keep state. */
export function setOption(ctx: L2Handle, key: number, value: number): void;
/** Read an option or a read-only counter: 4 MiB of repacked weight copies, 5 MiB of all weights on the device, 6 tokens decodeSample has picked,
 *  7 of those the ones picked by the reference's loop run as written (a running sum within the proven margin of its threshold). */
export function getOption(ctx: L2Handle, key: number): number;
export function deviceCount(): number;

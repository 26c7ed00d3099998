/*
 * llama2_hip.h -- C ABI of libllama2hip.so: the MI355X (gfx950) forward pass that drops in at the
 * single call `transformer(token, pos, config, state, weights)` of wizzard0/llama2.ts
 * (/root/reference/llama2.ts:468, body :205-303).
 *
 * The reference has no FFI/plugin interface (every function is module-private, llama2.ts:526 just
 * calls main()), so these entry points are what an N-API / bun:ffi / ctypes binding for that call
 * would bind; INTEGRATION.md shows the reference-side stub.
 *
 * Conventions: plain pointers and sizes, no C++ or torch types; every function returns 0 on success
 * or a negative L2_E_* code (l2_last_error() gives the text); nothing throws across the ABI; a
 * context is bound to one device, is not thread-safe, and distinct contexts are independent.  Host
 * pointers are only read/written during the call and never retained.
 */
#ifndef LLAMA2_HIP_H
#define LLAMA2_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3 (round 5): l2_bench_tokens, option keys 5-7, L2_TP_SOLO_ID / l2_tp_mode 5 and the L2_TP_FENCED switch joined the surface after 2 -- a
 * binding built against it refuses an older library at open() (l2_abi_version), not at the first call that is missing.
 * 4 (round 5): option key L2_OPT_AQL_QUEUE
 * 5 (round 6): option keys L2_OPT_PREFILL_F32_MFMA, L2_OPT_CHECK_POS; l2_dispatch_reason */
#define L2_ABI_VERSION 5

enum {
  L2_OK = 0,
  L2_E_ARG = -1,     /* bad argument (null pointer, index out of range, wrong size) */
  L2_E_CONFIG = -2,  /* unsupported header (dim % n_heads, odd head_size, non-positive sizes) */
  L2_E_HIP = -3,     /* a HIP runtime call failed */
  L2_E_STATE = -4,   /* call order violated (forward before all tensors uploaded, a kept-state read without L2_OPT_KEEP_STATE) */
  L2_E_NOGPU = -5,   /* no gfx950 device visible */
  L2_E_COMM = -6     /* RCCL failure (tensor-parallel contexts) */
};

/* Tensor kinds = fields of the reference's TransformerWeights in checkpoint order
 * (readWeights, llama2.ts:112-129). */
enum {
  L2_T_TOKEN_EMBEDDING = 0, /* token_embedding_table (V,d)   llama2.ts:114 */
  L2_T_RMS_ATT = 1,         /* rms_att_weight[l] (d)         :115 */
  L2_T_WQ = 2,              /* wq[l] (d,d)                   :116 */
  L2_T_WK = 3,              /* wk[l] (d,d)                   :117 */
  L2_T_WV = 4,              /* wv[l] (d,d)                   :118 */
  L2_T_WO = 5,              /* wo[l] (d,d)                   :119 */
  L2_T_RMS_FFN = 6,         /* rms_ffn_weight[l] (d)         :120 */
  L2_T_W1 = 7,              /* w1[l] (h,d)                   :121 */
  L2_T_W2 = 8,              /* w2[l] (d,h)                   :122 */
  L2_T_W3 = 9,              /* w3[l] (h,d)                   :123 */
  L2_T_RMS_FINAL = 10,      /* rms_final_weight (d)          :124 */
  L2_T_FREQ_REAL = 11,      /* freq_cis_real (S,hs/2)        :125 */
  L2_T_FREQ_IMAG = 12,      /* freq_cis_imag (S,hs/2)        :126 */
  L2_T_WCLS = 13,           /* wcls (V,d), only when the header's vocab_size < 0  :127 */
  L2_T_COUNT = 14
};

/* RunState fields (llama2.ts:131-146) readable with l2_read_state. */
enum {
  L2_S_X = 0, L2_S_XB = 1, L2_S_XB2 = 2, L2_S_HB = 3, L2_S_HB2 = 4, L2_S_Q = 5, L2_S_K = 6, L2_S_V = 7,
  L2_S_ATT = 8, L2_S_LOGITS = 9, L2_S_KEY_CACHE = 10, L2_S_VALUE_CACHE = 11
};

/* l2_set_option keys */
enum {
  L2_OPT_EXACT_ATTENTION = 1, /* 1: value-accumulate rounds to fp32 at every timestep in t order, exactly as
                                 llama2.ts:260-265 does (the reference's own rounding points in that loop; slower; > 99.9 % of logits
                                 come out bit-identical, the rest within 1 ulp: the tree-ordered fp64 sums elsewhere remain); 0 (default):
                                 fp64 partial sums, one rounding */
  L2_OPT_USE_GRAPH = 2,       /* 1 (default): the step of a context length level is recorded once and replayed per token (as packets on the
                                 library's own queue, L2_OPT_AQL_QUEUE, or as a captured hipGraph); 0: eager launches */
  L2_OPT_KEEP_STATE = 3,      /* 1: the RunState fields that only transformer() itself reads (llama2.ts:131-146: att, k, v, hb2, xb2,
                                 the xb of the FFN half, the final-normed x) are also written out for l2_read_state (parity
                                 tests); 0 (default): they stay on chip and reading them AFTER a forward returns L2_E_STATE -- q, hb, logits and
                                 the KV caches are always there */
  L2_OPT_PACKED_MIB = 4,      /* read-only (l2_get_option): MiB of device memory held by the repacked copies of the matrices the
                                 streaming kernels read; 0 before the first step and for models that need none */
  L2_OPT_WEIGHT_MIB = 5,      /* read-only: MiB of device memory held by ALL weights right now (row-major tensors + repacked copies).  After
                                 the first step a repacked matrix exists once: its row-major tensor has been given back */
  L2_OPT_SAMPLED_TOKENS = 6,  /* read-only: tokens l2_decode_sample has picked on this context with temperature != 0 (saturates at INT_MAX) */
  L2_OPT_SAMPLED_SERIAL = 7,  /* read-only: of those, the tokens whose running sums came within the proven margin of the threshold and were
                                 therefore picked by the reference's loop run as written (csrc/sampler_margin.hip.h); the others by the
                                 margin rule */
  L2_OPT_AQL_QUEUE = 8,       /* 1 (default): the recorded step of l2_decode_greedy, l2_decode_sample AND of the blocking l2_forward is submitted as
                                 hand-written AQL packets on a queue of the library's own (csrc/aql_queue.h: barrier bit, agent-scope release, no
                                 acquire between the launches of a token); 0: a replayed hipGraph per token.  Reading it tells what the context
                                 uses now: 1 the queue, 0 hipGraphs / eager launches (switched off, L2_USE_GRAPH=0, RCCL collectives in the step,
                                 a profiler's tool library in the process, or the queue was given up: l2_dispatch_reason() says which) */
  L2_OPT_PREFILL_F32_MFMA = 9, /* 1: l2_prefill's register-blocked GEMMs accumulate in fp32 on v_mfma_f32_16x16x4_f32 -- a k-ordered fmaf chain per
                                 element instead of the reference's fp64 accumulate (llama2.ts:196-203): faster prompt ingestion, logits within
                                 1e-4 on the fixtures, NOT bit-level parity with the reference (measured exactness);
                                 0 (default): fp64 MFMA, the reference's arithmetic.  Decode is never affected */
  L2_OPT_CHECK_POS = 10       /* 1 (or L2_CHECK_POS=1 in the environment at creation): l2_forward / l2_prefill refuse (L2_E_STATE) a position that
                                 neither restarts at 0 nor continues the sequence -- the reference's loop feeds pos = 0, 1, 2, ... (llama2.ts:464,
                                 496) and attention reads whatever rows 0 .. pos - 1 the cache holds; 0 (default): any position is accepted */
};

typedef struct l2_ctx l2_ctx;

int l2_abi_version(void);
int l2_device_count(void);                 /* number of visible HIP devices, or a negative code */
const char* l2_last_error(void);           /* thread-local text of the last failure */

/* Replaces readConfig + newRunState (llama2.ts:80-93, 147-163): `cfg` is the 7 header int32 verbatim
 * (sign of vocab_size kept).  Allocates weights, activations and the [L][S][d] KV caches in HBM. */
int l2_create(const int32_t cfg[7], int device, l2_ctx** out);
void l2_destroy(l2_ctx* ctx);

/* SURVEY.md 8(f4) -- checkpoints newer than the reference understands.  The reference parses n_kv_heads and ignores
 * it (llama2.ts:86, 117-118: wk / wv are always (d, d)) and reads the RoPE tables from the file (:125-126); l2_create
 * does the same.  l2_create_ex honours, on request:
 *   L2_F_GQA            n_kv_heads < n_heads: wk / wv are (n_kv_heads * head_size, d) per layer, the KV caches hold
 *                       n_kv_heads * head_size floats per position and query head h attends over cache head
 *                       h / (n_heads / n_kv_heads) -- grouped-query attention as llama2.c's run.c defines it;
 *   L2_F_GENERATE_ROPE  no freq_cis tensors will be uploaded: the tables are computed at creation the way run.c
 *                       computes them per position (fp32 powf / cosf / sinf).
 * Neither is pinned by the reference (it cannot run such checkpoints); the oracle's restatement is our own. */
enum { L2_F_GQA = 1, L2_F_GENERATE_ROPE = 2 };
int l2_create_ex(const int32_t cfg[7], int device, unsigned flags, l2_ctx** out);

/* Tensor-parallel context: rank `tp_rank` of `tp_size` owns heads / FFN rows
 * [rank*H/G, ...).  `nccl_id` is the 128-byte ncclUniqueId produced by l2_tp_unique_id on rank 0 and
 * handed to the others by the caller (e.g. over torch.distributed).  tp_size 1 == l2_create. */
int l2_tp_unique_id(void* id_out_128);
/* How this context's tensor-parallel step runs: 0 not tensor parallel, 1 eager launches with RCCL collectives (this RCCL
 * refused stream capture, or L2_USE_GRAPH=0), 2 one hipGraph per token with the RCCL collectives captured in it, 3 one hipGraph
 * per token with the one-shot peer-to-peer all-reduce, 4 loopback test group (L2_TEST_HOOKS), 5 shard-timing context (L2_TP_SOLO_ID). */
int l2_tp_mode(l2_ctx* ctx);
int l2_create_tp(const int32_t cfg[7], int device, int tp_rank, int tp_size, const void* nccl_id, l2_ctx** out);
/* Measurement only: a 128-byte id that starts with this text creates ONE rank of a tp_size group with no peers -- its shard of the
 * weights and of the step, the exchange kernels running against its own inbox (every wait satisfied at once).  It times a rank's
 * share of the step without a multi-GPU node (bench.py `tp_predicted`); what it decodes is meaningless (l2_tp_mode 5). */
#define L2_TP_SOLO_ID "L2-SOLO-SHARD-TIMING"

/* Replaces the hand-over of one Float32Array of TransformerWeights (readWeights, llama2.ts:112-129):
 * call once per array right after FileHandleReader.getF32Array returns it (llama2.ts:51-59).
 * `layer` is the index into the Float32Array[] for per-layer tensors, -1 otherwise.  The bytes are
 * copied to HBM before the call returns. */
int l2_upload(l2_ctx* ctx, int tensor_kind, int layer, const float* host, size_t n_floats);

/* Next row of SURVEY.md 8(f2): the checkpoint load path (FileHandleReader + readWeights, llama2.ts:44-68,
 * 112-129) done natively: reads the 28-byte header of the llama2.c-v0 file at `path`, creates the context and
 * streams every tensor to HBM through two pinned staging buffers (file reads overlap the host-to-device
 * copies), so host memory never holds more than 2 x 64 MiB of a 27 GB checkpoint.  Tensor-parallel ranks pass
 * their rank / size / id and receive only their slices; tp_size 1 ignores them.  `bytes_read` (optional)
 * returns the file bytes consumed.  Equivalent to l2_create + one l2_upload per Float32Array. */
int l2_load_checkpoint(const char* path, int device, int tp_rank, int tp_size, const void* nccl_id, l2_ctx** out,
                       uint64_t* bytes_read);
/* (l2_load_checkpoint also reads a llama2.c "version 1" file -- magic "ak42", 256-byte header, norms first, no freq_cis; fp32:
 * detected by its magic and loaded with L2_F_GQA | L2_F_GENERATE_ROPE.  A file without the magic is the v0 layout above.) */

/* The 7 header ints of a context (what readConfig parsed). */
int l2_get_header(l2_ctx* ctx, int32_t cfg_out[7]);

/* Fill every tensor on the device with the repo's deterministic synthetic generator (same values as
 * oracle/llama2_oracle.c:orc_synth_tensor); used by bench.py for shapes too large to ship. */
int l2_synth_fill(l2_ctx* ctx, uint32_t seed);
/* Read back part of a weight tensor (tests of l2_upload / l2_synth_fill). */
int l2_read_tensor(l2_ctx* ctx, int tensor_kind, int layer, size_t offset, float* out, size_t n_floats);

/* Replaces transformer(token, pos, p, s, w) (llama2.ts:205-303, call site :468).  Blocking: when it
 * returns, logits_out[0..V) holds state.logits for this position and the device KV cache holds rows
 * 0..pos.  pos must be 0 <= pos < seq_len; token in [0, V).  The reference's loop feeds pos = 0, 1, 2, ... (llama2.ts:464,
 * 496); any pos is accepted here -- attention then reads whatever rows 0..pos-1 the cache holds from earlier calls
 * (L2_OPT_CHECK_POS makes a position that skips ahead an error).  logits_out may be NULL (logits stay readable through l2_logits_host / l2_read_state). */
int l2_forward(l2_ctx* ctx, int token, int pos, float* logits_out);
/* Pinned host buffer (V floats) that l2_forward fills; a binding may wrap it as RunState.logits to
 * skip the copy into logits_out. */
float* l2_logits_host(l2_ctx* ctx);

/* Next row of SURVEY.md 8(f1): greedy decode (`-t 0`, argmax llama2.ts:364-366: first maximum) kept on
 * the device -- the loop llama2.ts:465-508 with temperature 0 and no prompt, minus printing.  Feeds
 * `first_token` at pos0, then each argmax; writes the `steps` chosen tokens to tokens_out[0..steps).
 * Does not stop at BOS (the caller truncates, llama2.ts:499). */
int l2_decode_greedy(l2_ctx* ctx, int first_token, int pos0, int steps, int32_t* tokens_out);

/* Rest of SURVEY.md 8(f1): the sampled branch of the loop (llama2.ts:480-493) kept on the device -- temperature
 * scaling and softmax in place of state.logits (:481-485), `sample` (:368-376) or, when 0 < topp < 1, `sample_topp`
 * (:378-394: stable descending sort, the element that crosses topp is never returned, fall-through returns id 0),
 * driven by the reference's xorshift* generator (:348-360).  `*rng_state` is the 64-bit state the reference keeps in
 * the BigInt `rng_seed` (what `-s` parsed, non-zero); it is advanced by one draw per sampled token and written back.
 * The SAME token ids come out for the same seed: what the reference's loops return is an INDEX -- the first element whose
 * sequentially accumulated running sum passes the threshold -- and the device decides every comparison those loops make from
 * tree sums plus a proven margin (csrc/margin_rule.h); a token whose running sum comes within that margin of its threshold
 * (about 4 in 100 000) is picked by the reference's loop run as written, element by element (L2_OPT_SAMPLED_SERIAL counts
 * them).  temperature == 0 is l2_decode_greedy (no draw).  Like l2_decode_greedy it does not stop at BOS. */
int l2_decode_sample(l2_ctx* ctx, int first_token, int pos0, int steps, double temperature, double topp,
                     uint64_t* rng_state, int32_t* tokens_out);

/* Diagnostic for the sampler's core: sums_out[i] = S_i with S_i = fl(S_{i-1} + values[i]) in fp64 (the reference's
 * `cumProb += x[i]` / `sum += x[i]` loops), computed on the device by the exact parallel algorithm of sampler.hip.
 * `values` must be finite and >= 0.  Tests compare it with the serial loop on adversarial vectors. */
int l2_debug_running_sums(int device, const float* values, size_t n, double* sums_out);

/* Next row of SURVEY.md 8(f3): prompt ingestion.  The reference runs one transformer() per prompt token and
 * ignores the logits (llama2.ts:471-473); this feeds `n_tokens` tokens at positions pos0 .. pos0+n_tokens-1 in
 * chunks of up to 64 (one, two or four 16-token MFMA tiles) that share every weight read (fp64 MFMA GEMMs), leaves the KV cache exactly as the n_tokens
 * separate calls would, and returns the logits of the LAST position in logits_out (may be NULL).  Shapes whose
 * dim / hidden_dim are not multiples of 16 fall back to n_tokens l2_forward calls. */
int l2_prefill(l2_ctx* ctx, const int32_t* tokens, int n_tokens, int pos0, float* logits_out);

/* Copy a RunState buffer to the host (parity tests).  For per-layer caches `layer` selects the
 * [S][d] slab (-1: all layers).  After a forward, X holds the final-normed x as in llama2.ts:299.
 * X, XB2, HB2, K, V and ATT need L2_OPT_KEEP_STATE (set before the forward), else L2_E_STATE. */
int l2_read_state(l2_ctx* ctx, int which, int layer, float* out, size_t n_floats);

int l2_set_option(l2_ctx* ctx, int key, int value);
int l2_get_option(l2_ctx* ctx, int key, int* value);
/* Why the library's own queue (L2_OPT_AQL_QUEUE) is not in use on this context, or "" when it is / has not been tried yet.  The text
 * lives in the context and is valid until the next call on it (l2_last_error is for failures only and is left alone). */
const char* l2_dispatch_reason(l2_ctx* ctx);

/* Measurement hooks (bench.py): HIP events on the context's own stream. */
int l2_timer_start(l2_ctx* ctx);
int l2_timer_stop(l2_ctx* ctx, float* elapsed_ms);   /* synchronises */
/* Launch only the dominant kernel (the weight-streaming GEMV of one matrix kind of one layer)
 * `iters` times back to back and return the average device time per launch.  Scratch use of the RunState
 * buffers: the activations and cache row `pos` are clobbered, call it outside a decode. */
int l2_bench_gemv(l2_ctx* ctx, int tensor_kind, int layer, int iters, float* avg_ms);
/* The same kernel timed in situ: `steps` greedy decode steps (eager launches) with a HIP start / stop event pair attached
 * to every dispatch of the rmsnorm + w1/w3 + SwiGLU kernel on the library's stream (hipExtLaunchKernelGGL: the events
 * bracket the kernel's execution, as a kernel trace does); mean duration in microseconds. */
int l2_bench_dominant_in_situ(l2_ctx* ctx, int first_token, int pos0, int steps, float* avg_us, int* launches);
/* `steps` forwards (greedy feed, device-resident), timed: total ms from the first submission to completion (the library's own queue: host
 * clock, first doorbell -> completion signal, spinning; hipGraph replays: events on the library's stream). */
int l2_bench_decode(l2_ctx* ctx, int first_token, int pos0, int steps, float* total_ms);
/* The first `n` tokens the last device-resident run (l2_bench_decode, l2_decode_greedy, l2_decode_sample) chose: lets a benchmark
 * check the very run it timed against the reference's tokens (llama2.ts:476-478 picks them on the host). */
int l2_bench_tokens(l2_ctx* ctx, int32_t* tokens_out, int n);

#ifdef __cplusplus
}
#endif
#endif

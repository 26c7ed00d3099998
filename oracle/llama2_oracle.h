/*
 * llama2_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, single thread) of the hot path of wizzard0/llama2.ts:
 * the per-token `transformer()` forward (/root/reference/llama2.ts:205-303) and the
 * helpers it calls (accum :168, rmsnorm :172, softmax :181, matmul :196), plus the
 * llama2.c-v0 checkpoint layout (readConfig :80-93, readWeights :112-129) and the
 * RunState buffers (newRunState :147-163); and, for the rows next to the path, the RNG and
 * samplers downstream of it (random_u32 / random_f32 :348-360, sample :368-376, sample_topp :378-394,
 * the temperature branch :476-493).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link, load or
 * call this.  The product (llama2.ts_amd/) never does: it fails loudly without the HIP library.
 *
 * Parity pin: the reference ships no tests and no golden vectors for this path (SURVEY.md
 * section 4), so this restatement is pinned against outputs of the reference itself, executed
 * in the build container under Node 12 (recipe: oracle/make_goldens.py, fixtures:
 * tests/golden/).  tests/test_oracle_golden.py checks every fixture bit-for-bit (logits of every position of
 * every run; sampled token ids of the reference's -t / -p / -s runs).
 */
#ifndef LLAMA2_ORACLE_H
#define LLAMA2_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Tensor kinds, in llama2.c-v0 file order (llama2.ts:114-127). Shared with include/llama2_hip.h. */
enum {
  ORC_T_TOKEN_EMBEDDING = 0, /* (V, d)        llama2.ts:114 */
  ORC_T_RMS_ATT = 1,         /* (L, d)        :115 */
  ORC_T_WQ = 2,              /* (L, d, d)     :116 */
  ORC_T_WK = 3,              /* (L, d, d)     :117 */
  ORC_T_WV = 4,              /* (L, d, d)     :118 */
  ORC_T_WO = 5,              /* (L, d, d)     :119 */
  ORC_T_RMS_FFN = 6,         /* (L, d)        :120 */
  ORC_T_W1 = 7,              /* (L, h, d)     :121 */
  ORC_T_W2 = 8,              /* (L, d, h)     :122 */
  ORC_T_W3 = 9,              /* (L, h, d)     :123 */
  ORC_T_RMS_FINAL = 10,      /* (d)           :124 */
  ORC_T_FREQ_REAL = 11,      /* (S, hs/2)     :125 */
  ORC_T_FREQ_IMAG = 12,      /* (S, hs/2)     :126 */
  ORC_T_WCLS = 13,           /* (V, d) only when header vocab_size < 0   :127 */
  ORC_T_COUNT = 14
};

/* RunState buffers readable through orc_state() (llama2.ts:131-146). */
enum {
  ORC_S_X = 0, ORC_S_XB = 1, ORC_S_XB2 = 2, ORC_S_HB = 3, ORC_S_HB2 = 4,
  ORC_S_Q = 5, ORC_S_K = 6, ORC_S_V = 7, ORC_S_ATT = 8, ORC_S_LOGITS = 9,
  ORC_S_KEY_CACHE = 10, ORC_S_VALUE_CACHE = 11
};

typedef struct orc_config {
  int dim, hidden_dim, n_layers, n_heads, n_kv_heads, vocab_size, seq_len; /* llama2.ts:82-89 (vocab_size = abs) */
  int shared_weights;                                                     /* :90 */
  int head_size;                                                          /* :91 */
} orc_config;

typedef struct orc_model orc_model;

/* readConfig (llama2.ts:80-93): 7 little-endian int32, sign of vocab_size = shared flag. */
/* grouped-query switch (parity unpinned by the reference): see llama2_oracle.c */
void orc_set_gqa(int on);
void orc_read_config(const int32_t hdr[7], orc_config* out);

/* Element counts / float-stream offsets of each tensor (layer = -1 for the whole tensor). */
uint64_t orc_tensor_count(const orc_config* c, int kind);             /* floats in the whole tensor (all layers) */
uint64_t orc_tensor_offset(const orc_config* c, int kind);            /* float index after the 28-byte header */
uint64_t orc_checkpoint_floats(const orc_config* c);                  /* total floats after the header */

/* ---- deterministic synthetic checkpoints (the repo's own generator) ---- */
/* value(g) = bias + float(c(g, seed)) * scale, c = centred sum of four 16-bit hash fields. */
void orc_synth_fill(float* out, uint64_t g0, uint64_t n, uint32_t seed, float scale, float bias);
void orc_synth_params(const orc_config* c, int kind, float* scale, float* bias); /* per-tensor law */
void orc_synth_freq(const orc_config* c, float* real, float* imag);             /* (S, hs/2) tables */
void orc_rope_runc(const orc_config* c, float* real, float* imag);              /* (S, hs/2) tables by llama2.c's run.c formula (8(f4)) */
/* Fill one tensor (layer >= 0: that layer's slice; -1: whole tensor) exactly as the file has it. */
void orc_synth_tensor(const orc_config* c, uint32_t seed, int kind, int layer, float* out);
/* Write a whole llama2.c-v0 checkpoint. Returns 0 or -1. */
int orc_synth_write(const int32_t hdr[7], uint32_t seed, const char* path);

/* ---- model + forward ---- */
/* Weights are one contiguous float stream in file order (what follows the header). Not copied. */
orc_model* orc_create(const int32_t hdr[7], const float* weights);
/* Same, but reads the file at `path` (mallocs + freads). */
orc_model* orc_open(const char* path);
/* Synthetic model generated in memory. */
orc_model* orc_create_synth(const int32_t hdr[7], uint32_t seed);
void orc_destroy(orc_model* m);
const orc_config* orc_get_config(const orc_model* m);
const float* orc_weights(const orc_model* m, int kind, int layer);
float* orc_state(orc_model* m, int which, size_t* n);

/* transformer(token,pos,...) (llama2.ts:205-303). logits_out may be NULL. */
void orc_forward(orc_model* m, int token, int pos, float* logits_out);
/* argmax (llama2.ts:364-366): first maximum, strict '>'. */
int orc_argmax(const float* v, int n);

/* Samplers downstream of the path.  `rng` is the 64-bit xorshift* state the reference keeps in
 * the BigInt `rng_seed` (llama2.ts:348-355), non-zero. */
uint32_t orc_random_u32(uint64_t* rng);                                            /* :349-354 */
float orc_random_f32(uint64_t* rng);                                               /* :357-360 (rounds, so 1.0f is possible) */
int orc_sample(const float* probs, int n, uint64_t* rng);                          /* :368-376 */
int orc_sample_topp(const float* probs, int n, double topp, uint64_t* rng);        /* :378-394 (stable sort, never returns element lastIdx) */
/* The branch at llama2.ts:476-493: greedy when temperature == 0, else logits /= temperature (in place, fp32
 * stores), softmax in place, then sample or sample_topp (topp <= 0 or >= 1: plain sample).  Mutates `logits`
 * exactly as the reference mutates state.logits. */
int orc_next_token(float* logits, int n, double temperature, double topp, uint64_t* rng);

/* Building blocks, exported so kernels can be checked one at a time. */
void orc_rmsnorm(float* o, const float* x, const float* w, int size);             /* :172-179 */
void orc_softmax(float* x, int size);                                               /* :181-194 */
void orc_matmul(float* xout, const float* x, const float* w, int n, int d);        /* :196-203 */

/* Tensor-parallel restatement: rank `r` of `g` computes its partial sums in
 * fp64; used by the world_size-2 gloo tests.  Computes one full forward by emulating all g ranks
 * in-process and summing fp64 partials in rank order, rounding once. */
void orc_forward_tp(orc_model* m, int token, int pos, int g, float* logits_out);

/* cpu_baseline helper: seconds for `steps` forwards starting at pos0 (greedy feed). */
double orc_time_forward(orc_model* m, int pos0, int steps, int* tokens_out);

#ifdef __cplusplus
}
#endif
#endif

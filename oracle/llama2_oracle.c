/*
 * llama2_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see llama2_oracle.h).
 *
 * Numeric contract restated here: in the reference every scalar temporary is a
 * JS `number` (IEEE double) and rounding to fp32 happens only on a Float32Array store.  So every
 * accumulator below is `double` and every store casts to `float` exactly where the reference
 * stores into a typed array.  Build with -ffp-contract=off: the reference has no fused multiply-add.
 */
#include "llama2_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------------------------ */
/* Config + layout                                                                             */

/* SURVEY.md 8(f4) -- PARITY UNPINNED BY THE REFERENCE: llama2.ts parses n_kv_heads and ignores it (llama2.ts:86,
 * 117-118), so it cannot run a grouped-query checkpoint.  With this switch on, the restatement honours the field the
 * way llama2.c's run.c does (wk / wv have n_kv_heads * head_size rows, query head h reads cache head
 * h / (n_heads / n_kv_heads)); everything else keeps the reference's arithmetic.  Off (default) = the reference. */
static int g_honour_kv_heads = 0;
void orc_set_gqa(int on) { g_honour_kv_heads = on; }
static int kv_dim(const orc_config* c) {
  return (g_honour_kv_heads && c->n_kv_heads > 0 && c->n_kv_heads < c->n_heads) ? c->n_kv_heads * c->head_size : c->dim;
}

void orc_read_config(const int32_t hdr[7], orc_config* c) {
  /* llama2.ts:80-93 */
  c->dim = hdr[0];
  c->hidden_dim = hdr[1];
  c->n_layers = hdr[2];
  c->n_heads = hdr[3];
  c->n_kv_heads = hdr[4];
  c->vocab_size = hdr[5] < 0 ? -hdr[5] : hdr[5];
  c->seq_len = hdr[6];
  c->shared_weights = hdr[5] > 0;
  c->head_size = c->dim / c->n_heads;
}

uint64_t orc_tensor_count(const orc_config* c, int kind) {
  const uint64_t d = (uint64_t)c->dim, h = (uint64_t)c->hidden_dim, L = (uint64_t)c->n_layers;
  const uint64_t V = (uint64_t)c->vocab_size, S = (uint64_t)c->seq_len, hs2 = (uint64_t)(c->head_size / 2);
  switch (kind) { /* llama2.ts:114-127 */
    case ORC_T_TOKEN_EMBEDDING: return V * d;
    case ORC_T_RMS_ATT: case ORC_T_RMS_FFN: return L * d;
    case ORC_T_WQ: case ORC_T_WO: return L * d * d;
    case ORC_T_WK: case ORC_T_WV: return L * (uint64_t)kv_dim(c) * d;
    case ORC_T_W1: case ORC_T_W3: return L * h * d;
    case ORC_T_W2: return L * d * h;
    case ORC_T_RMS_FINAL: return d;
    case ORC_T_FREQ_REAL: case ORC_T_FREQ_IMAG: return S * hs2;
    case ORC_T_WCLS: return c->shared_weights ? 0 : V * d;
    default: return 0;
  }
}

uint64_t orc_tensor_offset(const orc_config* c, int kind) {
  uint64_t off = 0;
  for (int k = 0; k < kind; ++k) off += orc_tensor_count(c, k);
  return off;
}

uint64_t orc_checkpoint_floats(const orc_config* c) { return orc_tensor_offset(c, ORC_T_COUNT); }

static int tensor_is_layered(int kind) {
  return (kind >= ORC_T_RMS_ATT && kind <= ORC_T_W3);
}

/* ------------------------------------------------------------------------------------------ */
/* Deterministic synthetic generator (integer hash -> centred sum of four 16-bit uniforms).     */
/* Only 32-bit integer ops and one fp32 multiply-add, so C / numpy / JS / HIP agree bit for bit. */

static inline uint32_t hash32(uint32_t a) {
  a ^= a >> 16; a *= 0x7feb352dU; a ^= a >> 15; a *= 0x846ca68bU; a ^= a >> 16;
  return a;
}

static inline int32_t synth_c(uint64_t g, uint32_t seed) {
  const uint32_t lo = (uint32_t)g, hi = (uint32_t)(g >> 32);
  const uint32_t k = hash32(hi ^ (seed * 0x9E3779B9U) ^ 0x85ebca6bU);
  const uint32_t h1 = hash32(lo ^ k);
  const uint32_t h2 = hash32(h1 + 0x9E3779B9U);
  return (int32_t)((h1 & 0xffffU) + (h1 >> 16) + (h2 & 0xffffU) + (h2 >> 16)) - 131070;
}

void orc_synth_fill(float* out, uint64_t g0, uint64_t n, uint32_t seed, float scale, float bias) {
  /* every element is a pure function of its index: the GENERATOR (never the forward pass) may use all host threads */
#ifdef _OPENMP
#pragma omp parallel for schedule(static) if (n > (1u << 20))
#endif
  for (uint64_t i = 0; i < n; ++i) {
    const float p = (float)synth_c(g0 + i, seed) * scale; /* one fp32 rounding */
    out[i] = bias + p;                                    /* one fp32 rounding */
  }
}

#define SYNTH_STD 37837.22723720648 /* sqrt(4 * (65536^2 - 1) / 12) */

void orc_synth_params(const orc_config* c, int kind, float* scale, float* bias) {
  double sigma = 0.0;
  *bias = 0.0f;
  switch (kind) {
    case ORC_T_TOKEN_EMBEDDING: case ORC_T_WCLS: sigma = 0.02; break;
    case ORC_T_RMS_ATT: case ORC_T_RMS_FFN: case ORC_T_RMS_FINAL: sigma = 0.1; *bias = 1.0f; break;
    case ORC_T_WQ: case ORC_T_WK: case ORC_T_WV: case ORC_T_WO: case ORC_T_W1: case ORC_T_W3:
      sigma = 1.0 / sqrt((double)c->dim); break;
    case ORC_T_W2: sigma = 1.0 / sqrt((double)c->hidden_dim); break;
    default: sigma = 0.0; break;
  }
  *scale = (float)(sigma / SYNTH_STD);
}

/* exp / sin / cos from IEEE basic operations only (deterministic in every language). */
static double det_exp(double x) { /* x in [-10, 0] */
  const double y = x / 1024.0;
  double t = 1.0, s = 1.0;
  for (int k = 1; k <= 14; ++k) { t = (t * y) / (double)k; s = s + t; }
  for (int i = 0; i < 10; ++i) s = s * s;
  return s;
}

static void det_sincos(double x, double* sn, double* cs) { /* |x| <= 1 */
  const double x2 = x * x;
  double ts = x, tc = 1.0, ss = x, cc = 1.0;
  for (int k = 1; k <= 12; ++k) {
    tc = ((-tc) * x2) / (double)((2 * k - 1) * (2 * k));
    cc = cc + tc;
    ts = ((-ts) * x2) / (double)((2 * k) * (2 * k + 1));
    ss = ss + ts;
  }
  *sn = ss; *cs = cc;
}

void orc_synth_freq(const orc_config* c, float* real, float* imag) {
  const int hs2 = c->head_size / 2;
  for (int j = 0; j < hs2; ++j) {
    const double theta = det_exp(-(((2.0 * (double)j) / (double)c->head_size) * 9.210340371976184));
    double st, ct;
    det_sincos(theta, &st, &ct);
    double cr = 1.0, ci = 0.0; /* angle t*theta by complex rotation */
    for (int t = 0; t < c->seq_len; ++t) {
      real[(size_t)t * hs2 + j] = (float)cr;
      imag[(size_t)t * hs2 + j] = (float)ci;
      const double nr = cr * ct - ci * st;
      const double ni = cr * st + ci * ct;
      cr = nr; ci = ni;
    }
  }
}

/* SURVEY.md 8(f4), beyond the reference: RoPE tables as newer llama2.c exports imply them (such files carry none; the
 * reference reads them from the file, llama2.ts:125-126).  llama2.c's run.c evaluates, per position and per pair, in fp32:
 * freq = 1 / powf(10000, head_dim / head_size) with head_dim = 2j, val = pos * freq, (cosf(val), sinf(val)).
 * Used to WRITE such tables into a v0 file the reference can run (oracle/make_goldens.py, the gqa_rope fixtures). */
void orc_rope_runc(const orc_config* c, float* real, float* imag) {
  const int hs2 = c->head_size / 2;
  for (int t = 0; t < c->seq_len; ++t)
    for (int j = 0; j < hs2; ++j) {
      const float freq = 1.0f / powf(10000.0f, (float)(2 * j) / (float)c->head_size);
      const float val = (float)t * freq;
      real[(size_t)t * hs2 + j] = cosf(val);
      imag[(size_t)t * hs2 + j] = sinf(val);
    }
}

void orc_synth_tensor(const orc_config* c, uint32_t seed, int kind, int layer, float* out) {
  const uint64_t total = orc_tensor_count(c, kind);
  if (total == 0) return;
  if (kind == ORC_T_FREQ_REAL || kind == ORC_T_FREQ_IMAG) {
    float* other = (float*)malloc(total * sizeof(float));
    if (kind == ORC_T_FREQ_REAL) orc_synth_freq(c, out, other); else orc_synth_freq(c, other, out);
    free(other);
    return;
  }
  float scale, bias;
  orc_synth_params(c, kind, &scale, &bias);
  uint64_t g0 = orc_tensor_offset(c, kind), n = total;
  if (layer >= 0 && tensor_is_layered(kind)) {
    n = total / (uint64_t)c->n_layers;
    g0 += n * (uint64_t)layer;
  }
  orc_synth_fill(out, g0, n, seed, scale, bias);
}

int orc_synth_write(const int32_t hdr[7], uint32_t seed, const char* path) {
  orc_config c;
  orc_read_config(hdr, &c);
  FILE* f = fopen(path, "wb");
  if (!f) return -1;
  if (fwrite(hdr, 4, 7, f) != 7) { fclose(f); return -1; }
  for (int kind = 0; kind < ORC_T_COUNT; ++kind) {
    const uint64_t total = orc_tensor_count(&c, kind);
    if (!total) continue;
    const int layered = tensor_is_layered(kind);
    const int parts = layered ? c.n_layers : 1;
    const uint64_t n = total / (uint64_t)parts;
    float* buf = (float*)malloc(n * sizeof(float));
    if (!buf) { fclose(f); return -1; }
    for (int p = 0; p < parts; ++p) {
      orc_synth_tensor(&c, seed, kind, layered ? p : -1, buf);
      if (fwrite(buf, sizeof(float), n, f) != n) { free(buf); fclose(f); return -1; }
    }
    free(buf);
  }
  return fclose(f) == 0 ? 0 : -1;
}

/* ------------------------------------------------------------------------------------------ */
/* Model                                                                                       */

struct orc_model {
  orc_config c;
  const float* w[ORC_T_COUNT]; /* start of each tensor in the float stream */
  float* owned;                /* non-NULL when we malloc'd the weights */
  /* RunState (llama2.ts:147-163) */
  float *x, *xb, *xb2, *hb, *hb2, *q, *k, *v, *att, *logits, *key_cache, *value_cache;
};

orc_model* orc_create(const int32_t hdr[7], const float* weights) {
  orc_model* m = (orc_model*)calloc(1, sizeof(orc_model));
  if (!m) return NULL;
  orc_read_config(hdr, &m->c);
  const orc_config* c = &m->c;
  for (int k = 0; k < ORC_T_COUNT; ++k) m->w[k] = weights + orc_tensor_offset(c, k);
  if (c->shared_weights) m->w[ORC_T_WCLS] = m->w[ORC_T_TOKEN_EMBEDDING]; /* alias, llama2.ts:127 */
  const size_t d = (size_t)c->dim, h = (size_t)c->hidden_dim;
  m->x = (float*)calloc(d, 4); m->xb = (float*)calloc(d, 4); m->xb2 = (float*)calloc(d, 4);
  m->hb = (float*)calloc(h, 4); m->hb2 = (float*)calloc(h, 4);
  m->q = (float*)calloc(d, 4); m->k = (float*)calloc(d, 4); m->v = (float*)calloc(d, 4);
  m->att = (float*)calloc((size_t)c->n_heads * c->seq_len, 4);
  m->logits = (float*)calloc((size_t)c->vocab_size, 4);
  m->key_cache = (float*)calloc((size_t)c->n_layers * c->seq_len * d, 4);
  m->value_cache = (float*)calloc((size_t)c->n_layers * c->seq_len * d, 4);
  return m;
}

orc_model* orc_open(const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) return NULL;
  int32_t hdr[7];
  if (fread(hdr, 4, 7, f) != 7) { fclose(f); return NULL; }
  orc_config c;
  orc_read_config(hdr, &c);
  const uint64_t n = orc_checkpoint_floats(&c);
  float* w = (float*)malloc(n * sizeof(float));
  if (!w || fread(w, sizeof(float), n, f) != n) { free(w); fclose(f); return NULL; }
  fclose(f);
  orc_model* m = orc_create(hdr, w);
  if (m) m->owned = w; else free(w);
  return m;
}

orc_model* orc_create_synth(const int32_t hdr[7], uint32_t seed) {
  orc_config c;
  orc_read_config(hdr, &c);
  const uint64_t n = orc_checkpoint_floats(&c);
  float* w = (float*)malloc(n * sizeof(float));
  if (!w) return NULL;
  for (int kind = 0; kind < ORC_T_COUNT; ++kind)
    if (orc_tensor_count(&c, kind)) orc_synth_tensor(&c, seed, kind, -1, w + orc_tensor_offset(&c, kind));
  orc_model* m = orc_create(hdr, w);
  if (m) m->owned = w; else free(w);
  return m;
}

void orc_destroy(orc_model* m) {
  if (!m) return;
  free(m->x); free(m->xb); free(m->xb2); free(m->hb); free(m->hb2); free(m->q); free(m->k); free(m->v);
  free(m->att); free(m->logits); free(m->key_cache); free(m->value_cache);
  free(m->owned);
  free(m);
}

const orc_config* orc_get_config(const orc_model* m) { return &m->c; }

const float* orc_weights(const orc_model* m, int kind, int layer) {
  const float* p = m->w[kind];
  if (layer > 0 && tensor_is_layered(kind))
    p += (orc_tensor_count(&m->c, kind) / (uint64_t)m->c.n_layers) * (uint64_t)layer;
  return p;
}

float* orc_state(orc_model* m, int which, size_t* n) {
  const orc_config* c = &m->c;
  const size_t d = (size_t)c->dim, h = (size_t)c->hidden_dim;
  const size_t kv = (size_t)c->n_layers * c->seq_len * d;
  float* p = NULL; size_t cnt = 0;
  switch (which) {
    case ORC_S_X: p = m->x; cnt = d; break;
    case ORC_S_XB: p = m->xb; cnt = d; break;
    case ORC_S_XB2: p = m->xb2; cnt = d; break;
    case ORC_S_HB: p = m->hb; cnt = h; break;
    case ORC_S_HB2: p = m->hb2; cnt = h; break;
    case ORC_S_Q: p = m->q; cnt = d; break;
    case ORC_S_K: p = m->k; cnt = d; break;
    case ORC_S_V: p = m->v; cnt = d; break;
    case ORC_S_ATT: p = m->att; cnt = (size_t)c->n_heads * c->seq_len; break;
    case ORC_S_LOGITS: p = m->logits; cnt = (size_t)c->vocab_size; break;
    case ORC_S_KEY_CACHE: p = m->key_cache; cnt = kv; break;
    case ORC_S_VALUE_CACHE: p = m->value_cache; cnt = kv; break;
    default: break;
  }
  if (n) *n = cnt;
  return p;
}

/* ------------------------------------------------------------------------------------------ */
/* Blocks                                                                                      */

static void accum(float* a, const float* b, int size) { /* llama2.ts:168-170 */
  for (int i = 0; i < size; ++i) a[i] = (float)((double)a[i] + (double)b[i]);
}

void orc_rmsnorm(float* o, const float* x, const float* w, int size) { /* llama2.ts:172-179 */
  double ss = 0.0;
  for (int j = 0; j < size; ++j) ss += (double)x[j] * (double)x[j];
  ss /= (double)size;
  ss = 1.0 / sqrt(1e-5 + ss);
  for (int j = 0; j < size; ++j) o[j] = (float)((double)w[j] * (ss * (double)x[j]));
}

void orc_softmax(float* x, int size) { /* llama2.ts:181-194 */
  float max_val = x[0];
  for (int i = 1; i < size; ++i) if (x[i] > max_val) max_val = x[i];
  for (int i = 0; i < size; ++i) x[i] = (float)exp((double)x[i] - (double)max_val);
  double sum = 0.0;
  for (int i = 0; i < size; ++i) sum += (double)x[i];
  for (int i = 0; i < size; ++i) x[i] = (float)((double)x[i] / sum);
}

void orc_matmul(float* xout, const float* x, const float* w, int n, int d) { /* llama2.ts:196-203 */
  for (int i = 0; i < d; ++i) {
    const float* row = w + (size_t)i * n;
    double sum = 0.0;
    for (int j = 0; j < n; ++j) sum += (double)row[j] * (double)x[j];
    xout[i] = (float)sum;
  }
}

/* fp64 partial dot over columns [j0, j1) -- the tensor-parallel column slice. */
static double dot_partial(const float* row, const float* x, int j0, int j1) {
  double sum = 0.0;
  for (int j = j0; j < j1; ++j) sum += (double)row[j] * (double)x[j];
  return sum;
}

int orc_argmax(const float* v, int n) { /* llama2.ts:364-366 */
  int best = 0;
  for (int i = 0; i < n; ++i) if (v[i] > v[best]) best = i;
  return best;
}

/* ------------------------------------------------------------------------------------------ */
/* Samplers (llama2.ts:348-394, 476-493)                                                       */

uint32_t orc_random_u32(uint64_t* rng) { /* llama2.ts:349-354; the BigInt product is not masked before >> 32, bits 32..63 are the same */
  uint64_t s = *rng;
  s ^= s >> 12;
  s ^= s << 25;
  s ^= s >> 27;
  *rng = s;
  return (uint32_t)((s * 0x2545F4914F6CDD1Dull) >> 32);
}

float orc_random_f32(uint64_t* rng) { /* llama2.ts:357-360: two fp64 divisions (exact), ONE rounding at the Float32Array store */
  const double v = ((double)orc_random_u32(rng) / 256.0) / 16777216.0;
  return (float)v;
}

int orc_sample(const float* probs, int n, uint64_t* rng) { /* llama2.ts:368-376 */
  double sum = 0.0;
  for (int i = 0; i < n; ++i) sum = sum + (double)probs[i];
  const double rand_value = (double)orc_random_f32(rng) * sum;
  double cum = 0.0;
  for (int i = 0; i < n; ++i) {
    cum += (double)probs[i];
    if (rand_value < cum) return i;
  }
  return 0;
}

typedef struct { int index; float prob; } orc_probindex;
static void orc_merge_sort_desc(orc_probindex* a, orc_probindex* tmp, int n) { /* stable, like V8's TimSort (Array.prototype.sort, Node >= 11) */
  if (n < 2) return;
  const int h = n / 2;
  orc_merge_sort_desc(a, tmp, h);
  orc_merge_sort_desc(a + h, tmp, n - h);
  int i = 0, j = h, k = 0;
  while (i < h && j < n) tmp[k++] = (a[j].prob > a[i].prob) ? a[j++] : a[i++];   /* comparator b.prob - a.prob: equal keeps the left one first */
  while (i < h) tmp[k++] = a[i++];
  while (j < n) tmp[k++] = a[j++];
  memcpy(a, tmp, (size_t)n * sizeof(*a));
}

int orc_sample_topp(const float* probs, int n, double topp, uint64_t* rng) { /* llama2.ts:378-394 */
  orc_probindex* pi = (orc_probindex*)malloc((size_t)n * sizeof(*pi) * 2);
  for (int i = 0; i < n; ++i) { pi[i].index = i; pi[i].prob = probs[i]; }
  orc_merge_sort_desc(pi, pi + n, n);
  double cum = 0.0;
  int last = 0;
  for (int i = 0; i < n; ++i) {
    cum += (double)pi[i].prob;
    if (cum > topp) { last = i; break; }
  }
  const double rand_value = (double)orc_random_f32(rng) * cum;
  cum = 0.0;
  int out = 0;
  for (int i = 0; i < last; ++i) {   /* i < lastIdx: the element that crossed topp is never returned (:390) */
    cum += (double)pi[i].prob;
    if (rand_value < cum) { out = pi[i].index; break; }
  }
  free(pi);
  return out;                         /* fall-through returns id 0 (:393) */
}

int orc_next_token(float* logits, int n, double temperature, double topp, uint64_t* rng) { /* llama2.ts:476-493 */
  if (temperature == 0.0) return orc_argmax(logits, n);
  for (int q = 0; q < n; ++q) logits[q] = (float)((double)logits[q] / temperature);
  orc_softmax(logits, n);
  if (topp <= 0 || topp >= 1) return orc_sample(logits, n, rng);
  return orc_sample_topp(logits, n, topp, rng);
}

/* ------------------------------------------------------------------------------------------ */
/* Forward                                                                                     */

static void rope_and_store(orc_model* m, int l, int pos) {
  const orc_config* p = &m->c;
  const int dim = p->dim, head_size = p->dim / p->n_heads, kvd = kv_dim(p);
  const float* fr = m->w[ORC_T_FREQ_REAL];
  const float* fi = m->w[ORC_T_FREQ_IMAG];
  for (int i = 0; i < dim; i += 2) { /* llama2.ts:224-235 */
    const size_t idx = (size_t)pos * head_size / 2 + (size_t)(i % head_size) / 2;
    const double fcr = fr[idx], fci = fi[idx];
    const double q0 = m->q[i], q1 = m->q[i + 1];
    m->q[i] = (float)(q0 * fcr - q1 * fci);
    m->q[i + 1] = (float)(q0 * fci + q1 * fcr);
    if (i < kvd) {   /* every pair when kvd == dim (the reference); grouped-query: only the cache heads' pairs */
      const double k0 = m->k[i], k1 = m->k[i + 1];
      m->k[i] = (float)(k0 * fcr - k1 * fci);
      m->k[i + 1] = (float)(k0 * fci + k1 * fcr);
    }
  }
  const size_t loff = (size_t)l * p->seq_len * kvd; /* llama2.ts:238-240 */
  memcpy(m->key_cache + loff + (size_t)pos * kvd, m->k, (size_t)kvd * 4);
  memcpy(m->value_cache + loff + (size_t)pos * kvd, m->v, (size_t)kvd * 4);
}

static void attention(orc_model* m, int l, int pos) {
  const orc_config* p = &m->c;
  const int head_size = p->dim / p->n_heads, dim = kv_dim(p) /* stride of a cache row */;
  const int kv_mul = p->dim / dim;                                  /* query heads per cache head: 1 in the reference */
  const size_t loff = (size_t)l * p->seq_len * dim;
  const double inv = sqrt((double)head_size);
  for (int hq = 0; hq < p->n_heads; ++hq) { /* llama2.ts:244-267 */
    const int h = hq / kv_mul;
    const float* q = m->q + (size_t)hq * head_size;
    float* att = m->att + (size_t)hq * p->seq_len;
    for (int t = 0; t <= pos; ++t) {
      const float* kk = m->key_cache + loff + (size_t)t * dim + (size_t)h * head_size;
      double score = 0.0;
      for (int i = 0; i < head_size; ++i) score += (double)q[i] * (double)kk[i];
      att[t] = (float)(score / inv);
    }
    orc_softmax(att, pos + 1);
    float* xb = m->xb + (size_t)hq * head_size;
    for (int i = 0; i < head_size; ++i) xb[i] = 0.0f;
    for (int t = 0; t <= pos; ++t) {
      const double a = att[t];
      const float* vv = m->value_cache + loff + (size_t)t * dim + (size_t)h * head_size;
      /* the accumulator lives in a Float32Array: one fp32 rounding per timestep (llama2.ts:263) */
      for (int i = 0; i < head_size; ++i) xb[i] = (float)((double)xb[i] + a * (double)vv[i]);
    }
  }
}

static void swiglu(orc_model* m) { /* llama2.ts:284-289: two fp32 roundings */
  const int hidden = m->c.hidden_dim;
  for (int i = 0; i < hidden; ++i) {
    const double v = m->hb[i];
    m->hb[i] = (float)(v * (1.0 / (1.0 + exp(-v))));
  }
  for (int i = 0; i < hidden; ++i) m->hb[i] = (float)((double)m->hb[i] * (double)m->hb2[i]);
}

void orc_forward(orc_model* m, int token, int pos, float* logits_out) {
  const orc_config* p = &m->c;
  const int dim = p->dim, hidden = p->hidden_dim;
  memcpy(m->x, m->w[ORC_T_TOKEN_EMBEDDING] + (size_t)token * dim, (size_t)dim * 4); /* :211 */
  for (int l = 0; l < p->n_layers; ++l) {
    orc_rmsnorm(m->xb, m->x, orc_weights(m, ORC_T_RMS_ATT, l), dim);           /* :216 */
    orc_matmul(m->q, m->xb, orc_weights(m, ORC_T_WQ, l), dim, dim);             /* :219 */
    orc_matmul(m->k, m->xb, orc_weights(m, ORC_T_WK, l), dim, kv_dim(p));       /* :220 (dim rows in the reference) */
    orc_matmul(m->v, m->xb, orc_weights(m, ORC_T_WV, l), dim, kv_dim(p));       /* :221 */
    rope_and_store(m, l, pos);                                                  /* :224-240 */
    attention(m, l, pos);                                                       /* :244-267 */
    orc_matmul(m->xb2, m->xb, orc_weights(m, ORC_T_WO, l), dim, dim);           /* :270 */
    accum(m->x, m->xb2, dim);                                                   /* :273 */
    orc_rmsnorm(m->xb, m->x, orc_weights(m, ORC_T_RMS_FFN, l), dim);           /* :276 */
    orc_matmul(m->hb, m->xb, orc_weights(m, ORC_T_W1, l), dim, hidden);         /* :280 */
    orc_matmul(m->hb2, m->xb, orc_weights(m, ORC_T_W3, l), dim, hidden);        /* :281 */
    swiglu(m);                                                                  /* :284-289 */
    orc_matmul(m->xb, m->hb, orc_weights(m, ORC_T_W2, l), hidden, dim);         /* :292 */
    accum(m->x, m->xb, dim);                                                    /* :295 */
  }
  orc_rmsnorm(m->x, m->x, m->w[ORC_T_RMS_FINAL], dim);                          /* :299 */
  orc_matmul(m->logits, m->x, m->w[ORC_T_WCLS], dim, p->vocab_size);            /* :302 */
  if (logits_out) memcpy(logits_out, m->logits, (size_t)p->vocab_size * 4);
}

/* Tensor-parallel restatement: G ranks; q/k/v/w1/w3/wcls row-sharded (full-length dots, so the
 * values equal the 1-rank ones), wo/w2 column-sharded: fp64 partial per rank, summed in rank
 * order, rounded to fp32 once. */
void orc_forward_tp(orc_model* m, int token, int pos, int g, float* logits_out) {
  const orc_config* p = &m->c;
  const int dim = p->dim, hidden = p->hidden_dim;
  memcpy(m->x, m->w[ORC_T_TOKEN_EMBEDDING] + (size_t)token * dim, (size_t)dim * 4);
  for (int l = 0; l < p->n_layers; ++l) {
    orc_rmsnorm(m->xb, m->x, orc_weights(m, ORC_T_RMS_ATT, l), dim);
    orc_matmul(m->q, m->xb, orc_weights(m, ORC_T_WQ, l), dim, dim);
    orc_matmul(m->k, m->xb, orc_weights(m, ORC_T_WK, l), dim, dim);
    orc_matmul(m->v, m->xb, orc_weights(m, ORC_T_WV, l), dim, dim);
    rope_and_store(m, l, pos);
    attention(m, l, pos);
    const float* wo = orc_weights(m, ORC_T_WO, l);
    for (int i = 0; i < dim; ++i) {
      double sum = 0.0;
      for (int r = 0; r < g; ++r) sum += dot_partial(wo + (size_t)i * dim, m->xb, r * (dim / g), (r + 1) * (dim / g));
      m->xb2[i] = (float)sum;
    }
    accum(m->x, m->xb2, dim);
    orc_rmsnorm(m->xb, m->x, orc_weights(m, ORC_T_RMS_FFN, l), dim);
    orc_matmul(m->hb, m->xb, orc_weights(m, ORC_T_W1, l), dim, hidden);
    orc_matmul(m->hb2, m->xb, orc_weights(m, ORC_T_W3, l), dim, hidden);
    swiglu(m);
    const float* w2 = orc_weights(m, ORC_T_W2, l);
    for (int i = 0; i < dim; ++i) {
      double sum = 0.0;
      for (int r = 0; r < g; ++r) sum += dot_partial(w2 + (size_t)i * hidden, m->hb, r * (hidden / g), (r + 1) * (hidden / g));
      m->xb[i] = (float)sum;
    }
    accum(m->x, m->xb, dim);
  }
  orc_rmsnorm(m->x, m->x, m->w[ORC_T_RMS_FINAL], dim);
  orc_matmul(m->logits, m->x, m->w[ORC_T_WCLS], dim, p->vocab_size);
  if (logits_out) memcpy(logits_out, m->logits, (size_t)p->vocab_size * 4);
}

double orc_time_forward(orc_model* m, int pos0, int steps, int* tokens_out) {
  struct timespec t0, t1;
  int token = 1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int s = 0; s < steps; ++s) {
    orc_forward(m, token, pos0 + s, NULL);
    token = orc_argmax(m->logits, m->c.vocab_size);
    if (tokens_out) tokens_out[s] = token;
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

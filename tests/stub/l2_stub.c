/* l2_stub.c -- TEST INFRASTRUCTURE: a host-memory stand-in for libllama2hip.so with the entry points the N-API addon binds
 * (llama2.ts_amd/host/l2_napi.cc), so the addon's pointer / length handling -- what replaces FileHandleReader.getF32Array views
 * and the state.logits hand-off of llama2.ts:44-68, 468 -- can run under AddressSanitizer / UBSan with Node on a box without a GPU.
 * It computes nothing of the model: every call touches exactly the bytes the real library would read or write (so an
 * out-of-bounds hand-off is an ASan report) and records what it was given (l2_stub_report). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/llama2_hip.h"

struct l2_ctx {
  int32_t hdr[7];
  int V;
  float* logits;          /* "pinned" logits, V floats */
  double upload_sum;      /* sum of every float uploaded (forces a read of every byte) */
  size_t upload_floats;
  int uploads;
  int last_kind, last_layer;
  float last_first, last_last;   /* first / last element of the last uploaded array: a view with byteOffset != 0 must hand over ITS bytes */
};

static char g_err[256] = "";
static int fail(int code, const char* msg) { snprintf(g_err, sizeof(g_err), "%s", msg); return code; }

int l2_abi_version(void) { return L2_ABI_VERSION; }
int l2_device_count(void) { return 1; }
const char* l2_last_error(void) { return g_err; }

static size_t tensor_floats(const l2_ctx* c, int kind) {
  const size_t d = c->hdr[0], h = c->hdr[1], H = c->hdr[3], V = c->V, S = c->hdr[6];
  switch (kind) {
    case L2_T_TOKEN_EMBEDDING: case L2_T_WCLS: return V * d;
    case L2_T_RMS_ATT: case L2_T_RMS_FFN: case L2_T_RMS_FINAL: return d;
    case L2_T_WQ: case L2_T_WK: case L2_T_WV: case L2_T_WO: return d * d;
    case L2_T_W1: case L2_T_W2: case L2_T_W3: return h * d;
    case L2_T_FREQ_REAL: case L2_T_FREQ_IMAG: return S * (d / H / 2);
    default: return 0;
  }
}

int l2_create(const int32_t cfg[7], int device, l2_ctx** out) {
  if (!cfg || !out) return fail(L2_E_ARG, "null argument");
  if (device != 0) return fail(L2_E_ARG, "device out of range");
  if (cfg[0] <= 0 || cfg[3] <= 0 || cfg[0] % cfg[3]) return fail(L2_E_CONFIG, "bad header");
  l2_ctx* c = (l2_ctx*)calloc(1, sizeof(l2_ctx));
  memcpy(c->hdr, cfg, sizeof(c->hdr));
  c->V = cfg[5] < 0 ? -cfg[5] : cfg[5];
  c->logits = (float*)calloc((size_t)c->V, sizeof(float));
  *out = c;
  return L2_OK;
}

void l2_destroy(l2_ctx* c) { if (c) { free(c->logits); free(c); } }

int l2_upload(l2_ctx* c, int kind, int layer, const float* host, size_t n) {
  if (!c || !host) return fail(L2_E_ARG, "null argument");
  if (kind < 0 || kind >= L2_T_COUNT) return fail(L2_E_ARG, "tensor kind out of range");
  if (n != tensor_floats(c, kind)) return fail(L2_E_ARG, "wrong float count for this tensor kind");
  double s = 0.0;
  for (size_t i = 0; i < n; ++i) s += host[i];          /* every byte of the array is read */
  c->upload_sum += s; c->upload_floats += n; c->uploads++;
  c->last_kind = kind; c->last_layer = layer; c->last_first = host[0]; c->last_last = host[n - 1];
  return L2_OK;
}

int l2_read_tensor(l2_ctx* c, int kind, int layer, size_t offset, float* out, size_t n) {
  (void)layer;
  if (!c || !out) return fail(L2_E_ARG, "null argument");
  if (kind < 0 || kind >= L2_T_COUNT) return fail(L2_E_ARG, "tensor kind out of range");
  if (offset + n > tensor_floats(c, kind)) return fail(L2_E_ARG, "read past the end of the tensor");
  for (size_t i = 0; i < n; ++i) out[i] = (float)(offset + i);     /* n floats are written */
  return L2_OK;
}

int l2_synth_fill(l2_ctx* c, uint32_t seed) { (void)seed; return c ? L2_OK : fail(L2_E_ARG, "null context"); }

static void fill_logits(l2_ctx* c, int token, int pos) {
  for (int i = 0; i < c->V; ++i) c->logits[i] = (float)((i * 31 + token * 7 + pos) % 1009) * 0.001f + c->last_first;
}

int l2_forward(l2_ctx* c, int token, int pos, float* logits_out) {
  if (!c) return fail(L2_E_ARG, "null context");
  if (pos < 0 || pos >= c->hdr[6]) return fail(L2_E_ARG, "pos out of range");
  if (token < 0 || token >= c->V) return fail(L2_E_ARG, "token out of range");
  fill_logits(c, token, pos);
  if (logits_out) memcpy(logits_out, c->logits, (size_t)c->V * sizeof(float));     /* V floats are written: a shorter array overflows */
  if (logits_out && getenv("L2_STUB_OVERRUN")) logits_out[c->V] = 1.0f;            /* negative control of the sanitizer test: one float too many */
  return L2_OK;
}

float* l2_logits_host(l2_ctx* c) { return c ? c->logits : NULL; }

int l2_decode_greedy(l2_ctx* c, int first, int pos0, int steps, int32_t* out) {
  if (!c || (!out && steps > 0)) return fail(L2_E_ARG, "null argument");
  if (steps < 0 || pos0 < 0 || pos0 + steps > c->hdr[6]) return fail(L2_E_ARG, "positions out of range");
  for (int s = 0; s < steps; ++s) out[s] = (first + pos0 + s) % c->V;
  return L2_OK;
}

int l2_decode_sample(l2_ctx* c, int first, int pos0, int steps, double t, double p, uint64_t* rng, int32_t* out) {
  (void)t; (void)p;
  if (!c || !rng || (!out && steps > 0)) return fail(L2_E_ARG, "null argument");
  if (steps < 0 || pos0 < 0 || pos0 + steps > c->hdr[6]) return fail(L2_E_ARG, "positions out of range");
  for (int s = 0; s < steps; ++s) { *rng ^= *rng >> 12; *rng ^= *rng << 25; *rng ^= *rng >> 27; out[s] = (int32_t)((*rng >> 33) % (uint64_t)c->V); }
  (void)first;
  return L2_OK;
}

int l2_read_state(l2_ctx* c, int which, int layer, float* out, size_t n) {
  (void)layer;
  if (!c || !out) return fail(L2_E_ARG, "null argument");
  size_t want = 0;
  switch (which) {
    case L2_S_X: case L2_S_XB: case L2_S_XB2: case L2_S_Q: case L2_S_K: case L2_S_V: want = c->hdr[0]; break;
    case L2_S_HB: case L2_S_HB2: want = c->hdr[1]; break;
    case L2_S_LOGITS: want = c->V; break;
    default: return fail(L2_E_ARG, "unknown state id");
  }
  if (n != want) return fail(L2_E_ARG, "wrong float count for this state");
  for (size_t i = 0; i < n; ++i) out[i] = (float)i;
  return L2_OK;
}

int l2_get_option(l2_ctx* c, int key, int* value) { if (!c || !value || key < 1 || key > 7) return fail(L2_E_ARG, "unknown option"); *value = 100 + key; return L2_OK; }
int l2_set_option(l2_ctx* c, int key, int value) { (void)value; return (c && key >= 1 && key <= 3) ? L2_OK : fail(L2_E_ARG, "unknown option"); }

int l2_get_header(l2_ctx* c, int32_t cfg_out[7]) {
  if (!c || !cfg_out) return fail(L2_E_ARG, "null argument");
  memcpy(cfg_out, c->hdr, sizeof(c->hdr));
  return L2_OK;
}

int l2_load_checkpoint(const char* path, int device, int r, int G, const void* id, l2_ctx** out, uint64_t* bytes) {
  (void)r; (void)G; (void)id;
  FILE* f = path ? fopen(path, "rb") : NULL;
  if (!f) return fail(L2_E_ARG, "cannot open checkpoint");
  int32_t hdr[7];
  const int ok = fread(hdr, 4, 7, f) == 7;
  fclose(f);
  if (!ok) return fail(L2_E_ARG, "checkpoint shorter than its header");
  if (bytes) *bytes = 28;
  return l2_create(hdr, device, out);
}

int l2_prefill(l2_ctx* c, const int32_t* tokens, int n, int pos0, float* logits_out) {
  if (!c || (!tokens && n > 0)) return fail(L2_E_ARG, "null argument");
  if (n <= 0 || pos0 < 0 || pos0 + n > c->hdr[6]) return fail(L2_E_ARG, "positions out of range");
  long s = 0;
  for (int i = 0; i < n; ++i) s += tokens[i];            /* every token is read */
  fill_logits(c, (int)(s % c->V), pos0 + n - 1);
  if (logits_out) memcpy(logits_out, c->logits, (size_t)c->V * sizeof(float));
  return L2_OK;
}

/* what the stub was handed so far (the test reads it back through a second dlopen of the same library) */
int l2_stub_report(l2_ctx* c, double* upload_sum, uint64_t* upload_floats, int* uploads, float* last_first, float* last_last) {
  if (!c) return L2_E_ARG;
  *upload_sum = c->upload_sum; *upload_floats = c->upload_floats; *uploads = c->uploads; *last_first = c->last_first; *last_last = c->last_last;
  return L2_OK;
}

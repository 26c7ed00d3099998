// llama2_hip.hip -- host runtime + C ABI (include/llama2_hip.h) of libllama2hip.so.
//
// Data layout in HBM (one context = one MI355X, 288 GB: every shape of SURVEY.md section 8 fits):
//   * one allocation per TransformerWeights field (llama2.ts:95-110), layers contiguous:
//     w[kind][layer][rows][cols] fp32 row-major exactly as the checkpoint has it (llama2.ts:112-129);
//   * RunState (llama2.ts:131-146): x, xb, xb2, hb, hb2, q, k, v, att, logits + KV caches [L][S][d];
//   * {token,pos,step} live in device memory so a captured hipGraph can be replayed for every
//     position without touching kernel arguments.
// One forward = 5 fused kernels per layer + the classifier, replayed as ONE hipGraph launch.
#include "../../include/llama2_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <chrono>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <algorithm>
#include <vector>

#include "kernels.hip.h"
#include "attention_inst.hip.h"      // attention.hip.h + its instances as extern templates (compiled in attention_inst.hip)
#include "prefill.hip.h"
#include "sampler.h"
#include "aql_queue.h"

using namespace l2k;

enum { NLEV = 3 };   // step levels by context length: two launches + one workgroup per head / the fused QKV + attention launch / two launches + 8 workgroups per head

#include "ctx.hip.h"

extern "C" int l2_abi_version(void) { return L2_ABI_VERSION; }
extern "C" const char* l2_last_error(void) { return g_err; }

extern "C" int l2_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return fail(L2_E_NOGPU, "hipGetDeviceCount: %s", hipGetErrorString(e));
  return n;
}

static void destroy_graphs(l2_ctx* c) {
  for (int i = 0; i < NLEV; ++i) {      // (their kernel arguments hold pointers and shapes of the step as it was)
    if (c->aql_greedy[i]) { aql_program_free(c->aql_greedy[i]); c->aql_greedy[i] = nullptr; }
    if (c->aql_step[i]) { aql_program_free(c->aql_step[i]); c->aql_step[i] = nullptr; }
    if (i == 0 && c->aql_last) { aql_program_free(c->aql_last); c->aql_last = nullptr; }
    for (int m = 0; m < 4; ++m) if (c->aql_sample[i][m]) { aql_program_free(c->aql_sample[i][m]); c->aql_sample[i][m] = nullptr; }
  }
  if (c->aql) aql_reset(c->aql);
  for (int i = 0; i < NLEV; ++i) {
    if (c->g_step[i]) { hipGraphExecDestroy(c->g_step[i]); c->g_step[i] = nullptr; }
    if (c->g_greedy[i]) { hipGraphExecDestroy(c->g_greedy[i]); c->g_greedy[i] = nullptr; }
    for (int m = 0; m < 4; ++m) if (c->g_sample[i][m]) { hipGraphExecDestroy(c->g_sample[i][m]); c->g_sample[i][m] = nullptr; }
  }
}

// Attention split level by context length.  A head's cache rows are read by ONE workgroup per split, and one CU pulls
// ~50-100 GB/s however many loads it keeps in flight, so beyond `split_rows` cached rows a head is split over 8
// workgroups (flash-decode merge by the last arriver, attention.hip.h).  The merge costs ~2.5 us per layer whatever
// the split count, so intermediate counts never win (tools/ctx_curve.py, 7B and 110M: 2 / 4 splits are slower than
// 8 at every position where they beat 1; the crossover is at 140-160 rows for 64- and 128-wide heads alike).
static const int kSplitLevels[NLEV] = {1, 1, 8};
static bool fused_shape_ok(const l2_ctx* c);
// Level of a step by its position.  Where the fused QKV + attention launch applies (launch.hip.h: fused_shape_ok) it takes the
// middle of the range: its attention workgroups request their cache rows while the GEMV runs, so ONE workgroup per head keeps up
// to 256 rows (two launches split a head over 8 workgroups from 144), but below `fuse_min_rows` the hand-off costs more than the
// launch boundary it replaces (stories110M: two launches win by 2-3 % up to 128 rows, the fused launch by 4 % from there to 256;
// stories15M: the fused launch wins from the first position -- profiles/r04/fused_qkv_attention_ab.txt, last block).
static int split_level(const l2_ctx* c, int pos) {
  if (c->attn_splits_forced > 0 || c->opt_exact) return 0;
  const bool fusable = fused_shape_ok(c);
  int unsplit_to = (c->split_rows_set || !fusable) ? c->split_rows : 256;
  if (fusable && unsplit_to > 256) unsplit_to = 256;      // the fused launch's attention role is built for one round of rows (attention.hip.h: attn_tile_dispatch)
  const int rows = pos + 1;
  if (rows > unsplit_to) return 2;
  return (fusable && rows > c->fuse_min_rows) ? 1 : 0;
}
static int splits_of(const l2_ctx* c, int level) { return c->attn_splits_forced > 0 ? c->attn_splits_forced : kSplitLevels[level]; }
// what a step of this level is enqueued with
static void set_level(l2_ctx* c, int level) {
  c->cur_splits = splits_of(c, level);
  c->cur_fused = level == 1 || (c->opt_fuse_splits && !c->opt_exact && fused_shape_ok(c));      // (the switch: fused wherever the shape allows -- tests)
}

extern "C" void l2_destroy(l2_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  destroy_graphs(c);
  if (c->aql) { aql_destroy(c->aql); c->aql = nullptr; }
  if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
  if (c->loop_tmp) hipFree(c->loop_tmp);
  for (void* m : c->p2p_opened) hipIpcCloseMemHandle(m);
  if (c->p2p_base) hipFree(c->p2p_base);
  if (c->p2p_epoch) hipFree(c->p2p_epoch);
  if (c->tp_push) hipFree(c->tp_push);
  if (c->awo_gran) hipFree(c->awo_gran);
  if (c->awo_ep) hipFree(c->awo_ep);
  if (c->p2p_err) hipHostFree(c->p2p_err);
  l2s::destroy(&c->samp);
  for (int k = 0; k < L2_T_COUNT; ++k)
    if (c->w[k] && !(k == L2_T_WCLS && c->shared)) hipFree(c->w[k]);  // shared wcls aliases the embedding table
  for (auto& p : c->packed) if (p.buf) hipFree(p.buf);
  float* bufs[] = {c->x, c->xb, c->xb2, c->hb, c->hb2, c->q, c->k, c->v, c->att, c->logits, c->kc, c->vc, c->xn};
  for (float* b : bufs) if (b) hipFree(b);
  if (c->logits_loc && c->logits_loc != c->logits) hipFree(c->logits_loc);
  if (c->partial) hipFree(c->partial);
  if (c->attn_part) hipFree(c->attn_part);
  if (c->attn_counter) hipFree(c->attn_counter);
  if (c->amax) hipFree(c->amax);
  if (c->gran) hipFree(c->gran);
  if (c->gran_ep) hipFree(c->gran_ep);
  if (c->h_herr) hipHostFree(c->h_herr);
  for (hipEvent_t e : c->probe) hipEventDestroy(e);
  { float* pb[] = {c->pf_x, c->pf_xn, c->pf_q, c->pf_xb, c->pf_hb}; for (float* b : pb) if (b) hipFree(b); if (c->pf_tok) hipFree(c->pf_tok); }
  if (c->pollute_sink) hipFree(c->pollute_sink);
  if (c->tokpos) hipFree(c->tokpos);
  if (c->d_tokens) hipFree(c->d_tokens);
  if (c->h_tokpos) hipHostFree(c->h_tokpos);
  if (c->h_logits) hipHostFree(c->h_logits);
  if (c->ev0) hipEventDestroy(c->ev0);
  if (c->ev1) hipEventDestroy(c->ev1);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
}

static int ensure_rowmajor(l2_ctx* c, bool unpack);
static int p2p_alloc(l2_ctx* c);
static int p2p_connect_ipc(l2_ctx* c);
static void p2p_set_peer(l2_ctx* c, int r, void* base, float* logits);
static int p2p_publish_table(l2_ctx* c);
static bool p2p_pushing(const l2_ctx* c);

static int create_impl(const int32_t cfg[7], int device, int rank, int G, const void* nccl_id, l2_ctx** out, unsigned flags = 0) {
  if (!cfg || !out) return fail(L2_E_ARG, "null argument");
  *out = nullptr;
  const int d = cfg[0], h = cfg[1], L = cfg[2], H = cfg[3], V = abs(cfg[5]), S = cfg[6];
  if (d <= 0 || h <= 0 || L <= 0 || H <= 0 || V <= 0 || S <= 0) return fail(L2_E_CONFIG, "non-positive size in header");
  if (d % H) return fail(L2_E_CONFIG, "dim %d not divisible by n_heads %d", d, H);
  if ((d / H) % 2) return fail(L2_E_CONFIG, "odd head_size %d (RoPE rotates adjacent pairs, llama2.ts:224)", d / H);
  if (G < 1 || rank < 0 || rank >= G) return fail(L2_E_ARG, "bad tensor-parallel rank %d of %d", rank, G);
  if (G > 1 && (H % G || h % G || V % G || ((d / G) % 2) || ((h / G) % 1)))
    return fail(L2_E_CONFIG, "shape does not shard over %d ranks (n_heads %d, hidden %d, vocab %d)", G, H, h, V);
  // the reference parses n_kv_heads and ignores it (llama2.ts:86, 117-118); it is honoured only on request (L2_F_GQA)
  const int KVH = (flags & L2_F_GQA) ? cfg[4] : H;
  if (KVH <= 0 || H % KVH || KVH % G) return fail(L2_E_CONFIG, "n_kv_heads %d does not divide n_heads %d (or the %d ranks)", KVH, H, G);
  // limits of the kernels, reported here instead of as an opaque launch failure: the attention tiles address a layer's
  // cache slab through a 32-bit buffer descriptor and keep one score per position of a split in LDS (160 KiB per CU);
  // the GEMV phases stage x (and the norm weight) in LDS
  if ((unsigned long long)S * (unsigned long long)(KVH * (d / H) / G) * 4ull >= (1ull << 32))
    return fail(L2_E_CONFIG, "seq_len %d x kv_dim %d: a layer's cache slab exceeds 4 GiB", S, KVH * (d / H) / G);
  if ((size_t)S * 4 + 40 * 1024 > 160 * 1024) return fail(L2_E_CONFIG, "seq_len %d: the attention kernel keeps one score per position in LDS (at most ~30 000)", S);
  if ((size_t)(d > h / G ? d : h / G) * 4 * 2 + 4096 > 160 * 1024) return fail(L2_E_CONFIG, "dim %d / hidden_dim %d: the input vector of a phase does not fit the 160 KiB LDS", d, h);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(L2_E_NOGPU, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(L2_E_ARG, "device %d out of range (%d visible)", device, ndev);
  HIPCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0 && !hook_int("L2_ALLOW_ANY_ARCH"))
    return fail(L2_E_NOGPU, "device %d is %s, this library is built for gfx950 only", device, prop.gcnArchName);

  l2_ctx* c = new l2_ctx();
  memcpy(c->hdr, cfg, sizeof(c->hdr));
  c->d = d; c->h = h; c->L = L; c->H = H; c->V = V; c->S = S; c->hs = d / H;
  c->shared = cfg[5] > 0;
  c->device = device;
  c->G = G; c->rank = rank;
  c->d_loc = d / G; c->h_loc = h / G; c->H_loc = H / G; c->V_loc = V / G;
  c->KVH = KVH; c->kvd = KVH * (d / H); c->kvd_loc = c->kvd / G; c->flags = flags;
  c->tune_R = dev_int("L2_TUNE_R", 0);
  c->tune_U = dev_int("L2_TUNE_U", 0);
  c->tune_nwaves = dev_int("L2_TUNE_NWAVES", 0);
  c->tune_gridcap = dev_int("L2_TUNE_GRIDCAP", 0);
  c->tune_rot = dev_int("L2_TUNE_ROT", 5);
  c->opt_packed = dev_int("L2_PACKED", 1);
  c->opt_one_copy = dev_int("L2_ONE_COPY", 1);
  c->opt_graph = env_int("L2_USE_GRAPH", (G == 1 && !hook_int("L2_TP_FORCE_COMM")) ? 1 : 0);
  c->profile_sync = dev_int("L2_PROFILE_SYNC", 0);
  c->opt_aql = env_int("L2_AQL", 1);
  c->opt_pollute = hook_int("L2_DEBUG_POLLUTE");
  c->debug_fail_aql = hook_int("L2_DEBUG_FAIL_AQL_RUN");
  c->opt_pos_check = env_int("L2_CHECK_POS", 0);
  // fences between the launches of a run on the library's own queue: no acquire, an agent-scope release, and an agent-scope acquire
  // on the first launch of every token (kernels.hip.h: the coherence rule).  L2_AQL_FENCE=<scope> sets both (1 = what a hipGraph
  // node carries); L2_AQL_ACQ / L2_AQL_REL / L2_AQL_TOKACQ one at a time (A/B, development switches)
  { const int f = dev_int("L2_AQL_FENCE", -1);
    c->aql_fence = dev_int("L2_AQL_ACQ", f < 0 ? 0 : f) + 4 * dev_int("L2_AQL_REL", f < 0 ? 1 : f) + 16 * dev_int("L2_AQL_TOKACQ", f < 0 ? 1 : 0); }
  c->p2p_fenced = env_int("L2_TP_FENCED", 0) ? 1 : 0;
  c->opt_push = dev_int("L2_TP_PUSH", 1) && !c->p2p_fenced;      // (the fenced form is the flag exchange)
  c->opt_awo = dev_int("L2_TP_ATTN_WO", 1);
  { const int ws = env_int("L2_TP_WAIT_S", 30); c->p2p_wait_ticks = (unsigned long long)(ws > 0 ? ws : 30) * 100000000ull; }

#define CK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { int rc_ = fail(L2_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); l2_destroy(c); return rc_; } } while (0)
  CK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  CK(hipEventCreate(&c->ev0));
  CK(hipEventCreate(&c->ev1));
  for (int k = 0; k < L2_T_COUNT; ++k) {
    const Slice s = tensor_slice(c, k);
    c->layers_of[k] = is_layered(k) ? L : 1;
    c->layer_elems[k] = s.rows * s.cols;
    if (k == L2_T_WCLS && c->shared) {
      if (G == 1) { c->w[k] = c->w[L2_T_TOKEN_EMBEDDING]; c->uploaded[k].assign(1, 0); continue; }
      // sharded classifier over a shared table: the rank's row slice of the (full) embedding table
      c->w[k] = c->w[L2_T_TOKEN_EMBEDDING] + (size_t)rank * c->V_loc * d;
      c->uploaded[k].assign(1, 0);
      continue;
    }
    CK(hipMalloc(&c->w[k], c->layer_elems[k] * c->layers_of[k] * sizeof(float)));
    c->uploaded[k].assign(c->layers_of[k], 0);
  }
  const size_t dl = c->d_loc, kvl = c->kvd_loc, kv = (size_t)L * S * kvl;
  CK(hipMalloc(&c->x, d * 4)); CK(hipMalloc(&c->xn, d * 4));
  CK(hipMalloc(&c->xb, dl * 4)); CK(hipMalloc(&c->xb2, d * 4));
  CK(hipMalloc(&c->hb, c->h_loc * 4)); CK(hipMalloc(&c->hb2, c->h_loc * 4));
  CK(hipMalloc(&c->q, dl * 4)); CK(hipMalloc(&c->k, kvl * 4)); CK(hipMalloc(&c->v, kvl * 4));
  CK(hipMalloc(&c->att, (size_t)c->H_loc * S * 4));
  c->tp_path = G > 1 || hook_int("L2_TP_FORCE_COMM");   // the latter: 1-rank communicator, exercises the RCCL path on one GPU
  // the gathered logits of a tensor-parallel rank are written by its peers (tp_p2p_gather_kernel): uncached memory
  if (c->tp_path) CK(hipExtMallocWithFlags((void**)&c->logits, (size_t)V * 4, hipDeviceMallocUncached));
  else CK(hipMalloc(&c->logits, (size_t)V * 4));
  if (c->tp_path) { CK(hipMalloc(&c->logits_loc, (size_t)c->V_loc * 4)); CK(hipMalloc(&c->partial, (size_t)d * 8)); }
  if (c->tp_path) {
    CK(hipMalloc(&c->awo_gran, (size_t)c->d_loc * 8)); CK(hipMemsetAsync(c->awo_gran, 0, (size_t)c->d_loc * 8, c->stream));
    CK(hipMalloc(&c->awo_ep, 16)); CK(hipMemsetAsync(c->awo_ep, 0, 16, c->stream));
  }
  else c->logits_loc = c->logits;
  CK(hipMalloc(&c->kc, kv * 4)); CK(hipMalloc(&c->vc, kv * 4));
  CK(hipMemsetAsync(c->kc, 0, kv * 4, c->stream)); CK(hipMemsetAsync(c->vc, 0, kv * 4, c->stream));
  float* zero[] = {c->x, c->xn, c->xb, c->xb2, c->hb, c->hb2, c->q, c->k, c->v};
  const size_t zn[] = {(size_t)d, (size_t)d, dl, (size_t)d, (size_t)c->h_loc, (size_t)c->h_loc, dl, kvl, kvl};
  for (int i = 0; i < 9; ++i) CK(hipMemsetAsync(zero[i], 0, zn[i] * 4, c->stream));
  CK(hipMemsetAsync(c->att, 0, (size_t)c->H_loc * S * 4, c->stream));
  CK(hipMemsetAsync(c->logits, 0, (size_t)V * 4, c->stream));
  // split attention scratch (sized for the largest split count)
  c->attn_splits_forced = dev_int("L2_ATTN_SPLITS", 0);
  c->split_rows = dev_int("L2_ATTN_SPLIT_ROWS", 144);
  c->split_rows_set = dev_int("L2_ATTN_SPLIT_ROWS", -1) >= 0;
  c->small_max = dev_int("L2_SMALL_MAX", 8 << 20);
  c->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  c->pf3 = dev_int("L2_PF3", 1);
  c->pf_attn = dev_int("L2_PF_ATTN", 1);
  if (c->attn_splits_forced > 64) c->attn_splits_forced = 64;
  {
    const size_t rec = ((size_t)c->hs + 2 + 15) & ~(size_t)15;
    const int maxs = c->attn_splits_forced > 8 ? c->attn_splits_forced : 8;
    CK(hipMalloc(&c->attn_part, (size_t)c->H_loc * maxs * rec * 8));
    CK(hipMalloc(&c->attn_counter, (size_t)c->H_loc * CTR_STRIDE * 4));
    CK(hipMemsetAsync(c->attn_counter, 0, (size_t)c->H_loc * CTR_STRIDE * 4, c->stream));
    CK(hipMalloc(&c->amax, 8 * 16 * 8));
    CK(hipMemsetAsync(c->amax, 0, 8 * 16 * 8, c->stream));
    CK(hipMalloc(&c->gran, ((size_t)c->d_loc + 2 * (size_t)c->kvd_loc) * 8));
    CK(hipMemsetAsync(c->gran, 0, ((size_t)c->d_loc + 2 * (size_t)c->kvd_loc) * 8, c->stream));
    CK(hipMalloc(&c->gran_ep, (size_t)c->H_loc * 4 + 16));
    CK(hipMemsetAsync(c->gran_ep, 0, (size_t)c->H_loc * 4 + 16, c->stream));
    CK(hipHostMalloc(&c->h_herr, sizeof(int), hipHostMallocMapped));
    *c->h_herr = 0;
    CK(hipHostGetDevicePointer((void**)&c->h_herr_dev, c->h_herr, 0));
    c->opt_fuse = dev_int("L2_FUSE_QKV_ATTN", 1);
    c->opt_fuse_splits = dev_int("L2_FUSE_SPLITS", 0);
    c->opt_fuse_four_waves = dev_int("L2_FUSE_FOUR_WAVES", 1);
    c->fuse_min_rows = dev_int("L2_FUSE_MIN_ROWS", c->d >= 512 ? 128 : 0);
  }
  if (c->opt_pollute) CK(hipMalloc(&c->pollute_sink, 64));
  CK(hipMalloc(&c->tokpos, 4 * sizeof(int)));
  CK(hipMemsetAsync(c->tokpos, 0, 4 * sizeof(int), c->stream));
  CK(hipMalloc(&c->d_tokens, (size_t)S * sizeof(int)));
  CK(hipHostMalloc(&c->h_tokpos, 4 * sizeof(int), hipHostMallocDefault));
  CK(hipHostGetDevicePointer((void**)&c->h_tokpos_dev, c->h_tokpos, 0));
  CK(hipHostMalloc(&c->h_logits, (size_t)V * 4, hipHostMallocMapped));
  memset(c->h_logits, 0, (size_t)V * 4);
  CK(hipHostGetDevicePointer((void**)&c->h_logits_dev, c->h_logits, 0));
  c->opt_zero_copy = env_int("L2_ZERO_COPY_LOGITS", G == 1 ? 1 : 0);
#ifdef L2_STAMPS
  CK(hipMalloc(&c->dbg, 8 * (66 * 108 + 64 * 2048)));
  CK(hipMemset(c->dbg, 0, 8 * (66 * 108 + 64 * 2048)));
#endif
  CK(hipStreamSynchronize(c->stream));
#undef CK
  if (c->tp_path && !hook_int("L2_TP_NO_COMM")) { const int rc_ = p2p_alloc(c); if (rc_) { l2_destroy(c); return rc_; } }
  if (G > 1 && nccl_id && !memcmp(nccl_id, L2_TP_SOLO_ID, sizeof(L2_TP_SOLO_ID) - 1)) {
    // shard timing (bench.py's tp_predicted): this rank's shard of the step alone on its GPU, the exchange kernels running against
    // its own inbox -- same launches, same stores, same flags, every wait satisfied at once.  The sums are G x the partial, so what
    // such a context decodes is meaningless; l2_tp_mode reports 5.
    if (!c->p2p_base) { l2_destroy(c); return fail(L2_E_COMM, "shard-timing context: no peer-to-peer inbox (G > %d, or L2_TP_ALLREDUCE=rccl)", (int)P2P_MAXG); }
    c->solo = true;
    for (int r = 0; r < G; ++r) p2p_set_peer(c, r, c->p2p_base, c->logits);
    c->p2p = true; c->p2p_peers_ready = true; c->p2p_synced = true;
    { const int rc_ = p2p_publish_table(c); if (rc_) { l2_destroy(c); return rc_; } }
  } else if (G > 1 && hook_int("L2_TP_NO_COMM")) {
    // shard-layout tests on a single GPU: the slices are real, the communicator is absent and every
    // forward on this context fails with L2_E_COMM
  } else if (G > 1 && hook_int("L2_TP_LOOPBACK")) {
    if (G > 16 || !nccl_id) { l2_destroy(c); return fail(L2_E_ARG, "loopback groups need an id and at most 16 ranks"); }
    if (hipMalloc(&c->loop_tmp, (size_t)d * 8) != hipSuccess) { l2_destroy(c); return fail(L2_E_HIP, "hipMalloc failed"); }
    std::lock_guard<std::mutex> lk(g_loop_mu);
    auto& grp = g_loop_groups[std::string((const char*)nccl_id, 128)];
    if (!grp) { grp = std::make_shared<LoopGroup>(); grp->G = G; }
    if (grp->G != G) { l2_destroy(c); return fail(L2_E_ARG, "loopback group size mismatch"); }
    c->loop = grp;
    grp->p2p_base[rank] = c->p2p_base; grp->p2p_logits[rank] = c->logits;
    c->p2p = c->p2p_base != nullptr;          // peers resolved at the first step, once every rank has registered
  } else if (G > 1 && ipc_dir_env()) {
    c->ipc_dir = ipc_dir_env();
    const int rc = p2p_connect_ipc(c);
    if (rc) { l2_destroy(c); return rc; }
  } else if (c->tp_path) {
    int rc = rccl_bind();
    if (rc) { l2_destroy(c); return rc; }
    nccl_uid uid;
    if (nccl_id) memcpy(&uid, nccl_id, sizeof(uid));
    else if (g_rccl.GetUniqueId(&uid) != 0) { l2_destroy(c); return fail(L2_E_COMM, "ncclGetUniqueId failed"); }
    // RCCL 2.26 prints a version banner to stdout at init; the host's stdout is the generated text
    // (llama2.ts:500) or bench.py's one JSON line, so the banner is sent to stderr instead
    fflush(stdout);
    const int saved_out = dup(1);
    if (saved_out >= 0) dup2(2, 1);
    int r = g_rccl.CommInitRank(&c->comm, G, uid, rank);
    fflush(stdout);
    if (saved_out >= 0) { dup2(saved_out, 1); close(saved_out); }
    if (r != 0) { l2_destroy(c); return fail(L2_E_COMM, "ncclCommInitRank failed: %d", r); }
    rc = p2p_connect_ipc(c);
    if (rc) { l2_destroy(c); return rc; }
  }
  if (c->p2p && !c->loop && !getenv("L2_USE_GRAPH")) c->opt_graph = 1;   // nothing but kernels in the step: one hipGraph per token
  if (c->tp_path && !c->p2p && !c->loop && c->comm && !getenv("L2_USE_GRAPH")) { c->opt_graph = 1; c->rccl_graph = true; }   // RCCL collectives captured with the step (eager if capture is refused)
  if (c->loop) c->opt_graph = 0;                                          // host barriers between the halves of an exchange
  *out = c;
  return L2_OK;
}

extern "C" int l2_create(const int32_t cfg[7], int device, l2_ctx** out) { return create_impl(cfg, device, 0, 1, nullptr, out); }

// RoPE tables the way llama2.c's run.c computes them per position when the checkpoint carries none (newer export
// versions): freq = 1 / 10000^(j / (head_size / 2))... in fp32, angle = pos * freq, cosf / sinf.
static int generate_rope(l2_ctx* c) {
  const int hs2 = c->hs / 2;
  std::vector<float> re((size_t)c->S * hs2), im((size_t)c->S * hs2);
  for (int t = 0; t < c->S; ++t)
    for (int j = 0; j < hs2; ++j) {
      const float freq = 1.0f / powf(10000.0f, (float)(2 * j) / (float)c->hs);
      const float val = (float)t * freq;
      re[(size_t)t * hs2 + j] = cosf(val);
      im[(size_t)t * hs2 + j] = sinf(val);
    }
  HIPCHK(hipMemcpy(c->w[L2_T_FREQ_REAL], re.data(), re.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->w[L2_T_FREQ_IMAG], im.data(), im.size() * 4, hipMemcpyHostToDevice));
  c->uploaded[L2_T_FREQ_REAL][0] = 1; c->uploaded[L2_T_FREQ_IMAG][0] = 1;
  return L2_OK;
}

extern "C" int l2_create_ex(const int32_t cfg[7], int device, unsigned flags, l2_ctx** out) {
  if (flags & ~(unsigned)(L2_F_GQA | L2_F_GENERATE_ROPE)) return fail(L2_E_ARG, "unknown flag bits 0x%x", flags);
  int rc = create_impl(cfg, device, 0, 1, nullptr, out, flags);
  if (rc) return rc;
  if (flags & L2_F_GENERATE_ROPE) { rc = generate_rope(*out); if (rc) { l2_destroy(*out); *out = nullptr; return rc; } }
  return L2_OK;
}

extern "C" int l2_tp_unique_id(void* id_out_128) {
  if (!id_out_128) return fail(L2_E_ARG, "null argument");
  int rc = rccl_bind();
  if (rc) return rc;
  nccl_uid uid;
  NCCLCHK(g_rccl.GetUniqueId(&uid));
  memcpy(id_out_128, &uid, sizeof(uid));
  return L2_OK;
}

extern "C" int l2_tp_mode(l2_ctx* c) {
  if (!c || !c->tp_path) return 0;
  if (c->solo) return 5;
  if (c->loop) return 4;
  if (c->p2p) return 3;
  return (c->rccl_graph && c->opt_graph) ? 2 : 1;
}

extern "C" int l2_create_tp(const int32_t cfg[7], int device, int tp_rank, int tp_size, const void* nccl_id, l2_ctx** out) {
  if (tp_size > 1 && !nccl_id) return fail(L2_E_ARG, "nccl_id required for tp_size > 1");
  return create_impl(cfg, device, tp_rank, tp_size, nccl_id, out);
}

// ------------------------------------------------------------------------------------------------
static int check_tensor(l2_ctx* c, int kind, int layer, int* layer_idx) {
  if (!c) return fail(L2_E_ARG, "null context");
  if (kind < 0 || kind >= L2_T_COUNT) return fail(L2_E_ARG, "tensor kind %d out of range", kind);
  if (kind == L2_T_WCLS && c->shared) return fail(L2_E_ARG, "wcls aliases token_embedding_table for this checkpoint (vocab_size > 0, llama2.ts:127)");
  if (is_layered(kind)) {
    if (layer < 0 || layer >= c->L) return fail(L2_E_ARG, "layer %d out of range for tensor kind %d", layer, kind);
    *layer_idx = layer;
  } else {
    if (layer != -1 && layer != 0) return fail(L2_E_ARG, "tensor kind %d is not per-layer", kind);
    *layer_idx = 0;
  }
  return L2_OK;
}

extern "C" int l2_upload(l2_ctx* c, int kind, int layer, const float* host, size_t n_floats) {
  int li = 0;
  int rc = check_tensor(c, kind, layer, &li);
  if (rc) return rc;
  if (!host) return fail(L2_E_ARG, "null host pointer");
  const Slice s = tensor_slice(c, kind);
  if (n_floats != s.full_rows * s.full_cols)
    return fail(L2_E_ARG, "tensor kind %d expects %zu floats, got %zu", kind, s.full_rows * s.full_cols, n_floats);
  HIPCHK(hipSetDevice(c->device));
  if (c->released[kind]) { rc = ensure_rowmajor(c, true); if (rc) return rc; }     // the row-major tensor was given back after packing: bring it back
  float* dst = c->w[kind] + c->layer_elems[kind] * (size_t)li;
  const float* src = host + s.row0 * s.full_cols + s.col0;
  if (s.cols == s.full_cols) {
    HIPCHK(hipMemcpy(dst, src, s.rows * s.cols * sizeof(float), hipMemcpyHostToDevice));
  } else {  // column slice, repacked contiguous (tensor-parallel wo / w2)
    HIPCHK(hipMemcpy2D(dst, s.cols * sizeof(float), src, s.full_cols * sizeof(float), s.cols * sizeof(float), s.rows, hipMemcpyHostToDevice));
  }
  c->uploaded[kind][li] = 1;
  mark_dirty(c, kind, li);
  return L2_OK;
}

#include "loader.hip.h"
#include "launch.hip.h"
#include "tp_exchange.hip.h"


// Enqueue one transformer() call (llama2.ts:205-303) reading {token,pos} from device memory.
// PhaseArgs of each phase of layer l
static const float* packed_of(const l2_ctx* c, int mode, int l) {
  const l2_ctx::Packed& p = c->packed[mode];
  return (c->packed_valid && p.buf) ? p.buf + p.layer_elems * (size_t)l : nullptr;
}
// layer l of a per-layer tensor, or null once its row-major copy has been given back (one copy of the weights: the launch then
// reads the repacked copy, and a null here makes a launch that cannot fail loudly instead of reading freed memory)
static const float* wptr(const l2_ctx* c, int kind, int l) { return c->w[kind] ? c->w[kind] + c->layer_elems[kind] * (size_t)l : nullptr; }
static PhaseArgs base_args(const l2_ctx* c) {
  PhaseArgs a;
  memset(&a, 0, sizeof(a));
  a.tokpos = c->tokpos; a.fr = c->w[L2_T_FREQ_REAL]; a.fi = c->w[L2_T_FREQ_IMAG];
  a.head_size = c->hs; a.dim = c->d; a.inv_n = 1.0 / (double)c->d;   // every normed phase has n == dim
  return a;
}
static PhaseArgs qkv_args(const l2_ctx* c, int l) {   // rmsnorm + q,k,v GEMVs + RoPE + KV-cache store (llama2.ts:216-240)
  PhaseArgs a = base_args(c);
  const size_t loff = (size_t)l * c->S * c->kvd_loc;
  a.w0 = wptr(c, L2_T_WQ, l);
  a.w1 = wptr(c, L2_T_WK, l);
  a.w2 = wptr(c, L2_T_WV, l);
  a.in = c->x; a.emb = (l == 0) ? c->w[L2_T_TOKEN_EMBEDDING] : nullptr;
  a.rmsw = c->w[L2_T_RMS_ATT] + (size_t)c->d * l;
  a.out = c->q; a.out_k = c->kc + loff; a.out_v = c->vc + loff;
  if (c->opt_keep_state) { a.aux = c->k; a.aux2 = c->v; }     // RunState.k / v: the cache rows are what attention reads
  a.n = c->d; a.rows = c->d_loc + 2 * c->kvd_loc; a.dim = c->d_loc; a.kv_dim = c->kvd_loc;
  a.wp = packed_of(c, MODE_QKV, l);
  return a;
}
static PhaseArgs wo_args(const l2_ctx* c, int l) {    // wo GEMV + residual (llama2.ts:270-273)
  PhaseArgs a = base_args(c);
  a.w0 = wptr(c, L2_T_WO, l);
  a.in = c->xb; a.emb = (l == 0) ? c->w[L2_T_TOKEN_EMBEDDING] : nullptr; a.res = c->x; a.out = c->x; a.aux = c->opt_keep_state ? c->xb2 : nullptr;
  a.n = c->d_loc; a.rows = c->d;
  if (c->tp_path) a.partial = c->partial;
  if (p2p_pushing(c)) a.push = c->tp_push;      // rows go straight into the peers' inboxes (tp_exchange.hip.h)
  a.wp = packed_of(c, MODE_WO, l);
  return a;
}
static PhaseArgs w13_args(const l2_ctx* c, int l) {   // rmsnorm + w1,w3 GEMVs + SwiGLU (llama2.ts:276-289)
  PhaseArgs a = base_args(c);
  a.w0 = wptr(c, L2_T_W1, l);
  a.w1 = wptr(c, L2_T_W3, l);
  a.in = c->x; a.rmsw = c->w[L2_T_RMS_FFN] + (size_t)c->d * l;
  a.out = c->hb; a.aux = c->opt_keep_state ? c->hb2 : nullptr;
  a.n = c->d; a.rows = c->h_loc;
  a.wp = packed_of(c, MODE_W13, l);
  return a;
}
static PhaseArgs w2_args(const l2_ctx* c, int l) {    // w2 GEMV + residual (llama2.ts:292-295)
  PhaseArgs a = base_args(c);
  a.w0 = wptr(c, L2_T_W2, l);
  a.in = c->hb; a.res = c->x; a.out = c->x; a.aux = (c->tp_path || !c->opt_keep_state) ? nullptr : c->xb;
  a.n = c->h_loc; a.rows = c->d;
  if (c->tp_path) a.partial = c->partial;
  if (p2p_pushing(c)) a.push = c->tp_push;
  a.wp = packed_of(c, MODE_W2, l);
  return a;
}
static PhaseArgs cls_args(const l2_ctx* c, bool to_host) {   // final rmsnorm + classifier (llama2.ts:299-302)
  PhaseArgs a = base_args(c);
  a.w0 = c->w[L2_T_WCLS];
  a.in = c->x; a.rmsw = c->w[L2_T_RMS_FINAL]; a.out = c->logits_loc; a.aux = c->opt_keep_state ? c->xn : nullptr;
  a.aux2 = (to_host && c->opt_zero_copy && !c->tp_path) ? c->h_logits_dev : nullptr;
  a.n = c->d; a.rows = c->V_loc;
  a.wp = packed_of(c, MODE_CLS, 0);
  return a;
}

// Build (or rebuild after an upload) the repacked copies of the matrices the streaming form reads: kernels.hip.h, phase_body PK.
// A phase is packed when its launch is the streaming form with two 64-lane sub-batches per batch and more than one batch per row
// (n % 256 == 0, n > 512: every layer matrix and the classifier of Llama-2-7B; the small models' phases take the latency form).
// No memory for the second copy: that phase streams the row-major tensor as before.
template <int MODE>
static int pack_phase(l2_ctx* c, int layers, PhaseArgs (*args_of)(const l2_ctx*, int)) {
  l2_ctx::Packed& p = c->packed[MODE];
  PhaseArgs a0 = args_of(c, 0);
  const bool small = use_small(c, MODE, a0.rows, a0.n);
  const Geo g = pick_geo(c, MODE, a0.rows, a0.n, a0.dim);
  const int n4 = a0.n / 4, rpg = (MODE == MODE_W13) ? 1 : 2, groups = (a0.rows + rpg - 1) / rpg;
  const bool want = c->opt_packed && !small && g.vec && g.U == 2 && n4 % 64 == 0 && n4 > 128;
  // (a copy that goes away takes the captured graphs with it: they hold its address)
  if (!want) { if (p.buf) { destroy_graphs(c); hipFree(p.buf); p.buf = nullptr; } p.dirty.clear(); return L2_OK; }
  const size_t elems = (size_t)groups * 2 * a0.n;
  if (p.buf && !(p.layer_elems == elems && p.U == g.U && p.nwaves == g.nwaves && p.grid == g.grid)) { destroy_graphs(c); hipFree(p.buf); p.buf = nullptr; }
  if (!p.buf) {
    if (hipMalloc(&p.buf, elems * layers * sizeof(float)) != hipSuccess) {
      (void)hipGetLastError(); p.buf = nullptr; p.dirty.clear();
      if (!p.noted) {   // said once: the phase still runs (it streams the row-major tensor), only slower
        fprintf(stderr, "libllama2hip: no device memory for the repacked copy of phase %d (%zu MiB): streaming the row-major tensors instead\n", MODE, elems * layers * sizeof(float) >> 20);
        p.noted = true;
      }
      return L2_OK;
    }
    p.layer_elems = elems; p.U = g.U; p.nwaves = g.nwaves; p.grid = g.grid;
    p.dirty.assign(layers, 1);
  }
  if ((int)p.dirty.size() != layers) p.dirty.assign(layers, 1);
  for (int l = 0; l < layers; ++l) {
    if (!p.dirty[l]) continue;       // only the slices whose matrices were uploaded since the last step
    const PhaseArgs a = args_of(c, l);
    hipLaunchKernelGGL((pack_kernel<MODE, 2>), dim3(groups, (2 * n4 + 255) / 256), dim3(256), 0, c->stream, a, reinterpret_cast<f4*>(p.buf + elems * (size_t)l), g.U, g.grid * g.nwaves);
    p.dirty[l] = 0;
  }
  LCHK(hipGetLastError());
  return L2_OK;
}
static PhaseArgs cls_args_l(const l2_ctx* c, int) { return cls_args(c, false); }
// tensor kinds whose only steady-state reader is phase `mode` (the shared classifier's matrix is the embedding table: it stays)
static int phase_kinds(const l2_ctx* c, int mode, int (&kinds)[3]) {
  switch (mode) {
    case MODE_QKV: kinds[0] = L2_T_WQ; kinds[1] = L2_T_WK; kinds[2] = L2_T_WV; return 3;
    case MODE_WO: kinds[0] = L2_T_WO; return 1;
    case MODE_W13: kinds[0] = L2_T_W1; kinds[1] = L2_T_W3; return 2;
    case MODE_W2: kinds[0] = L2_T_W2; return 1;
    default: if (c->shared) return 0; kinds[0] = L2_T_WCLS; return 1;
  }
}

static bool phase_has_dirty(const l2_ctx* c, int m) { for (uint8_t b : c->packed[m].dirty) if (b) return true; return false; }

// ONE copy of the weights: a packed phase reads nothing but its repacked copy (decode, prompt ingestion), so the row-major tensors it
// was built from go back to the allocator (Llama-2-7B: 25 GB).  Captured graphs keep working: their launches carry the repacked
// addresses, which do not move.
static int release_packed_sources(l2_ctx* c) {
  c->rerelease = false;
  if (!c->opt_one_copy) return L2_OK;
  for (int m = 0; m < 5; ++m) {
    if (!c->packed[m].buf) continue;
    int kinds[3];
    const int nk = phase_kinds(c, m, kinds);
    for (int i = 0; i < nk; ++i) {
      const int k = kinds[i];
      if (c->w[k] && !c->released[k]) { HIPCHK(hipFree(c->w[k])); c->w[k] = nullptr; c->released[k] = true; }
    }
  }
  return L2_OK;
}

static int ensure_packed(l2_ctx* c) {
  // nothing to repack; row-major tensors that l2_read_tensor brought back for a parity read are given away again
  if (c->packed_valid) return c->rerelease ? release_packed_sources(c) : L2_OK;
  // A dirty slice is rebuilt from the row-major tensors of ITS phase: those must be there (an upload brings them back before it marks
  // anything).  Other phases may have given theirs away -- a shared classifier's matrix is the embedding table, which never goes:
  // re-uploading it after a step dirties the classifier phase only.
  for (int m = 0; m < 5; ++m) {
    if (!phase_has_dirty(c, m)) continue;
    int kinds[3];
    const int nk = phase_kinds(c, m, kinds);
    for (int i = 0; i < nk; ++i)
      if (c->released[kinds[i]]) return fail(L2_E_STATE, "internal: tensor kind %d is marked dirty while its row-major copy is gone", kinds[i]);
  }
  int rc;
  if ((rc = pack_phase<MODE_QKV>(c, c->L, qkv_args))) return rc;
  if ((rc = pack_phase<MODE_WO>(c, c->L, wo_args))) return rc;
  if ((rc = pack_phase<MODE_W13>(c, c->L, w13_args))) return rc;
  if ((rc = pack_phase<MODE_W2>(c, c->L, w2_args))) return rc;
  if ((rc = pack_phase<MODE_CLS>(c, 1, cls_args_l))) return rc;
  HIPCHK(hipStreamSynchronize(c->stream));
  c->packed_valid = true;
  return release_packed_sources(c);
}

template <int MODE>
static int unpack_phase(l2_ctx* c, int layers, PhaseArgs (*args_of)(const l2_ctx*, int)) {
  const l2_ctx::Packed& p = c->packed[MODE];
  int kinds[3];
  const int nk = phase_kinds(c, MODE, kinds);
  bool any = false;
  for (int i = 0; i < nk; ++i) any = any || c->released[kinds[i]];
  if (!any) return L2_OK;
  if (!p.buf) return fail(L2_E_STATE, "internal: phase %d has neither a row-major nor a repacked copy", MODE);
  if (phase_has_dirty(c, MODE)) return fail(L2_E_STATE, "internal: the repacked copy of phase %d is stale and its row-major tensors are gone", MODE);
  const PhaseArgs a0 = args_of(c, 0);
  const int n4 = a0.n / 4, rpg = (MODE == MODE_W13) ? 1 : 2, groups = (a0.rows + rpg - 1) / rpg;
  for (int l = 0; l < layers; ++l) {
    const PhaseArgs a = args_of(c, l);
    hipLaunchKernelGGL((unpack_kernel<MODE, 2>), dim3(groups, (2 * n4 + 255) / 256), dim3(256), 0, c->stream, a, reinterpret_cast<const f4*>(p.buf + p.layer_elems * (size_t)l), p.U, p.grid * p.nwaves);
  }
  LCHK(hipGetLastError());
  return L2_OK;
}

// Bring back the row-major tensors that were given away after packing (`unpack`: with their contents, out of the repacked copies;
// false: the caller overwrites all of them anyway).  They are given away again after the next repack.
static int ensure_rowmajor(l2_ctx* c, bool unpack) {
  bool any = false;
  for (int k = 0; k < L2_T_COUNT; ++k) {
    if (!c->released[k]) continue;
    any = true;
    if (hipMalloc(&c->w[k], c->layer_elems[k] * c->layers_of[k] * sizeof(float)) != hipSuccess) {
      (void)hipGetLastError();
      c->w[k] = nullptr;
      for (int j = 0; j < k; ++j) if (c->released[j] && c->w[j]) { hipFree(c->w[j]); c->w[j] = nullptr; }      // all or nothing: the repacked copies stay the only ones
      return fail(L2_E_HIP, "no device memory to bring the row-major copy of tensor kind %d back (one copy of the weights is held; a re-upload or l2_read_tensor needs a second one for a moment)", k);
    }
  }
  if (!any) return L2_OK;
  int rc = L2_OK;
  if (unpack) {
    // (a phase whose tensors are gone is never stale: unpack_phase checks; another phase may be -- a shared classifier after the
    //  embedding table was uploaded again -- and that one still has its source)
    if ((rc = unpack_phase<MODE_QKV>(c, c->L, qkv_args))) return rc;
    if ((rc = unpack_phase<MODE_WO>(c, c->L, wo_args))) return rc;
    if ((rc = unpack_phase<MODE_W13>(c, c->L, w13_args))) return rc;
    if ((rc = unpack_phase<MODE_W2>(c, c->L, w2_args))) return rc;
    if ((rc = unpack_phase<MODE_CLS>(c, 1, cls_args_l))) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  for (int k = 0; k < L2_T_COUNT; ++k) c->released[k] = false;
  return L2_OK;
}

// fold: 0 logits only; 1 the classifier also folds its best logit into the argmax keys (the sampled loop's maximum); 2 the greedy pick
// folded into the step (kernels.hip.h: greedy_token_from_keys) -- layer 0's q / k / v and wo launches and the classifier launch share it
static bool greedy_folds(const l2_ctx* c) { return !c->tp_path && c->amax && c->d % 4 == 0 && c->h % 4 == 0; }      // (the scalar kernels of n % 4 != 0 shapes keep the pick's own launch)
static int enqueue_forward_impl(l2_ctx* c, hipStream_t st, bool to_host, int fold = 0) {
  auto greedy = [&](PhaseArgs& a) { if (fold == 2) { a.amax = c->amax; a.tok_out = c->d_tokens; } };
  for (int l = 0; l < c->L; ++l) {
    PhaseArgs a = qkv_args(c, l);
    if (l == 0) greedy(a);
    if (fused_qkv_attn_ok(c)) {
      LCHK(launch_qkv_attn(c, a, l, st));        // the head-local edge inside one launch (attention.hip.h: qkv_attn_small_kernel)
    } else {
      LCHK(launch_phase<MODE_QKV>(c, a, st));
      if (attn_wo_ok(c)) {
        LCHK(launch_attn_wo(c, l, wo_args(c, l), st));      // attention and the rank's wo shard in ONE launch (attention.hip.h: attn_wo_kernel)
      } else {
        LCHK(launch_attn(c, l, st));
        a = wo_args(c, l);
        if (l == 0) greedy(a);
        LCHK(launch_phase<MODE_WO>(c, a, st));
      }
    }
    if (fused_qkv_attn_ok(c)) {
      a = wo_args(c, l);
      if (l == 0) greedy(a);
      LCHK(launch_phase<MODE_WO>(c, a, st));
    }
    if (c->p2p) {
      const int rc_ = p2p_reduce(c, st, (l == 0) ? c->w[L2_T_TOKEN_EMBEDDING] : nullptr, c->opt_keep_state ? c->xb2 : nullptr, attn_wo_ok(c) ? c->awo_ep : nullptr);
      if (rc_) return rc_;
    } else if (c->tp_path) {
      { const int rc_ = tp_all_reduce(c, st); if (rc_) return rc_; }
      hipLaunchKernelGGL(tp_residual_kernel, dim3((c->d + 255) / 256), dim3(256), 0, st, c->x, (l == 0) ? c->w[L2_T_TOKEN_EMBEDDING] : nullptr, c->partial, c->opt_keep_state ? c->xb2 : nullptr, c->tokpos, c->d);
      LCHK(hipGetLastError());
    }
    a = w13_args(c, l);
    LCHK(launch_phase<MODE_W13>(c, a, st));      // the in-situ probe attaches its events to this dispatch (launch_probed)
    a = w2_args(c, l);
    LCHK(launch_phase<MODE_W2>(c, a, st));
    if (c->p2p) {
      const int rc_ = p2p_reduce(c, st, nullptr, nullptr, nullptr);
      if (rc_) return rc_;
    } else if (c->tp_path) {
      { const int rc_ = tp_all_reduce(c, st); if (rc_) return rc_; }
      hipLaunchKernelGGL(tp_residual_kernel, dim3((c->d + 255) / 256), dim3(256), 0, st, c->x, nullptr, c->partial, nullptr, c->tokpos, c->d);
      LCHK(hipGetLastError());
    }
  }
  PhaseArgs a = cls_args(c, to_host);
  if (fold) a.amax = c->amax;
  if (fold == 2) a.tok_out = c->d_tokens;
  LCHK(launch_phase<MODE_CLS>(c, a, st));
  if (c->p2p) {
    const dim3 grid(p2p_grid(c->V_loc));
    if (!c->loop) l2_launch(c, tp_p2p_gather_kernel<0>, grid, dim3(256), 0, st, p2p_args(c, c->V_loc), (const float*)c->logits_loc);
    else {
      hipLaunchKernelGGL(tp_p2p_gather_kernel<1>, grid, dim3(256), 0, st, p2p_args(c, c->V_loc), c->logits_loc);
      HIPCHK(hipStreamSynchronize(st));
      if (!c->loop->wait()) return fail(L2_E_COMM, "loopback gather: a rank never arrived");
      hipLaunchKernelGGL(tp_p2p_gather_kernel<2>, grid, dim3(256), 0, st, p2p_args(c, c->V_loc), c->logits_loc);
    }
    LCHK(hipGetLastError());
  } else if (c->tp_path) { const int rc_ = tp_all_gather_logits(c, st); if (rc_) return rc_; }
  return L2_OK;
}

static int enqueue_forward_host(l2_ctx* c, hipStream_t st) { return enqueue_forward_impl(c, st, true); }
// ... behind the launch that fetches {token, pos} of the call from pinned host memory (the blocking call on the library's own queue)
static int enqueue_forward_call(l2_ctx* c, hipStream_t st) {
  l2_launch(c, set_tokpos_kernel, dim3(1), dim3(64), 0, st, (const int*)c->h_tokpos_dev, c->tokpos);
  return enqueue_forward_impl(c, st, true);
}

static int ensure_ready(l2_ctx* c) {
  // everything below may allocate and launch (ensure_packed): the context's device first, whatever the calling thread had current
  HIPCHK(hipSetDevice(c->device));
  if (c->broken) return fail(L2_E_COMM, "tensor-parallel context is unusable: an earlier peer-to-peer exchange timed out");
  if (c->tp_path && !c->comm && !c->loop && !c->solo && !(c->p2p && !c->ipc_dir.empty())) return fail(L2_E_COMM, "tensor-parallel context has no communicator (L2_TP_NO_COMM)");
  if (c->loop && c->p2p && !c->p2p_peers_ready) {
    if (!c->loop->wait()) return fail(L2_E_COMM, "loopback group: a rank never arrived");
    for (int r = 0; r < c->G; ++r) {
      if (!c->loop->p2p_base[r]) return fail(L2_E_COMM, "loopback group: rank %d has no peer-to-peer inbox", r);
      p2p_set_peer(c, r, c->loop->p2p_base[r], c->loop->p2p_logits[r]);
    }
    c->p2p_peers_ready = true;
  }
  for (int k = 0; k < L2_T_COUNT; ++k) {
    if (k == L2_T_WCLS && c->shared) continue;
    for (size_t l = 0; l < c->uploaded[k].size(); ++l)
      if (!c->uploaded[k][l]) return fail(L2_E_STATE, "tensor kind %d layer %zu was never uploaded", k, l);
  }
  if (c->p2p && !c->p2p_synced) { const int rc_ = p2p_first_sync(c); if (rc_) return rc_; }
  return ensure_packed(c);
}

static int enqueue_greedy(l2_ctx* c, hipStream_t st) {  // device-resident step: forward, argmax, advance
  // the classifier's workgroups fold their logits into 8 argmax keys, one wave finishes; a tensor-parallel rank has
  // only its slice of the logits before the all-gather and takes the maximum over the gathered vector instead
  // ... or, on one GPU with vector-form phases, the pick folded into the next token's first launch (no launch of its own; the run's last
  // pick: enqueue_greedy_last)
  const bool fold = !c->tp_path;
  int rc = enqueue_forward_impl(c, st, false, greedy_folds(c) ? 2 : (fold ? 1 : 0));
  if (rc) return rc;
  if (greedy_folds(c)) return L2_OK;
  if (fold) l2_launch(c, argmax_finish_kernel, dim3(1), dim3(64), 0, st, c->amax, c->tokpos, c->d_tokens);
  else l2_launch(c, argmax_advance_kernel, dim3(1), dim3(1024), 0, st, (const float*)c->logits, c->V, c->tokpos, c->d_tokens);
  LCHK(hipGetLastError());
  return L2_OK;
}
static int enqueue_greedy_last(l2_ctx* c, hipStream_t st) {      // once per run: the last token's pick (kernels.hip.h: argmax_last_kernel)
  if (!greedy_folds(c)) return L2_OK;
  l2_launch(c, argmax_last_kernel, dim3(1), dim3(64), 0, st, c->amax, c->tokpos, c->d_tokens);
  LCHK(hipGetLastError());
  return L2_OK;
}

static int enqueue_sample(l2_ctx* c, hipStream_t st) {  // device-resident sampled step: forward, temperature/softmax/sample(_topp), advance
  int rc = enqueue_forward_impl(c, st, false, c->samp_amax ? 1 : 0);
  if (rc) return rc;
  if (c->aql_rec) {
    // the sampler's launches (their own translation unit) are handed over by its recorder; its kernels read logits, keys and
    // {token, pos} through their CU's caches: each of them acquires (AQL_LAUNCH_ACQUIRES)
    l2s::set_recorder([](void* user, const void* fn, dim3 grid, dim3 block, size_t lds, hipStream_t s2, const void* args, size_t nb) -> bool {
      l2_ctx* m = (l2_ctx*)user;
      const char* name = hipKernelNameRefByPtr(fn, s2);
      const unsigned g[3] = {grid.x, grid.y, grid.z}, b[3] = {block.x, block.y, block.z};
      return name && !aql_record(m->aql, m->aql_rec, name, g, b, (unsigned)lds, args, nb, AQL_LAUNCH_ACQUIRES);
    }, c);
    const hipError_t e = l2s::enqueue(c->samp, c->logits, c->samp_mode == 1, c->tokpos, c->d_tokens, c->samp_amax ? c->amax : nullptr, st);
    if (l2s::recorder_failed()) c->aql_rec_failed = true;
    l2s::set_recorder(nullptr, nullptr);
    LCHK(e);
    return L2_OK;
  }
  LCHK(l2s::enqueue(c->samp, c->logits, c->samp_mode == 1, c->tokpos, c->d_tokens, c->samp_amax ? c->amax : nullptr, st));
  return L2_OK;
}

enum { L2_RUN_EAGER = 1 };   // capture(): the step cannot be captured on this context (RCCL collectives that refuse capture): launch it eagerly from now on

static int capture_impl(l2_ctx* c, int (*enq)(l2_ctx*, hipStream_t), hipGraphExec_t* out) {
  hipGraph_t graph = nullptr;
  LCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
  int rc = enq(c, c->stream);
  hipError_t e = hipStreamEndCapture(c->stream, &graph);
  if (rc) { if (graph) hipGraphDestroy(graph); return rc; }
  if (e != hipSuccess) return fail(L2_E_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
  e = hipGraphInstantiate(out, graph, nullptr, nullptr, 0);
  hipGraphDestroy(graph);
  if (e != hipSuccess) return fail(L2_E_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
  return L2_OK;
}

// A tensor-parallel step whose exchanges are RCCL collectives is captured like any other (one graph launch per token instead
// of ~290 host-issued launches); if this RCCL refuses stream capture the context falls back to eager launches for good.
static int capture(l2_ctx* c, int (*enq)(l2_ctx*, hipStream_t), hipGraphExec_t* out) {
  const int rc = capture_impl(c, enq, out);
  if (rc && c->rccl_graph) {
    (void)hipGetLastError();
    *out = nullptr;
    c->rccl_graph = false; c->opt_graph = 0;
    destroy_graphs(c);
    hipStreamSynchronize(c->stream);
    return L2_RUN_EAGER;
  }
  return rc;
}

// ---- the greedy loop on the library's own AQL queue (aql_queue.h) ---------------------------------------------------------------
// One GPU, the step a pure chain of kernels (no RCCL, no loopback barriers), graphs enabled, no probe: everything a hipGraph would
// replay is replayed as hand-written packets instead.
static bool aql_usable(const l2_ctx* c) {
  // (a tensor-parallel rank: only when its exchanges are kernels too -- the one-shot peer-to-peer form; RCCL collectives are the runtime's)
  if (c->tp_path && !(c->p2p && !c->loop && c->p2p_peers_ready)) return false;
  return c->opt_aql && c->opt_graph && !c->loop && !c->probe_on && !c->profile_sync && (c->aql || !c->aql_tried);
}

// A profiler's tool library wraps every HSA queue of the process in an intercept queue and rewrites the packets it finds there;
// rocprofv3 (ROCm 7.2) crashes on packets written by hand (profiles/r05/aql_under_rocprof.txt).  Under one, the loop stays with HIP launches.
static bool hsa_tools_loaded() {
  for (const char* k : {"HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES", "ROCP_TOOL_LIBRARY", "ROCPROFILER_REGISTER_FORCE_LOAD"}) { const char* v = getenv(k); if (v && *v) return true; }
  const char* pre = getenv("LD_PRELOAD");
  return pre && (strstr(pre, "rocprof") || strstr(pre, "roctracer"));
}

// (tests, no GPU needed) how many gfx950 code objects the AQL queue would load out of this library's own file
extern "C" int l2_debug_aql_code_objects(void) {
  Dl_info info;
  if (!dladdr((const void*)&l2_abi_version, &info) || !info.dli_fname) return -1;
  return aql_count_code_objects(info.dli_fname);
}

static int aql_open(l2_ctx* c) {
  c->aql_tried = true;
  if (hsa_tools_loaded()) { c->aql_note = "a profiler's tool library intercepts the HSA queues of this process"; return -1; }
  char bus[64] = "";
  if (hipDeviceGetPCIBusId(bus, sizeof(bus), c->device) != hipSuccess) { c->aql_note = "hipDeviceGetPCIBusId failed"; return -1; }
  unsigned dom = 0, b = 0, d = 0, f = 0;
  if (sscanf(bus, "%x:%x:%x.%x", &dom, &b, &d, &f) != 4) { c->aql_note = std::string("cannot parse PCI bus id ") + bus; return -1; }
  Dl_info info;
  if (!dladdr((const void*)&l2_abi_version, &info) || !info.dli_fname) { c->aql_note = "dladdr: the library cannot find its own file"; return -1; }
  char err[512] = "";
  c->aql = aql_create((int)dom, (int)b, (int)d, (int)f, info.dli_fname, err, sizeof(err));
  if (!c->aql) { c->aql_note = err; return -1; }
  return 0;
}

static int enqueue_greedy(l2_ctx* c, hipStream_t st);
static int enqueue_greedy_last(l2_ctx* c, hipStream_t st);

// Record the step of level `lvl` (once per level: destroy_graphs drops the recordings with the graphs) -- the same enqueue a hipGraph
// captures, every launch turned into a packet by l2_launch.  L2_RUN_EAGER: the queue is given up (the note says why).
static int aql_record_level(l2_ctx* c, int lvl, int (*enq)(l2_ctx*, hipStream_t), AqlProgram** slot) {
  if (*slot) return L2_OK;
  set_level(c, lvl);
  AqlProgram* p = aql_program_new(c->aql);
  c->aql_rec = p; c->aql_rec_failed = false;
  const int rc = enq(c, c->stream);
  c->aql_rec = nullptr;
  if (!rc && !c->aql_rec_failed && !aql_upload(c->aql)) { *slot = p; return L2_OK; }
  c->aql_note = std::string("recording the step failed: ") + (rc ? l2_last_error() : aql_last_error(c->aql));
  aql_program_free(p);
  destroy_graphs(c);
  aql_destroy(c->aql); c->aql = nullptr;
  return L2_RUN_EAGER;
}

// A run on the queue failed (no progress within L2_QUEUE_WAIT_S, or the runtime reported a queue error): the ring may still hold its
// packets, so the queue is never submitted to again -- it is destroyed with its recordings and the context goes on with HIP launches
// (aql_tried stays set; the note says why).  The failed call itself reports the error.
// (test hook, L2_TEST_HOOKS only: L2_DEBUG_FAIL_AQL_RUN=n makes the n-th run on the queue REPORT a failure after it has completed, so that the
// recovery below -- queue retired, recordings dropped, the context going on with hipGraphs -- is exercised without hanging a GPU)
static int aql_run_c(l2_ctx* c, int ntok, AqlProgram* const* per, double* us) {
  const int rc = aql_run(c->aql, ntok, per, c->aql_fence, us);
  if (!rc && c->debug_fail_aql > 0 && --c->debug_fail_aql == 0) return -1;
  return rc;
}
static int aql_give_up(l2_ctx* c) {
  const int rc = fail(L2_E_HIP, "AQL queue: %s", aql_last_error(c->aql));
  c->aql_note = std::string("given up after a failed run: ") + aql_last_error(c->aql);
  destroy_graphs(c);
  aql_destroy(c->aql); c->aql = nullptr;
  return rc;
}

static int run_greedy_aql(l2_ctx* c, int pos0, int steps, bool timed, float* ms) {
  if (!c->aql && aql_open(c)) return L2_RUN_EAGER;
  for (int s = 0; s < steps; ++s) {
    const int lvl = split_level(c, pos0 + s);
    if (aql_record_level(c, lvl, enqueue_greedy, &c->aql_greedy[lvl])) return L2_RUN_EAGER;
  }
  if (steps > 0 && greedy_folds(c) && aql_record_level(c, 0, enqueue_greedy_last, &c->aql_last)) return L2_RUN_EAGER;
  HIPCHK(hipStreamSynchronize(c->stream));      // {token, pos}, the argmax keys: the HIP stream's work is done before the queue starts
  std::vector<AqlProgram*> per(steps);
  for (int s = 0; s < steps; ++s) per[s] = c->aql_greedy[split_level(c, pos0 + s)];
  if (steps > 0 && c->aql_last) per.push_back(c->aql_last);      // the run's last pick
  double us = 0.0;
  if (aql_run_c(c, (int)per.size(), per.data(), timed ? &us : nullptr)) return aql_give_up(c);      // (only a timed run spins on the completion signal)
  if (timed && ms) *ms = (float)(us * 1e-3);
  c->ran_forward = true;
  return check_p2p(c);
}

extern "C" int l2_forward(l2_ctx* c, int token, int pos, float* logits_out) {
  if (!c) return fail(L2_E_ARG, "null context");
  if (pos < 0 || pos >= c->S) return fail(L2_E_ARG, "pos %d outside [0, seq_len=%d)", pos, c->S);
  if (token < 0 || token >= c->V) return fail(L2_E_ARG, "token %d outside [0, vocab_size=%d)", token, c->V);
  // L2_CHECK_POS=1 (a debugging host): the reference's loop feeds pos = 0, 1, 2, ... (llama2.ts:464, 496) and attention reads whatever rows
  // 0 .. pos - 1 the cache holds -- a position that neither restarts at 0 nor continues where the last call (or device loop) left off
  // would read rows no call of this sequence wrote
  if (c->opt_pos_check && pos != 0 && pos > c->next_pos)
    return fail(L2_E_STATE, "L2_CHECK_POS: pos %d skips ahead of the sequence (cache rows 0 .. %d have been written)", pos, c->next_pos - 1);
  int rc = ensure_ready(c);
  if (rc) return rc;
  HIPCHK(hipSetDevice(c->device));
  c->next_pos = pos + 1;
  c->h_tokpos[0] = token; c->h_tokpos[1] = pos; c->h_tokpos[2] = 0; c->h_tokpos[3] = 0;
  const int lvl = split_level(c, pos);
  set_level(c, lvl);
  if (aql_usable(c) && c->opt_zero_copy && !c->tp_path && (c->aql || !aql_open(c)) && !aql_record_level(c, lvl, enqueue_forward_call, &c->aql_step[lvl])) {
    // the library's own queue: {token, pos} read from pinned host memory by the first launch, logits written straight into the
    // host's buffer by the classifier, one doorbell, one signal
    HIPCHK(hipStreamSynchronize(c->stream));      // (uploads, an earlier graph's work)
    if (aql_run_c(c, 1, &c->aql_step[lvl], nullptr)) return aql_give_up(c);
    c->ran_forward = true;
    rc = check_p2p(c);
    if (rc) return rc;
    if (logits_out) memcpy(logits_out, c->h_logits, (size_t)c->V * 4);
    return L2_OK;
  }
  set_level(c, lvl);
  // Replayed hipGraph (or eager launches): {token, pos} of the call are fetched from pinned host memory by the step's FIRST kernel
  // (set_tokpos_kernel, as on the library's queue) -- not by a stream copy in front of the graph launch.  Round 5 saw the blocking call hand
  // back another step's logits under AMD_DIRECT_DISPATCH=0 (the runtime submits from a thread of its own): a graph launch there is not
  // reliably ordered behind a copy enqueued on the same stream just before it; with the fetch INSIDE the graph there is nothing in front of
  // it to be ordered against (tests/test_aql_gpu.py runs the L2_AQL=0 call under that setting).  Logits come back through the host-mapped
  // buffer the classifier writes; only contexts without it (tensor parallel, L2_ZERO_COPY_LOGITS=0) copy them behind the graph.
  int (*enq)(l2_ctx*, hipStream_t) = c->tp_path ? enqueue_forward_host : enqueue_forward_call;
  if (c->tp_path) HIPCHK(hipMemcpyAsync(c->tokpos, c->h_tokpos, 4 * sizeof(int), hipMemcpyHostToDevice, c->stream));      // (its step may hold RCCL collectives: kept as it was)
  if (c->opt_graph) {
    if (!c->g_step[lvl]) { rc = capture(c, enq, &c->g_step[lvl]); if (rc && rc != L2_RUN_EAGER) return rc; }
  }
  if (c->opt_graph) {
    HIPCHK(hipGraphLaunch(c->g_step[lvl], c->stream));
  } else {
    rc = enq(c, c->stream);
    if (rc) return rc;
  }
  if (!(c->opt_zero_copy && !c->tp_path))
    HIPCHK(hipMemcpyAsync(c->h_logits, c->logits, (size_t)c->V * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->ran_forward = true;
  rc = check_p2p(c);
  if (rc) return rc;
  if (logits_out) memcpy(logits_out, c->h_logits, (size_t)c->V * 4);
  return L2_OK;
}

extern "C" float* l2_logits_host(l2_ctx* c) { return c ? c->h_logits : nullptr; }

#include "prefill_host.hip.h"

static int run_greedy(l2_ctx* c, int first_token, int pos0, int steps, bool timed, float* ms) {
  if (!c) return fail(L2_E_ARG, "null context");
  if (steps < 0 || pos0 < 0 || pos0 + steps > c->S) return fail(L2_E_ARG, "pos0 %d + steps %d exceeds seq_len %d", pos0, steps, c->S);
  if (first_token < 0 || first_token >= c->V) return fail(L2_E_ARG, "token out of range");
  int rc = ensure_ready(c);
  if (rc) return rc;
  HIPCHK(hipSetDevice(c->device));
  if (pos0 + steps > c->next_pos || pos0 == 0) c->next_pos = pos0 + steps;      // (L2_CHECK_POS: rows 0 .. pos0 + steps - 1 are written when this returns)
  c->h_tokpos[0] = first_token; c->h_tokpos[1] = pos0; c->h_tokpos[2] = 0; c->h_tokpos[3] = 0;
  HIPCHK(hipMemcpyAsync(c->tokpos, c->h_tokpos, 4 * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemsetAsync(c->amax, 0, 8 * 16 * 8, c->stream));   // argmax keys: zero at the start of every run (an aborted sampled step may have left some)
  if (aql_usable(c)) {
    const int rc_a = run_greedy_aql(c, pos0, steps, timed, ms);
    if (rc_a != L2_RUN_EAGER) return rc_a;      // (L2_RUN_EAGER: the queue could not be had -- the note says why -- and the run goes on below)
  }
  if (c->opt_graph) {   // capture what this run needs before the timed region
    for (int s = 0; s < steps; ++s) {
      const int lvl = split_level(c, pos0 + s);
      if (!c->g_greedy[lvl]) { set_level(c, lvl); rc = capture(c, enqueue_greedy, &c->g_greedy[lvl]); if (rc == L2_RUN_EAGER) break; if (rc) return rc; }
    }
  }
  if (timed) HIPCHK(hipEventRecord(c->ev0, c->stream));
  for (int s = 0; s < steps; ++s) {
    const int lvl = split_level(c, pos0 + s);
    set_level(c, lvl);
    if (c->opt_graph) HIPCHK(hipGraphLaunch(c->g_greedy[lvl], c->stream));
    else { rc = enqueue_greedy(c, c->stream); if (rc) return rc; }
    // rocprofv3 (ROCm 7.2) segfaults with thousands of un-synchronised dispatches queued behind it:
    // L2_PROFILE_SYNC=1 drains the stream after every token (kernel durations are unaffected)
    if (c->profile_sync) HIPCHK(hipStreamSynchronize(c->stream));
  }
  if (steps > 0) { rc = enqueue_greedy_last(c, c->stream); if (rc) return rc; }
  if (timed) {
    HIPCHK(hipEventRecord(c->ev1, c->stream));
    HIPCHK(hipEventSynchronize(c->ev1));
    HIPCHK(hipEventElapsedTime(ms, c->ev0, c->ev1));
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  c->ran_forward = true;
  return check_p2p(c);
}

extern "C" int l2_decode_sample(l2_ctx* c, int first_token, int pos0, int steps, double temperature, double topp,
                                uint64_t* rng_state, int32_t* tokens_out) {
  if (!c || (!tokens_out && steps > 0) || !rng_state) return fail(L2_E_ARG, "null argument");
  if (temperature == 0.0) return l2_decode_greedy(c, first_token, pos0, steps, tokens_out);   // llama2.ts:477-479, no RNG draw
  if (!(temperature == temperature) || !(topp == topp)) return fail(L2_E_ARG, "temperature / topp is NaN");
  int rc = ensure_ready(c);
  if (rc) return rc;
  if (steps < 0 || pos0 < 0 || pos0 + steps > c->S) return fail(L2_E_ARG, "positions %d..%d outside [0, %d)", pos0, pos0 + steps, c->S);
  if (first_token < 0 || first_token >= c->V) return fail(L2_E_ARG, "token %d outside [0, %d)", first_token, c->V);
  if (steps == 0) return L2_OK;
  HIPCHK(hipSetDevice(c->device));
  if (!c->samp.V) {
    if (c->V > l2s::MAX_VOCAB) return fail(L2_E_CONFIG, "device sampler supports vocabularies up to %d", (int)l2s::MAX_VOCAB);
    HIPCHK(l2s::create(&c->samp, c->V));
  }
  const double params[2] = {temperature, topp};
  c->samp_mode = (topp <= 0 || topp >= 1) ? 0 : 1;        // llama2.ts:486: plain sample unless 0 < topp < 1
  // the classifier's argmax keys give the softmax its maximum; the serial A/B form takes its own and would leave them stale
  c->samp_amax = !c->tp_path && temperature > 0 && c->amax && !c->samp.serial;
  HIPCHK(hipMemsetAsync(c->amax, 0, 8 * 16 * 8, c->stream));   // "zero between tokens" holds whatever an earlier (aborted) run left
  c->h_tokpos[0] = first_token; c->h_tokpos[1] = pos0; c->h_tokpos[2] = 0; c->h_tokpos[3] = 0;
  HIPCHK(hipMemcpyAsync(c->tokpos, c->h_tokpos, 4 * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->samp.params, params, sizeof(params), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->samp.rng, rng_state, sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));                 // the host sources above are stack / caller memory
  if (aql_usable(c) && (c->aql || !aql_open(c))) {
    // the sampled loop on the library's own queue: forward + sampler launches of every token as packets
    const int mode = c->samp_mode + (c->samp_amax ? 2 : 0);
    bool ok = true;
    std::vector<AqlProgram*> per(steps);
    for (int s = 0; s < steps && ok; ++s) {
      const int lvl = split_level(c, pos0 + s);
      if (aql_record_level(c, lvl, enqueue_sample, &c->aql_sample[lvl][mode])) ok = false;
      else per[s] = c->aql_sample[lvl][mode];
    }
    if (ok) {
      if (aql_run_c(c, steps, per.data(), nullptr)) return aql_give_up(c);
      HIPCHK(hipMemcpyAsync(rng_state, c->samp.rng, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipMemcpyAsync(tokens_out, c->d_tokens, (size_t)steps * sizeof(int), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      c->ran_forward = true;
      return check_p2p(c);
    }
  }
  const bool graph = c->opt_graph && !c->loop;
  for (int s = 0; s < steps; ++s) {
    const int lvl = split_level(c, pos0 + s);
    set_level(c, lvl);
    if (graph) {
      hipGraphExec_t& g = c->g_sample[lvl][c->samp_mode + (c->samp_amax ? 2 : 0)];
      if (!g) { rc = capture(c, enqueue_sample, &g); if (rc && rc != L2_RUN_EAGER) return rc; }
    }
    if (graph && c->opt_graph) {
      HIPCHK(hipGraphLaunch(c->g_sample[lvl][c->samp_mode + (c->samp_amax ? 2 : 0)], c->stream));
    } else {
      rc = enqueue_sample(c, c->stream);
      if (rc) return rc;
    }
  }
  HIPCHK(hipMemcpyAsync(rng_state, c->samp.rng, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(tokens_out, c->d_tokens, (size_t)steps * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->ran_forward = true;
  return check_p2p(c);
}

extern "C" int l2_debug_running_sums(int device, const float* values, size_t n, double* sums_out) {
  if (!values || !sums_out || n == 0 || n > (size_t)l2s::MAX_VOCAB) return fail(L2_E_ARG, "bad argument (1 .. %d values)", (int)l2s::MAX_VOCAB);
  HIPCHK(hipSetDevice(device));
  float* dx = nullptr; double* dp = nullptr;
  HIPCHK(hipMalloc(&dx, n * 4));
  if (hipMalloc(&dp, n * 8) != hipSuccess) { hipFree(dx); return fail(L2_E_HIP, "hipMalloc failed"); }
  hipError_t e = hipMemcpy(dx, values, n * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = l2s::running_sums(dx, (int)n, dp, nullptr);
  if (e == hipSuccess) e = hipMemcpy(sums_out, dp, n * 8, hipMemcpyDeviceToHost);
  hipFree(dx); hipFree(dp);
  if (e != hipSuccess) return fail(L2_E_HIP, "running sums: %s", hipGetErrorString(e));
  return L2_OK;
}

extern "C" int l2_decode_greedy(l2_ctx* c, int first_token, int pos0, int steps, int32_t* tokens_out) {
  if (!tokens_out && steps > 0) return fail(L2_E_ARG, "null tokens_out");
  int rc = run_greedy(c, first_token, pos0, steps, false, nullptr);
  if (rc) return rc;
  if (steps > 0) HIPCHK(hipMemcpy(tokens_out, c->d_tokens, (size_t)steps * sizeof(int), hipMemcpyDeviceToHost));
  return L2_OK;
}

extern "C" int l2_read_state(l2_ctx* c, int which, int layer, float* out, size_t n_floats) {
  if (!c || !out) return fail(L2_E_ARG, "null argument");
  HIPCHK(hipSetDevice(c->device));
  const float* src = nullptr;
  size_t n = 0;
  const size_t slab = (size_t)c->S * c->kvd_loc;
  // xb after transformer() holds the w2 result in the reference (llama2.ts:292); on chip it is the attention output
  // unless the state is kept (and never on a tensor-parallel rank, whose w2 emits fp64 partials)
  if (which == L2_S_XB && c->ran_forward && c->tp_path)
    return fail(L2_E_STATE, "RunState.xb (the w2 result, llama2.ts:292) is not materialised on a tensor-parallel rank");
  if (!c->opt_keep_state && (((which == L2_S_X || which == L2_S_XB) && c->ran_forward) || which == L2_S_XB2 || which == L2_S_HB2 || which == L2_S_K || which == L2_S_V || which == L2_S_ATT))
    return fail(L2_E_STATE, "this RunState field is only read by transformer() itself and stays on chip: set L2_OPT_KEEP_STATE before the forward");
  switch (which) {
    case L2_S_X: src = c->ran_forward ? c->xn : c->x; n = c->d; break;   // after a forward: the final-normed x (kept state)
    case L2_S_XB: src = c->xb; n = c->d_loc; break;
    case L2_S_XB2: src = c->xb2; n = c->d; break;
    case L2_S_HB: src = c->hb; n = c->h_loc; break;
    case L2_S_HB2: src = c->hb2; n = c->h_loc; break;
    case L2_S_Q: src = c->q; n = c->d_loc; break;
    case L2_S_K: src = c->k; n = c->kvd_loc; break;
    case L2_S_V: src = c->v; n = c->kvd_loc; break;
    case L2_S_ATT: src = c->att; n = (size_t)c->H_loc * c->S; break;
    case L2_S_LOGITS: src = c->logits; n = c->V; break;
    case L2_S_KEY_CACHE: case L2_S_VALUE_CACHE: {
      const float* base = which == L2_S_KEY_CACHE ? c->kc : c->vc;
      if (layer < 0) { src = base; n = slab * c->L; }
      else { if (layer >= c->L) return fail(L2_E_ARG, "layer out of range"); src = base + slab * layer; n = slab; }
      break;
    }
    default: return fail(L2_E_ARG, "unknown state id %d", which);
  }
  if (n_floats != n) return fail(L2_E_ARG, "state %d has %zu floats, caller asked for %zu", which, n, n_floats);
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(out, src, n * 4, hipMemcpyDeviceToHost));
  return L2_OK;
}

extern "C" int l2_set_option(l2_ctx* c, int key, int value) {
  if (!c) return fail(L2_E_ARG, "null context");
  switch (key) {
    case L2_OPT_EXACT_ATTENTION:
      if (c->opt_exact != !!value) {
        c->opt_exact = !!value;
        destroy_graphs(c);
      }
      return L2_OK;
    case L2_OPT_USE_GRAPH: c->opt_graph = !!value; return L2_OK;
    case L2_OPT_KEEP_STATE: if (c->opt_keep_state != !!value) { c->opt_keep_state = !!value; destroy_graphs(c); } return L2_OK;
    case L2_OPT_AQL_QUEUE: c->opt_aql = !!value; if (value) c->aql_tried = false; return L2_OK;
    case L2_OPT_PREFILL_F32_MFMA: c->opt_pf_f32 = !!value; return L2_OK;
    case L2_OPT_CHECK_POS: c->opt_pos_check = !!value; return L2_OK;
    case L2_OPT_PACKED_MIB: case L2_OPT_WEIGHT_MIB: case L2_OPT_SAMPLED_TOKENS: case L2_OPT_SAMPLED_SERIAL:
      return fail(L2_E_ARG, "option %d is read-only", key);
    default: return fail(L2_E_ARG, "unknown option %d", key);
  }
}

extern "C" int l2_get_option(l2_ctx* c, int key, int* value) {
  if (!c || !value) return fail(L2_E_ARG, "null argument");
  switch (key) {
    case L2_OPT_EXACT_ATTENTION: *value = c->opt_exact; return L2_OK;
    case L2_OPT_USE_GRAPH: *value = c->opt_graph; return L2_OK;
    case L2_OPT_KEEP_STATE: *value = c->opt_keep_state; return L2_OK;
    case L2_OPT_AQL_QUEUE: *value = aql_usable(c) ? 1 : 0; return L2_OK;      // (why not: l2_dispatch_reason -- l2_last_error is for failures)
    case L2_OPT_PREFILL_F32_MFMA: *value = c->opt_pf_f32; return L2_OK;
    case L2_OPT_CHECK_POS: *value = c->opt_pos_check; return L2_OK;
    case L2_OPT_PACKED_MIB: {
      size_t floats = 0;
      if (c->packed_valid) for (int m = 0; m < 5; ++m) if (c->packed[m].buf) floats += c->packed[m].layer_elems * (size_t)(m == MODE_CLS ? 1 : c->L);
      *value = (int)(floats * sizeof(float) >> 20);
      return L2_OK;
    }
    case L2_OPT_WEIGHT_MIB: {      // every byte of weights on the device right now: row-major tensors still held + repacked copies
      size_t floats = 0;
      for (int k = 0; k < L2_T_COUNT; ++k) if (c->w[k] && !(k == L2_T_WCLS && c->shared)) floats += c->layer_elems[k] * (size_t)c->layers_of[k];
      for (int m = 0; m < 5; ++m) if (c->packed[m].buf) floats += c->packed[m].layer_elems * (size_t)(m == MODE_CLS ? 1 : c->L);
      *value = (int)(floats * sizeof(float) >> 20);
      return L2_OK;
    }
    case L2_OPT_SAMPLED_TOKENS: case L2_OPT_SAMPLED_SERIAL: {
      unsigned long long st[2] = {0, 0};
      if (c->samp.V) { HIPCHK(hipSetDevice(c->device)); HIPCHK(l2s::read_stats(c->samp, st, c->stream)); }
      const unsigned long long v = st[key == L2_OPT_SAMPLED_SERIAL ? 1 : 0];
      *value = v > 0x7fffffffull ? 0x7fffffff : (int)v;
      return L2_OK;
    }
    default: return fail(L2_E_ARG, "unknown option %d", key);
  }
}

extern "C" const char* l2_dispatch_reason(l2_ctx* c) {
  if (!c || aql_usable(c)) return "";
  if (!c->opt_aql) c->dispatch_why = "switched off (L2_AQL=0 / L2_OPT_AQL_QUEUE)";
  else if (!c->opt_graph) c->dispatch_why = "the step is not recorded (L2_USE_GRAPH=0): eager launches";
  else if (c->tp_path && !(c->p2p && !c->loop && c->p2p_peers_ready)) c->dispatch_why = "a tensor-parallel step whose exchanges are not kernels of the library (RCCL collectives, or the loopback test group)";
  else if (c->loop) c->dispatch_why = "the loopback test group (host barriers inside the step)";
  else if (c->profile_sync) c->dispatch_why = "L2_PROFILE_SYNC=1";
  else c->dispatch_why = c->aql_note;      // the queue could not be had, or was given up: why
  return c->dispatch_why.c_str();
}

#include "bench_hooks.hip.h"

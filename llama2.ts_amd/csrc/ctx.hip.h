// ctx.hip.h -- error text, RCCL binding, test-group plumbing and the context structure of libllama2hip.so
// Part of the one translation unit llama2_hip.hip (included there, in order); not a stand-alone header.
#pragma once

// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIPCHK(expr)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) return fail(L2_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// ---- RCCL, bound lazily so the library has no link-time dependency on it ------------------------
typedef struct { char internal[128]; } nccl_uid;
typedef void* nccl_comm;
struct Rccl {
  void* so = nullptr;
  int (*GetUniqueId)(nccl_uid*) = nullptr;
  int (*CommInitRank)(nccl_comm*, int, nccl_uid, int) = nullptr;
  int (*CommDestroy)(nccl_comm) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, nccl_comm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
static Rccl g_rccl;
enum { NCCL_FLOAT32 = 7, NCCL_FLOAT64 = 8, NCCL_SUM = 0 };

static int rccl_bind() {
  if (g_rccl.so) return L2_OK;
  const char* names[] = {getenv("L2_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* so = nullptr;
  for (const char* n : names) {
    if (!n) continue;
    so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (so) break;
  }
  if (!so) return fail(L2_E_COMM, "cannot dlopen RCCL: %s", dlerror());
  g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(so, "ncclGetUniqueId");
  g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(so, "ncclCommInitRank");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(so, "ncclCommDestroy");
  g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(so, "ncclAllReduce");
  g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(so, "ncclAllGather");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(so, "ncclGetErrorString");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.AllGather || !g_rccl.CommDestroy) {
    dlclose(so);
    return fail(L2_E_COMM, "RCCL symbols missing");
  }
  g_rccl.so = so;
  return L2_OK;
}

#define NCCLCHK(expr)                                                                             \
  do {                                                                                            \
    int r_ = (expr);                                                                              \
    if (r_ != 0) return fail(L2_E_COMM, "%s failed: %s", #expr, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?"); \
  } while (0)

// ---- Loopback communicator (L2_TP_LOOPBACK=1): a TEST HOOK, not a product path ---------------------
// The driver's multi-GPU node is the only place RCCL runs with more than one rank, and it is not reachable from
// the 1-GPU development boxes.  With L2_TP_LOOPBACK=1 the G ranks of a tensor-parallel group are G contexts of
// ONE process on ONE device, each driven by its own host thread (tests/test_tp_gpu.py); the two collectives are
// then plain device work between host-side thread barriers: every rank sums the G published fp64 partial vectors
// in rank order (all-reduce) or copies the G logits slices (all-gather).  Everything else -- shard slicing,
// fp64 partial GEMVs, the single rounding in tp_residual_kernel, the greedy loop on gathered logits -- is the
// code the RCCL path runs.  Groups are keyed by the 128-byte id the caller passes to l2_create_tp.
struct LoopGroup {
  int G = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  unsigned gen = 0;
  bool broken = false;
  const void* ptrs[16] = {};
  void* p2p_base[16] = {};     // peer-to-peer exchange: every rank's inbox / logits, registered at create
  float* p2p_logits[16] = {};
  bool wait() {   // generation barrier; false after 60 s (a rank died: fail instead of hanging the box)
    std::unique_lock<std::mutex> lk(mu);
    if (broken) return false;
    const unsigned g = gen;
    if (++arrived == G) { arrived = 0; ++gen; cv.notify_all(); return true; }
    if (!cv.wait_for(lk, std::chrono::seconds(60), [&] { return gen != g || broken; })) { broken = true; cv.notify_all(); return false; }
    return !broken;
  }
};
static std::mutex g_loop_mu;
static std::map<std::string, std::shared_ptr<LoopGroup>> g_loop_groups;

struct LoopPtrs { const double* p[16]; };
__global__ void loop_sum_kernel(double* out, const LoopPtrs in, int G, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = in.p[0][i];
  for (int r = 1; r < G; ++r) s += in.p[r][i];
  out[i] = s;
}

// one-shot peer-to-peer exchange (kernels and protocol: tp_p2p_* below)
enum { P2P_MAXG = 8, P2P_FB = 64 };   // ranks; flag words per (parity, source): blocks of the widest exchange
struct P2PPeers { unsigned long long* flags[P2P_MAXG]; double* inbox[P2P_MAXG]; float* logits[P2P_MAXG]; unsigned long long* gin[P2P_MAXG]; };   // gin: granule inbox (pushed partials)
struct P2PArgs {
  P2PPeers pr;
  unsigned long long* epoch;   // this rank's exchange counters, one per block
  unsigned* ticket;            // (unused)
  int* err;
  int G, rank, n;              // n: elements of this exchange (d, or V_loc)
  unsigned long long wait_ticks;   // bound of a flag wait, in 100 MHz ticks
  int fenced;                  // L2_TP_FENCED=1: system-scope fences around the flags (tp_exchange.hip.h)
  int solo;                    // shard-timing context (l2_create_tp with the L2_TP_SOLO_ID id): every "peer" is this rank itself
};

// ------------------------------------------------------------------------------------------------
struct l2_ctx {
  int32_t hdr[7];
  int d, h, L, H, V, S, hs;
  bool shared;
  int device;
  hipStream_t stream = nullptr;
  // tensor parallel shard (G == 1: everything local)
  int G = 1, rank = 0;
  int d_loc, h_loc, H_loc, V_loc;
  int KVH, kvd, kvd_loc;             // cache heads honoured (== H unless L2_F_GQA), floats of a cache row, per rank
  unsigned flags = 0;
  nccl_comm comm = nullptr;
  l2s::Sampler samp;                 // device sampler (l2_decode_sample), created on first use
  hipGraphExec_t g_sample[NLEV][4] = {};  // [attention split level][plain sample / top-p, + 2: maximum taken from the classifier's argmax keys]
  bool samp_amax = false;
  int samp_mode = 0;
  std::shared_ptr<LoopGroup> loop;   // L2_TP_LOOPBACK test hook (see LoopGroup)
  double* loop_tmp = nullptr;
  bool tp_path = false;   // WO/W2 write fp64 partials + all-reduce; logits all-gathered (G > 1, or forced for tests)
  // one-shot peer-to-peer exchange (tp_p2p_*): this rank's inbox + flags, the peers' mappings
  bool p2p = false;
  void* p2p_base = nullptr;          // uncached: [2][8][64] flag words, then [2][8][d] doubles
  unsigned long long* p2p_epoch = nullptr;   // [2][P2P_FB] exchange counters, one per block (device): the flag exchange's and the logits gather's, then the pushed exchange's own
  int* p2p_err = nullptr;            // pinned + mapped
  int* p2p_err_dev = nullptr;
  P2PPeers p2p_peers = {};
  std::vector<void*> p2p_opened;     // IPC mappings to close
  bool p2p_peers_ready = false;
  // fused attention + wo launch of a tensor-parallel rank (attention.hip.h: attn_wo_kernel)
  unsigned long long* awo_gran = nullptr;   // [d / G] hand-off granules: the attention output of this rank's heads
  unsigned* awo_ep = nullptr;               // its launch counter (advanced by the combine launch that follows)
  int opt_awo = 1;                          // L2_TP_ATTN_WO=0: attention and wo as two launches (A/B, development switch)
  TpPush* tp_push = nullptr;         // device table of the peers' granule inboxes for the GEMV epilogues (kernels.hip.h: tp_push_row)
  int opt_push = 1;                  // L2_TP_PUSH=0: partials through c->partial and the flag exchange (round-4 form; A/B, development switch)
  bool rccl_graph = false;           // the RCCL collectives of the step are captured into the per-token hipGraph (cleared if capture is refused)
  unsigned long long p2p_wait_ticks = 3000000000ull;   // L2_TP_WAIT_S (default 30 s) on the 100 MHz clock
  int p2p_fenced = 0;                // L2_TP_FENCED=1: the peer-to-peer exchange with system-scope fences around its flags
  bool solo = false;                 // shard-timing context: one rank of G alone, the exchange kernels run against its own inbox (timing only, sums are G x the partial)
  bool p2p_synced = false;           // the ranks have met once (host side) right before the first exchange of a step
  bool broken = false;               // a peer-to-peer wait gave up: this rank's epochs no longer match its peers'

  float* w[L2_T_COUNT] = {};
  size_t layer_elems[L2_T_COUNT] = {};  // LOCAL floats per layer (or whole tensor when unlayered)
  int layers_of[L2_T_COUNT] = {};
  std::vector<uint8_t> uploaded[L2_T_COUNT];

  float *x = nullptr, *xb = nullptr, *xb2 = nullptr, *hb = nullptr, *hb2 = nullptr, *q = nullptr, *k = nullptr,
        *v = nullptr, *att = nullptr, *logits = nullptr, *logits_loc = nullptr, *kc = nullptr, *vc = nullptr, *xn = nullptr;
  double* partial = nullptr;
  double* attn_part = nullptr;      // split attention partials [H][NS][rec]
  unsigned* attn_counter = nullptr; // [H] arrival tickets, zero between launches
  unsigned long long* gran = nullptr;   // fused QKV + attention launch: [d + 2 kvd] hand-off granules (kernels.hip.h: granule_store)
  unsigned* gran_ep = nullptr;          // {launch counter the granule tags come from, heads done in this launch}
  int* h_herr = nullptr;                // pinned + mapped: a granule wait gave up
  int* h_herr_dev = nullptr;
  int opt_fuse = 1;                     // L2_FUSE_QKV_ATTN=0: two launches (A/B, development switch)
  int opt_fuse_four_waves = 1;          // L2_FUSE_FOUR_WAVES=0: the fused launch's attention workgroups always on eight waves (A/B)
  int opt_fuse_splits = 0;              // L2_FUSE_SPLITS=1: the fused launch also at the split level (A/B, tests: it is slower there)
  unsigned long long* amax = nullptr;   // greedy loop: 8 argmax keys, one per 128-byte line, zero between tokens
  int attn_splits_forced = 0;       // L2_ATTN_SPLITS: fixed split count (tests); 0 = by position
  int pf3 = 1;                      // L2_PF3: 1 (default) register-blocked prefill GEMMs where the shape allows, 0: the 16-row-tile kernels everywhere (A/B, tests)
  int pf_attn = 1;                  // L2_PF_ATTN: 1 (default) prompt attention on the fp64 MFMA (16 queries per workgroup), 0: the decode kernel per (head, query)
  int cur_splits = 1;               // split count of the step being enqueued / captured
  bool cur_fused = false;           // ... and whether its QKV + attention are the fused launch
  int fuse_min_rows = 0;            // L2_FUSE_MIN_ROWS: the fused launch from this many cached rows on (0 for inputs below 512 floats, else 128)
  int split_rows = 144;             // L2_ATTN_SPLIT_ROWS: cached rows of a head beyond which attention runs 8 workgroups per head
  bool split_rows_set = false;      // the switch was given: it also overrides the fused launch's 256
  int small_max = 0;                // L2_SMALL_MAX: largest matrix (floats) that takes the latency-form GEMV
  int n_cus = 256;
  std::string ipc_dir;              // L2_TP_IPC_DIR: ranks are processes that meet through files (test hook)
  std::vector<hipEvent_t> probe;    // in-situ probe: event pairs around every launch of the dominant kernel
  size_t probe_used = 0;
  bool probe_on = false;
  // prefill (prefill.hip.h): 16-token chunk buffers
  float *pf_x = nullptr, *pf_xn = nullptr, *pf_q = nullptr, *pf_xb = nullptr, *pf_hb = nullptr;
  int* pf_tok = nullptr;
  int* tokpos = nullptr;    // device {token,pos,step,0}
  int* h_tokpos = nullptr;  // pinned
  int* d_tokens = nullptr;  // device, S ints
  float* h_logits = nullptr;      // pinned + mapped: the classifier kernel writes it directly
  float* h_logits_dev = nullptr;  // device alias of h_logits
  int opt_zero_copy = 1;
  int profile_sync = 0;
  unsigned long long* dbg = nullptr;  // L2_STAMPS builds

  hipGraphExec_t g_step[NLEV] = {}, g_greedy[NLEV] = {};   // one captured graph per attention split level
  // The greedy loop as hand-written AQL packets on a queue of the library's own (aql_queue.h): the step of a level is RECORDED once
  // (the same enqueue that a hipGraph captures) and replayed for every token with fence-free packet headers.
  AqlQueue* aql = nullptr;
  AqlProgram* aql_greedy[NLEV] = {};
  AqlProgram* aql_sample[NLEV][4] = {};   // the sampled loop (same indices as g_sample)
  AqlProgram* aql_last = nullptr;   // the greedy run's last pick (one launch, once per run)
  AqlProgram* aql_step[NLEV] = {};  // the blocking call: {token, pos} from pinned host memory, the step, logits straight into the host's buffer
  int* h_tokpos_dev = nullptr;      // device alias of h_tokpos
  AqlProgram* aql_rec = nullptr;    // non-null while enqueue_* records instead of launching
  bool aql_rec_failed = false;
  bool aql_tried = false;
  int opt_aql = 1;                  // L2_AQL=0: replayed hipGraphs instead of the library's own queue (greedy loop, sampled loop and the blocking call alike)
  int aql_fence = 20;               // acquire scope + 4 * release scope + 16 * (agent acquire on a token's first launch): aql_queue.h, create_impl
  std::string aql_note;             // why the queue could not be had / was given up
  std::string dispatch_why;         // l2_dispatch_reason's text (valid until the next call)
  int opt_exact = 0, opt_graph = 1, opt_keep_state = 0;
  int opt_pf_f32 = 0;               // L2_OPT_PREFILL_F32_MFMA: the register-blocked prompt GEMMs accumulate in fp32 on v_mfma_f32_16x16x4_f32 (opt-in; prefill.hip.h)
  int opt_pollute = 0;              // L2_DEBUG_POLLUTE=1 (test hook): l1_pollute_kernel behind every launch of the step (kernels.hip.h)
  float* pollute_sink = nullptr;
  int debug_fail_aql = 0;           // L2_DEBUG_FAIL_AQL_RUN=n (test hook): the n-th run on the queue reports a failure (the recovery path, llama2_hip.hip: aql_give_up)
  int opt_pos_check = 0;            // L2_CHECK_POS=1: l2_forward refuses a position that does not continue the sequence (llama2.ts:464, 496)
  int next_pos = 0;
  bool ran_forward = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // tuning overrides (env)
  int tune_R = 0, tune_U = 0, tune_nwaves = 0, tune_gridcap = 0, tune_rot = 5;
  // Streaming-form matrices repacked in consumption order (kernels.hip.h, phase_body PK): a second copy, per phase (MODE_*),
  // built on the device from the row-major tensors (which prefill, the embedding gather and l2_read_tensor keep using).
  struct Packed { float* buf = nullptr; size_t layer_elems = 0; int U = 0, nwaves = 0, grid = 0; std::vector<uint8_t> dirty; bool noted = false; } packed[5];
  bool packed_valid = false;        // false after an upload of a matrix a packed phase reads: its (phase, layer) slices are rebuilt before the next step
  int opt_packed = 1;               // L2_PACKED=0: stream the row-major tensors (A/B)
  // One copy of the weights: once a phase's matrices are repacked, their row-major tensors are given back (w[kind] = null); the prompt
  // GEMMs read the repacked copy too, l2_read_tensor and a later l2_upload unpack it first (ensure_rowmajor).
  bool released[L2_T_COUNT] = {};
  bool rerelease = false;           // l2_read_tensor brought the row-major tensors back for a parity read: the next step gives them away again (no repack)
  int opt_one_copy = 1;             // L2_ONE_COPY=0: keep both copies (A/B, development switch)
};

static bool is_layered(int kind) { return kind >= L2_T_RMS_ATT && kind <= L2_T_W3; }

// An upload dirties only the repacked slice (phase, layer) that holds that matrix; vectors and tables dirty nothing.
// layer < 0: every layer of the tensor.
static void mark_dirty(l2_ctx* c, int kind, int layer) {
  int m = -1;
  switch (kind) {
    case L2_T_WQ: case L2_T_WK: case L2_T_WV: m = MODE_QKV; break;
    case L2_T_WO: m = MODE_WO; break;
    case L2_T_W1: case L2_T_W3: m = MODE_W13; break;
    case L2_T_W2: m = MODE_W2; break;
    case L2_T_WCLS: m = MODE_CLS; layer = 0; break;
    case L2_T_TOKEN_EMBEDDING: if (c->shared) { m = MODE_CLS; layer = 0; } break;
    default: break;
  }
  if (m < 0) return;
  std::vector<uint8_t>& d = c->packed[m].dirty;
  if (layer < 0) { for (auto& b : d) b = 1; }
  else if ((size_t)layer < d.size()) d[layer] = 1;
  c->packed_valid = false;
}

// Local (per-rank) shape of one layer of a tensor: rows x cols, plus where the slice sits in the
// full tensor (row0/col0) so l2_upload can cut it out of the caller's full array.
struct Slice { size_t rows, cols, full_rows, full_cols, row0, col0; };

// (pure: the shape, the rank and the group size are all it needs -- l2_debug_tensor_slice below hands it to a CPU test, which holds it
//  against llama2_ts_amd/tp.py and, through that, against the numpy restatement tests/test_tp_gloo.py runs over gloo)
static Slice tensor_slice_pure(size_t d, size_t h, size_t V, size_t S, size_t hs, size_t kvd, size_t G, size_t r, int kind) {
  const size_t hs2 = hs / 2, dl = d / G, hl = h / G, Vl = V / G, kvl = kvd / G;
  switch (kind) {
    case L2_T_TOKEN_EMBEDDING: return {V, d, V, d, 0, 0};
    case L2_T_RMS_ATT: case L2_T_RMS_FFN: case L2_T_RMS_FINAL: return {1, d, 1, d, 0, 0};
    case L2_T_WQ: return {dl, d, d, d, r * dl, 0};  // whole heads
    case L2_T_WK: case L2_T_WV: return {kvl, d, kvd, d, r * kvl, 0};
    case L2_T_WO: return {d, dl, d, d, 0, r * dl};                                 // columns, repacked
    case L2_T_W1: case L2_T_W3: return {hl, d, h, d, r * hl, 0};
    case L2_T_W2: return {d, hl, d, h, 0, r * hl};
    case L2_T_FREQ_REAL: case L2_T_FREQ_IMAG: return {S, hs2, S, hs2, 0, 0};
    case L2_T_WCLS: return {Vl, d, V, d, r * Vl, 0};
    default: return {0, 0, 0, 0, 0, 0};
  }
}
static Slice tensor_slice(const l2_ctx* c, int kind) {
  return tensor_slice_pure((size_t)c->d, (size_t)c->h, (size_t)c->V, (size_t)c->S, (size_t)c->hs, (size_t)c->kvd, (size_t)c->G, (size_t)c->rank, kind);
}
// (tests, no GPU needed) rank `rank` of `G`'s slice of one layer of tensor `kind` for the header `cfg`: out = {rows, cols, full_rows, full_cols, row0, col0}
extern "C" int l2_debug_tensor_slice(const int32_t cfg[7], int honour_kv_heads, int kind, int rank, int G, long long out[6]) {
  if (!cfg || !out || G < 1 || rank < 0 || rank >= G || kind < 0 || kind >= L2_T_COUNT || cfg[3] <= 0) return L2_E_ARG;
  const size_t d = (size_t)cfg[0], H = (size_t)cfg[3], hs = d / H, kvd = honour_kv_heads ? (size_t)cfg[4] * hs : d;
  const Slice s_ = tensor_slice_pure(d, (size_t)cfg[1], (size_t)abs(cfg[5]), (size_t)cfg[6], hs, kvd, (size_t)G, (size_t)rank, kind);
  out[0] = (long long)s_.rows; out[1] = (long long)s_.cols; out[2] = (long long)s_.full_rows; out[3] = (long long)s_.full_cols; out[4] = (long long)s_.row0; out[5] = (long long)s_.col0;
  return L2_OK;
}

// ------------------------------------------------------------------------------------------------
#define LCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(L2_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)

static int env_int(const char* name, int dflt) {
  const char* s = getenv(name);
  return s && *s ? atoi(s) : dflt;
}

// Test hooks (tensor-parallel groups on ONE GPU: L2_TP_FORCE_COMM, L2_TP_NO_COMM, L2_TP_LOOPBACK, L2_TP_IPC_DIR) exist only behind
// L2_TEST_HOOKS=1, read once per process: a production host cannot trip them through a stray variable.
static bool hooks_on() { static const bool on = env_int("L2_TEST_HOOKS", 0) != 0; return on; }
static int hook_int(const char* name) { return hooks_on() ? env_int(name, 0) : 0; }
// development switches (launch geometry, attention split policy, A/B forms): the same gate, with the shipped default otherwise
static int dev_int(const char* name, int dflt) { return hooks_on() ? env_int(name, dflt) : dflt; }
// The file rendezvous of a tensor-parallel group (ranks that cannot share an RCCL communicator meet in a directory) has a switch of
// its own, L2_TP_FILE_RENDEZVOUS=1 + L2_TP_IPC_DIR=<dir>, read at every create: bench.py's last-resort attempt needs it without
// opening the whole development gate in a measured run.
static const char* ipc_dir_env() {
  if (!hooks_on() && env_int("L2_TP_FILE_RENDEZVOUS", 0) == 0) return nullptr;
  const char* s = getenv("L2_TP_IPC_DIR");
  return (s && *s) ? s : nullptr;
}

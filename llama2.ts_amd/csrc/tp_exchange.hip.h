// tp_exchange.hip.h -- the two exchanges of the tensor-parallel step: one-shot peer-to-peer over xGMI, RCCL baseline, one-GPU test groups
// Part of the one translation unit llama2_hip.hip (included there, in order); not a stand-alone header.
#pragma once

// ---- One-shot peer-to-peer exchange over xGMI -------------------------------------------------
// The two all-reduces of a layer move d fp64 partials (32 KB at 7B) and the logits gather V/G floats per rank:
// latency-bound messages, for which a ring or tree collective pays several launches and hops.  Here every rank
// WRITES its contribution straight into a slot of every peer's inbox (peer-mapped, uncached memory), raises one
// flag per (peer, block), waits for the G flags of its own inbox and sums the G slots IN RANK ORDER -- every rank
// adds the same numbers in the same order, so x stays bit-identical across ranks and is rounded to fp32 once
// (llama2.ts:201), exactly as the RCCL path and the oracle's orc_forward_tp do.  One kernel = exchange + residual;
// nothing but kernels, so the whole tensor-parallel step is captured in one hipGraph.
//   * epochs: block b of every rank handles the same elements and only ever talks to block b of its peers, so every block
//     counts ITS exchanges in a device word of its own (all ranks run the same sequence of launches); a flag holds the epoch of
//     the exchange that last wrote its slot; slots alternate by epoch parity -- block b of a rank cannot start exchange e + 2
//     before block b of every peer has finished reading exchange e, because e + 1 needs their contribution first, and they raise
//     that only in the NEXT launch;
//   * no fences (round 4; the first version bracketed the exchange with __threadfence_system(): ~8 us per exchange for a rank
//     ALONE on its GPU, 30 % of the 8-rank step -- profiles/r04/tp_shard_step_kernels.txt): inbox, flags and the gathered
//     logits are UNCACHED memory, every access to them is a system-scope relaxed atomic (sc0 sc1: past L1 and L2, straight to
//     the memory that holds them, over xGMI for a peer's), so the release is "every storing wave has drained its stores
//     (s_waitcnt vmcnt(0)), workgroup barrier, then the flag stores" -- writes to one peer arrive in order -- and the acquire is
//     "poll the flag words, workgroup barrier, load" (MI355X guide, hand-off recipe R1 with every load L1-bypassing).  Polls are
//     bounded: a rank that never arrives sets `err` instead of hanging the GPU.
//   * L2_TP_FENCED=1 (an ordinary environment switch, read at creation, not behind the development gate) puts the fences back:
//     __threadfence_system() between the barrier and the flag stores and behind the poll.  No multi-GPU box has run either form;
//     the fence-free one is guarded by the soak at creation (below) and by bench.py's golden check before it times anything -- a
//     deployment that sees ranks diverge has this switch before it has a new build.

// (past L1 and not through the scalar cache: the launch before this one advanced it -- the coherence rule of kernels.hip.h)
__device__ __forceinline__ unsigned long long p2p_begin(const P2PArgs& a) {
  const unsigned long long e = __hip_atomic_load(a.epoch + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(e >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)e)) + 1;
}

__device__ __forceinline__ void p2p_store(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double p2p_load(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}
__device__ __forceinline__ void p2p_store(float* p, float v) {
  __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// flags of this block up on every peer (solo: the G flags this rank would receive, raised by itself in its own inbox) ...
__device__ __forceinline__ void p2p_raise(const P2PArgs& a, unsigned long long e, int tid) {
  const int par = (int)(e & 1), b = blockIdx.x;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // EVERY storing wave: its payload stores have left
  __syncthreads();
  if (a.fenced && tid < a.G) __threadfence_system();   // L2_TP_FENCED=1: a system-scope release in front of the flags (the round-3 form: ~8 us per exchange)
  if (tid < a.G) __hip_atomic_store(a.pr.flags[a.solo ? a.rank : tid] + ((size_t)(par * P2P_MAXG + (a.solo ? tid : a.rank)) * P2P_FB + b), e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// ... then wait for every source's flag in the local inbox
__device__ __forceinline__ void p2p_wait(const P2PArgs& a, unsigned long long e, int tid) {
  const int par = (int)(e & 1), b = blockIdx.x;
  if (tid < a.G) {
    const unsigned long long* mine = a.pr.flags[a.rank] + ((size_t)(par * P2P_MAXG + tid) * P2P_FB + b);
    unsigned spins = 0;
    unsigned long long t0 = 0;
    while (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != e) {
      __builtin_amdgcn_s_sleep(1);
      // bounded by WALL time on the constant 100 MHz clock (a.wait_ticks, default 30 s, L2_TP_WAIT_S): ordinary rank skew
      // -- a peer still capturing its graph, a slower checkpoint read -- must not trip it; a rank that died must.
      // The host then marks the context broken (check_p2p): epochs and flags no longer match the peers'.
      if ((++spins & 255u) == 0) {
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        if (!t0) t0 = now;
        else if (now - t0 > a.wait_ticks) { *a.err = 1; break; }
      }
    }
    if (a.fenced) __threadfence_system();              // ... and a system-scope acquire behind the poll
  }
  __syncthreads();
}

__device__ __forceinline__ void p2p_end(const P2PArgs& a, unsigned long long e, int tid) {
  if (tid == 0) a.epoch[blockIdx.x] = e;      // this block's count, read by the same block of the NEXT launch: a plain store nobody waits for
}

// PART 0: the whole exchange in one kernel (product path).  PART 1 / 2: its two halves -- contribute, then wait + combine
// -- as separate launches with a host barrier in between: the loopback test group runs all ranks on ONE GPU, where
// G kernels that wait for each other are not guaranteed to be resident together (they deadlock until the bounded
// wait gives up when two ranks' streams share a hardware queue); on a node every rank has its own GPU.
// all-reduce(sum) of the d fp64 partials + ONE fp32 rounding + residual accumulate (llama2.ts:201, 168-170)
template <int PART>
__global__ void __launch_bounds__(256) tp_p2p_reduce_kernel(const P2PArgs a, const double* partial, float* x, const float* res_emb,
                                                            float* mv_out, const int* tokpos) {
  const int tid = threadIdx.x, stride = gridDim.x * 256;
  const unsigned long long e = p2p_begin(a);
  const size_t slot = (size_t)((int)(e & 1) * P2P_MAXG) * a.n;      // (e: this block's epoch -- every block of a launch has the same one in a reduce)
  if (PART != 2) {
    for (int i = blockIdx.x * 256 + tid; i < a.n; i += stride) {
      const double v = ld_sc1(partial + i);
      for (int p = 0; p < a.G; ++p) p2p_store(a.pr.inbox[a.solo ? a.rank : p] + slot + (size_t)(a.solo ? p : a.rank) * a.n + i, v);
    }
  }
  if (PART == 1) { p2p_raise(a, e, tid); return; }
  if (PART == 0) p2p_raise(a, e, tid);
  p2p_wait(a, e, tid);
  for (int i = blockIdx.x * 256 + tid; i < a.n; i += stride) {
    const double* in = a.pr.inbox[a.rank] + slot + i;
    double part[P2P_MAXG];
#pragma unroll
    for (int r = 0; r < P2P_MAXG; ++r) part[r] = p2p_load(in + (size_t)(r < a.G ? r : 0) * a.n);      // all requested at once
    double s = part[0];
#pragma unroll
    for (int r = 1; r < P2P_MAXG; ++r) if (r < a.G) s += part[r];      // rank order on every rank
    const float xr = res_emb ? res_emb[(size_t)tokpos[0] * a.n + i] : ld_sc1(x + i);
    const float mv = (float)s;
    x[i] = xr + mv;
    if (mv_out) mv_out[i] = mv;
  }
  p2p_end(a, e, tid);
}

// The pushed form of the same exchange (round 5): the wo / w2 GEMV's own waves have already stored every row's fp64 partial into
// every peer's granule inbox (kernels.hip.h: tp_push_row -- two tagged 8-byte halves in one 16-byte system-scope store, the data is
// its own flag), the moment the row was reduced: what is left for this launch is to wait for the G granule pairs of every element in
// the LOCAL inbox, add them in rank order, round once, add the residual.  No payload load + G stores, no drain, no flag round trip
// (4.5 us -> ~3 for a rank alone on its GPU), and on a node the xGMI hop runs while the GEMV's later rows are still being computed.
// Slots alternate by the exchange's parity exactly as the flag form's do (a rank cannot push exchange e + 2 before every peer has
// finished reading e: e + 1 needs their contribution first, and they push that only after their own e is combined).
__global__ void __launch_bounds__(256) tp_p2p_combine_kernel(const P2PArgs a, float* x, const float* res_emb, float* mv_out, const int* tokpos, unsigned* bump) {
  const int tid = threadIdx.x, stride = gridDim.x * 256;
  // the launch counter of the fused attention + wo launch in front of this one: every workgroup of that launch has read it (the
  // launch is over), the next one reads the new number
  if (bump && blockIdx.x == 0 && tid == 0) __hip_atomic_store(bump, __hip_atomic_load(bump, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long e = p2p_begin(a);
  const unsigned tag = (unsigned)e;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(a.pr.gin[a.rank], 0, (unsigned)((size_t)2 * P2P_MAXG * a.n * 16), 0x00020000);
  for (int i = blockIdx.x * 256 + tid; i < a.n; i += stride) {
    const float xr = res_emb ? res_emb[(size_t)tokpos[0] * a.n + i] : ld_sc1(x + i);      // requested before the wait
    const unsigned base = (unsigned)(((size_t)((unsigned)(e & 1) * P2P_MAXG) * a.n + i) * 16);
    u32x4 g[P2P_MAXG];
    unsigned spins = 0;
    unsigned long long t0 = 0;
    for (;;) {
#pragma unroll
      for (int r = 0; r < P2P_MAXG; ++r)      // all requested at once; sources beyond G re-read source 0
        g[r] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, base + (unsigned)(r < a.G ? r : 0) * (unsigned)a.n * 16u, 0, 17));      // aux 17 = sc0 sc1
      bool ok = true;
#pragma unroll
      for (int r = 0; r < P2P_MAXG; ++r) ok = ok & (g[r].y == tag) & (g[r].w == tag);
      if (ok) break;
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 255u) == 0) {
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        if (!t0) t0 = now;
        else if (now - t0 > a.wait_ticks) { *a.err = 1; break; }
      }
    }
    double s = __longlong_as_double((long long)(((unsigned long long)g[0].z << 32) | g[0].x));
#pragma unroll
    for (int r = 1; r < P2P_MAXG; ++r)
      if (r < a.G) s += __longlong_as_double((long long)(((unsigned long long)g[r].z << 32) | g[r].x));      // rank order on every rank
    const float mv = (float)s;
    x[i] = xr + mv;
    if (mv_out) mv_out[i] = mv;
  }
  p2p_end(a, e, tid);
}

// all-gather of the logits slices: every rank writes its V/G floats into every peer's (uncached) logits vector
template <int PART>
__global__ void __launch_bounds__(256) tp_p2p_gather_kernel(const P2PArgs a, const float* mine) {
  const int tid = threadIdx.x, stride = gridDim.x * 256;
  const unsigned long long e = p2p_begin(a);
  if (PART != 2) {
    for (int i = blockIdx.x * 256 + tid; i < a.n; i += stride) {
      const float v = ld_sc1(mine + i);
      for (int p = 0; p < a.G; ++p) p2p_store(a.pr.logits[a.solo ? a.rank : p] + (size_t)(a.solo ? p : a.rank) * a.n + i, v);
    }
  }
  if (PART == 1) { p2p_raise(a, e, tid); return; }
  if (PART == 0) p2p_raise(a, e, tid);
  p2p_wait(a, e, tid);
  p2p_end(a, e, tid);
}

__global__ void tp_residual_kernel(float* x, const float* res_emb, const double* sum, float* mv_out, const int* tokpos, int d) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d) return;
  const float xr = res_emb ? res_emb[(size_t)tokpos[0] * d + i] : x[i];
  const float mv = (float)sum[i];   // ONE rounding of the all-reduced fp64 sum (llama2.ts:201)
  x[i] = xr + mv;                   // accum, llama2.ts:168-170
  if (mv_out) mv_out[i] = mv;
}

// ---- peer-to-peer exchange: setup -----------------------------------------------------------------------------
// flags | inbox of fp64 partials (flag exchange) | inbox of granule pairs (pushed partials: 16 bytes per element)
static size_t p2p_bytes(const l2_ctx* c) { return (size_t)2 * P2P_MAXG * P2P_FB * 8 + (size_t)2 * P2P_MAXG * c->d * 8 + (size_t)2 * P2P_MAXG * c->d * 16; }
static void p2p_set_peer(l2_ctx* c, int r, void* base, float* logits) {
  c->p2p_peers.flags[r] = (unsigned long long*)base;
  c->p2p_peers.inbox[r] = (double*)((char*)base + (size_t)2 * P2P_MAXG * P2P_FB * 8);
  c->p2p_peers.gin[r] = (unsigned long long*)((char*)base + (size_t)2 * P2P_MAXG * P2P_FB * 8 + (size_t)2 * P2P_MAXG * c->d * 8);
  c->p2p_peers.logits[r] = logits;
}
// the device table the GEMV epilogues read (once every peer is known)
static int p2p_publish_table(l2_ctx* c) {
  TpPush t;
  memset(&t, 0, sizeof(t));
  for (int r = 0; r < c->G && r < P2P_MAXG; ++r) t.gin[r] = c->p2p_peers.gin[r];
  t.G = c->G; t.rank = c->rank; t.n = c->d; t.solo = c->solo ? 1 : 0; t.epoch = c->p2p_epoch + P2P_FB;      // (the combine launches' own counters: p2p_combine_args)
  if (!c->tp_push) HIPCHK(hipMalloc(&c->tp_push, sizeof(TpPush)));
  HIPCHK(hipMemcpy(c->tp_push, &t, sizeof(t), hipMemcpyHostToDevice));
  return L2_OK;
}
static bool p2p_pushing(const l2_ctx* c) { return c->p2p && !c->loop && c->opt_push && c->tp_push; }
static P2PArgs p2p_args(const l2_ctx* c, int n) {
  P2PArgs a;
  a.pr = c->p2p_peers; a.epoch = c->p2p_epoch; a.ticket = nullptr; a.err = c->p2p_err_dev;
  a.G = c->G; a.rank = c->rank; a.n = n; a.wait_ticks = c->p2p_wait_ticks; a.solo = c->solo ? 1 : 0; a.fenced = c->p2p_fenced;
  return a;
}
// The pushed exchange counts on words of ITS OWN (the second half of p2p_epoch): the pushing GEMV reads word 0 for every row, and the
// combine launch's block b reads word b -- the two agree only if every block of every launch that advances these words advances all of
// them.  The combine launches do (their grid is always p2p_grid(d)); the logits gather, whose grid is p2p_grid(V_loc), does not, so it
// keeps the flag exchange's words to itself (a vocabulary shard of fewer 256-element blocks than d has would otherwise leave the
// combine's upper blocks one exchange behind block 0: wrong-parity slots with matching stale tags, or waits that never end).
static P2PArgs p2p_combine_args(const l2_ctx* c) { P2PArgs a = p2p_args(c, c->d); a.epoch = c->p2p_epoch + P2P_FB; return a; }
static int p2p_grid(int n) { const int b = (n + 255) / 256; return b > P2P_FB ? P2P_FB : (b < 1 ? 1 : b); }

// Own buffers (every tensor-parallel context): inbox + flags and the gathered-logits vector are UNCACHED device
// memory, because peers write them while this GPU's L2 knows nothing about it.
static int p2p_alloc(l2_ctx* c) {
  if (c->G > P2P_MAXG) return L2_OK;
  const char* mode = getenv("L2_TP_ALLREDUCE");
  if (mode && !strcmp(mode, "rccl")) return L2_OK;
  if (hipExtMallocWithFlags(&c->p2p_base, p2p_bytes(c), hipDeviceMallocUncached) != hipSuccess) { c->p2p_base = nullptr; (void)hipGetLastError(); return L2_OK; }
  HIPCHK(hipMemset(c->p2p_base, 0, p2p_bytes(c)));
  HIPCHK(hipMalloc(&c->p2p_epoch, (size_t)2 * P2P_FB * 8));      // [0, FB): flag exchange + logits gather; [FB, 2 FB): pushed exchange (p2p_combine_args)
  HIPCHK(hipMemset(c->p2p_epoch, 0, (size_t)2 * P2P_FB * 8));
  HIPCHK(hipHostMalloc(&c->p2p_err, sizeof(int), hipHostMallocMapped));
  *c->p2p_err = 0;
  HIPCHK(hipHostGetDevicePointer((void**)&c->p2p_err_dev, c->p2p_err, 0));
  return L2_OK;
}

// Multi-process group: IPC handles of every rank's buffers travel through one RCCL all-gather; then ONE exchange on
// a known vector is checked against the closed form, and the ranks agree (all-reduce of a flag) whether the
// peer-to-peer path is used -- any rank that cannot map or complete it sends everybody back to the RCCL collectives.
enum { NCCL_UINT8 = 1, NCCL_INT32 = 2, NCCL_MIN = 3 };
__global__ void p2p_selftest_fill(double* partial, float* x, int rank, int n, int k) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { partial[i] = (double)((rank + 1) * (k + 1)) + 0.5 * (double)i; x[i] = 0.0f; }
}
// the pushed form's stand-in for a GEMV epilogue: one wave per 64 elements, lane r < G hands element i to rank r (kernels.hip.h: tp_push_row)
__global__ void p2p_selftest_push(const TpPush* p, float* x, int rank, int n, int k) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const PushCtx pc = tp_push_ctx(p, lane);
  for (int j = 0; j < 64; ++j) {
    const int i = wave * 64 + j;
    if (i < n) { tp_push_row(pc, i, (double)((rank + 1) * (k + 1)) + 0.5 * (double)i, lane); if (lane == 0) x[i] = 0.0f; }
  }
}
// ... and the check of what the exchange left in x, on the device: the soak below runs its exchanges back to back, like the
// decode step does, not one per host round trip
__global__ void p2p_selftest_check(const float* x, int G, int n, int k, int* bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && x[i] != (float)(0.5 * G * (G + 1) * (k + 1) + 0.5 * (double)i * G)) atomicAdd(bad, 1);
}
__global__ void p2p_selftest_fill_logits(float* mine, int rank, int n, int k) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) mine[i] = (float)(rank * 4096 + k) + (float)i * 0.25f;
}
__global__ void p2p_selftest_check_logits(const float* all, int G, int n, int k, int* bad) {      // all: the gathered (uncached) vector of G slices
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < G * n) { const int r = i / n, j = i - r * n; if (all[i] != (float)(r * 4096 + k) + (float)j * 0.25f) atomicAdd(bad, 1); }
}
// Test hook (L2_TP_IPC_DIR=<directory>): the ranks are separate PROCESSES that meet through files instead of an RCCL
// communicator, so the IPC mapping, the self-test and the peer-to-peer exchange run between processes on a box with one GPU
// (RCCL refuses two ranks on one device).  No fallback in this mode: the exchange works or creation fails.
static bool file_exchange(const char* dir, const char* tag, int rank, int G, const void* mine, void* all, size_t bytes) {
  char path[512];
  snprintf(path, sizeof(path), "%s/%s.%d.tmp", dir, tag, rank);
  FILE* f = fopen(path, "wb");
  if (!f) return false;
  const bool wrote = fwrite(mine, 1, bytes, f) == bytes;
  fclose(f);
  char final_path[512];
  snprintf(final_path, sizeof(final_path), "%s/%s.%d", dir, tag, rank);
  if (!wrote || rename(path, final_path) != 0) return false;
  for (int r = 0; r < G; ++r) {
    snprintf(path, sizeof(path), "%s/%s.%d", dir, tag, r);
    bool got = false;
    for (int tries = 0; tries < 6000 && !got; ++tries) {            // 60 s
      f = fopen(path, "rb");
      if (f) { got = fread((char*)all + (size_t)r * bytes, 1, bytes, f) == bytes; fclose(f); }
      if (!got) usleep(10000);
    }
    if (!got) return false;
  }
  return true;
}

static int p2p_connect_ipc(l2_ctx* c) {
  if (!c->p2p_base) return c->ipc_dir.empty() ? L2_OK : fail(L2_E_COMM, "L2_TP_IPC_DIR: no peer-to-peer inbox was allocated");
  const int G = c->G;
  const char* dir = c->ipc_dir.empty() ? nullptr : c->ipc_dir.c_str();
  int round = 0;
  auto all_min = [&](int v, int* out) -> int {                      // every rank learns the minimum of v
    if (dir) {
      std::vector<int> vs(G, 0);
      char tag[32]; snprintf(tag, sizeof(tag), "min%d", round++);
      if (!file_exchange(dir, tag, c->rank, G, &v, vs.data(), sizeof(int))) return fail(L2_E_COMM, "L2_TP_IPC_DIR: a rank did not arrive");
      *out = *std::min_element(vs.begin(), vs.end());
      return L2_OK;
    }
    int* d_v = nullptr;
    HIPCHK(hipMalloc(&d_v, sizeof(int)));
    HIPCHK(hipMemcpy(d_v, &v, 4, hipMemcpyHostToDevice));
    NCCLCHK(g_rccl.AllReduce(d_v, d_v, 1, NCCL_INT32, NCCL_MIN, c->comm, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out, d_v, 4, hipMemcpyDeviceToHost));
    hipFree(d_v);
    return L2_OK;
  };
  struct Rec { hipIpcMemHandle_t base, logits; };
  static_assert(sizeof(Rec) == 128, "two 64-byte IPC handles");
  Rec mine;
  bool ok = hipIpcGetMemHandle(&mine.base, c->p2p_base) == hipSuccess && hipIpcGetMemHandle(&mine.logits, c->logits) == hipSuccess;
  (void)hipGetLastError();
  std::vector<Rec> all(G);
  if (dir) {
    if (!file_exchange(dir, "handles", c->rank, G, &mine, all.data(), sizeof(Rec))) return fail(L2_E_COMM, "L2_TP_IPC_DIR: a rank did not arrive");
  } else {
    Rec* d_all = nullptr;
    HIPCHK(hipMalloc(&d_all, sizeof(Rec) * (G + 1)));
    HIPCHK(hipMemcpy(d_all + G, &mine, sizeof(Rec), hipMemcpyHostToDevice));
    NCCLCHK(g_rccl.AllGather(d_all + G, d_all, sizeof(Rec), NCCL_UINT8, c->comm, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(all.data(), d_all, sizeof(Rec) * G, hipMemcpyDeviceToHost));
    hipFree(d_all);
  }
  for (int r = 0; r < G && ok; ++r) {
    if (r == c->rank) { p2p_set_peer(c, r, c->p2p_base, c->logits); continue; }
    void *pb = nullptr, *pl = nullptr;
    if (hipIpcOpenMemHandle(&pb, all[r].base, hipIpcMemLazyEnablePeerAccess) != hipSuccess ||
        hipIpcOpenMemHandle(&pl, all[r].logits, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { ok = false; (void)hipGetLastError(); break; }
    c->p2p_opened.push_back(pb); c->p2p_opened.push_back(pl);
    p2p_set_peer(c, r, pb, (float*)pl);
  }
  if (ok && p2p_publish_table(c)) ok = false;
  // every rank learns whether every rank mapped everything BEFORE anybody waits on a peer
  int h_ok = ok ? 1 : 0;
  { const int rc_ = all_min(h_ok, &h_ok); if (rc_) return rc_; }
  if (h_ok) {   // four exchanges (each inbox slot is reused once) on known vectors: sum over ranks of ((rank + 1)(k + 1) + i / 2)
    const int n = c->d;
    std::vector<float> got(n);
    // (the form the decode step will use: rows pushed by the "GEMV" + the combine launch, or partials + the flag exchange)
    auto exchange = [&](int k) {
      if (p2p_pushing(c)) {
        hipLaunchKernelGGL(p2p_selftest_push, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->tp_push, c->xb2, c->rank, n, k);
        hipLaunchKernelGGL(tp_p2p_combine_kernel, dim3(p2p_grid(n)), dim3(256), 0, c->stream, p2p_combine_args(c), c->xb2, nullptr, nullptr, c->tokpos, (unsigned*)nullptr);
      } else {
        hipLaunchKernelGGL(p2p_selftest_fill, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->partial, c->xb2, c->rank, n, k);
        hipLaunchKernelGGL(tp_p2p_reduce_kernel<0>, dim3(p2p_grid(n)), dim3(256), 0, c->stream, p2p_args(c, n), c->partial, c->xb2, nullptr, nullptr, c->tokpos);
      }
    };
    c->p2p = true;      // (p2p_pushing looks at it; settled below)
    for (int k = 0; k < 4 && h_ok; ++k) {
      exchange(k);
      HIPCHK(hipStreamSynchronize(c->stream));
      HIPCHK(hipMemcpy(got.data(), c->xb2, (size_t)n * 4, hipMemcpyDeviceToHost));
      if (*c->p2p_err) h_ok = 0;
      for (int i = 0; i < n && h_ok; ++i) if (got[i] != (float)(0.5 * G * (G + 1) * (k + 1) + 0.5 * (double)i * G)) h_ok = 0;
    }
    // ... then a soak: 96 all-reduces and 32 logits gathers BACK TO BACK on the stream, every element of every result checked on
    // the device.  The exchange has no fences (its ordering rests on drained write-through stores, see the top of this file), and
    // no multi-GPU box has ever run it: a peer whose payload could become visible after its flag has 128 chances to show it here,
    // in the timing the decode step has -- and sends the whole group to the RCCL collectives if it does (a group that met through
    // files, L2_TP_FILE_RENDEZVOUS / L2_TP_IPC_DIR, has no communicator to fall back to: there creation FAILS, below, and bench.py
    // moves on to independent replicas).
    if (h_ok) {
      int* d_bad = nullptr;
      HIPCHK(hipMalloc(&d_bad, sizeof(int)));
      HIPCHK(hipMemsetAsync(d_bad, 0, sizeof(int), c->stream));
      const int nl = c->V_loc;
      for (int k = 4; k < 100; ++k) {
        exchange(k);
        hipLaunchKernelGGL(p2p_selftest_check, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->xb2, G, n, k, d_bad);
        if (k % 3 == 0) {
          hipLaunchKernelGGL(p2p_selftest_fill_logits, dim3((nl + 255) / 256), dim3(256), 0, c->stream, c->logits_loc, c->rank, nl, k);
          hipLaunchKernelGGL(tp_p2p_gather_kernel<0>, dim3(p2p_grid(nl)), dim3(256), 0, c->stream, p2p_args(c, nl), c->logits_loc);
          hipLaunchKernelGGL(p2p_selftest_check_logits, dim3((G * nl + 255) / 256), dim3(256), 0, c->stream, c->logits, G, nl, k, d_bad);
        }
      }
      int bad = 0;
      HIPCHK(hipStreamSynchronize(c->stream));
      HIPCHK(hipMemcpy(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost));
      hipFree(d_bad);
      if (bad || *c->p2p_err) h_ok = 0;
    }
    *c->p2p_err = 0;
    HIPCHK(hipMemset(c->xb2, 0, (size_t)n * 4));
    HIPCHK(hipMemset(c->logits, 0, (size_t)c->V * 4));
    { const int rc_ = all_min(h_ok, &h_ok); if (rc_) return rc_; }
  }
  c->p2p = h_ok != 0;
  c->p2p_peers_ready = true;
  if (!c->p2p && dir) return fail(L2_E_COMM, "L2_TP_IPC_DIR: the peer-to-peer exchange between the processes failed its self-test");
  if (!c->p2p && getenv("L2_TP_ALLREDUCE") && !strcmp(getenv("L2_TP_ALLREDUCE"), "p2p"))
    return fail(L2_E_COMM, "L2_TP_ALLREDUCE=p2p but the peer-to-peer exchange could not be set up on every rank");
  return L2_OK;
}

// Enqueue one transformer() call (llama2.ts:205-303) reading {token,pos} from device memory.
// The two collectives of the tensor-parallel step: RCCL, or the loopback test hook.
static int tp_all_reduce(l2_ctx* c, hipStream_t st) {
  if (!c->loop) { NCCLCHK(g_rccl.AllReduce(c->partial, c->partial, (size_t)c->d, NCCL_FLOAT64, NCCL_SUM, c->comm, st)); return L2_OK; }
  LoopGroup& g = *c->loop;
  HIPCHK(hipStreamSynchronize(st));
  { std::lock_guard<std::mutex> lk(g.mu); g.ptrs[c->rank] = c->partial; }
  if (!g.wait()) return fail(L2_E_COMM, "loopback all-reduce: a rank never arrived");
  LoopPtrs in;
  for (int r = 0; r < g.G; ++r) in.p[r] = (const double*)g.ptrs[r];
  hipLaunchKernelGGL(loop_sum_kernel, dim3((c->d + 255) / 256), dim3(256), 0, st, c->loop_tmp, in, g.G, c->d);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st));
  if (!g.wait()) return fail(L2_E_COMM, "loopback all-reduce: a rank never arrived");   // every rank has read every partial
  HIPCHK(hipMemcpyAsync(c->partial, c->loop_tmp, (size_t)c->d * 8, hipMemcpyDeviceToDevice, st));
  return L2_OK;
}
static int tp_all_gather_logits(l2_ctx* c, hipStream_t st) {
  if (!c->loop) { NCCLCHK(g_rccl.AllGather(c->logits_loc, c->logits, (size_t)c->V_loc, NCCL_FLOAT32, c->comm, st)); return L2_OK; }
  LoopGroup& g = *c->loop;
  HIPCHK(hipStreamSynchronize(st));
  { std::lock_guard<std::mutex> lk(g.mu); g.ptrs[c->rank] = c->logits_loc; }
  if (!g.wait()) return fail(L2_E_COMM, "loopback all-gather: a rank never arrived");
  for (int r = 0; r < g.G; ++r)
    HIPCHK(hipMemcpyAsync(c->logits + (size_t)r * c->V_loc, g.ptrs[r], (size_t)c->V_loc * 4, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipStreamSynchronize(st));
  if (!g.wait()) return fail(L2_E_COMM, "loopback all-gather: a rank never arrived");
  return L2_OK;
}

// one all-reduce + residual of the tensor-parallel step: ONE kernel; the loopback test group (all ranks on one GPU)
// runs its two halves around a host barrier instead (see tp_p2p_reduce_kernel)
static int p2p_reduce(l2_ctx* c, hipStream_t st, const float* res_emb, float* mv_out, unsigned* bump) {
  const dim3 grid(p2p_grid(c->d));
  if (p2p_pushing(c)) {
    l2_launch(c, tp_p2p_combine_kernel, grid, dim3(256), 0, st, p2p_combine_args(c), c->x, res_emb, mv_out, (const int*)c->tokpos, bump);
  } else if (!c->loop) {
    l2_launch(c, tp_p2p_reduce_kernel<0>, grid, dim3(256), 0, st, p2p_args(c, c->d), (const double*)c->partial, c->x, res_emb, mv_out, (const int*)c->tokpos);
  } else {
    hipLaunchKernelGGL(tp_p2p_reduce_kernel<1>, grid, dim3(256), 0, st, p2p_args(c, c->d), c->partial, c->x, res_emb, mv_out, c->tokpos);
    HIPCHK(hipStreamSynchronize(st));
    if (!c->loop->wait()) return fail(L2_E_COMM, "loopback all-reduce: a rank never arrived");
    hipLaunchKernelGGL(tp_p2p_reduce_kernel<2>, grid, dim3(256), 0, st, p2p_args(c, c->d), c->partial, c->x, res_emb, mv_out, c->tokpos);
  }
  LCHK(hipGetLastError());
  return L2_OK;
}

// The ranks meet once on the HOST right before their first peer-to-peer step: whatever happened between creation and now
// (per-rank checkpoint I/O, synthetic fill) is skew the in-kernel flag wait should not have to absorb.
static int p2p_first_sync(l2_ctx* c) {
  if (c->loop) { if (!c->loop->wait()) return fail(L2_E_COMM, "loopback group: a rank never arrived"); }
  else if (!c->ipc_dir.empty()) {
    int mine = 1; std::vector<int> all(c->G, 0);
    if (!file_exchange(c->ipc_dir.c_str(), "first", c->rank, c->G, &mine, all.data(), sizeof(int))) return fail(L2_E_COMM, "L2_TP_IPC_DIR: a rank did not arrive for the first step");
  } else if (c->comm) {
    int* d_v = nullptr;
    HIPCHK(hipMalloc(&d_v, sizeof(int)));
    HIPCHK(hipMemsetAsync(d_v, 0, sizeof(int), c->stream));
    NCCLCHK(g_rccl.AllReduce(d_v, d_v, 1, NCCL_INT32, NCCL_SUM, c->comm, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    hipFree(d_v);
  }
  c->p2p_synced = true;
  return L2_OK;
}

static int check_p2p(l2_ctx* c) {   // after a stream sync: did a peer-to-peer wait (or a hand-off wait of the fused QKV + attention launch) give up?
  if (c->h_herr && *c->h_herr) {
    *c->h_herr = 0;
    hipMemsetAsync(c->gran, 0, ((size_t)c->d_loc + 2 * (size_t)c->kvd_loc) * 8, c->stream);      // tags of the broken launch: gone (the counters only ever grow)
    hipMemsetAsync(c->gran_ep + c->H_loc, 0, 4, c->stream);                                      // the device-side "a wait gave up" words: armed again
    if (c->awo_ep) { hipMemsetAsync(c->awo_ep + 1, 0, 4, c->stream); hipMemsetAsync(c->awo_gran, 0, (size_t)c->d_loc * 8, c->stream); }
    hipStreamSynchronize(c->stream);
    // a tensor-parallel rank has pushed the rows of that invalid step to its peers with valid tags: they have added garbage without an error of
    // their own, so the group's state is no longer the reference's -- this rank fails fast from now on (its peers time out at their next exchange)
    if (c->tp_path) c->broken = true;
    return fail(L2_E_HIP, "a hand-off granule inside a fused launch never arrived (bounded wait gave up); the step's results are invalid%s", c->tp_path ? " and so are its peers' (the context is unusable)" : "");
  }
  if (c->p2p_err && *c->p2p_err) {
    *c->p2p_err = 0;
    c->broken = true;   // the device-side epoch / flag state is out of step with the peers for good: fail fast from now on
    return fail(L2_E_COMM, "peer-to-peer exchange: a rank never raised its flag (wait of %.0f s gave up); the context is unusable", (double)c->p2p_wait_ticks / 1e8);
  }
  return L2_OK;
}

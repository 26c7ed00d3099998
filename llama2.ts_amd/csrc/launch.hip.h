// launch.hip.h -- launch geometry of the GEMV phases and the attention kernel
// Part of the one translation unit llama2_hip.hip (included there, in order); not a stand-alone header.
#pragma once

// Launch geometry.  A wave owns R rows at a time and loads U x 64 float4 per row per batch; see
// phase_kernel.  (R, U) is picked so one batch is ~16 loads per lane and a short row is one batch; the
// grid is capped at what is co-resident so every wave loops over several row groups with its two
// register sets always full (the per-workgroup prologue is then amortised as well).
struct Geo { int R, U, pre, nwaves, grid; bool vec; };

// The tuning knobs a context carries (development switches) and the chip: everything pick_geo needs besides the shape, so that the
// selection is a pure function (l2_debug_pick_geo: a CPU test walks every shape of configs.py, the tensor-parallel shards and a few
// hundred random ones through it and checks that each selected template point is one the library instantiates).
struct GeoKnobs { int n_cus, tune_U, tune_nwaves, tune_gridcap; };
static GeoKnobs geo_knobs(const l2_ctx* c) { return {c->n_cus, c->tune_U, c->tune_nwaves, c->tune_gridcap}; }

// Template points of the streaming kernel: U (64-lane float4 sub-batches per row and batch) in {2, 4} -- 3 for classifiers of
// 513 .. 768 columns -- and PRE (float4 per thread per staging round) in {1, 4, 12}.  (Round 5: U = 1 and PRE = 2 are gone -- a
// row of at most 64 float4 takes the two-sub-batch kernel, whose second sub-batch re-reads the row's last float4 against zeros of x;
// a vector of up to two rounds of float4 is staged by the four-per-thread round -- 116 -> 65 instances of this kernel.)
static Geo pick_geo_pure(const GeoKnobs& kb, int mode, int rows, int n) {
  Geo g;
  g.vec = (n % 4) == 0;
  const int n4 = n / 4;
  const int pair = (mode == MODE_W13) ? 2 : 1;  // W13: R covers R/2 rows of w1 + R/2 of w3
  // Measured on MI355X (tools/sweep_gemv.py, 7B shapes): small batches at high occupancy win -- R = 2 rows,
  // U = 2..4 (8..16 KiB in flight per wave, <= 64 VGPRs => 8 waves per SIMD) reach 6.0-6.4 TB/s, R = 4 / U = 8
  // variants (more bytes per wave, fewer waves) stay below 5.5.
  g.R = 2;
  int U = 2;
  if (n4 > 128 && n4 <= 256) U = 4;
  if (mode == MODE_CLS && n4 > 128 && n4 <= 192) U = 3;   // 768 columns (stories110M): three float4 per lane cover a row exactly; with U = 4 a quarter of the lanes re-read the last one (16.8 -> 16.5 us)
  if (kb.tune_U == 2 || kb.tune_U == 4 || (kb.tune_U == 3 && mode == MODE_CLS && n4 > 128 && n4 <= 192)) U = kb.tune_U;      // (U = 3 exists for the widths it covers exactly)
  g.U = U;
  const int groups = (rows * pair + g.R - 1) / g.R;
  // 4 waves per workgroup from 512 row groups on (2 up to round 3: the q / k / v shard of an 8-rank group -- 768 groups -- runs
  // 9.0 -> 7.1 us with 4: fewer, fuller workgroups share the staged x; tools/tp_shard_sweep.py)
  g.nwaves = groups >= 512 ? 4 : (groups >= 256 ? 2 : 1);
  if (kb.tune_nwaves == 1 || kb.tune_nwaves == 2 || kb.tune_nwaves == 4) g.nwaves = kb.tune_nwaves;
  // q / k / v of 513 .. 1024 columns (U = 4) exist with the one-per-thread staging round only, i.e. on four waves: with whole (d, d)
  // matrices such a width always has the 512 row groups that select them, but a grouped-query or tensor-parallel phase has
  // d_loc + 2 kv_dim_loc rows (d = 640 with 128-wide k / v: 448 groups), and L2_TUNE_NWAVES can ask for fewer
  if (mode == MODE_QKV && U == 4) g.nwaves = 4;
  // staging: PRE float4 per thread per round, one round if it can cover the (padded) vector
  const int cpi = 64 * U, npad4 = ((n4 + cpi - 1) / cpi) * cpi, nth = 64 * g.nwaves;
  // one staging round whenever 12 float4 per thread cover the vector (w2 of Llama-2-7B: 11008 floats = 2752 float4 on 256
  // threads): every extra round is one more dependent L2 round trip in front of the first FMA
  g.pre = (npad4 <= nth) ? 1 : (npad4 <= 4 * nth ? 4 : 12);
  int grid = (groups + g.nwaves - 1) / g.nwaves;
  // persistent grid: 2 workgroups (8 waves) per CU, each wave looping over row groups with both register sets
  // full, measured best on the 7B shapes (129.6 us of GEMV per layer vs 134.1 at 6 per CU)
  const int cap = kb.tune_gridcap > 0 ? kb.tune_gridcap : kb.n_cus * 2;
  if (grid > cap) {
    // balanced: every wave gets the same number k of row groups (w1/w3 of 7B: 5504 groups on 2048 waves would
    // leave a third of the chip idle in the last round; 459 workgroups x 4 waves x 3 groups covers it evenly).
    // Measured against a full grid that deals the odd groups evenly over the CUs (the 53 CUs with one workgroup are done
    // after 44 us, the others after 53-57: tools/stamps.py STAMPS_WG=2): the full grid's last round ran as slowly as any
    // other, 216.5 vs 219.3 tok/s.
    const int waves_cap = cap * g.nwaves;
    const int k = (groups + waves_cap - 1) / waves_cap;
    grid = (groups + g.nwaves * k - 1) / (g.nwaves * k);
  }
  g.grid = grid < 1 ? 1 : grid;
  return g;
}
static Geo pick_geo(const l2_ctx* c, int mode, int rows, int n, int dim) { (void)dim; return pick_geo_pure(geo_knobs(c), mode, rows, n); }

#ifdef L2_STAMPS
static int g_stamp_slot = 0;   // each launch of the enqueue gets its own 36-stamp slot
extern "C" int l2_debug_stamps(l2_ctx* c, unsigned long long* out, size_t n) {
  hipStreamSynchronize(c->stream);
  return hipMemcpy(out, c->dbg, n * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
}
#endif


// LDS a launch may ask for: 160 KiB per CU on gfx950, opted into per kernel (the default cap is 64 KiB).
template <class K>
static hipError_t lds_opt_in(K kernel, size_t lds) {
  if (lds <= 64 * 1024) return hipSuccess;
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

// Every launch of the decode step goes through here: a HIP launch (eager, or captured into a hipGraph), or -- while the step of a
// level is being recorded for the library's own AQL queue (aql_queue.h) -- one packet of that recording: the kernel by its name in
// the code object, the explicit arguments packed as the kernel-argument segment lays them out (each at its natural alignment).
template <class T>
static void aql_pack(char* buf, size_t& off, const T& v) {
  off = (off + alignof(T) - 1) & ~(alignof(T) - 1);
  memcpy(buf + off, &v, sizeof(T));
  off += sizeof(T);
}
static void pollute_behind(const l2_ctx* c, hipStream_t st);
template <class... KA, class... A>
static void l2_launch_one(const l2_ctx* c, void (*kernel)(KA...), dim3 grid, dim3 block, size_t lds, hipStream_t st, const A&... a);
template <class... KA, class... A>
static void l2_launch(const l2_ctx* c, void (*kernel)(KA...), dim3 grid, dim3 block, size_t lds, hipStream_t st, const A&... a) {
  l2_launch_one(c, kernel, grid, block, lds, st, a...);
  if (c->opt_pollute) pollute_behind(c, st);      // test hook: the coherence rule's adversary behind every launch (kernels.hip.h: l1_pollute_kernel)
}
template <class... KA, class... A>
static void l2_launch_one(const l2_ctx* c, void (*kernel)(KA...), dim3 grid, dim3 block, size_t lds, hipStream_t st, const A&... a) {
  static_assert(sizeof...(KA) == sizeof...(A), "argument count");
  if (c->aql_rec) {
    l2_ctx* m = const_cast<l2_ctx*>(c);
    char buf[1024];
    size_t off = 0;
    static_assert((sizeof(KA) + ... + 0) + 8 * sizeof...(KA) <= sizeof(buf), "kernel arguments exceed the packing buffer");
    (aql_pack<KA>(buf, off, static_cast<KA>(a)), ...);
    const char* name = hipKernelNameRefByPtr(reinterpret_cast<const void*>(kernel), st);
    const unsigned g[3] = {grid.x, grid.y, grid.z}, b[3] = {block.x, block.y, block.z};
    if (!name || aql_record(m->aql, m->aql_rec, name, g, b, (unsigned)lds, buf, off)) m->aql_rec_failed = true;
    return;
  }
  hipLaunchKernelGGL(kernel, grid, block, lds, st, a...);
}

static void pollute_behind(const l2_ctx* c, hipStream_t st) {
  PolluteArgs p;
  memset(&p, 0, sizeof(p));
  int nb = 0;
  auto add = [&](const void* b, size_t bytes) { if (b && bytes && nb < 20) { p.buf[nb] = (const float*)b; p.bytes[nb] = (unsigned)(bytes > 16384 ? 16384 : bytes); ++nb; } };
  add(c->x, (size_t)c->d * 4); add(c->xn, (size_t)c->d * 4); add(c->xb, (size_t)c->d_loc * 4); add(c->xb2, (size_t)c->d * 4);
  add(c->hb, (size_t)c->h_loc * 4); add(c->hb2, (size_t)c->h_loc * 4); add(c->q, (size_t)c->d_loc * 4); add(c->k, (size_t)c->kvd_loc * 4); add(c->v, (size_t)c->kvd_loc * 4);
  add(c->logits_loc, (size_t)c->V_loc * 4); add(c->amax, 8 * 16 * 8); add(c->attn_counter, (size_t)c->H_loc * CTR_STRIDE * 4);
  add(c->attn_part, (size_t)c->H_loc * 8 * (((size_t)c->hs + 2 + 15) & ~(size_t)15) * 8); add(c->gran, ((size_t)c->d_loc + 2 * (size_t)c->kvd_loc) * 8);
  add(c->gran_ep, (size_t)c->H_loc * 4 + 16); add(c->tokpos, 16); add(c->d_tokens, (size_t)c->S * 4); add(c->att, (size_t)c->H_loc * c->S * 4);
  p.nb = nb; p.kc = c->kc; p.vc = c->vc; p.tokpos = c->tokpos; p.sink = c->pollute_sink; p.L = c->L; p.S = c->S; p.kvd = c->kvd_loc;
  l2_launch_one(c, l1_pollute_kernel, dim3(c->n_cus * 4), dim3(256), 0, st, p);      // four workgroups per CU: every CU gets some
}

// Launch with optional start / stop events on THE DISPATCH (hipExtLaunchKernelGGL): their elapsed time is the kernel's
// own execution time, as a kernel trace reports it -- no launch boundary, no event-record latency (the in-situ probe).
template <class K, class A>
static void launch_probed(const l2_ctx* c, K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t st, const A& a, bool probe) {
  l2_ctx* m = const_cast<l2_ctx*>(c);
  if (probe && m->probe_on && m->probe_used + 2 <= m->probe.size()) {
    hipEvent_t e0 = m->probe[m->probe_used], e1 = m->probe[m->probe_used + 1];
    m->probe_used += 2;
    hipExtLaunchKernelGGL(kernel, grid, block, lds, st, e0, e1, 0, a);
  } else {
    l2_launch(c, kernel, grid, block, lds, st, a);
  }
}

// Latency form (kernels.hip.h: phase_small_kernel) for matrices of at most `small_max` floats whose input vector fits
// 8 float4 per lane; everything else (Llama-2-7B's phases, every classifier) streams through phase_kernel.
static bool use_small_pure(long long small_max, int mode, int rows, int n) {
  if (n % 4 || n > 2048 || mode == MODE_CLS) return false;
  if (mode == MODE_W13 && n > 1536) return false;      // two matrices x 8 float4 of x per lane do not fit the register file (the instance spilled): stream
  const long long elems = (long long)rows * n * (mode == MODE_W13 ? 2 : 1);
  return elems <= small_max;
}
static bool use_small(const l2_ctx* c, int mode, int rows, int n) { return use_small_pure((long long)c->small_max, mode, rows, n); }

// Geometry of the latency form (pure): XV template point (float4 of x per lane: 1 .. 4, 6, 8), rows per wave and group (1 for a single
// matrix while one row per wave still leaves waves idle, else 2 -- RoPE neighbours / a (w1, w3) row pair always travel together) and the
// grid: one workgroup per CU as soon as there are that many row groups -- fewer busy waves per CU beat fewer CUs (the CU's 64 B / clock
// address path is what a 7-wave workgroup queues on): stories15M 8 660 -> 8 945 tok/s, stories110M +0.7 % (profiles/r03/ab_small_kernel_grid.txt).
// ONE function for the launcher and the debug hook below.
struct SmallGeo { int xv, R, grid; };
static SmallGeo small_geo_pure(int n_cus, int mode, int rows, int n) {
  const int xv = (n / 4 + 63) / 64, xvt = xv <= 4 ? xv : (xv <= 6 ? 6 : 8);
  const bool pair = (mode == MODE_QKV || mode == MODE_W13);
  const bool r1 = !pair && rows <= n_cus * 7;      // seven compute waves per workgroup (kernels.hip.h)
  const int rpg = (mode == MODE_W13) ? 1 : (r1 ? 1 : 2);
  const int groups = (rows + rpg - 1) / rpg;
  int grid = groups < n_cus ? groups : n_cus;
  if (grid < 1) grid = 1;
  return {xvt, r1 ? 1 : 2, grid};
}

// What launch_phase would launch for a phase of `rows` x `n` (pure: no context, no GPU): out = {form, U or XV, PRE or R, waves per workgroup,
// grid, packed-capable}; form 0 = streaming vector kernel, 1 = latency form, 2 = scalar kernel (n % 4 != 0).
extern "C" int l2_debug_pick_geo(int mode, int rows, int n, int n_cus, int small_max, int out[6]) {
  if (!out || mode < 0 || mode > MODE_CLS || rows <= 0 || n <= 0 || n_cus <= 0) return L2_E_ARG;
  if (use_small_pure(small_max, mode, rows, n)) {
    const SmallGeo sg = small_geo_pure(n_cus, mode, rows, n);
    out[0] = 1; out[1] = sg.xv; out[2] = sg.R; out[3] = 8; out[4] = sg.grid; out[5] = 0;
    return L2_OK;
  }
  const GeoKnobs kb = {n_cus, 0, 0, 0};
  const Geo g = pick_geo_pure(kb, mode, rows, n);
  out[0] = g.vec ? 0 : 2; out[1] = g.U; out[2] = g.pre; out[3] = g.nwaves; out[4] = g.grid;
  out[5] = (g.vec && g.U == 2 && (n / 4) % 64 == 0 && n / 4 > 128) ? 1 : 0;
  return L2_OK;
}

template <int MODE, int XV>
static hipError_t launch_small_xv(const l2_ctx* c, const PhaseArgs& a, const SmallGeo& sg, hipStream_t st) {
  constexpr bool pair = (MODE == MODE_QKV || MODE == MODE_W13);       // row pairs: RoPE neighbours / (w1, w3)
  const size_t lds = (size_t)XV * 64 * 16;
  if (!pair && sg.R == 1) launch_probed(c, phase_small_kernel<MODE, XV, pair ? 2 : 1>, dim3(sg.grid), dim3(512), lds, st, a, MODE == MODE_W13);
  else launch_probed(c, phase_small_kernel<MODE, XV, 2>, dim3(sg.grid), dim3(512), lds, st, a, MODE == MODE_W13);
  return hipGetLastError();
}

template <int MODE>
static hipError_t launch_small(const l2_ctx* c, const PhaseArgs& a, hipStream_t st) {
  const SmallGeo sg = small_geo_pure(c->n_cus, MODE, a.rows, a.n);
  switch (sg.xv) {
    case 1: return launch_small_xv<MODE, 1>(c, a, sg, st);
    case 2: return launch_small_xv<MODE, 2>(c, a, sg, st);
    case 3: return launch_small_xv<MODE, 3>(c, a, sg, st);
    case 4: return launch_small_xv<MODE, 4>(c, a, sg, st);
    case 6: return launch_small_xv<MODE, 6>(c, a, sg, st);
    default: if constexpr (MODE == MODE_W13) return hipErrorInvalidValue; else return launch_small_xv<MODE, 8>(c, a, sg, st);   // use_small() keeps W13 out
  }
}

template <int MODE>
static hipError_t launch_phase(const l2_ctx* c, const PhaseArgs& a_in, hipStream_t st) {
  PhaseArgs a = a_in;
  a.rot = c->tune_rot;
#ifdef L2_STAMPS
  a.dbg_wg = c->dbg + 66 * 108 + (size_t)(g_stamp_slot % 64) * 2048;
  a.dbg = c->dbg + (size_t)(g_stamp_slot++ % 64) * 108;
#endif
  if (use_small(c, MODE, a.rows, a.n)) return launch_small<MODE>(c, a, st);
  const Geo g = pick_geo(c, MODE, a.rows, a.n, a.dim);
  if (!g.vec) a.wp = nullptr;
  const dim3 grid(g.grid), block(64 * g.nwaves);
  if (!g.vec) {
    const size_t lds = (((size_t)a.n * 4 + 15) & ~(size_t)15) + 64;
    hipError_t e = lds_opt_in(&phase_kernel_scalar<MODE>, lds);
    if (e != hipSuccess) return e;
    l2_launch(c, phase_kernel_scalar<MODE>, grid, block, lds, st, a);
    return hipGetLastError();
  }
  const int n4 = a.n / 4, cpi = 64 * g.U;
  const int npad4 = ((n4 + cpi - 1) / cpi) * cpi;
  const bool norm = (MODE == MODE_QKV || MODE == MODE_W13 || MODE == MODE_CLS);
  const int round4 = g.pre * 64 * g.nwaves;                   // PRE * nthreads (kernels.hip.h)
  const int nstage4 = ((npad4 + round4 - 1) / round4) * round4;
  const size_t lds = (size_t)nstage4 * (norm ? 2 : 1) * 16 + 64;
  // the repacked copy is only good for the geometry it was packed for (U = 2: the only batch width with more than one batch per row)
  if (a.wp && !(g.U == 2 && g.U == c->packed[MODE].U && g.nwaves == c->packed[MODE].nwaves && g.grid == c->packed[MODE].grid)) a.wp = nullptr;
  if (!a.wp && !a.w0) return hipErrorInvalidValue;       // one copy of the weights: the row-major tensor is gone and this launch cannot read the repacked one
  // (tensor-parallel push: an instance of its own for wo / w2, see phase_body)
#define L2_LAUNCH_K(KERNEL) do { hipError_t e_ = lds_opt_in(&KERNEL, lds); if (e_ != hipSuccess) return e_; launch_probed(c, KERNEL, grid, block, lds, st, a, MODE == MODE_W13); } while (0)
  // (a repacked row has more than 256 float4: its staging round is never the one-per-thread one -- no repacked instance with PRE = 1)
#define L2_LAUNCH(UU, PP) do { constexpr bool pk_ = (UU == 2 && PP != 1); \
                               if constexpr (MODE == MODE_WO || MODE == MODE_W2) { if (a.push) { \
                                 if constexpr (pk_) { if (a.wp) { L2_LAUNCH_K((phase_kernel<MODE, 2, 2, PP, pk_, true>)); break; } } \
                                 L2_LAUNCH_K((phase_kernel<MODE, 2, UU, PP, false, true>)); break; } } \
                               if constexpr (pk_) { if (a.wp) { L2_LAUNCH_K((phase_kernel<MODE, 2, 2, PP, pk_>)); break; } } \
                               if (a.wp) return hipErrorInvalidValue; \
                               L2_LAUNCH_K((phase_kernel<MODE, 2, UU, PP>)); } while (0)
#define L2_LAUNCH_U(UU) do { if (g.pre == 1) L2_LAUNCH(UU, 1); else if (g.pre == 4) L2_LAUNCH(UU, 4); else L2_LAUNCH(UU, 12); } while (0)
  if (g.U == 3) { if constexpr (MODE == MODE_CLS) { if (g.pre == 1) L2_LAUNCH(3, 1); else L2_LAUNCH(3, 4); } }
  else if (g.U == 2) L2_LAUNCH_U(2);
  else if (g.pre == 1) L2_LAUNCH(4, 1);      // (a row of at most 256 float4 never needs more than four per thread ...
  else if constexpr (MODE != MODE_QKV) L2_LAUNCH(4, 4);      //  ... and q / k / v of such a width always run on four waves -- pick_geo_pure -- : one round)
  else return hipErrorInvalidValue;
#undef L2_LAUNCH_U
#undef L2_LAUNCH
#undef L2_LAUNCH_K
  return hipGetLastError();
}

static bool attn_vec(const l2_ctx* c) { return (c->hs % 4 == 0) && (c->d_loc % 4 == 0) && (c->kvd_loc % 4 == 0) && c->hs <= 256; }

static void fill_attn_args(const l2_ctx* c, int l, AttnArgs& a) {
  const size_t loff = (size_t)l * c->S * c->kvd_loc;
  memset(&a, 0, sizeof(a));
  a.q = c->q; a.kc = c->kc + loff; a.vc = c->vc + loff; a.att = c->opt_keep_state ? c->att : nullptr; a.xb = c->xb;
  a.tokpos = c->tokpos; a.part = c->attn_part; a.counter = c->attn_counter;
  a.dim = c->d_loc; a.head_size = c->hs; a.seq_len = c->S; a.n_heads = c->H_loc; a.nsplit = c->cur_splits;
  a.kv_dim = c->kvd_loc; a.kv_mul = c->H / c->KVH;
  a.exact = c->opt_exact;
  a.inv_sqrt_hs = 1.0 / sqrt((double)c->hs);
#ifdef L2_STAMPS
  a.dbg = c->dbg + 64 * 108;   // attention stamps live behind the phase-kernel slots (last launch wins)
#endif
}

// Lanes per cache row: head_size / 4 rounded up to a power of two (attention.hip.h); waves per workgroup: 8 for heads of
// 65 .. 128 floats (a round is then 256 rows), else 4 -- heads wider than 128 (a whole wave per row) keep the 4-wave form:
// their 8-wave instance needs more than 256 registers and spilled.
static int attn_lr(int hs) { int l = 4; while (l * 4 < hs) l <<= 1; return l; }
static int attn_nw(const l2_ctx* c) { return (c->hs > 64 && c->hs <= 128) ? 8 : 4; }

// One launch of the tile kernel; ny = splits (decode) or queries of the chunk (prefill, pos0 >= 0).
static hipError_t launch_attn_tile(const l2_ctx* c, const AttnArgs& a, int ny, int pos0, hipStream_t st) {
  const int lr = attn_lr(c->hs), nw = attn_nw(c);
  const size_t lds = attn_tile_lds(c->S, pos0 >= 0 ? 1 : a.nsplit, nw, nw == 8 ? 8 : 16);
  const dim3 grid(c->H_loc, ny), block(64 * nw);
  // 4 waves x 16 tiles (one wave per SIMD) for heads up to 64 floats and beyond 128, 8 waves x 8 tiles (two per SIMD) for 65 .. 128
#define L2_AT(LR, NW, NT) do { if (pos0 >= 0) { hipError_t e_ = lds_opt_in(&pf_attn_tile_kernel<LR, NW, NT>, lds); if (e_ != hipSuccess) return e_; \
                                            hipLaunchKernelGGL((pf_attn_tile_kernel<LR, NW, NT>), grid, block, lds, st, a, pos0); } \
                           else { hipError_t e_ = lds_opt_in(&attn_tile_kernel<LR, NW, NT>, lds); if (e_ != hipSuccess) return e_; \
                                  l2_launch(c, attn_tile_kernel<LR, NW, NT>, grid, block, lds, st, a); } } while (0)
  (void)nw;      // (one form per row width: the L2_ATTN_NW override and its four extra instances per kernel went in round 5)
  switch (lr) {
    case 4: L2_AT(4, 4, 16); break;
    case 8: L2_AT(8, 4, 16); break;
    case 16: L2_AT(16, 4, 16); break;
    case 32: L2_AT(32, 8, 8); break;
    default: L2_AT(64, 4, 16); break;
  }
#undef L2_AT
  return hipGetLastError();
}

// ---- the fused QKV + attention launch (attention.hip.h: qkv_attn_small_kernel) --------------------------------------------
// Taken when the QKV phase is the latency form, heads are 33 .. 64 floats wide (16 lanes per cache row), the input vector is
// 257 .. 1024 floats (stories15M, stories110M), one cache head per query head, one GPU, and the reference's own value accumulate
// is not asked for.  L2_FUSE_QKV_ATTN=0 (development switch) keeps the two launches.
static int fused_attn_blocks(const l2_ctx* c, int nsplit) { return c->H * nsplit; }
static bool fused_shape_ok(const l2_ctx* c) {      // whatever the position
  if (!c->opt_fuse || c->tp_path || c->KVH != c->H || !attn_vec(c) || !c->gran) return false;
  if (!use_small(c, MODE_QKV, c->d + 2 * c->kvd, c->d) || attn_lr(c->hs) != 16) return false;
  const int xv = (c->d / 4 + 63) / 64;
  return xv >= 2 && xv <= 4;
}
// ... and for the step being enqueued: one workgroup per head only (with 8 splits per head the fused form loses to two launches:
// 96 of the CUs are then attention workgroups that wait while the rest do the GEMV)
static bool fused_qkv_attn_ok(const l2_ctx* c) {
  if (!c->cur_fused || !fused_shape_ok(c) || c->opt_exact) return false;
  const int ns = c->cur_splits;
  return fused_attn_blocks(c, ns) * 2 <= c->n_cus && (size_t)attn_tile_lds(c->S, ns, 8, 8) <= 160 * 1024;
}

template <int XV>
static hipError_t launch_qkv_attn_xv(const l2_ctx* c, const PhaseArgs& qa, const AttnArgs& at, hipStream_t st) {
  const int nattn = fused_attn_blocks(c, at.nsplit);
  int nq = c->n_cus - nattn;
  const int groups = (qa.rows + 1) / 2;
  if (nq > groups) nq = groups;
  const size_t lds_q = (size_t)XV * 64 * 16, lds_a = attn_tile_lds(c->S, at.nsplit, 8, 8);      // (the four-wave form needs less)
  const size_t lds = lds_q > lds_a ? lds_q : lds_a;
  hipError_t e = lds_opt_in(&qkv_attn_small_kernel<XV, 16, 8>, lds);
  if (e != hipSuccess) return e;
  l2_launch(c, qkv_attn_small_kernel<XV, 16, 8>, dim3(nq + nattn), dim3(512), lds, st, qa, at, nattn);
  return hipGetLastError();
}

static hipError_t launch_qkv_attn(const l2_ctx* c, const PhaseArgs& qa_in, int l, hipStream_t st) {
  PhaseArgs qa = qa_in;
  qa.gran = c->gran; qa.gran_ep = c->gran_ep;
  qa.gran_hmagic = (unsigned)(((1u << 20) + (unsigned)c->hs - 1u) / (unsigned)c->hs);
#ifdef L2_STAMPS
  qa.dbg = c->dbg + (size_t)(g_stamp_slot++ % 64) * 108;
#endif
  AttnArgs at;
  fill_attn_args(c, l, at);
  at.gran = c->gran; at.gran_ep = c->gran_ep; at.herr = c->h_herr_dev; at.wait_ticks = 200000000ull;      // 2 s
  at.fused_four_waves = c->opt_fuse_four_waves;
  switch ((c->d / 4 + 63) / 64) {
    case 2: return launch_qkv_attn_xv<2>(c, qa, at, st);
    case 3: return launch_qkv_attn_xv<3>(c, qa, at, st);
    default: return launch_qkv_attn_xv<4>(c, qa, at, st);
  }
}

static hipError_t launch_attn(const l2_ctx* c, int l, hipStream_t st) {   // attention (llama2.ts:244-267)
  AttnArgs a;
  fill_attn_args(c, l, a);
  if (!attn_vec(c)) {
    const size_t lds = (size_t)((c->S + 3) & ~3) * 4 + 64;
    hipError_t e = lds_opt_in(&attn_scalar_kernel, lds);
    if (e != hipSuccess) return e;
    l2_launch(c, attn_scalar_kernel, dim3(c->H_loc, 1), dim3(256), lds, st, a, 0);
    return hipGetLastError();
  }
  if (c->opt_exact) a.nsplit = 1;
  return launch_attn_tile(c, a, a.nsplit, -1, st);
}

// ---- the fused attention + wo launch of a tensor-parallel rank (attention.hip.h: attn_wo_kernel) ------------------------------
// Taken when the rank's wo shard is the latency form with an input of 512 / 1024 / 2048 floats (Llama-2-7B at 8 / 4 / 2 ranks), heads
// are 128 wide (the eight-wave tile kernel), the rows leave through the peer-to-peer push (the combine launch behind it advances
// the launch counter) and the reference's own value accumulate is not asked for.  L2_TP_ATTN_WO=0 keeps the two launches.
static bool attn_wo_ok(const l2_ctx* c) {
  if (!c->opt_awo || !c->tp_path || !p2p_pushing(c) || !c->awo_gran || c->opt_exact || !attn_vec(c)) return false;
  if (c->hs != 128 || attn_nw(c) != 8 || !use_small(c, MODE_WO, c->d, c->d_loc) || c->d <= c->n_cus * 7) return false;
  const int xv = (c->d_loc / 4 + 63) / 64;
  if (!(xv == 2 || xv == 4 || xv == 8) || c->d_loc % 256) return false;
  return c->H_loc * c->cur_splits * 2 <= c->n_cus && (size_t)attn_tile_lds(c->S, c->cur_splits, 8, 8) <= 160 * 1024;
}

static hipError_t launch_attn_wo(const l2_ctx* c, int l, const PhaseArgs& wo_in, hipStream_t st) {
  AttnArgs at;
  fill_attn_args(c, l, at);
  at.gout = c->awo_gran; at.gout_ep = c->awo_ep;
  PhaseArgs wo = wo_in;
  wo.gran = c->awo_gran; wo.gran_ep = c->awo_ep; wo.gin_herr = c->h_herr_dev;
  const int nattn = c->H_loc * at.nsplit, xv = (c->d_loc / 4 + 63) / 64;
  const int groups = (wo.rows + 1) / 2;
  int nwo = c->n_cus - nattn;                 // one workgroup per CU, the attention workgroups' CUs left to them
  if (nwo < c->n_cus / 2) nwo = c->n_cus / 2;
  if (nwo > groups) nwo = groups;
  const size_t lds_a = attn_tile_lds(c->S, at.nsplit, 8, 8), lds_w = (size_t)xv * 64 * 16, lds = lds_a > lds_w ? lds_a : lds_w;
  const dim3 grid(nattn + nwo), block(512);
#define L2_AWO(XVV) do { hipError_t e_ = lds_opt_in(&attn_wo_kernel<XVV, 2, 32, 8, 8>, lds); if (e_ != hipSuccess) return e_; \
                         l2_launch(c, attn_wo_kernel<XVV, 2, 32, 8, 8>, grid, block, lds, st, at, wo, nattn); } while (0)
  if (xv == 2) L2_AWO(2); else if (xv == 4) L2_AWO(4); else L2_AWO(8);
#undef L2_AWO
  return hipGetLastError();
}

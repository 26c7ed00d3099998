// sampler_chain.hip.h -- the whole-chip form: tiled exact running sums (runs + checked chain, exact_sum.h), exps / normalise kernels, the pick
// Part of sampler.hip (included there inside namespace l2s, in order); not a stand-alone header.
#pragma once

// ------------------------------------------------------------------------------------------------
// The whole-chip form.  A vector is cut into tiles of TILE = 1024 consecutive elements, one 256-thread workgroup per
// tile, IT = 4 consecutive elements per thread.
constexpr int TN = 256, IT = 4, TILE = TN * IT, NWV = TN / 64;
constexpr int RUN_CAP = 1024;                  // runs the chain stages in LDS (a 32 000-element softmax has ~50)
static_assert(MAX_VOCAB <= TN * TILE, "one chain thread per tile");

using xs::Comp;
using xs::Run;
using xs::E_NONE;

// Scan element: composite of the open run (q0, d), its grid E, "a run starts inside the range" (flag) and the number of
// serial elements in the range (cnt), the last four packed into one word so that a lane exchange moves three dwords:
//   meta = d + 1 (bits 0-1) | flag (bit 2) | cnt (bits 4-15) | E (bits 16-31)
struct Seg { unsigned long long q0; int meta; };
constexpr int SEG_CNT = 0xfff0, SEG_FLAG = 4, SEG_ID = 1 | (int)((unsigned)E_NONE << 16);   // identity: d = 0, no grid

__device__ __forceinline__ Seg seg_identity() { Seg s; s.q0 = 0; s.meta = SEG_ID; return s; }
__device__ __forceinline__ Seg seg_make(const Comp& c, int E, bool flag, bool serial) {
  Seg s; s.q0 = c.q0; s.meta = (c.d + 1) | (flag ? SEG_FLAG : 0) | (serial ? 16 : 0) | (int)((unsigned)E << 16);
  return s;
}
__device__ __forceinline__ int seg_d(const Seg& s) { return (s.meta & 3) - 1; }
__device__ __forceinline__ int seg_E(const Seg& s) { return s.meta >> 16; }
__device__ __forceinline__ int seg_cnt(const Seg& s) { return (s.meta & SEG_CNT) >> 4; }
__device__ __forceinline__ Seg seg_op(const Seg& a, const Seg& b) {            // a first, then b
  const int cnt = (a.meta & SEG_CNT) + (b.meta & SEG_CNT);
  Seg r;
  if (b.meta & SEG_FLAG) { r.q0 = b.q0; r.meta = (b.meta & ~SEG_CNT) | cnt; return r; }
  Comp ca; ca.q0 = a.q0; ca.d = seg_d(a);
  Comp cb; cb.q0 = b.q0; cb.d = seg_d(b);
  const Comp c = xs::compose(ca, cb);
  const int e = (seg_E(b) != E_NONE) ? (b.meta & (int)0xffff0000) : (a.meta & (int)0xffff0000);
  r.q0 = c.q0; r.meta = e | cnt | (a.meta & SEG_FLAG) | (c.d + 1);
  return r;
}

// Lane exchanges of the wave scans: data-parallel primitives, no LDS.  `old` is what a lane without a source keeps.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, false); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {                          // identity 0.0
  return __hiloint2double(dpp_i32<CTRL, ROW_MASK>(0, __double2hiint(v)), dpp_i32<CTRL, ROW_MASK>(0, __double2loint(v)));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ Seg dpp_seg(const Seg& v) {
  Seg r;
  r.q0 = ((unsigned long long)(unsigned)dpp_i32<CTRL, ROW_MASK>(0, (int)(v.q0 >> 32)) << 32) | (unsigned)dpp_i32<CTRL, ROW_MASK>(0, (int)v.q0);
  r.meta = dpp_i32<CTRL, ROW_MASK>(SEG_ID, v.meta);
  return r;
}
// inclusive scans over the 64 lanes: row_shr 1, 2, 4, 8, then row 0 -> 1 and 2 -> 3 (row_bcast:15), rows 0-1 -> 2-3 (row_bcast:31)
__device__ __forceinline__ double wave_scan_f64(double v) {
  v += dpp_f64<0x111, 0xf>(v); v += dpp_f64<0x112, 0xf>(v); v += dpp_f64<0x114, 0xf>(v); v += dpp_f64<0x118, 0xf>(v);
  v += dpp_f64<0x142, 0xa>(v); v += dpp_f64<0x143, 0xc>(v);
  return v;
}
__device__ __forceinline__ Seg wave_scan_seg(Seg v) {
  v = seg_op(dpp_seg<0x111, 0xf>(v), v); v = seg_op(dpp_seg<0x112, 0xf>(v), v); v = seg_op(dpp_seg<0x114, 0xf>(v), v);
  v = seg_op(dpp_seg<0x118, 0xf>(v), v); v = seg_op(dpp_seg<0x142, 0xa>(v), v); v = seg_op(dpp_seg<0x143, 0xc>(v), v);
  return v;
}

struct TileShared {
  double wsum[NWV];
  int wser[NWV];
  Seg wagg[NWV];
};

__device__ __forceinline__ void load_tile(const float* x, int V, int tile, float (&v)[IT]) {
  const int i0 = tile * TILE + threadIdx.x * IT;
  if (i0 + IT <= V && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    const float4 q = *reinterpret_cast<const float4*>(x + i0);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
  } else {
#pragma unroll
    for (int k = 0; k < IT; ++k) v[k] = (i0 + k < V) ? x[i0 + k] : 0.0f;
  }
}

// Approximate sum of one tile (any fixed order); every thread gets the same value.
__device__ __forceinline__ double tile_total(const float (&v)[IT], double* wsum) {
  double t = ((double)v[0] + (double)v[1]) + ((double)v[2] + (double)v[3]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = t;
  __syncthreads();
  return (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// Approximate sum of the tiles in front of `tile`; the same instruction sequence in every kernel that needs it, and a
// butterfly of commutative adds, so every lane of every wave holds the same bits.
__device__ __forceinline__ double tile_base(const double* part, int tile) {
  double s = 0.0;
  for (int j = threadIdx.x & 63; j < tile; j += 64) s += part[j];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  return s;
}

struct Elems {
  Seg inc[IT];        // inclusive segmented scan at the element: composite of the regular elements of its run up to it, .cnt serial elements up to it
  bool serial[IT];
};

// Classify the tile's elements (exact_sum.h) and scan their grid composites run by run.
__device__ __forceinline__ void tile_scan(const float (&v)[IT], double base, TileShared& sh, Elems& o, int mb = 32) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double a[IT];
  a[0] = (double)v[0];
#pragma unroll
  for (int k = 1; k < IT; ++k) a[k] = a[k - 1] + (double)v[k];
  const double incl = wave_scan_f64(a[IT - 1]);
  __syncthreads();                                         // sh may still be read by a previous call
  if (lane == 63) sh.wsum[wave] = incl;
  __syncthreads();
  double wbase = 0.0;
  for (int w = 0; w < wave; ++w) wbase += sh.wsum[w];
  const double tb = (base + wbase) + dpp_f64<0x138, 0xf>(incl);   // wave_shr:1 -- the lane in front, 0 for lane 0

  bool ser[IT];
  int eE[IT];
  // the common case first: the whole thread sits in one binade, nothing to decide per element
  int Eq;
  const bool quiet = !xs::classify(tb, tb + a[IT - 1], 1.0f, &Eq, mb);
#pragma unroll
  for (int k = 0; k < IT; ++k) {
    if (quiet) { ser[k] = false; eE[k] = (v[k] == 0.0f) ? E_NONE : Eq; }
    else ser[k] = xs::classify(k ? tb + a[k - 1] : tb, tb + a[k], v[k], &eE[k], mb);
  }
  const int last = ser[IT - 1] ? 1 : 0;
  if (lane == 63) sh.wser[wave] = last;
  __syncthreads();
  int prev = dpp_i32<0x138, 0xf>(0, last);
  if (lane == 0) prev = wave ? sh.wser[wave - 1] : 1;      // a run starts with the tile

  Seg l[IT];
#pragma unroll
  for (int k = 0; k < IT; ++k) {
    o.serial[k] = ser[k];
    const Comp c = (ser[k] || eE[k] == E_NONE) ? xs::identity() : xs::on_grid(v[k], eE[k]);
    const Seg e = seg_make(c, ser[k] ? E_NONE : eE[k], k ? ser[k - 1] : prev != 0, ser[k]);
    l[k] = k ? seg_op(l[k - 1], e) : e;
  }
  const Seg agg = wave_scan_seg(l[IT - 1]);
  if (lane == 63) sh.wagg[wave] = agg;
  Seg pre = dpp_seg<0x138, 0xf>(agg);                       // exclusive: identity for lane 0
  __syncthreads();
  Seg wpre = seg_identity();
  for (int w = 0; w < wave; ++w) wpre = seg_op(wpre, sh.wagg[w]);
  pre = seg_op(wpre, pre);
#pragma unroll
  for (int k = 0; k < IT; ++k) o.inc[k] = seg_op(pre, l[k]);
}

__device__ __forceinline__ unsigned order_key(float x) { const unsigned b = __float_as_uint(x); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float order_value(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// max over state.logits[q] / temperature (:482, softmax :183-186)
__global__ void __launch_bounds__(TN) scaled_max_kernel(const float* logits, int V, const double* params, unsigned* mxkey) {
  float v[IT];
  load_tile(logits, V, blockIdx.x, v);
  const double T = params[0];
  const int i0 = blockIdx.x * TILE + threadIdx.x * IT;
  unsigned key = 0;
#pragma unroll
  for (int k = 0; k < IT; ++k) if (i0 + k < V) key = max(key, order_key((float)((double)v[k] / T)));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) key = max(key, (unsigned)__shfl_xor((int)key, off, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(mxkey, key);
}

// probs[i] = (float)exp(x_i - max)  (:187) and the tile sums for the approximate prefix
// (amax != null, temperature > 0: the classifier already folded max(logits) into its argmax keys -- kernels.hip.h
// argmax_key -- and x -> (float)(x / T) is monotone, so the maximum of the scaled logits is the scaled maximum)
__global__ void __launch_bounds__(TN) exp_kernel(const float* logits, int V, const double* params, const unsigned* mxkey, const unsigned long long* amax,
                                                  float* probs, double* part) {
  __shared__ double wsum[NWV];
  float v[IT];
  load_tile(logits, V, blockIdx.x, v);
  const double T = params[0];
  float mx;
  if (amax) {
    unsigned long long k = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const unsigned long long kj = amax[(size_t)j * 16]; k = kj > k ? kj : k; }
    mx = (float)((double)order_value((unsigned)(k >> 32)) / T);
  } else {
    mx = order_value(*mxkey);
  }
  const int i0 = blockIdx.x * TILE + threadIdx.x * IT;
#pragma unroll
  for (int k = 0; k < IT; ++k) {
    const float x = (float)((double)v[k] / T);
    v[k] = (i0 + k < V) ? (float)exp((double)x - (double)mx) : 0.0f;
    if (i0 + k < V) probs[i0 + k] = v[k];
  }
  const double t = tile_total(v, wsum);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

__global__ void __launch_bounds__(TN) tile_sums_kernel(const float* x, int V, double* part) {
  __shared__ double wsum[NWV];
  float v[IT];
  load_tile(x, V, blockIdx.x, v);
  const double t = tile_total(v, wsum);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// One record per run of the tile: recs[tile * (TILE + 1) + r], cnt[tile] of them; and (COMP) per element its composite
// since the start of its run (cq) with {d + 1, serial, grid} packed into cm, which is what the search needs to turn the
// exact sum in front of a run into the exact running sum at any element of it.
__device__ __forceinline__ int pack_meta(const Seg& s, bool serial) { return (s.meta & 3) | (serial ? 4 : 0) | (s.meta & (int)0xffff0000); }

// A run record / a count as three / one write-through store(s) and L1-bypassing loads: what a workgroup of the SAME launch may read once the
// writers have drained their stores and taken a ticket (MI355X_MICROARCH.md, hand-off forms; Run is three 8-byte words).
static_assert(sizeof(Run) == 24, "a run record is three 8-byte words");
__device__ __forceinline__ void run_store_through(Run* dst, const Run& r) {
  unsigned long long w[3];
  memcpy(w, &r, 24);
  unsigned long long* d = reinterpret_cast<unsigned long long*>(dst);
#pragma unroll
  for (int k = 0; k < 3; ++k) __hip_atomic_store(d + k, w[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool THROUGH>
__device__ __forceinline__ Run run_load(const Run* src) {
  if (!THROUGH) return *src;
  unsigned long long w[3];
  const unsigned long long* s = reinterpret_cast<const unsigned long long*>(src);
#pragma unroll
  for (int k = 0; k < 3; ++k) w[k] = __hip_atomic_load(s + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  Run r;
  memcpy(&r, w, 24);
  return r;
}

template <bool COMP, bool THROUGH = false>
__device__ __forceinline__ void emit_runs(const Elems& el, const float (&v)[IT], int V, int tile, Run* recs, int* cnt, unsigned long long* cq, int* cm) {
  Run* out = recs + (size_t)tile * (TILE + 1);
  const int i0 = tile * TILE + threadIdx.x * IT;
#pragma unroll
  for (int k = 0; k < IT; ++k) {
    const Seg& s = el.inc[k];
    if (COMP) { cq[i0 + k] = s.q0; cm[i0 + k] = pack_meta(s, el.serial[k]); }
    const bool tile_end = threadIdx.x == TN - 1 && k == IT - 1;
    if (el.serial[k] || tile_end) {
      Run r; r.q0 = s.q0; r.d = seg_d(s); r.E = seg_E(s); r.x = el.serial[k] ? v[k] : 0.0f; r.end = min(i0 + k, V - 1);
      Run* at = out + (el.serial[k] ? seg_cnt(s) - 1 : seg_cnt(s));
      const int total_runs = seg_cnt(s) + (el.serial[k] ? 0 : 1);
      if (THROUGH) { run_store_through(at, r); if (tile_end) __hip_atomic_store(cnt + tile, total_runs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      else { *at = r; if (tile_end) cnt[tile] = total_runs; }
    }
  }
}

template <bool COMP>
__global__ void __launch_bounds__(TN) runs_kernel(const float* x, int V, const double* part, Run* recs, int* cnt, unsigned long long* cq, int* cm) {
  __shared__ TileShared sh;
  float v[IT];
  load_tile(x, V, blockIdx.x, v);
  Elems el;
  tile_scan(v, tile_base(part, blockIdx.x), sh, el);
  emit_runs<COMP>(el, v, V, blockIdx.x, recs, cnt, cq, cm);
}

struct ChainArgs {
  const float* x;            // the vector being accumulated
  int V, G;
  const double* part;
  const Run* recs;
  const int* cnt;
  int* off;                  // (G + 1) first run of every tile
  double* S;                 // per run: exact sum after it
  int* End;                  // per run: index of its last element
  int* Bad;                  // per run: prediction failed, its elements were added one by one
  const double* params;
  unsigned long long* rng;
  int* tokpos;
  int* tokens_out;
  const unsigned long long* cq;   // per element: composite since the start of its run
  const int* cm;
  const int* ids;            // top-p: token ids in sorted order
  unsigned* mxkey;           // reset for the next token
  unsigned long long* amax;  // or: the classifier's 8 argmax keys (llama2_hip.hip) supplied the maximum; reset those
  double* part_sorted;       // top-p: tile sums the rank merge accumulates, zero between tokens
};

struct ChainShared {
  TileShared tile;
  Run rec[RUN_CAP];
  double S[RUN_CAP];         // per-run state when the runs fit (else ChainArgs' arrays in global memory)
  int End[RUN_CAP];
  int Bad[RUN_CAP];
  int off[TN + 1];
  int wtot[NWV];
  int slot;
  double val;
};

// First index i < limit whose exact running sum satisfies pred (pred is monotone in S); -1 if none.  *at = that sum.
// Per-run state of the chain: LDS when the runs fit there, the arrays in global memory otherwise (and for the diagnostic).
// A compile-time choice, so that no access turns into a flat instruction (those wait on both memory counters).
template <bool IN_LDS>
struct RunState {
  ChainShared& sh;
  const ChainArgs& a;
  __device__ __forceinline__ double& S(int k) const { if (IN_LDS) return sh.S[k]; return a.S[k]; }
  __device__ __forceinline__ int& End(int k) const { if (IN_LDS) return sh.End[k]; return a.End[k]; }
  __device__ __forceinline__ int& Bad(int k) const { if (IN_LDS) return sh.Bad[k]; return a.Bad[k]; }
};

template <bool IN_LDS, class Pred>
__device__ __forceinline__ int find_first(const ChainArgs& a, ChainShared& sh, const RunState<IN_LDS>& rs, int T, Pred pred, int limit, double* at) {
  const int tid = threadIdx.x;
  __syncthreads();
  if (tid == 0) sh.slot = 0x7fffffff;
  __syncthreads();
  int f = 0x7fffffff;
  for (int k = tid; k < T; k += TN) if (pred(rs.S(k))) { f = k; break; }
  if (f != 0x7fffffff) atomicMin(&sh.slot, f);
  __syncthreads();
  const int kr = sh.slot;
  __syncthreads();
  if (kr == 0x7fffffff) return -1;
  const int start = kr ? rs.End(kr - 1) + 1 : 0, end = rs.End(kr);
  const double S0 = kr ? rs.S(kr - 1) : 0.0;
  if (tid == 0) sh.slot = 0x7fffffff;
  __syncthreads();
  if (rs.Bad(kr)) {
    if (tid == 0) {
      double S = S0;
      for (int j = start; j <= end && j < a.V; ++j) { S += (double)a.x[j]; if (pred(S)) { sh.slot = j; sh.val = S; break; } }
    }
  } else {
    const int tile = start / TILE, i0 = tile * TILE + tid * IT;
    unsigned long long q[IT];
    int m[IT];
#pragma unroll
    for (int k = 0; k < IT; ++k) { q[k] = a.cq[i0 + k]; m[k] = a.cm[i0 + k]; }
    const double Send = rs.S(kr);
    double mine_S = 0.0;
    int mine = 0x7fffffff;
#pragma unroll
    for (int k = IT - 1; k >= 0; --k) {
      const int i = i0 + k;
      Comp c; c.q0 = q[k]; c.d = (m[k] & 3) - 1;
      const double Si = (m[k] & 4) ? Send : xs::value_at(S0, c, m[k] >> 16);
      if (i >= start && i <= end && pred(Si)) { mine = i; mine_S = Si; }
    }
    if (mine != 0x7fffffff) atomicMin(&sh.slot, mine);
    __syncthreads();
    if (mine != 0x7fffffff && mine == sh.slot) sh.val = mine_S;
  }
  __syncthreads();
  const int hit = sh.slot;
  *at = sh.val;
  return (hit != 0x7fffffff && hit < limit) ? hit : -1;
}

enum { CHAIN_SAMPLE = 1, CHAIN_TOPP = 2, CHAIN_DEBUG = 3 };

// Walk the runs in order with the exact fp64 state; leaves the total in sh.val (read it after a barrier).
template <bool IN_LDS, bool THROUGH = false>
__device__ __forceinline__ void chain_walk(const ChainArgs& a, ChainShared& sh, int T) {
  const int tid = threadIdx.x;
  const RunState<IN_LDS> rs{sh, a};
  // Fast walk (runs staged in LDS), 64 runs at a time, lane l of wave 0 holding run l.  Runs that merely end with their tile
  // sit on the grid of the run behind them, so a segmented scan first composes every stretch of runs up to the next serial
  // element (~3 stretches per binade crossing instead of one step per tile); then the state S (uniform) takes one step per
  // stretch: "add the increment picked by the parity of S to the BIT PATTERN of S" (that many grid steps inside the binade),
  // then the ordinary add of the serial element -- ~5 dependent instructions.  Every lane then derives the exact sum after
  // its own run from the sum in front of its stretch and checks the prediction it rested on; if any check fails (never
  // observed) the generic loop below redoes the walk run by run with the element-wise fallback.
  bool fast_ok = false;
  if (IN_LDS && tid < 64) {
    double S = 0.0;
    bool allok = true;
    for (int c0 = 0; c0 < T; c0 += 64) {
      const int k = c0 + tid;
      Run r; r.q0 = 0; r.d = 0; r.E = E_NONE; r.x = 0.0f; r.end = 0;
      if (k < T) r = sh.rec[k];
      const bool ender = k < T && (r.x != 0.0f || k == T - 1 || tid == 63);
      int prev_ender = dpp_i32<0x138, 0xf>(1, ender ? 1 : 0);            // wave_shr:1; a stretch starts at lane 0
      Comp rc; rc.q0 = r.q0; rc.d = r.d;
      const bool own = (r.q0 | (unsigned long long)(unsigned)r.d) != 0;
      const Seg me = seg_make(rc, own ? r.E : E_NONE, prev_ender != 0, false);
      const Seg inc = wave_scan_seg(me);
      const Seg before = dpp_seg<0x138, 0xf>(inc);
      // one grid per stretch: a run on another grid than the runs composed in front of it would be a wrong prediction
      allok = allok && r.q0 < xs::TWO53 && (!own || prev_ender || seg_E(before) == E_NONE || seg_E(before) == r.E);
      const unsigned long long inc0 = inc.q0, inc1 = inc.q0 + (unsigned long long)(long long)seg_d(inc);
      const bool has = (inc.q0 | (unsigned long long)(unsigned)seg_d(inc)) != 0;
      const int eb = has ? seg_E(inc) + 1023 : -1;
      const double x = (double)r.x;
      double Sfront = 0.0;
      int prev = -1;
      for (unsigned long long todo = __ballot(ender); todo; todo &= todo - 1) {
        const int j = __builtin_ctzll(todo);
        const unsigned long long i0 = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(inc0 >> 32), j) << 32) | (unsigned)__builtin_amdgcn_readlane((int)inc0, j);
        const unsigned long long i1 = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(inc1 >> 32), j) << 32) | (unsigned)__builtin_amdgcn_readlane((int)inc1, j);
        const double xj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), j), __builtin_amdgcn_readlane(__double2loint(x), j));
        asm volatile("" : "+v"(S));                           // keep the state in vector registers: no scalar round trip per step
        Sfront = (tid > prev && tid <= j) ? S : Sfront;       // the lanes of this stretch start from here
        const unsigned long long sb = (unsigned long long)__double_as_longlong(S);
        S = __longlong_as_double((long long)(sb + ((sb & 1) ? i1 : i0))) + xj;
        prev = j;
      }
      const unsigned long long fb = (unsigned long long)__double_as_longlong(Sfront);
      const unsigned long long fb2 = fb + ((fb & 1) ? inc1 : inc0);
      const double Sk = __longlong_as_double((long long)fb2) + x;   // x = 0 unless the run ends with a serial element
      allok = allok && (k >= T || eb < 0 || ((int)(fb >> 52) == eb && (int)(fb2 >> 52) == eb));
      if (k < T) { rs.S(k) = Sk; rs.End(k) = r.end; rs.Bad(k) = 0; }
    }
    fast_ok = __all(allok);
    if (tid == 0 && fast_ok) {
      sh.val = S;
    }
  }
  if (tid == 0 && !fast_ok) {
    double S = 0.0;
    int start = 0;
    auto step = [&](int k, const Run& rec) {
      bool ok;
      double S2 = xs::chain_step(S, rec, &ok);
      if (!ok) {
        S2 = S;
        for (int j = start; j <= rec.end && j < a.V; ++j) S2 += (double)a.x[j];
      }
      rs.S(k) = S2; rs.End(k) = rec.end; rs.Bad(k) = ok ? 0 : 1;
      S = S2; start = rec.end + 1;
    };
    if (IN_LDS) {
      for (int k = 0; k < T; ++k) step(k, sh.rec[k]);
    } else {
      int k = 0;
      for (int t = 0; t < a.G; ++t)
        for (int r = 0, n = sh.off[t + 1] - sh.off[t]; r < n; ++r, ++k) step(k, run_load<THROUGH>(a.recs + (size_t)t * (TILE + 1) + r));
    }
    sh.val = S;
  }
}

// Order the tiles' runs (first run of every tile in sh.off, records staged in LDS when they fit) and walk them.
// Returns the number of runs; *in_lds says where the per-run state went.  Every thread of the workgroup calls it.
template <bool THROUGH = false>
__device__ __forceinline__ int chain_total(const ChainArgs& a, ChainShared& sh, bool* in_lds) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = tid < a.G ? (THROUGH ? __hip_atomic_load(a.cnt + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : a.cnt[tid]) : 0;
  Run r0, r1;                                              // nearly every tile has one or two runs: fetched together with the count
  if (tid < a.G) { r0 = run_load<THROUGH>(a.recs + (size_t)tid * (TILE + 1)); r1 = run_load<THROUGH>(a.recs + (size_t)tid * (TILE + 1) + 1); }
  int incl = c;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
  if (lane == 63) sh.wtot[wave] = incl;
  __syncthreads();
  int wb = 0;
  for (int w = 0; w < wave; ++w) wb += sh.wtot[w];
  const int first = wb + incl - c;
  sh.off[tid] = first;
  if (tid == TN - 1) sh.off[TN] = first + c;
  __syncthreads();
  const int T = sh.off[TN];
  *in_lds = T <= RUN_CAP;
  if (*in_lds) {
    if (tid < a.G) {
      if (c > 0) sh.rec[first] = r0;
      if (c > 1) sh.rec[first + 1] = r1;
      for (int r = 2; r < c; ++r) sh.rec[first + r] = run_load<THROUGH>(a.recs + (size_t)tid * (TILE + 1) + r);
    }
    __syncthreads();
    chain_walk<true>(a, sh, T);
  } else {
    chain_walk<false, THROUGH>(a, sh, T);
  }
  __syncthreads();
  return T;
}

// What sample() / sample_topp() do with the running sums (llama2.ts:368-394).
template <int MODE, bool IN_LDS>
__device__ __forceinline__ void chain_pick(const ChainArgs& a, ChainShared& sh, int T) {
  const int tid = threadIdx.x;
  const RunState<IN_LDS> rs{sh, a};
  const double total = sh.val;
  __syncthreads();
  double at = 0.0;
  if (MODE == CHAIN_SAMPLE) {
    if (tid == 0) sh.val = (double)random_f32(a.rng) * total;                    // randValue = random_f32() * sum (:370)
    __syncthreads();
    const double r = sh.val;
    const int hit = find_first(a, sh, rs, T, [r](double S) { return r < S; }, a.V, &at);   // :373
    if (tid == 0) { advance(a.tokpos, a.tokens_out, hit < 0 ? 0 : hit); *a.mxkey = 0; }   // fall-through returns 0 (:375)
    if (a.amax && tid < 8) a.amax[(size_t)tid * 16] = 0ull;
  } else {
    const double topp = a.params[1];
    const int cross = find_first(a, sh, rs, T, [topp](double S) { return S > topp; }, a.V, &at);   // :385
    const int last = cross < 0 ? 0 : cross;                                      // never crossed: lastIdx stays 0 (:383)
    __syncthreads();
    if (tid == 0) sh.val = (double)random_f32(a.rng) * (cross < 0 ? total : at);  // cumProb as the loop left it (:388)
    __syncthreads();
    const double r = sh.val;
    const int hit = find_first(a, sh, rs, T, [r](double S) { return r < S; }, last, &at);   // i < lastIdx only (:390)
    if (tid == 0) { advance(a.tokpos, a.tokens_out, hit < 0 ? 0 : a.ids[hit]); *a.mxkey = 0; }
    if (a.amax && tid < 8) a.amax[(size_t)tid * 16] = 0ull;
    if (a.part_sorted && tid < a.G) a.part_sorted[tid] = 0.0;
  }
}


// One workgroup: every run's state for the diagnostic (CHAIN_DEBUG), or the sampled token.
template <int MODE>
__global__ void __launch_bounds__(TN) chain_kernel(ChainArgs a) {
  __shared__ ChainShared sh;
  bool in_lds;
  const int T = chain_total(a, sh, &in_lds);
  if (MODE == CHAIN_DEBUG) {                                   // everything prefix_kernel needs, in global memory
    const int tid = threadIdx.x;
    if (tid < a.G) a.off[tid] = sh.off[tid];
    if (tid == 0) a.off[a.G] = T;
    if (in_lds) for (int k = tid; k < T; k += TN) { a.S[k] = sh.S[k]; a.End[k] = sh.End[k]; a.Bad[k] = sh.Bad[k]; }
    return;
  }
  if (in_lds) chain_pick<MODE, true>(a, sh, T); else chain_pick<MODE, false>(a, sh, T);
}

// Runs of the exps and, by the last workgroup to finish, their exact total (the softmax denominator, :189): the records go out as
// write-through stores, every wave drains them, one lane takes a ticket; the workgroup that gets the last one walks the chain.
__global__ void __launch_bounds__(TN) runs_total_kernel(ChainArgs a, Run* recs, int* cnt, unsigned* ticket, double* total) {
  __shared__ ChainShared sh;
  float v[IT];
  load_tile(a.x, a.V, blockIdx.x, v);
  Elems el;
  tile_scan(v, tile_base(a.part, blockIdx.x), sh.tile, el);
  emit_runs<false, true>(el, v, a.V, blockIdx.x, recs, cnt, nullptr, nullptr);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) sh.slot = (int)__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (sh.slot != a.G - 1) return;
  __syncthreads();
  bool in_lds;
  chain_total<true>(a, sh, &in_lds);
  if (threadIdx.x == 0) { *total = sh.val; __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}

// The exact softmax denominator, recomputed by every workgroup of the kernel that needs it next (a walk over ~50 runs is
// cheaper than a launch boundary): probabilities = exps / total (:192), then straight into their own tile scan.  The
// approximate prefix in front of a tile is the exps' prefix over the same total -- within 2^-24 relative of the sum of the
// rounded quotients, hence the 20-bit margin (exact_sum.h).  Writes the probabilities and the runs of THEIR running sums.
__global__ void __launch_bounds__(TN) normalise_runs_kernel(ChainArgs a, float* probs_n, Run* recs_n, int* cnt_n, unsigned long long* cq, int* cm) {
  __shared__ ChainShared sh;
  bool in_lds;
  chain_total(a, sh, &in_lds);
  const double total = sh.val;
  float v[IT];
  load_tile(a.x, a.V, blockIdx.x, v);
  const int i0 = blockIdx.x * TILE + threadIdx.x * IT;
#pragma unroll
  for (int k = 0; k < IT; ++k) {
    v[k] = (i0 + k < a.V) ? (float)((double)v[k] / total) : 0.0f;
    if (i0 + k < a.V) probs_n[i0 + k] = v[k];
  }
  Elems el;
  tile_scan(v, tile_base(a.part, blockIdx.x) / total, sh.tile, el, 20);
  emit_runs<true>(el, v, a.V, blockIdx.x, recs_n, cnt_n, cq, cm);
}

// Diagnostic (l2_debug_running_sums): every running sum, from the chain's per-run state.
__global__ void __launch_bounds__(TN) prefix_kernel(const float* x, int V, const double* part, const int* off, const double* S, const int* Bad,
                                                     const int* End, double* prefix) {
  __shared__ TileShared sh;
  float v[IT];
  load_tile(x, V, blockIdx.x, v);
  Elems el;
  tile_scan(v, tile_base(part, blockIdx.x), sh, el);
  const int i0 = blockIdx.x * TILE + threadIdx.x * IT, k0 = off[blockIdx.x];
#pragma unroll
  for (int k = 0; k < IT; ++k) {
    const int i = i0 + k;
    if (i >= V) continue;
    const int run = k0 + seg_cnt(el.inc[k]) - (el.serial[k] ? 1 : 0);
    if (Bad[run]) continue;
    Comp c; c.q0 = el.inc[k].q0; c.d = seg_d(el.inc[k]);
    prefix[i] = el.serial[k] ? S[run] : xs::value_at(run ? S[run - 1] : 0.0, c, seg_E(el.inc[k]));
  }
  if (threadIdx.x == 0) {
    for (int run = k0; run < off[blockIdx.x + 1]; ++run) {
      if (!Bad[run]) continue;
      double acc = run ? S[run - 1] : 0.0;
      for (int j = run ? End[run - 1] + 1 : 0; j <= End[run] && j < V; ++j) { acc += (double)x[j]; prefix[j] = acc; }
    }
  }
}

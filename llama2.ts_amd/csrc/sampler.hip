// Device-side sampling for libllama2hip.so (gfx950): temperature scaling, softmax, sample / sample_topp and the
// xorshift* RNG of wizzard0/llama2.ts (llama2.ts:348-394, 476-493), restated so that the SAME token comes out.
//
// What makes that non-trivial: the reference's sums are sequential fp64 accumulations of fp32 values
// (softmax :189, sample :369/:373, sample_topp :385/:391) and the chosen index depends on comparing a random
// threshold against those running sums, so a tree sum (different in the last bits) can flip a token.
//
// Default form (whole chip; 4 launches per token for sample, 6 for top-p; +31 / +59 us per token at stories110M):
//   exp + tile sums (the maximum comes from the classifier's argmax keys) -> runs of the exps
//   sample:  [exact total -> probabilities -> their runs] -> chain: exact running sums, threshold, search, advance
//   top-p:   [exact total -> probabilities -> sorted tiles] -> rank merge (+ tile sums) -> runs -> chain
// "runs" / "chain" are exact_sum.h: every 1024-element tile turns its elements into integer increments on the grid its
// approximate prefix predicts, one wave walks the ~50 runs of a 32 000-element vector with exact fp64 state (stretches of
// runs on one grid composed by a scan first), every prediction is checked, and the searched index is evaluated inside
// the one run that contains it.  The bracketed steps share a launch: each of their workgroups repeats the walk for the
// total instead of waiting for a launch that would hand it over.  Bit-identical to the serial loop by construction
// (tests/test_exact_sum_cpu.py on the host, l2_debug_running_sums on the GPU).
// The descending stable sort of sample_topp (Array.prototype.sort is stable in V8 >= 7.0) is a bitonic sort of
// (probability, id) keys per tile followed by one rank-by-binary-search merge of the 32 sorted tiles out of LDS.
//
// L2_SAMPLER_SERIAL=1 keeps the straightforward form for A/B: ONE lane adds in index order (~10 cycles per element,
// ~130 us per pass over 32 000 values) inside a single 1024-thread workgroup, rocPRIM's radix sort for top-p.
#include "sampler.h"
#include "exact_sum.h"

#include <hipcub/hipcub.hpp>
#include <stdlib.h>

namespace l2s {

#pragma clang fp contract(off)

constexpr int NT = 1024;     // threads of the one workgroup of the serial form
constexpr int CH = 4096;     // values staged in LDS per chunk, widened to fp64 (32 KB)
constexpr int SEG = 512;     // spacing of recorded running sums
constexpr int MAXSEG = MAX_VOCAB / SEG;

__device__ __forceinline__ float random_f32(unsigned long long* rng) {   // llama2.ts:349-360
  unsigned long long s = *rng;
  s ^= s >> 12;
  s ^= s << 25;
  s ^= s >> 27;
  *rng = s;
  const unsigned u = (unsigned)((s * 0x2545F4914F6CDD1Dull) >> 32);
  return (float)(((double)u / 256.0) / 16777216.0);      // one rounding at the Float32Array store (:358)
}

__device__ __forceinline__ void advance(int* tokpos, int* tokens_out, int next) {
  const int step = tokpos[2];
  tokens_out[step] = next;
  tokpos[0] = next; tokpos[1] = tokpos[1] + 1; tokpos[2] = step + 1;
}

// Running sum of buf[0..n) added to `acc` in index order by the calling lane; records the value after every
// element whose global index + 1 is a multiple of SEG (g0 is a multiple of SEG).  Stops at the first element at
// which acc > limit (limit = +inf: never) and returns that local index, else -1.  The values were widened to fp64
// by the threads that staged them (exact), so the serial lane issues nothing but the dependent v_add_f64 chain;
// blocks of 16 with the next block's LDS reads in flight.  The sums are monotone (values >= 0), so testing the
// limit at the end of a block finds the block of the first crossing, which is replayed from its starting value.
typedef double d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int seq_accumulate(const double* buf, int n, int g0, double& acc, double* bound, double limit) {
  double a = acc;
  int i = 0;
  const d2* b2 = reinterpret_cast<const d2*>(buf);
  d2 v[8], w[8];
  if (n >= 16) {
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = b2[k];
  }
  for (; i + 16 <= n; i += 16) {
    const int nx = (i + 32 <= n) ? (i + 16) / 2 : i / 2;     // next block (or this one again: never read past n)
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = b2[nx + k];
    const double start = a;
#pragma unroll
    for (int k = 0; k < 8; ++k) { a += v[k].x; a += v[k].y; }
    if (a > limit) {
      a = start;
      for (int j = i;; ++j) { a += buf[j]; if (a > limit) { acc = a; return j; } }
    }
    if (bound && ((g0 + i + 16) & (SEG - 1)) == 0) bound[(g0 + i) / SEG] = a;
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = w[k];
  }
  for (; i < n; ++i) {
    a += buf[i];
    if (bound && ((g0 + i + 1) & (SEG - 1)) == 0) bound[(g0 + i) / SEG] = a;
    if (a > limit) { acc = a; return i; }
  }
  acc = a;
  return -1;
}

// logits -> probabilities exactly as llama2.ts:481-485 + softmax :181-194 does it in place on state.logits.
__device__ __forceinline__ void softmax_in_place(const float* logits, int V, double T, float* probs, int* idx, double* buf, float* redf, double* shd) {
  const int tid = threadIdx.x;
  float mx = -INFINITY;
  for (int i = tid; i < V; i += NT) {
    const float x = (float)((double)logits[i] / T);        // state.logits[q] /= temperature (:482)
    probs[i] = x;
    mx = fmaxf(mx, x);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  if ((tid & 63) == 0) redf[tid >> 6] = mx;
  __syncthreads();
  mx = redf[0];
  for (int w = 1; w < NT / 64; ++w) mx = fmaxf(mx, redf[w]);
  double sum = 0.0;                                          // lane 0 only
  for (int c0 = 0; c0 < V; c0 += CH) {
    const int n = min(CH, V - c0);
    for (int i = tid; i < n; i += NT) {
      const float e = (float)exp((double)probs[c0 + i] - (double)mx);   // stored to fp32 (:187)
      probs[c0 + i] = e;
      buf[i] = (double)e;
    }
    __syncthreads();
    if (tid == 0) seq_accumulate(buf, n, c0, sum, nullptr, INFINITY);    // sum of the ROUNDED values, in order (:189)
    __syncthreads();
  }
  if (tid == 0) shd[0] = sum;
  __syncthreads();
  sum = shd[0];
  for (int i = tid; i < V; i += NT) {
    probs[i] = (float)((double)probs[i] / sum);             // :192
    if (idx) idx[i] = i;
  }
  __syncthreads();
}

// After lane 0 knows the threshold r and the recorded boundary sums: first index i < limit_idx with
// r < (running sum through i), or -1.  `vals` are the values in accumulation order.
__device__ __forceinline__ int first_crossing(const float* vals, int limit_idx, double r, const double* bound, double* buf, int* shi) {
  const int tid = threadIdx.x;
  if (tid == 0) {
    int seg = -1;
    for (int s = 0; s * SEG < limit_idx; ++s) {
      const bool complete = (s + 1) * SEG <= limit_idx;      // its last element is below limit_idx
      if (!complete || r < bound[s]) { seg = s; break; }     // partial last segment: scan it; complete one: crossing is inside
    }
    shi[0] = seg;
  }
  __syncthreads();
  const int seg = shi[0];
  if (seg < 0) return -1;
  const int g0 = seg * SEG, n = min(SEG, limit_idx - g0);
  for (int i = tid; i < n; i += NT) buf[i] = (double)vals[g0 + i];
  __syncthreads();
  if (tid == 0) {
    double a = seg ? bound[seg - 1] : 0.0;                   // exact running sum at the segment start
    int hit = -1;
    for (int i = 0; i < n; ++i) {
      a += buf[i];
      if (r < a) { hit = g0 + i; break; }
    }
    shi[1] = hit;
  }
  __syncthreads();
  return shi[1];
}

// temperature + softmax + sample (llama2.ts:480-487, 368-376) + advance.
__global__ void __launch_bounds__(NT) sample_kernel(const float* logits, int V, const double* params, float* probs,
                                                     unsigned long long* rng, int* tokpos, int* tokens_out) {
  __shared__ __attribute__((aligned(16))) double buf[CH];
  __shared__ float redf[NT / 64];
  __shared__ double shd[2];
  __shared__ int shi[2];
  __shared__ double bound[MAXSEG];
  const int tid = threadIdx.x;
  softmax_in_place(logits, V, params[0], probs, nullptr, buf, redf, shd);
  double cum = 0.0;
  for (int c0 = 0; c0 < V; c0 += CH) {
    const int n = min(CH, V - c0);
    for (int i = tid; i < n; i += NT) buf[i] = (double)probs[c0 + i];
    __syncthreads();
    if (tid == 0) seq_accumulate(buf, n, c0, cum, bound, INFINITY);
    __syncthreads();
  }
  if (tid == 0) {
    if (V & (SEG - 1)) bound[V / SEG] = cum;                 // close the last, partial segment
    shd[1] = (double)random_f32(rng) * cum;                  // randValue = random_f32() * sum (:370)
  }
  __syncthreads();
  const int hit = first_crossing(probs, V, shd[1], bound, buf, shi);
  if (tid == 0) advance(tokpos, tokens_out, hit < 0 ? 0 : hit);   // fall-through returns 0 (:375)
}

// Stage 1 of the top-p branch: temperature + softmax, and the identity permutation for the sort.
__global__ void __launch_bounds__(NT) softmax_kernel(const float* logits, int V, const double* params, float* probs, int* idx) {
  __shared__ __attribute__((aligned(16))) double buf[CH];
  __shared__ float redf[NT / 64];
  __shared__ double shd[2];
  softmax_in_place(logits, V, params[0], probs, idx, buf, redf, shd);
}

// Stage 3: sample_topp (llama2.ts:378-394) on the sorted pairs + advance.
__global__ void __launch_bounds__(NT) topp_kernel(const float* sorted, const int* sorted_idx, int V, const double* params,
                                                   unsigned long long* rng, int* tokpos, int* tokens_out) {
  __shared__ __attribute__((aligned(16))) double buf[CH];
  __shared__ double shd[2];
  __shared__ int shi[3];
  __shared__ double bound[MAXSEG];
  const int tid = threadIdx.x;
  const double topp = params[1];
  double cum = 0.0;
  if (tid == 0) shi[2] = -1;
  __syncthreads();
  for (int c0 = 0; c0 < V; c0 += CH) {                      // cumProb until it exceeds topp (:384-386)
    const int n = min(CH, V - c0);
    for (int i = tid; i < n; i += NT) buf[i] = (double)sorted[c0 + i];
    __syncthreads();
    if (tid == 0) {
      const int at = seq_accumulate(buf, n, c0, cum, bound, topp);
      if (at >= 0) shi[2] = c0 + at;
    }
    __syncthreads();
    if (shi[2] >= 0) break;
  }
  const int last = shi[2] < 0 ? 0 : shi[2];                  // never crossed: lastIdx stays 0 (:383)
  if (tid == 0) shd[1] = (double)random_f32(rng) * cum;      // cumProb as the loop left it (:388)
  __syncthreads();
  const int hit = first_crossing(sorted, last, shd[1], bound, buf, shi);   // i < lastIdx only (:390)
  if (tid == 0) advance(tokpos, tokens_out, hit < 0 ? 0 : sorted_idx[hit]);
}




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

template <bool COMP>
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
      out[el.serial[k] ? seg_cnt(s) - 1 : seg_cnt(s)] = r;
      if (tile_end) cnt[tile] = seg_cnt(s) + (el.serial[k] ? 0 : 1);
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
template <bool IN_LDS>
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
        for (int r = 0, n = sh.off[t + 1] - sh.off[t]; r < n; ++r, ++k) step(k, a.recs[(size_t)t * (TILE + 1) + r]);
    }
    sh.val = S;
  }
}

// Order the tiles' runs (first run of every tile in sh.off, records staged in LDS when they fit) and walk them.
// Returns the number of runs; *in_lds says where the per-run state went.  Every thread of the workgroup calls it.
__device__ __forceinline__ int chain_total(const ChainArgs& a, ChainShared& sh, bool* in_lds) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = tid < a.G ? a.cnt[tid] : 0;
  Run r0, r1;                                              // nearly every tile has one or two runs: fetched together with the count
  if (tid < a.G) { r0 = a.recs[(size_t)tid * (TILE + 1)]; r1 = a.recs[(size_t)tid * (TILE + 1) + 1]; }
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
      for (int r = 2; r < c; ++r) sh.rec[first + r] = a.recs[(size_t)tid * (TILE + 1) + r];
    }
    __syncthreads();
    chain_walk<true>(a, sh, T);
  } else {
    chain_walk<false>(a, sh, T);
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

// ---- top-p: descending stable order = ascending order of the key (~probability bits, id)
typedef unsigned long long u64;

__device__ __forceinline__ void order_pair(u64& a, u64& b, bool up) {
  const bool sw = (a > b) == up;
  const u64 lo = sw ? b : a, hi = sw ? a : b;
  a = lo; b = hi;
}

// Bitonic sort of one sort tile: SIT consecutive positions per thread, so the small strides stay inside a thread, the
// middle ones are lane exchanges inside a wave, and only the strides >= 64 * SIT go through LDS.  Measured: 8 keys per
// thread (2048-key tiles, half as many for the rank merge to search) take 12 us longer here and save 4 us there.
constexpr int SIT = 4, STILE = TN * SIT;
template <bool FUSED>
__global__ void __launch_bounds__(TN) sort_tile_kernel(ChainArgs a, const float* probs, int V, float* run_p, int* run_id) {
  __shared__ union { ChainShared sh; u64 xch[STILE]; } lds;      // the chain is over before the first exchange (a barrier in between)
  ChainShared& sh = lds.sh;
  u64* xch = lds.xch;
  const int tid = threadIdx.x, base = blockIdx.x * STILE, p0 = tid * SIT;
  double s = 1.0;
  if (FUSED) {                                                 // probs holds the exps: every workgroup derives their exact total itself
    bool in_lds;
    chain_total(a, sh, &in_lds);
    s = sh.val;
  }
  u64 v[SIT];
#pragma unroll
  for (int k = 0; k < SIT; ++k) {
    const int i = base + p0 + k;
    const float e = (i < V) ? probs[i] : 0.0f;
    const float p = FUSED ? (float)((double)e / s) : e;                      // :192
    v[k] = (i < V) ? (((u64)(0xffffffffu - __float_as_uint(p)) << 32) | (unsigned)i) : ~0ull;
  }
#pragma unroll
  for (int k2 = 2; k2 <= STILE; k2 <<= 1) {
#pragma unroll
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      if (j < SIT) {
#pragma unroll
        for (int k = 0; k < SIT; ++k) if ((k & j) == 0) order_pair(v[k], v[k | j], ((p0 + k) & k2) == 0);
      } else {
        const bool keep_min = ((p0 & j) == 0) == ((p0 & k2) == 0);
        u64 o[SIT];
        if (j < 64 * SIT) {
#pragma unroll
          for (int k = 0; k < SIT; ++k) o[k] = __shfl_xor(v[k], j / SIT, 64);
        } else {
          __syncthreads();
#pragma unroll
          for (int k = 0; k < SIT; ++k) xch[p0 + k] = v[k];
          __syncthreads();
#pragma unroll
          for (int k = 0; k < SIT; ++k) o[k] = xch[(p0 ^ j) + k];
        }
#pragma unroll
        for (int k = 0; k < SIT; ++k) v[k] = keep_min ? (v[k] < o[k] ? v[k] : o[k]) : (v[k] > o[k] ? v[k] : o[k]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < SIT; ++k) {
    const bool pad = v[k] == ~0ull;
    run_p[base + p0 + k] = pad ? -1.0f : __uint_as_float(0xffffffffu - (unsigned)(v[k] >> 32));
    run_id[base + p0 + k] = pad ? -1 : (int)(unsigned)v[k];
  }
}

// Every element's place in the merged order: its place in its own tile + the number of elements of every other tile
// in front of it (ties: the tile with the smaller ids first), by binary search in the G sorted tiles held in LDS.
constexpr int RT = 512;                                           // threads = elements per workgroup of the rank merge (256 and 1024: the same time)
// Also adds every element to the sum of the tile of the merged order it lands in (part[], zero on entry): the approximate
// prefix of the next stage.  fp64 atomics in no fixed order -- the prefix only has to be approximate (exact_sum.h).
__global__ void __launch_bounds__(RT) sort_rank_kernel(const float* run_p, const int* run_id, int GS, int G, float* sorted, int* ids, double* part) {
  extern __shared__ int lds_p[];                                // GS * STILE probability bit patterns (pads: negative), then G tile sums
  const int tid = threadIdx.x, n = GS * STILE;
  double* lpart = reinterpret_cast<double*>(lds_p + n);
  if (tid < G) lpart[tid] = 0.0;
  // every workgroup pulls all G tiles; measured: one 16-byte load in flight per thread (8 KB per workgroup) beats 4, 8 and 16
  // (+5 / +5 / +8 us) -- the same lines are wanted by every CU at once and deeper queues only lengthen the wait behind them;
  // an LDS-DMA fill (global_load_lds_dwordx4, every 1 KB chunk in flight at once) takes the same time as this loop
  for (int j = tid * 4; j < n; j += RT * 4) *reinterpret_cast<int4*>(lds_p + j) = *reinterpret_cast<const int4*>(reinterpret_cast<const int*>(run_p) + j);
  const int e = blockIdx.x * RT + tid;
  const int my_id = e < n ? run_id[e] : -1;
  __syncthreads();
  const int mine = e < n ? lds_p[e] : -1;
  if (mine >= 0) {                                              // not a pad
  const int own = e / STILE;
  int rank = e - own * STILE;
  constexpr int U = 8;                                          // searches in flight per thread
  for (int b0 = 0; b0 < GS; b0 += U) {
    int lo[U], thr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int b = min(b0 + u, GS - 1);
      lo[u] = b * STILE;
      thr[u] = mine - (b < own ? 1 : 0);                        // earlier tile: elements >= mine come first; later tile: only > mine
    }
#pragma unroll
    for (int s = STILE / 2; s > 0; s >>= 1) {
#pragma unroll
      for (int u = 0; u < U; ++u) if (lds_p[lo[u] + s - 1] > thr[u]) lo[u] += s;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int b = min(b0 + u, GS - 1);
      int cnt = lo[u] - b * STILE;
      if (cnt == STILE - 1 && lds_p[lo[u]] > thr[u]) cnt = STILE;
      if (b0 + u < GS && b != own) rank += cnt;
    }
  }
  sorted[rank] = __int_as_float(mine);
  ids[rank] = my_id;
  atomicAdd(lpart + rank / TILE, (double)__int_as_float(mine));
  }
  __syncthreads();
  if (tid < G && lpart[tid] != 0.0) atomicAdd(part + tid, lpart[tid]);
}

// Stage 1 = the exps (recs / cnt), stage 2 = the probabilities, in index or in sorted order (recs2 / cnt2, cq / cm): two
// sets of run records because the fused kernels write stage 2 while other workgroups still read stage 1.
static ChainArgs chain_args(const Sampler& s, const float* x, const double* part, bool stage2) {
  ChainArgs a = {};
  a.x = x; a.V = s.V; a.G = s.G; a.part = part;
  a.recs = (const Run*)(stage2 ? s.recs2 : s.recs); a.cnt = stage2 ? s.cnt2 : s.cnt;
  a.off = s.off; a.S = s.runS; a.End = s.runEnd; a.Bad = s.runBad; a.cq = s.cq; a.cm = s.cm;
  a.params = s.params; a.rng = s.rng; a.mxkey = s.mxkey;
  return a;
}

static hipError_t launch_chain(const ChainArgs& a, int mode, hipStream_t st) {
  if (mode == CHAIN_SAMPLE) hipLaunchKernelGGL(chain_kernel<CHAIN_SAMPLE>, dim3(1), dim3(TN), 0, st, a);
  else if (mode == CHAIN_TOPP) hipLaunchKernelGGL(chain_kernel<CHAIN_TOPP>, dim3(1), dim3(TN), 0, st, a);
  else hipLaunchKernelGGL(chain_kernel<CHAIN_DEBUG>, dim3(1), dim3(TN), 0, st, a);
  return hipGetLastError();
}

hipError_t running_sums(const float* x_dev, int n, double* prefix_dev, hipStream_t st) {
  Sampler s;
  hipError_t e = create(&s, n);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(tile_sums_kernel, dim3(s.G), dim3(TN), 0, st, x_dev, n, s.part);
  hipLaunchKernelGGL(runs_kernel<false>, dim3(s.G), dim3(TN), 0, st, x_dev, n, s.part, (Run*)s.recs, s.cnt, s.cq, s.cm);
  e = launch_chain(chain_args(s, x_dev, s.part, false), CHAIN_DEBUG, st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(prefix_kernel, dim3(s.G), dim3(TN), 0, st, x_dev, n, s.part, s.off, s.runS, s.runBad, s.runEnd, prefix_dev);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  destroy(&s);
  return e;
}

hipError_t create(Sampler* s, int V) {
  if (V <= 0 || V > MAX_VOCAB) return hipErrorInvalidValue;
  s->V = V;
  s->G = (V + TILE - 1) / TILE;
  hipError_t e;
#define L2S(x) do { e = (x); if (e != hipSuccess) { destroy(s); return e; } } while (0)
  const size_t padded = (size_t)((V + STILE - 1) / STILE) * STILE, max_runs = (size_t)s->G * (TILE + 1);
  L2S(hipMalloc(&s->probs, padded * 4));
  L2S(hipMalloc(&s->probs_n, padded * 4));
  L2S(hipMalloc(&s->probs_sorted, padded * 4));
  L2S(hipMalloc(&s->idx, padded * 4));
  L2S(hipMalloc(&s->idx_sorted, padded * 4));
  L2S(hipMalloc(&s->run_p, padded * 4));
  L2S(hipMalloc(&s->params, 2 * sizeof(double)));
  L2S(hipMalloc(&s->rng, sizeof(unsigned long long)));
  L2S(hipMalloc(&s->part, (size_t)s->G * sizeof(double)));
  L2S(hipMalloc(&s->part_sorted, (size_t)s->G * sizeof(double)));
  L2S(hipMemset(s->part_sorted, 0, (size_t)s->G * sizeof(double)));
  L2S(hipMalloc(&s->recs, max_runs * sizeof(Run)));
  L2S(hipMalloc(&s->recs2, max_runs * sizeof(Run)));
  L2S(hipMalloc(&s->cnt, (size_t)s->G * sizeof(int)));
  L2S(hipMalloc(&s->cnt2, (size_t)s->G * sizeof(int)));
  L2S(hipMalloc(&s->off, (size_t)(s->G + 1) * sizeof(int)));
  L2S(hipMalloc(&s->runS, max_runs * sizeof(double)));
  L2S(hipMalloc(&s->runEnd, max_runs * sizeof(int)));
  L2S(hipMalloc(&s->runBad, max_runs * sizeof(int)));
  L2S(hipMalloc(&s->cq, padded * sizeof(unsigned long long)));
  L2S(hipMalloc(&s->cm, padded * sizeof(int)));
  L2S(hipMalloc(&s->mxkey, sizeof(unsigned)));
  L2S(hipMemset(s->mxkey, 0, sizeof(unsigned)));
  { const char* g_ = getenv("L2_TEST_HOOKS"); const char* e_ = getenv("L2_SAMPLER_SERIAL"); s->serial = g_ && atoi(g_) != 0 && e_ && atoi(e_) != 0; }   // A/B form, development gate
  // the rank merge holds every tile in LDS: 4 bytes per (padded) element of the 160 KB
  const size_t rank_lds = padded * 4 + (size_t)s->G * 8;
  s->own_sort = !s->serial && rank_lds <= 160 * 1024;
  if (s->own_sort) L2S(hipFuncSetAttribute(reinterpret_cast<const void*>(sort_rank_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rank_lds));
  s->sort_temp_bytes = 0;
  if (!s->own_sort) {
    L2S(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, s->sort_temp_bytes, s->probs_n, s->probs_sorted, s->idx, s->idx_sorted, V, 0, 32, nullptr));
    L2S(hipMalloc(&s->sort_temp, s->sort_temp_bytes ? s->sort_temp_bytes : 16));
  }
#undef L2S
  return hipSuccess;
}

void destroy(Sampler* s) {
  void* bufs[] = {s->probs, s->probs_n, s->probs_sorted, s->idx, s->idx_sorted, s->run_p, s->params, s->rng, s->sort_temp, s->part, s->part_sorted,
                  s->recs, s->recs2, s->cnt, s->cnt2, s->off, s->runS, s->runEnd, s->runBad, s->cq, s->cm, s->mxkey};
  for (void* b : bufs) if (b) (void)hipFree(b);
  *s = Sampler();
}

__global__ void iota_kernel(int* idx, int V) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < V) idx[i] = i;
}

hipError_t enqueue(const Sampler& s, const float* logits, bool topp_mode, int* tokpos, int* tokens_out, unsigned long long* amax, hipStream_t st) {
  hipError_t e;
  if (s.serial) {
    if (!topp_mode) {
      hipLaunchKernelGGL(sample_kernel, dim3(1), dim3(NT), 0, st, logits, s.V, s.params, s.probs, s.rng, tokpos, tokens_out);
      return hipGetLastError();
    }
    hipLaunchKernelGGL(softmax_kernel, dim3(1), dim3(NT), 0, st, logits, s.V, s.params, s.probs, s.idx);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    size_t bytes = s.sort_temp_bytes;
    e = hipcub::DeviceRadixSort::SortPairsDescending(s.sort_temp, bytes, s.probs, s.probs_sorted, s.idx, s.idx_sorted, s.V, 0, 32, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(topp_kernel, dim3(1), dim3(NT), 0, st, s.probs_sorted, s.idx_sorted, s.V, s.params, s.rng, tokpos, tokens_out);
    return hipGetLastError();
  }
  // temperature + exp (:481-483, :183-188), runs of the exps' running sum
  if (!amax) hipLaunchKernelGGL(scaled_max_kernel, dim3(s.G), dim3(TN), 0, st, logits, s.V, s.params, s.mxkey);
  hipLaunchKernelGGL(exp_kernel, dim3(s.G), dim3(TN), 0, st, logits, s.V, s.params, s.mxkey, amax, s.probs, s.part);
  hipLaunchKernelGGL(runs_kernel<false>, dim3(s.G), dim3(TN), 0, st, s.probs, s.V, s.part, (Run*)s.recs, s.cnt, s.cq, s.cm);
  const ChainArgs exps = chain_args(s, s.probs, s.part, false);
  ChainArgs pick;
  if (!topp_mode) {
    // exact total -> probabilities -> their runs, in one launch; then sample (:368-376)
    hipLaunchKernelGGL(normalise_runs_kernel, dim3(s.G), dim3(TN), 0, st, exps, s.probs_n, (Run*)s.recs2, s.cnt2, s.cq, s.cm);
    pick = chain_args(s, s.probs_n, s.part, true);
  } else if (s.own_sort) {
    // sample_topp (:378-394): exact total -> probabilities -> sorted tiles in one launch, rank merge, runs of the sorted order
    const int gs = (s.V + STILE - 1) / STILE, n = gs * STILE;
    hipLaunchKernelGGL(sort_tile_kernel<true>, dim3(gs), dim3(TN), 0, st, exps, s.probs, s.V, s.run_p, s.idx);
    hipLaunchKernelGGL(sort_rank_kernel, dim3((n + RT - 1) / RT), dim3(RT), (size_t)n * 4 + s.G * sizeof(double), st, s.run_p, s.idx, gs, s.G, s.probs_sorted,
                       s.idx_sorted, s.part_sorted);
    hipLaunchKernelGGL(runs_kernel<true>, dim3(s.G), dim3(TN), 0, st, s.probs_sorted, s.V, s.part_sorted, (Run*)s.recs2, s.cnt2, s.cq, s.cm);
    pick = chain_args(s, s.probs_sorted, s.part_sorted, true);
    pick.part_sorted = s.part_sorted;
  } else {
    // vocabularies beyond the rank merge's LDS: rocPRIM's radix sort
    hipLaunchKernelGGL(normalise_runs_kernel, dim3(s.G), dim3(TN), 0, st, exps, s.probs_n, (Run*)s.recs2, s.cnt2, s.cq, s.cm);
    hipLaunchKernelGGL(iota_kernel, dim3((s.V + 255) / 256), dim3(256), 0, st, s.idx, s.V);
    size_t bytes = s.sort_temp_bytes;
    e = hipcub::DeviceRadixSort::SortPairsDescending(s.sort_temp, bytes, s.probs_n, s.probs_sorted, s.idx, s.idx_sorted, s.V, 0, 32, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(tile_sums_kernel, dim3(s.G), dim3(TN), 0, st, s.probs_sorted, s.V, s.part);
    hipLaunchKernelGGL(runs_kernel<true>, dim3(s.G), dim3(TN), 0, st, s.probs_sorted, s.V, s.part, (Run*)s.recs2, s.cnt2, s.cq, s.cm);
    pick = chain_args(s, s.probs_sorted, s.part, true);
  }
  if ((e = hipGetLastError()) != hipSuccess) return e;
  pick.tokpos = tokpos; pick.tokens_out = tokens_out; pick.amax = amax;
  pick.ids = topp_mode ? s.idx_sorted : nullptr;
  return launch_chain(pick, topp_mode ? CHAIN_TOPP : CHAIN_SAMPLE, st);
}

}  // namespace l2s

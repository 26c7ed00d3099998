// Device-side sampling for libllama2hip.so (gfx950): temperature scaling, softmax, sample / sample_topp and the
// xorshift* RNG of wizzard0/llama2.ts (llama2.ts:348-394, 476-493), restated so that the SAME token comes out.
//
// What makes that non-trivial: the reference's sums are sequential fp64 accumulations of fp32 values
// (softmax :189, sample :369/:373, sample_topp :385/:391) and the chosen index depends on comparing a random
// threshold against those running sums.  A parallel (tree) sum differs in the last bits and can flip an index,
// so every running sum here is produced by ONE lane adding in index order -- ~10 cycles per element, ~130 us per
// pass over 32 000 values -- while everything that is elementwise (divide, exp, normalise, search) uses the
// whole 1024-thread workgroup.  The index search avoids a second full sequential pass: the running sum is
// recorded at every 512th element; once the threshold is known the (monotone) boundary values locate the
// segment of the first crossing and only that segment is re-accumulated from its exact starting value.
// The descending stable sort of sample_topp (Array.prototype.sort is stable in V8 >= 7.0) is rocPRIM's radix sort
// on (probability, id) pairs: stable, so ties stay in id order.
#include "sampler.h"

#include <hipcub/hipcub.hpp>
#include <stdlib.h>

namespace l2s {

#pragma clang fp contract(off)

constexpr int NT = 1024;     // threads of the one workgroup
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
// The same running sums, exactly, but in parallel.
//
// S_i = fl(S_{i-1} + x_i) with x_i >= 0 fp32 and S fp64 is a serial recurrence, and on this chip a dependent
// v_add_f64 costs ~19 cycles: 2 x 32 000 of them are 0.5 ms per sampled token.  But while the exponent E of S
// does not change, S lives on the grid g = 2^(E-52), S = M g with 2^52 <= M < 2^53, and rounding S + x to
// nearest is M + rn(x / g) -- integer arithmetic, associative -- except that a tie (x / g = k + 1/2) goes to the
// EVEN neighbour, which depends on the parity of the running M.  After a tie the sum is even by construction, so
// the parity seen by every element is a segmented XOR scan with ties as reset points.  So, per window of 4096
// elements and per exponent: (1) every thread reduces its elements to (k, tie?) on the current grid,
// (2) a block scan gives each element the parity in front of it, which settles the ties, (3) a saturating block
// scan of the integer increments gives every M_i at once, (4) the first element that reaches 2^53 (S crosses a
// power of two -- ~17 times over a whole pass) is added the ordinary way and the grid is re-based there.
// Bit-identical to the serial loop by construction; checked against the oracle's loop on adversarial vectors
// (ties, power-of-two crossings, zeros, subnormals) through l2_debug_running_sums.
constexpr int IT = 4, WIN = NT * IT;   // measured: 4 beats 8 and 16 (per-thread work, not barriers, sets the iteration time)
constexpr unsigned long long TWO52 = 1ull << 52, TWO53 = 1ull << 53, CAP = 1ull << 54;

struct SatAdd {
  __device__ __forceinline__ unsigned long long operator()(unsigned long long a, unsigned long long b) const {
    const unsigned long long s = a + b;
    return s > CAP ? CAP : s;
  }
};
struct ParityOp {   // bit 1: a tie (reset to even) lies inside the range; bit 0: XOR of the increments after the last reset
  __device__ __forceinline__ int operator()(int a, int b) const { return (b & 2) ? b : ((a & 2) | ((a ^ b) & 1)); }
};

typedef hipcub::BlockScan<int, NT, hipcub::BLOCK_SCAN_WARP_SCANS> ParScan;
typedef hipcub::BlockScan<unsigned long long, NT, hipcub::BLOCK_SCAN_WARP_SCANS> SumScan;
struct ExactShared {
  union { typename ParScan::TempStorage par; typename SumScan::TempStorage sum; } scan;
  double S;
  int first;
};

__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
  return v;
}

// prefix[i] (may be null) = S_i for i in [0, n); returns S_{n-1} (0 for n = 0).  Whole workgroup, uniform control flow.
// Windows are fixed ([w0, w0 + WIN), values held in registers); inside a window `pos` moves forward at every
// power-of-two crossing and elements in front of it are final.  The first HEAD elements are added by one lane:
// while S is still small nearly every add crosses a power of two, so the grid trick has nothing to work with there.
constexpr int HEAD = 512;
__device__ __forceinline__ double exact_running_sums(const float* x, int n, double* prefix, ExactShared& sh, double* headbuf) {
  const int tid = threadIdx.x;
  const int nh = min(n, HEAD);
  for (int i = tid; i < nh; i += NT) headbuf[i] = (double)x[i];
  __syncthreads();
  if (tid == 0) {
    double a = 0.0;
    for (int i = 0; i < nh; ++i) { a += headbuf[i]; headbuf[i] = a; }
    sh.S = a;
  }
  __syncthreads();
  if (prefix) for (int i = tid; i < nh; i += NT) prefix[i] = headbuf[i];
  double S = sh.S;
  __syncthreads();
  for (int w0 = nh; w0 < n; w0 += WIN) {
    const int wend = min(n, w0 + WIN);
    const int i0 = w0 + tid * IT;                      // blocked arrangement: thread t owns IT consecutive elements
    float xv[IT];
#pragma unroll
    for (int k = 0; k < IT; ++k) xv[k] = (i0 + k < wend) ? x[i0 + k] : 0.0f;
    int pos = w0;
    while (pos < wend) {
      if (tid == 0) sh.first = 0x7fffffff;
      __syncthreads();
      if (S == 0.0) {                                  // leading zeros: 0 + x is exact, the first non-zero value becomes S
        int f = 0x7fffffff;
#pragma unroll
        for (int k = IT - 1; k >= 0; --k) if (i0 + k >= pos && i0 + k < wend && xv[k] != 0.0f) f = i0 + k;
        f = wave_min_i32(f);
        if ((tid & 63) == 0 && f != 0x7fffffff) atomicMin(&sh.first, f);
        __syncthreads();
        const int first = sh.first;
        const int upto = min(first, wend);
#pragma unroll
        for (int k = 0; k < IT; ++k) {
          const int i = i0 + k;
          if (i >= pos && i < upto && prefix) prefix[i] = 0.0;
          if (i == first && i < wend) { sh.S = (double)xv[k]; if (prefix) prefix[i] = (double)xv[k]; }
        }
        __syncthreads();
        if (first < wend) { S = sh.S; pos = first + 1; } else pos = wend;
        __syncthreads();
        continue;
      }
      const unsigned long long sb = (unsigned long long)__double_as_longlong(S);
      const int E = (int)((sb >> 52) & 0x7ff) - 1023;   // S is a normal double here (>= 2^-149)
      const unsigned long long M = (sb & (TWO52 - 1)) | TWO52;
      unsigned long long r[IT];
      int fv[IT];
#pragma unroll
      for (int k = 0; k < IT; ++k) {                    // x = m 2^lsb on the grid 2^(E-52); elements before pos are done: 0
        const unsigned xb = (i0 + k >= pos) ? __float_as_uint(xv[k]) : 0u;
        const int ef = (int)((xb >> 23) & 0xff);
        const unsigned m = (xb & 0x7fffffu) | (ef ? 0x800000u : 0u);
        const int shift = (ef ? ef - 127 : -126) - 23 - (E - 52);
        unsigned long long q = 0;
        int tie = 0;
        if (m != 0) {
          if (shift >= 0) q = (shift > 30) ? CAP : ((unsigned long long)m << shift);
          else if (shift > -25) {
            const int sft = -shift;
            const unsigned rem = m & ((1u << sft) - 1u), half = 1u << (sft - 1);
            q = (unsigned long long)(m >> sft) + (rem > half ? 1u : 0u);
            tie = (rem == half);
          }
        }
        r[k] = q;
        fv[k] = tie ? 2 : (int)(q & 1);
      }
      int pin[IT];
      ParScan(sh.scan.par).ExclusiveScan(fv, pin, 0, ParityOp());
      __syncthreads();
      const int p0 = (int)(M & 1);
#pragma unroll
      for (int k = 0; k < IT; ++k) {
        if (fv[k] & 2) {                                 // tie: round half to even
          const int before = (pin[k] & 2) ? (pin[k] & 1) : (p0 ^ (pin[k] & 1));
          r[k] += (unsigned long long)((before + (int)(r[k] & 1)) & 1);
        }
      }
      unsigned long long tex[IT];
      SumScan(sh.scan.sum).ExclusiveScan(r, tex, 0ull, SatAdd());
      int ov = 0x7fffffff;
#pragma unroll
      for (int k = IT - 1; k >= 0; --k) if (i0 + k >= pos && i0 + k < wend && M + SatAdd()(tex[k], r[k]) >= TWO53) ov = i0 + k;
      ov = wave_min_i32(ov);
      if ((tid & 63) == 0 && ov != 0x7fffffff) atomicMin(&sh.first, ov);
      __syncthreads();
      const int first = sh.first;                         // first element at which S reaches the next power of two
      const int upto = min(first, wend);
      const double g = __longlong_as_double((long long)(E - 52 + 1023) << 52);
#pragma unroll
      for (int k = 0; k < IT; ++k) {
        const int i = i0 + k;
        if (i >= pos && i < upto) {
          const double Si = (double)(M + tex[k] + r[k]) * g;   // < 2^53: exact
          if (prefix) prefix[i] = Si;
          if (i == wend - 1) sh.S = Si;
        } else if (i == first && i < wend) {
          const double Sn = (double)(M + tex[k]) * g + (double)xv[k];   // the ordinary add re-bases the grid
          if (prefix) prefix[i] = Sn;
          sh.S = Sn;
        }
      }
      __syncthreads();
      S = sh.S;
      pos = (first < wend) ? first + 1 : wend;
      __syncthreads();
    }
  }
  return S;
}

__device__ __forceinline__ float block_max_f32(float mx, float* redf) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  if ((tid & 63) == 0) redf[tid >> 6] = mx;
  __syncthreads();
  mx = redf[0];
  for (int w = 1; w < NT / 64; ++w) mx = fmaxf(mx, redf[w]);
  return mx;
}

__device__ __forceinline__ void softmax_in_place_par(const float* logits, int V, double T, float* probs, int* idx, float* redf, ExactShared& sh, double* headbuf) {
  const int tid = threadIdx.x;
  float mx = -INFINITY;
  for (int i = tid; i < V; i += NT) {
    const float x = (float)((double)logits[i] / T);        // :482
    probs[i] = x;
    mx = fmaxf(mx, x);
  }
  mx = block_max_f32(mx, redf);
  for (int i = tid; i < V; i += NT) probs[i] = (float)exp((double)probs[i] - (double)mx);   // :187
  __syncthreads();
  const double sum = exact_running_sums(probs, V, nullptr, sh, headbuf);                               // :189
  for (int i = tid; i < V; i += NT) {
    probs[i] = (float)((double)probs[i] / sum);             // :192
    if (idx) idx[i] = i;
  }
  __syncthreads();
}

// first i in [0, limit) with pred(prefix[i]); 0x7fffffff if none
template <class Pred>
__device__ __forceinline__ int first_index(const double* prefix, int limit, Pred pred, int* slot) {
  const int tid = threadIdx.x;
  if (tid == 0) *slot = 0x7fffffff;
  __syncthreads();
  int f = 0x7fffffff;
  for (int i = tid; i < limit; i += NT) if (pred(prefix[i])) { f = i; break; }
  if (f != 0x7fffffff) atomicMin(slot, f);
  __syncthreads();
  const int out = *slot;
  __syncthreads();
  return out;
}

__global__ void __launch_bounds__(NT) sample_par_kernel(const float* logits, int V, const double* params, float* probs, double* prefix,
                                                         unsigned long long* rng, int* tokpos, int* tokens_out) {
  __shared__ ExactShared sh;
  __shared__ float redf[NT / 64];
  __shared__ double shr;
  __shared__ int slot;
  __shared__ double headbuf[HEAD];
  softmax_in_place_par(logits, V, params[0], probs, nullptr, redf, sh, headbuf);
  const double total = exact_running_sums(probs, V, prefix, sh, headbuf);
  if (threadIdx.x == 0) shr = (double)random_f32(rng) * total;     // :370
  __syncthreads();
  const double r = shr;
  const int hit = first_index(prefix, V, [r](double c) { return r < c; }, &slot);   // :373
  if (threadIdx.x == 0) advance(tokpos, tokens_out, hit == 0x7fffffff ? 0 : hit);
}

__global__ void __launch_bounds__(NT) softmax_par_kernel(const float* logits, int V, const double* params, float* probs, int* idx) {
  __shared__ ExactShared sh;
  __shared__ float redf[NT / 64];
  __shared__ double headbuf[HEAD];
  softmax_in_place_par(logits, V, params[0], probs, idx, redf, sh, headbuf);
}

__global__ void __launch_bounds__(NT) topp_par_kernel(const float* sorted, const int* sorted_idx, int V, const double* params, double* prefix,
                                                       unsigned long long* rng, int* tokpos, int* tokens_out) {
  __shared__ ExactShared sh;
  __shared__ double shr;
  __shared__ int slot;
  __shared__ double headbuf[HEAD];
  const double topp = params[1];
  const double total = exact_running_sums(sorted, V, prefix, sh, headbuf);
  const int cross = first_index(prefix, V, [topp](double c) { return c > topp; }, &slot);   // :385
  const int last = cross == 0x7fffffff ? 0 : cross;                                        // never crossed: lastIdx stays 0
  if (threadIdx.x == 0) shr = (double)random_f32(rng) * (cross == 0x7fffffff ? total : prefix[cross]);   // :388
  __syncthreads();
  const double r = shr;
  const int hit = first_index(prefix, last, [r](double c) { return r < c; }, &slot);       // i < lastIdx (:390)
  if (threadIdx.x == 0) advance(tokpos, tokens_out, hit == 0x7fffffff ? 0 : sorted_idx[hit]);
}

// Diagnostic: the running sums of an arbitrary vector (tests of exact_running_sums against the oracle's serial loop).
__global__ void __launch_bounds__(NT) running_sums_kernel(const float* x, int n, double* prefix) {
  __shared__ ExactShared sh;
  __shared__ double headbuf[HEAD];
  exact_running_sums(x, n, prefix, sh, headbuf);
}

hipError_t running_sums(const float* x_dev, int n, double* prefix_dev, hipStream_t st) {
  hipLaunchKernelGGL(running_sums_kernel, dim3(1), dim3(NT), 0, st, x_dev, n, prefix_dev);
  return hipGetLastError();
}

hipError_t create(Sampler* s, int V) {
  if (V <= 0 || V > MAX_VOCAB) return hipErrorInvalidValue;
  s->V = V;
  hipError_t e;
#define L2S(x) do { e = (x); if (e != hipSuccess) { destroy(s); return e; } } while (0)
  L2S(hipMalloc(&s->probs, (size_t)V * 4));
  L2S(hipMalloc(&s->probs_sorted, (size_t)V * 4));
  L2S(hipMalloc(&s->idx, (size_t)V * 4));
  L2S(hipMalloc(&s->idx_sorted, (size_t)V * 4));
  L2S(hipMalloc(&s->params, 2 * sizeof(double)));
  L2S(hipMalloc(&s->rng, sizeof(unsigned long long)));
  L2S(hipMalloc(&s->prefix, (size_t)V * sizeof(double)));
  { const char* e_ = getenv("L2_SAMPLER_SERIAL"); s->serial = e_ && atoi(e_) != 0; }
  s->sort_temp_bytes = 0;
  L2S(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, s->sort_temp_bytes, s->probs, s->probs_sorted, s->idx, s->idx_sorted, V, 0, 32, nullptr));
  L2S(hipMalloc(&s->sort_temp, s->sort_temp_bytes ? s->sort_temp_bytes : 16));
#undef L2S
  return hipSuccess;
}

void destroy(Sampler* s) {
  void* bufs[] = {s->probs, s->probs_sorted, s->idx, s->idx_sorted, s->params, s->rng, s->sort_temp, s->prefix};
  for (void* b : bufs) if (b) (void)hipFree(b);
  *s = Sampler();
}

hipError_t enqueue(const Sampler& s, const float* logits, bool topp_mode, int* tokpos, int* tokens_out, hipStream_t st) {
  if (!topp_mode) {
    if (s.serial) hipLaunchKernelGGL(sample_kernel, dim3(1), dim3(NT), 0, st, logits, s.V, s.params, s.probs, s.rng, tokpos, tokens_out);
    else hipLaunchKernelGGL(sample_par_kernel, dim3(1), dim3(NT), 0, st, logits, s.V, s.params, s.probs, s.prefix, s.rng, tokpos, tokens_out);
    return hipGetLastError();
  }
  if (s.serial) hipLaunchKernelGGL(softmax_kernel, dim3(1), dim3(NT), 0, st, logits, s.V, s.params, s.probs, s.idx);
  else hipLaunchKernelGGL(softmax_par_kernel, dim3(1), dim3(NT), 0, st, logits, s.V, s.params, s.probs, s.idx);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  size_t bytes = s.sort_temp_bytes;
  e = hipcub::DeviceRadixSort::SortPairsDescending(s.sort_temp, bytes, s.probs, s.probs_sorted, s.idx, s.idx_sorted, s.V, 0, 32, st);
  if (e != hipSuccess) return e;
  if (s.serial) hipLaunchKernelGGL(topp_kernel, dim3(1), dim3(NT), 0, st, s.probs_sorted, s.idx_sorted, s.V, s.params, s.rng, tokpos, tokens_out);
  else hipLaunchKernelGGL(topp_par_kernel, dim3(1), dim3(NT), 0, st, s.probs_sorted, s.idx_sorted, s.V, s.params, s.prefix, s.rng, tokpos, tokens_out);
  return hipGetLastError();
}

}  // namespace l2s

// sampler_serial.hip.h -- the one-workgroup A/B form of the device sampler: one lane accumulates in index order (L2_SAMPLER_SERIAL)
// Part of sampler.hip (included there inside namespace l2s, in order); not a stand-alone header.
#pragma once

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

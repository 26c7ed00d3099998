// kernels.hip.h -- gfx950 (CDNA4, wave64) device code for the llama2.ts forward pass.
//
// One token's forward (llama2.ts:205-303) is a chain of batch-1 GEMVs over row-major fp32 weights
// (matmul, llama2.ts:196-203), so every kernel here is HBM-bound: weights are streamed exactly once
// with 16-byte non-temporal loads straight into VGPRs (no LDS hop: nothing is reused across waves),
// the input vector is staged once per workgroup in LDS, and all the elementwise work of the
// reference (rmsnorm, RoPE, KV store, residual add, SwiGLU) is fused into the prologue/epilogue of
// the GEMV that produces or consumes it.
//
// Numeric contract: the reference computes in JS doubles and rounds to fp32 only
// on Float32Array stores.  Every accumulator here is fp64 and every store rounds exactly where the
// reference stores.  All products are fp32 x fp32 (exact in fp64), so FMA contraction cannot change
// a result; only the summation ORDER differs (tree vs sequential), which moves an fp64 sum by ~1e-16
// relative and almost never changes the fp32 it rounds to.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace l2k {

typedef float f4 __attribute__((ext_vector_type(4)));

enum { MODE_QKV = 0, MODE_WO = 1, MODE_W13 = 2, MODE_W2 = 3, MODE_CLS = 4 };

struct PhaseArgs {
  const float* w0;    // QKV: wq[l]   W13: w1[l]   else: the matrix
  const float* w1;    // QKV: wk[l]   W13: w3[l]
  const float* w2;    // QKV: wv[l]
  const float* in;    // input vector: x (QKV, W13, CLS), xb (WO), hb (W2)
  const float* emb;   // token_embedding_table when the input/residual is the embedding row (layer 0), else null
  const float* rmsw;  // rmsnorm weight (QKV, W13, CLS) or null
  float* out;         // QKV: q   WO/W2: x   W13: hb   CLS: logits
  float* out_k;       // QKV: key_cache   + l*S*d
  float* out_v;       // QKV: value_cache + l*S*d
  float* aux;         // CLS: final-normed x (llama2.ts:299)  QKV: k  W13: hb2  WO: xb2  W2: xb (parity reads; may be null)
  float* aux2;        // QKV: v (parity reads; may be null)   CLS: host-mapped pinned logits (zero-copy hand-off) or null
  const float* res;   // WO/W2: residual source x
  const float* fr;    // freq_cis_real
  const float* fi;    // freq_cis_imag
  const int* tokpos;  // {token, pos, step, _}
  int n;              // input length (columns)
  int rows;           // output rows (QKV: 3*dim, W13: hidden (pairs of w1/w3 rows))
  int dim;
  int head_size;
  // tensor parallel: column-sharded WO/W2 write fp64 partials instead of x
  double* partial;    // non-null (WO/W2 of a tensor-parallel rank): out[i] is not written, partial[i] = fp64 sum
};

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ double block_sum(double v, double* red, int tid, int nthreads) {
  v = wave_sum(v);
  const int nw = nthreads >> 6;
  if (nw == 1) return v;
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < nw; ++w) s += red[w];
  return s;
}

__device__ __forceinline__ f4 ldg_nt(const float* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
}

// ------------------------------------------------------------------------------------------------
// One dependency phase of a layer = prologue (stage input, optional rmsnorm) + GEMV rows + epilogue.
// A wave owns R consecutive output rows at a time and strides over the columns 16 B per lane.
template <int MODE, int R, bool VEC>
__global__ void __launch_bounds__(256) phase_kernel(const PhaseArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* xs = reinterpret_cast<float*>(smem);
  double* red = reinterpret_cast<double*>(smem + (((size_t)a.n * 4 + 15) & ~(size_t)15));

  const int tid = threadIdx.x, nthreads = blockDim.x, lane = tid & 63, wave = tid >> 6, nwaves = nthreads >> 6;
  const int n = a.n;
  const int token = a.tokpos[0], pos = a.tokpos[1];

  // ---- prologue: input vector -> LDS (rmsnorm fused, llama2.ts:172-179)
  const float* src = a.in;
  if ((MODE == MODE_QKV) && a.emb) src = a.emb + (size_t)token * n;  // x.set(embedding row), llama2.ts:211
  if (MODE == MODE_QKV || MODE == MODE_W13 || MODE == MODE_CLS) {
    double ss = 0.0;
    for (int j = tid; j < n; j += nthreads) { const double v = src[j]; ss += v * v; }
    ss = block_sum(ss, red, tid, nthreads);
    ss /= (double)n;
    ss = 1.0 / sqrt(1e-5 + ss);
    for (int j = tid; j < n; j += nthreads) {
      const float o = (float)((double)a.rmsw[j] * (ss * (double)src[j]));
      xs[j] = o;
      if (MODE == MODE_CLS && blockIdx.x == 0) a.aux[j] = o;  // rmsnorm(x, x, ...) in place, llama2.ts:299
    }
  } else {
    for (int j = tid; j < n; j += nthreads) xs[j] = src[j];
  }
  __syncthreads();

  // ---- GEMV: groups of R rows per wave
  // QKV: rows [0,3*dim) = q rows, k rows, v rows.  W13: group g = rows [g*R/2, ...) of BOTH w1 and w3.
  const int rows_per_group = (MODE == MODE_W13) ? R / 2 : R;
  const int groups = (a.rows + rows_per_group - 1) / rows_per_group;
  for (int g = blockIdx.x * nwaves + wave; g < groups; g += gridDim.x * nwaves) {
    const float* rp[R];
    int row0 = g * rows_per_group;
    if (MODE == MODE_QKV) {
      const int m = row0 / a.dim, i0 = row0 - m * a.dim;
      const float* base = (m == 0) ? a.w0 : (m == 1) ? a.w1 : a.w2;
#pragma unroll
      for (int r = 0; r < R; ++r) rp[r] = base + (size_t)min(i0 + r, a.dim - 1) * n;
    } else if (MODE == MODE_W13) {
#pragma unroll
      for (int r = 0; r < R / 2; ++r) {
        const size_t ro = (size_t)min(row0 + r, a.rows - 1) * n;
        rp[r] = a.w0 + ro;
        rp[R / 2 + r] = a.w1 + ro;
      }
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) rp[r] = a.w0 + (size_t)min(row0 + r, a.rows - 1) * n;
    }

    double acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0;

    if (VEC) {
      const int n4 = n >> 2;
#pragma unroll 4
      for (int c = lane; c < n4; c += 64) {
        const f4 xv = reinterpret_cast<const f4*>(xs)[c];
        f4 wv[R];
#pragma unroll
        for (int r = 0; r < R; ++r) wv[r] = ldg_nt(rp[r] + 4 * c);
        const double x0 = xv.x, x1 = xv.y, x2 = xv.z, x3 = xv.w;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          acc[r] += (double)wv[r].x * x0;
          acc[r] += (double)wv[r].y * x1;
          acc[r] += (double)wv[r].z * x2;
          acc[r] += (double)wv[r].w * x3;
        }
      }
    } else {
#pragma unroll 4
      for (int c = lane; c < n; c += 64) {
        const double xv = xs[c];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] += (double)__builtin_nontemporal_load(rp[r] + c) * xv;
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]);

    // ---- epilogue (every lane holds every sum; lane p finishes output / pair p)
    if (MODE == MODE_QKV) {
      const int m = row0 / a.dim, i0 = row0 - m * a.dim;
#pragma unroll
      for (int p = 0; p < R / 2; ++p) {
        if (lane == p && i0 + 2 * p < a.dim) {
          const int i = i0 + 2 * p;
          const float s0 = (float)acc[2 * p], s1 = (float)acc[2 * p + 1];  // matmul store, llama2.ts:201
          if (m == 2) {  // v: straight into the cache row (llama2.ts:240)
            float* vc = a.out_v + (size_t)pos * a.dim;
            vc[i] = s0; vc[i + 1] = s1;
            if (a.aux2) { a.aux2[i] = s0; a.aux2[i + 1] = s1; }
          } else {       // RoPE on the adjacent pair (llama2.ts:224-235)
            const int idx = pos * (a.head_size / 2) + (i % a.head_size) / 2;
            const double fcr = a.fr[idx], fci = a.fi[idx];
            const float o0 = (float)((double)s0 * fcr - (double)s1 * fci);
            const float o1 = (float)((double)s0 * fci + (double)s1 * fcr);
            if (m == 0) { a.out[i] = o0; a.out[i + 1] = o1; }
            else {        // k: cache row (llama2.ts:239)
              float* kc = a.out_k + (size_t)pos * a.dim;
              kc[i] = o0; kc[i + 1] = o1;
              if (a.aux) { a.aux[i] = o0; a.aux[i + 1] = o1; }
            }
          }
        }
      }
    } else if (MODE == MODE_W13) {
#pragma unroll
      for (int p = 0; p < R / 2; ++p) {
        if (lane == p && row0 + p < a.rows) {
          const float h1 = (float)acc[p], h3 = (float)acc[R / 2 + p];       // llama2.ts:280-281
          const double v = h1;
          const float s = (float)(v * (1.0 / (1.0 + exp(-v))));              // llama2.ts:285 (store #1)
          a.out[row0 + p] = (float)((double)s * (double)h3);                  // llama2.ts:289 (store #2)
          if (a.aux) a.aux[row0 + p] = h3;
        }
      }
    } else if (MODE == MODE_CLS) {
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (lane == r && row0 + r < a.rows) {
          const float lg = (float)acc[r];                                     // llama2.ts:302
          a.out[row0 + r] = lg;
          if (a.aux2) a.aux2[row0 + r] = lg;   // straight into the host's RunState.logits (pinned, mapped)
        }
    } else {  // WO / W2: matmul store then residual accum (llama2.ts:270-273, 292-295)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (lane == r && row0 + r < a.rows) {
          const int i = row0 + r;
          if (a.partial) {
            a.partial[i] = acc[r];
          } else {
            const float xr = (MODE == MODE_WO && a.emb) ? a.emb[(size_t)token * a.dim + i] : a.res[i];
            const float mv = (float)acc[r];   // xb2 (WO) / xb (W2) as the reference stores it
            a.out[i] = xr + mv;
            if (a.aux) a.aux[i] = mv;
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Multi-head attention for one layer (llama2.ts:244-267): one workgroup per head.
struct AttnArgs {
  const float* q;        // (dim) rotated q
  const float* kc;       // key_cache   + l*S*d
  const float* vc;       // value_cache + l*S*d
  float* att;            // (H, S) scores / probabilities (kept for parity reads)
  float* xb;             // (dim) out
  const int* tokpos;
  int dim, head_size, seq_len;
  int exact;             // 1: fp32-rounded t-sequential value accumulate (llama2.ts:263)
  int lpr;               // lanes per timestep row (power of two >= ceil(head_size / vecw))
};

template <bool VEC>
__global__ void __launch_bounds__(256) attn_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int S = a.seq_len, hs = a.head_size, dim = a.dim;
  float* att = reinterpret_cast<float*>(smem);                                   // S floats
  float* qs = att + ((S + 3) & ~3);                                               // hs floats
  double* red = reinterpret_cast<double*>(qs + ((hs + 3) & ~3));                  // 8 doubles
  double* pacc = red + 8;                                                         // G * hs doubles

  const int tid = threadIdx.x, h = blockIdx.x;
  const int pos = a.tokpos[1];
  constexpr int W = VEC ? 4 : 1;
  const int lpr = a.lpr, sub = tid & (lpr - 1), grp = tid / lpr, G = 256 / lpr;
  const int e0 = sub * W;                       // first element of this lane inside the head
  const bool live = e0 < hs;

  for (int i = tid; i < hs; i += 256) qs[i] = a.q[(size_t)h * hs + i];
  __syncthreads();

  // ---- scores (llama2.ts:249-254)
  double qv[W];
#pragma unroll
  for (int j = 0; j < W; ++j) qv[j] = live ? (double)qs[e0 + j] : 0.0;
  const double rsq = sqrt((double)hs);
  const float* kbase = a.kc + (size_t)h * hs + e0;
  for (int t0 = 0; t0 <= pos; t0 += G) {
    const int t = t0 + grp;
    double p = 0.0;
    if (live && t <= pos) {
      const float* kp = kbase + (size_t)t * dim;
      if (VEC) {
        const f4 kv = *reinterpret_cast<const f4*>(kp);
        p = qv[0] * (double)kv.x;
        p += qv[1] * (double)kv.y;
        p += qv[2] * (double)kv.z;
        p += qv[3] * (double)kv.w;
      } else {
        p = qv[0] * (double)kp[0];
      }
    }
    for (int off = lpr >> 1; off > 0; off >>= 1) p += __shfl_xor(p, off, 64);
    if (sub == 0 && t <= pos) att[t] = (float)(p / rsq);
  }
  __syncthreads();

  // ---- softmax (llama2.ts:181-194)
  float mx = -INFINITY;
  for (int t = tid; t <= pos; t += 256) mx = fmaxf(mx, att[t]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  float* redf = reinterpret_cast<float*>(red);
  if ((tid & 63) == 0) redf[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
  double lsum = 0.0;
  for (int t = tid; t <= pos; t += 256) {
    const float e = (float)exp((double)att[t] - (double)mx);   // stored to fp32 (llama2.ts:187)
    att[t] = e;
    lsum += (double)e;                                          // sum of the ROUNDED values (:190)
  }
  const double sum = block_sum(lsum, red, tid, 256);
  for (int t = tid; t <= pos; t += 256) {
    const float pr = (float)((double)att[t] / sum);             // llama2.ts:192
    att[t] = pr;
    a.att[(size_t)h * S + t] = pr;
  }
  __syncthreads();

  // ---- weighted sum of values (llama2.ts:257-265)
  const float* vbase = a.vc + (size_t)h * hs + e0;
  if (a.exact) {
    // bit-faithful: the accumulator is a Float32Array element, rounded at every timestep, t ascending
    if (grp == 0 && live) {
      float o[W];
#pragma unroll
      for (int j = 0; j < W; ++j) o[j] = 0.0f;
      for (int t = 0; t <= pos; ++t) {
        const double at = att[t];
        const float* vp = vbase + (size_t)t * dim;
        if (VEC) {
          const f4 vv = *reinterpret_cast<const f4*>(vp);
          o[0] = (float)((double)o[0] + at * (double)vv.x);
          o[1] = (float)((double)o[1] + at * (double)vv.y);
          o[2] = (float)((double)o[2] + at * (double)vv.z);
          o[3] = (float)((double)o[3] + at * (double)vv.w);
        } else {
          o[0] = (float)((double)o[0] + at * (double)vp[0]);
        }
      }
#pragma unroll
      for (int j = 0; j < W; ++j) a.xb[(size_t)h * hs + e0 + j] = o[j];
    }
  } else {
    double o[W];
#pragma unroll
    for (int j = 0; j < W; ++j) o[j] = 0.0;
    if (live) {
      for (int t = grp; t <= pos; t += G) {
        const double at = att[t];
        const float* vp = vbase + (size_t)t * dim;
        if (VEC) {
          const f4 vv = *reinterpret_cast<const f4*>(vp);
          o[0] += at * (double)vv.x; o[1] += at * (double)vv.y; o[2] += at * (double)vv.z; o[3] += at * (double)vv.w;
        } else {
          o[0] += at * (double)vp[0];
        }
      }
#pragma unroll
      for (int j = 0; j < W; ++j) pacc[(size_t)grp * hs + e0 + j] = o[j];
    }
    __syncthreads();
    for (int i = tid; i < hs; i += 256) {
      double s = 0.0;
      for (int g2 = 0; g2 < G; ++g2) s += pacc[(size_t)g2 * hs + i];
      a.xb[(size_t)h * hs + i] = (float)s;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// argmax (llama2.ts:364-366: first maximum, strict '>') + advance {token,pos,step}: keeps the greedy
// loop (llama2.ts:465-508 at -t 0) on the device.
__global__ void __launch_bounds__(1024) argmax_advance_kernel(const float* logits, int V, int* tokpos, int* tokens_out) {
  __shared__ float sv[16];
  __shared__ int si[16];
  const int tid = threadIdx.x;
  float bv = -INFINITY; int bi = 0x7fffffff;
  for (int i = tid; i < V; i += 1024) { const float v = logits[i]; if (v > bv) { bv = v; bi = i; } }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(bv, off, 64); const int oi = __shfl_xor(bi, off, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if ((tid & 63) == 0) { sv[tid >> 6] = bv; si[tid >> 6] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 16; ++w) if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
    if (bi == 0x7fffffff) bi = 0;   // all -inf / NaN: reduce() keeps index 0
    const int step = tokpos[2];
    tokens_out[step] = bi;
    tokpos[0] = bi; tokpos[1] = tokpos[1] + 1; tokpos[2] = step + 1;
  }
}

// ------------------------------------------------------------------------------------------------
// Deterministic synthetic weights (same law as oracle/llama2_oracle.c:orc_synth_fill).
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

// One launch fills all layers of one tensor's LOCAL slice: local element (layer, r, c) takes the value of
// global element g0 + layer*full_layer + (row0 + r)*full_cols + col0 + c of the checkpoint's float stream.
struct SynthSlice { uint64_t g0, full_layer, rows, cols, full_cols, row0, col0, n; };

__global__ void synth_fill_kernel(float* out, SynthSlice s, uint32_t seed, float scale, float bias) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint64_t per_layer = s.rows * s.cols;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < s.n; i += stride) {
    const uint64_t layer = i / per_layer, rem = i - layer * per_layer;
    const uint64_t r = rem / s.cols, cc = rem - r * s.cols;
    const uint64_t g = s.g0 + layer * s.full_layer + (s.row0 + r) * s.full_cols + s.col0 + cc;
    const uint32_t lo = (uint32_t)g, hi = (uint32_t)(g >> 32);
    const uint32_t k = hash32(hi ^ (seed * 0x9E3779B9U) ^ 0x85ebca6bU);
    const uint32_t h1 = hash32(lo ^ k);
    const uint32_t h2 = hash32(h1 + 0x9E3779B9U);
    const int c = (int)((h1 & 0xffffU) + (h1 >> 16) + (h2 & 0xffffU) + (h2 >> 16)) - 131070;
    float v;
    {
#pragma clang fp contract(off)   // two fp32 roundings like the host generator: never an FMA
      const float p = (float)c * scale;
      v = bias + p;
    }
    out[i] = v;
  }
}

}  // namespace l2k

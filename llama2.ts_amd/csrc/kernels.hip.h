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

// Hand-off counters sit one per 128-byte line: atomics on one line serialise at ~12 ns each at the memory side
// (6144 adds on one line cost a 7B layer 74 us; spread over 32 lines they overlap).
enum { CTR_STRIDE = 32 };

// Diagnostic build only (-DL2_STAMPS, tools/stamps.py): shader-clock stamps of wave 0 of a few workgroups go
// to a buffer nothing else reads.  In the product build STAMP() is empty.
#ifdef L2_STAMPS
#define L2_NSTAMP 12
#define STAMP(k)                                                                                        \
  do {                                                                                                  \
    if (a.dbg && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2 || blockIdx.x == gridDim.x - 1)) { \
      unsigned long long t_;                                                                            \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                        \
      a.dbg[((blockIdx.x == 0) ? 0 : (blockIdx.x == gridDim.x - 1) ? 2 : 1) * L2_NSTAMP + (k)] = t_;  \
    }                                                                                                   \
  } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

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
  unsigned long long* dbg;  // diagnostic stamps (L2_STAMPS builds), else null
  unsigned* head_done;      // QKV fused with attention: per-head count of finished row groups (else null)
  // chain launch (one kernel per token, phases ordered by block id): dependency + completion counters
  const unsigned* wait_shard; // the previous phase's 16 shard counters (CTR_STRIDE apart); null: no wait
  int wait_blocks;            // workgroups of the previous phase
  unsigned* done_shard;       // 16 shard counters of THIS phase
  int* err;                  // set to 1 when a bounded wait gives up
};

// ------------------------------------------------------------------------------------------------
// Cross-lane reductions on the DPP path (register-to-register, ~10 cycles a step) instead of
// ds_bpermute shuffles (an LDS round trip per step, twice for a 64-bit value): quad swaps, half-row and
// row mirrors leave every lane with its 16-lane row total, row_bcast15 / row_bcast31 fold the four rows
// into lane 63, which is then broadcast through a scalar register.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {   // lanes outside ROW_MASK receive 0
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int nlo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
  const int nhi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(nhi, nlo);
}

__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_f64<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
  v += dpp_f64<0x141, 0xf>(v);   // row_half_mirror
  v += dpp_f64<0x140, 0xf>(v);   // row_mirror
  v += dpp_f64<0x142, 0xa>(v);   // row_bcast15 -> rows 1, 3
  v += dpp_f64<0x143, 0xc>(v);   // row_bcast31 -> rows 2, 3; lane 63 = total
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ float wave_max(float v) {
#define L2_DPP_MAX(CTRL, RM)                                                                              \
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, RM, 0xf, false)))
  L2_DPP_MAX(0xB1, 0xf);
  L2_DPP_MAX(0x4E, 0xf);
  L2_DPP_MAX(0x141, 0xf);
  L2_DPP_MAX(0x140, 0xf);
  L2_DPP_MAX(0x142, 0xa);   // rows not written keep their own value (old = v)
  L2_DPP_MAX(0x143, 0xc);
#undef L2_DPP_MAX
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
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
// One dependency phase of a layer = prologue (stage the input vector in LDS, optional rmsnorm) +
// GEMV over this phase's matrix rows + fused epilogue.
//
// Work split: a wave owns R consecutive output rows ("row group") at a time and walks the columns in
// batches of U x 64 float4 per row, so one batch = R*U independent 16-byte non-temporal loads per lane
// (R*U KiB per wave).  Two register sets (A/B) are filled alternately: batch b+1 is always issued
// BEFORE batch b is consumed, across row-group boundaries too, and the first batch is issued before the
// prologue -- the weight stream never depends on the activations, only the FMAs do.  hipcc turns the
// in-order load queue into counted `s_waitcnt vmcnt(R*U)` waits, so ~2*R*U KiB per wave stay in flight.
// write-through (sc1) stores / L1-bypassing loads: relaxed agent-scope atomics (hand-offs inside a launch)
__device__ __forceinline__ void st_sc1(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_sc1(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_sc1(float* p, float v) {
  __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_sc1(const float* p) {
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}


typedef unsigned u4 __attribute__((ext_vector_type(4)));
// 16-byte L1-bypassing load (buffer_load_dwordx4 ... sc1) of element idx4 of a float4 array of `n4` elements
__device__ __forceinline__ f4 ld16_sc1(const float* base, int idx4, int n4) {
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, n4 * 16, 0x00020000);
  return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, idx4 * 16, 0, 16));
}

// Chain launch hand-off.  A finished workgroup adds 1 (fire-and-forget, no returned value: nothing waits on
// the atomic's round trip) to one of 16 shard counters on separate lines, after its write-through stores have
// drained.  A waiting workgroup polls all 16 shards with one wave instruction (lane s reads shard s) until
// every shard holds its share of the previous phase's workgroups.  Bounded: gives up after ~0.5 s and sets err.
__device__ __forceinline__ void chain_wait(const unsigned* shards, int prev_blocks, int* err, int tid) {
  if (shards) {
    if (tid < 64) {
      const int s16 = tid & 15;
      const unsigned want = (unsigned)((prev_blocks - s16 + 15) / 16);   // workgroups of the previous phase in shard s16
      const unsigned* f = shards + (size_t)s16 * CTR_STRIDE;
      unsigned spins = 0;
      for (;;) {
        const unsigned v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all(v >= want)) break;
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1u << 22)) { if (tid == 0) *err = 1; break; }   // never hang: the host falls back to separate launches
      }
    }
    __syncthreads();
  }
}

__device__ __forceinline__ void chain_signal(unsigned* shard, int vblock, int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave: its write-through stores have reached memory
  __syncthreads();
  if (tid == 0) __hip_atomic_fetch_add(shard + (size_t)(vblock & 15) * CTR_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// QKV row groups are ordered head-major: all q, k and v rows of head 0, then head 1, ... so a head's three
// projections finish together and attention on it can start while later heads are still streaming.
__device__ __forceinline__ void qkv_group(const PhaseArgs& a, int g, int R, int& m, int& i0) {
  if (a.head_done) {   // fused with attention only: head-major (two integer divisions per batch are not free)
    const int per_mat = a.head_size / R, per_head = 3 * per_mat;
    const int head = g / per_head, rem = g - head * per_head;
    m = rem / per_mat;
    i0 = head * a.head_size + (rem - m * per_mat) * R;
  } else {             // all q rows, then k, then v
    const int row0 = g * R;
    m = (row0 >= a.dim) + (row0 >= 2 * a.dim);   // compares, not a division: this runs once per batch
    i0 = row0 - m * a.dim;
  }
}

template <int MODE, int R>
__device__ __forceinline__ void row_ptrs(const PhaseArgs& a, int g, int n, const float* (&rp)[R]) {
  if (MODE == MODE_QKV) {
    int m, i0;
    qkv_group(a, g, R, m, i0);
    const float* base = (m == 0) ? a.w0 : (m == 1) ? a.w1 : a.w2;
#pragma unroll
    for (int r = 0; r < R; ++r) rp[r] = base + (size_t)min(i0 + r, a.dim - 1) * n;
  } else if (MODE == MODE_W13) {
    const int row0 = g * (R / 2);
#pragma unroll
    for (int r = 0; r < R / 2; ++r) {
      const size_t ro = (size_t)min(row0 + r, a.rows - 1) * n;
      rp[r] = a.w0 + ro;
      rp[R / 2 + r] = a.w1 + ro;
    }
  } else {
    const int row0 = g * R;
#pragma unroll
    for (int r = 0; r < R; ++r) rp[r] = a.w0 + (size_t)min(row0 + r, a.rows - 1) * n;
  }
}

// Epilogue of one row group; every lane holds every reduced sum, lane p finishes output / pair p.
template <int MODE, int R, bool CHAIN>
__device__ __forceinline__ void finish_group(const PhaseArgs& a, int g, const double (&acc)[R], int lane, int token, int pos) {
  if (MODE == MODE_QKV) {
    int m, i0;
    qkv_group(a, g, R, m, i0);
    const bool hand = CHAIN || a.head_done != nullptr;   // consumed inside this launch: publish write-through
#pragma unroll
    for (int p = 0; p < R / 2; ++p) {
      if (lane == p && i0 + 2 * p < a.dim) {
        const int i = i0 + 2 * p;
        const float s0 = (float)acc[2 * p], s1 = (float)acc[2 * p + 1];  // matmul store, llama2.ts:201
        if (m == 2) {  // v: straight into the cache row (llama2.ts:240)
          float* vc = a.out_v + (size_t)pos * a.dim;
          vc[i] = s0; vc[i + 1] = s1;
          if (hand) { st_sc1(a.aux2 + i, s0); st_sc1(a.aux2 + i + 1, s1); }
          else if (a.aux2) { a.aux2[i] = s0; a.aux2[i + 1] = s1; }
        } else {       // RoPE on the adjacent pair (llama2.ts:224-235)
          const int idx = pos * (a.head_size / 2) + (i % a.head_size) / 2;
          const double fcr = a.fr[idx], fci = a.fi[idx];
          const float o0 = (float)((double)s0 * fcr - (double)s1 * fci);
          const float o1 = (float)((double)s0 * fci + (double)s1 * fcr);
          if (m == 0) {
            if (hand) { st_sc1(a.out + i, o0); st_sc1(a.out + i + 1, o1); }
            else { a.out[i] = o0; a.out[i + 1] = o1; }
          } else {        // k: cache row (llama2.ts:239)
            float* kc = a.out_k + (size_t)pos * a.dim;
            kc[i] = o0; kc[i + 1] = o1;
            if (hand) { st_sc1(a.aux + i, o0); st_sc1(a.aux + i + 1, o1); }
            else if (a.aux) { a.aux[i] = o0; a.aux[i + 1] = o1; }
          }
        }
      }
    }
    if (a.head_done != nullptr) {   // this wave's stores are out (write-through) before its ticket counts for the head
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_fetch_add(a.head_done + (size_t)(i0 / a.head_size) * CTR_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else if (MODE == MODE_W13) {
    const int row0 = g * (R / 2);
#pragma unroll
    for (int p = 0; p < R / 2; ++p) {
      if (lane == p && row0 + p < a.rows) {
        const float h1 = (float)acc[p], h3 = (float)acc[R / 2 + p];       // llama2.ts:280-281
        const double v = h1;
        const float sl = (float)(v * (1.0 / (1.0 + exp(-v))));            // llama2.ts:285 (store #1)
        const float hv = (float)((double)sl * (double)h3);                  // llama2.ts:289 (store #2)
        if (CHAIN) st_sc1(a.out + row0 + p, hv); else a.out[row0 + p] = hv;
        if (a.aux) a.aux[row0 + p] = h3;
      }
    }
  } else if (MODE == MODE_CLS) {
    const int row0 = g * R;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (lane == r && row0 + r < a.rows) {
        const float lg = (float)acc[r];                                     // llama2.ts:302
        if (CHAIN) st_sc1(a.out + row0 + r, lg); else a.out[row0 + r] = lg;
        if (a.aux2) a.aux2[row0 + r] = lg;   // straight into the host's RunState.logits (pinned, mapped)
      }
    }
  } else {  // WO / W2: matmul store then residual accum (llama2.ts:270-273, 292-295)
    const int row0 = g * R;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (lane == r && row0 + r < a.rows) {
        const int i = row0 + r;
        if (a.partial) {
          a.partial[i] = acc[r];
        } else {
          const float xr = (MODE == MODE_WO && a.emb) ? a.emb[(size_t)token * a.dim + i] : (CHAIN ? ld_sc1(a.res + i) : a.res[i]);
          const float mv = (float)acc[r];   // xb2 (WO) / xb (W2) as the reference stores it
          if (CHAIN) st_sc1(a.out + i, xr + mv); else a.out[i] = xr + mv;
          if (a.aux) a.aux[i] = mv;
        }
      }
    }
  }
}

template <int MODE>
__device__ __forceinline__ constexpr bool mode_has_norm() { return MODE == MODE_QKV || MODE == MODE_W13 || MODE == MODE_CLS; }

// Vector path: n % 4 == 0 (every real checkpoint).  LDS: xs[npad4] float4 (zero padded to whole batches),
// ws[n4] float4 (norm weight, norm modes only), 8 doubles of reduction scratch.
template <int MODE, int R, int U, int PRE, bool CHAIN>
__device__ __forceinline__ void phase_body(const PhaseArgs& a, char* smem, const int vblock, const int vgrid) {
  constexpr int CPI = 64 * U;                       // float4 per row per batch
  const int n = a.n, n4 = n >> 2;
  const int nchunks = (n4 + CPI - 1) / CPI;
  const int npad4 = nchunks * CPI;
  // PRE = float4 per thread per staging round, picked by the host so one round covers the vector
  const int tid = threadIdx.x, nthreads = blockDim.x, lane = tid & 63, wave = tid >> 6, nwaves = nthreads >> 6;
  const int nstage4 = ((npad4 + PRE * nthreads - 1) / (PRE * nthreads)) * (PRE * nthreads);  // whole staging rounds
  f4* xs4 = reinterpret_cast<f4*>(smem);
  f4* ws4 = xs4 + nstage4;
  double* red = reinterpret_cast<double*>(smem + (size_t)(nstage4 * (mode_has_norm<MODE>() ? 2 : 1)) * 16);

  const int rows_per_group = (MODE == MODE_W13) ? R / 2 : R;
  const int groups = (a.rows + rows_per_group - 1) / rows_per_group;
  const int wstride = vgrid * nwaves;

  STAMP(0);
  f4 bufA[R][U], bufB[R][U];
  auto issue = [&](f4 (&buf)[R][U], int gi, int ci) {
    const float* rp[R];
    row_ptrs<MODE, R>(a, gi, n, rp);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = min(ci * CPI + u * 64 + lane, n4 - 1);   // tail lanes re-read the last float4 (x there is 0)
#pragma unroll
      for (int r = 0; r < R; ++r) buf[r][u] = ldg_nt(rp[r] + 4 * c);
    }
  };
  double acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = 0.0;
  auto consume = [&](const f4 (&buf)[R][U], int ci) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const f4 xv = xs4[ci * CPI + u * 64 + lane];
      const double x0 = xv.x, x1 = xv.y, x2 = xv.z, x3 = xv.w;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        acc[r] += (double)buf[r][u].x * x0;
        acc[r] += (double)buf[r][u].y * x1;
        acc[r] += (double)buf[r][u].z * x2;
        acc[r] += (double)buf[r][u].w * x3;
      }
    }
  };
  // the flattened (row group, column batch) sequence of this wave: batch k -> (gk, ck); prefetches past the
  // end are clamped to the last valid batch (unconditional loads keep hipcc's waits counted, never vmcnt(0))
  auto next = [&](int& gi, int& ci, bool& hv) {
    if (++ci == nchunks) { ci = 0; gi += wstride; }
    hv = gi < groups;
  };
  int g0 = vblock * nwaves + wave, c0 = 0;
  bool h0 = g0 < groups;
  int g1 = g0, c1 = c0;
  bool h1 = h0;
  if (h0) next(g1, c1, h1);

  // ---- prologue: input vector -> LDS (rmsnorm fused, llama2.ts:172-179).
  // Vector-memory results return in issue order.  Separate launches: the activations are requested FIRST and
  // two weight batches right behind them, so waiting for x costs one L2 round trip while the weight stream
  // (which depends on nothing) is already in flight.  Chain launch: the two weight batches go out before the
  // wait on the previous phase -- the HBM pipe stays busy across the dependency -- and x is read (L1-bypassing)
  // once the flag is up.
  int token = 0, pos = 0;
  if (MODE == MODE_QKV || MODE == MODE_WO) { token = a.tokpos[0]; pos = a.tokpos[1]; }
  const float* src = a.in;
  if (MODE == MODE_QKV) { if (a.emb) src = a.emb + (size_t)token * n; }
  const f4* src4 = reinterpret_cast<const f4*>(src);
  const f4* rw4 = reinterpret_cast<const f4*>(a.rmsw);
  const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
  double ss = 0.0, ss1 = 0.0, ss2 = 0.0, ss3 = 0.0;   // four chains: fp64 FMA latency is not on the path
  auto stage_load = [&](f4 (&xr)[PRE], f4 (&wr)[PRE], int base) {
#pragma unroll
    for (int k = 0; k < PRE; ++k) {
      const int cc = min(base + tid + k * nthreads, n4 - 1);
      xr[k] = CHAIN ? ld16_sc1(src, cc, n4) : src4[cc];
      if (mode_has_norm<MODE>()) wr[k] = rw4[cc];
    }
  };
  auto stage_store = [&](const f4 (&xr)[PRE], const f4 (&wr)[PRE], int base) {
#pragma unroll
    for (int k = 0; k < PRE; ++k) {
      const int c = base + tid + k * nthreads;   // < nstage4: the LDS arrays hold whole staging rounds, so no
      const f4 xv = (c < n4) ? xr[k] : zero4;    // branch here (a branch lets hipcc sink the load behind the weights)
      xs4[c] = xv;
      if (mode_has_norm<MODE>()) {
        ws4[c] = wr[k];
        ss += (double)xv.x * (double)xv.x;
        ss1 += (double)xv.y * (double)xv.y;
        ss2 += (double)xv.z * (double)xv.z;
        ss3 += (double)xv.w * (double)xv.w;
      }
    }
  };
  if (CHAIN) {
    issue(bufA, h0 ? g0 : groups - 1, h0 ? c0 : 0);
    issue(bufB, h1 ? g1 : (h0 ? g0 : groups - 1), h1 ? c1 : (h0 ? c0 : 0));
    chain_wait(a.wait_shard, a.wait_blocks, a.err, tid);
    f4 xr[PRE], wr[PRE];
    stage_load(xr, wr, 0);
    stage_store(xr, wr, 0);
  } else {
    f4 xr[PRE], wr[PRE];
    stage_load(xr, wr, 0);
    issue(bufA, h0 ? g0 : groups - 1, h0 ? c0 : 0);
    STAMP(1);
    stage_store(xr, wr, 0);
    STAMP(2);
  }
  for (int base = PRE * nthreads; base < npad4; base += PRE * nthreads) {
    f4 xr[PRE], wr[PRE];
    stage_load(xr, wr, base);
    stage_store(xr, wr, base);
  }
  if (mode_has_norm<MODE>()) {
    ss = block_sum((ss + ss1) + (ss2 + ss3), red, tid, nthreads);
    STAMP(3);
    ss /= (double)n;
    ss = 1.0 / sqrt(1e-5 + ss);
    for (int c = tid; c < n4; c += nthreads) {   // same thread, same elements as above: no barrier needed in between
      const f4 xv = xs4[c], wv = ws4[c];
      f4 o;
      o.x = (float)((double)wv.x * (ss * (double)xv.x));
      o.y = (float)((double)wv.y * (ss * (double)xv.y));
      o.z = (float)((double)wv.z * (ss * (double)xv.z));
      o.w = (float)((double)wv.w * (ss * (double)xv.w));
      xs4[c] = o;
      if (MODE == MODE_CLS && vblock == 0) reinterpret_cast<f4*>(a.aux)[c] = o;  // rmsnorm(x, x, ...) in place, llama2.ts:299
    }
  }
  __syncthreads();
  STAMP(4);

  auto finish = [&](int gi) {
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]);
    STAMP(6);
    finish_group<MODE, R, CHAIN>(a, gi, acc, lane, token, pos);
    STAMP(7);
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0;
  };
  if (!CHAIN) {
    // ---- GEMV, double buffered: batch k+1 is issued, then batch k consumed (A holds batch 0 on entry).
    // (Pre-issuing two batches costs ~20 VGPRs and one wave per SIMD of occupancy: measured slower.)
    int g = g0, ch = c0;
    bool have = h0;
    while (have) {
      int g2 = g, ch2 = ch;
      bool have2 = true;
      next(g2, ch2, have2);
      issue(bufB, have2 ? g2 : g, have2 ? ch2 : ch);   // unconditional: keeps the wait counts uniform
      consume(bufA, ch);
      STAMP(5);
      if (ch == nchunks - 1) finish(g);
      if (!have2) break;
      int g3 = g2, ch3 = ch2;
      bool have3 = true;
      next(g3, ch3, have3);
      issue(bufA, have3 ? g3 : g2, have3 ? ch3 : ch2);
      consume(bufB, ch2);
      if (ch2 == nchunks - 1) finish(g2);
      g = g3; ch = ch3; have = have3;
    }
  } else {
    // ---- chain launch: A = batch k, B = batch k+1 are in flight on entry (issued before the dependency wait)
    while (h0) {
      int g2 = g1, c2 = c1;
      bool h2 = h1;
      if (h1) next(g2, c2, h2);
      consume(bufA, c0);
      if (c0 == nchunks - 1) finish(g0);
      if (!h1) break;
      issue(bufA, h2 ? g2 : g1, h2 ? c2 : c1);           // refill A with batch k+2
      int g3 = g2, c3 = c2;
      bool h3 = h2;
      if (h2) next(g3, c3, h3);
      consume(bufB, c1);
      if (c1 == nchunks - 1) finish(g1);
      if (!h2) break;
      issue(bufB, h3 ? g3 : g2, h3 ? c3 : c2);           // refill B with batch k+3
      g0 = g2; c0 = c2; h0 = true;
      g1 = g3; c1 = c3; h1 = h3;
    }
  }
  if (CHAIN) chain_signal(a.done_shard, vblock, tid);
}

template <int MODE, int R, int U, int PRE>
__global__ void __launch_bounds__(256) phase_kernel(const PhaseArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  phase_body<MODE, R, U, PRE, false>(a, smem, blockIdx.x, gridDim.x);
}

// Scalar path for shapes with n % 4 != 0 (rows are not 16-byte aligned): correctness only.
template <int MODE>
__global__ void __launch_bounds__(256) phase_kernel_scalar(const PhaseArgs a) {
  constexpr int R = 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* xs = reinterpret_cast<float*>(smem);
  double* red = reinterpret_cast<double*>(smem + (((size_t)a.n * 4 + 15) & ~(size_t)15));
  const int tid = threadIdx.x, nthreads = blockDim.x, lane = tid & 63, wave = tid >> 6, nwaves = nthreads >> 6;
  const int n = a.n;
  const int token = a.tokpos[0], pos = a.tokpos[1];
  const float* src = a.in;
  if ((MODE == MODE_QKV) && a.emb) src = a.emb + (size_t)token * n;
  if (mode_has_norm<MODE>()) {
    double ss = 0.0;
    for (int j = tid; j < n; j += nthreads) { const double v = src[j]; ss += v * v; }
    ss = block_sum(ss, red, tid, nthreads);
    ss /= (double)n;
    ss = 1.0 / sqrt(1e-5 + ss);
    for (int j = tid; j < n; j += nthreads) {
      const float o = (float)((double)a.rmsw[j] * (ss * (double)src[j]));
      xs[j] = o;
      if (MODE == MODE_CLS && blockIdx.x == 0) a.aux[j] = o;
    }
  } else {
    for (int j = tid; j < n; j += nthreads) xs[j] = src[j];
  }
  __syncthreads();
  const int rows_per_group = (MODE == MODE_W13) ? R / 2 : R;
  const int groups = (a.rows + rows_per_group - 1) / rows_per_group;
  for (int g = blockIdx.x * nwaves + wave; g < groups; g += gridDim.x * nwaves) {
    const float* rp[R];
    row_ptrs<MODE, R>(a, g, n, rp);
    double acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0;
#pragma unroll 4
    for (int c = lane; c < n; c += 64) {
      const double xv = xs[c];
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] += (double)__builtin_nontemporal_load(rp[r] + c) * xv;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]);
    finish_group<MODE, R, false>(a, g, acc, lane, token, pos);
  }
}

// ------------------------------------------------------------------------------------------------
// Multi-head attention for one layer (llama2.ts:244-267).
//
// Two forms share AttnArgs: attn_body (one workgroup per head; keeps every rounding of the reference, and the
// bit-faithful value accumulate when `exact`) and attn_split_body (timesteps split over `nsplit` workgroups
// per head, flash-decode merge by the last arriver -- long contexts).  Either runs as its own kernel or as the
// leading workgroups of qkv_attn_kernel, where it waits on `head_done[h]` for the head's q / k / v rows of
// THIS position to be produced by the GEMV waves of the same launch (`fused`); those rows (q, and the new k and
// v from the k / v scratch vectors, written write-through) are read with L1-bypassing loads into LDS, every
// older cache row was written by earlier launches and is read normally.
struct AttnArgs {
  const float* q;        // (dim) rotated q
  const float* knew;     // (dim) this position's k (RunState.k)   -- same values as cache row `pos`
  const float* vnew;     // (dim) this position's v (RunState.v)
  const float* kc;       // key_cache   + l*S*d
  const float* vc;       // value_cache + l*S*d
  float* att;            // (H, S) scores / probabilities (kept for parity reads)
  float* xb;             // (dim) out
  const int* tokpos;
  double* part;          // split form: [H][nsplit][rec] doubles, rec = round_up(hs + 2, 16)
  unsigned* counter;     // split form: [H] merge tickets, zero between launches
  unsigned* head_done;   // fused form: [H] finished q/k/v row groups of the head, zero between launches
  int* err;              // set to 1 if a bounded wait gives up
  const unsigned* wait_shard;  // chain launch (fused == 2): shard counters of the QKV phase of this layer
  int wait_blocks;
  unsigned long long* dbg;  // diagnostic stamps (L2_STAMPS builds), else null
  unsigned expect;       // fused form: row groups per head (3 * head_size / R)
  int fused;
  int dim, head_size, seq_len, n_heads, nsplit;
  int exact;             // 1: fp32-rounded t-sequential value accumulate (llama2.ts:263)
  int pos_plus1;         // prefill: position of this query + 1 (0: read it from tokpos)
  int lpr;               // lanes per timestep row (power of two >= ceil(head_size / vecw))
  int xb_sc1;            // 1: publish xb write-through (consumed by other workgroups of the same launch: attn_wo_kernel)
};

__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// q.k over one head in fp64.  Four independent chains (one per float4 component): a single accumulator makes
// the sweep a 128-long dependent v_fma_f64 chain (~64 cycles an element measured), which -- not memory -- was
// 60 % of the attention kernel.  Order of fp64 additions differs from the reference's (llama2.ts:252) as in
// every other reduction here; the fp32 score it rounds to does not.
typedef double d2 __attribute__((ext_vector_type(2)));

template <bool VEC, class KP>
__device__ __forceinline__ double head_dot(const double* qd, KP kp, int hs) {   // q is staged widened to fp64
  if (VEC) {
    const d2* q2 = reinterpret_cast<const d2*>(qd);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll 16
    for (int i = 0; i < hs / 4; ++i) {
      const f4 kv = reinterpret_cast<const f4*>(kp)[i];
      const d2 qa = q2[2 * i], qb = q2[2 * i + 1];
      s0 += qa.x * (double)kv.x;
      s1 += qa.y * (double)kv.y;
      s2 += qb.x * (double)kv.z;
      s3 += qb.y * (double)kv.w;
    }
    return (s0 + s1) + (s2 + s3);
  }
  double sc = 0.0;
  for (int i = 0; i < hs; ++i) sc += qd[i] * (double)kp[i];
  return sc;
}

// Wait (bounded) until the GEMV waves of this launch have finished every q/k/v row group of head h, then
// stage q and the new k / v rows in LDS.  Not fused: plain copies, no wait.
__device__ __forceinline__ void attn_stage(const AttnArgs& a, int h, int tid, double* qs, float* kn, float* vn) {
  const int hs = a.head_size;
  if (a.fused == 2) {
    chain_wait(a.wait_shard, a.wait_blocks, a.err, tid);
    for (int i = tid; i < hs; i += 256) {
      qs[i] = (double)ld_sc1(a.q + (size_t)h * hs + i);
      kn[i] = ld_sc1(a.knew + (size_t)h * hs + i);
      vn[i] = ld_sc1(a.vnew + (size_t)h * hs + i);
    }
  } else if (a.fused) {
    if (tid == 0) {
      unsigned spins = 0;
      while (ld_sc1(a.head_done + (size_t)h * CTR_STRIDE) < a.expect) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1u << 22)) { *a.err = 1; break; }   // never hang the GPU: give up, the host reports it
      }
    }
    __syncthreads();
    for (int i = tid; i < hs; i += 256) {
      qs[i] = (double)ld_sc1(a.q + (size_t)h * hs + i);
      kn[i] = ld_sc1(a.knew + (size_t)h * hs + i);
      vn[i] = ld_sc1(a.vnew + (size_t)h * hs + i);
    }
  } else {
    for (int i = tid; i < hs; i += 256) {
      qs[i] = (double)a.q[(size_t)h * hs + i];
      kn[i] = a.knew[(size_t)h * hs + i];
      vn[i] = a.vnew[(size_t)h * hs + i];
    }
  }
  __syncthreads();
}

// One batch of NB passes of the value sweep: all NB row loads are issued before the first use (addresses clamped,
// never predicated: a predicated load makes hipcc serialise the batch behind vmcnt(0) waits).
template <int NB>
__device__ __forceinline__ void value_batch(const float* vbase, int dim, int tb, int G, int grp, int pos, const float* att, double (&o)[4]) {
  f4 vr[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) vr[b] = *reinterpret_cast<const f4*>(vbase + (size_t)min(tb + b * G + grp, pos - 1) * dim);
  double e[4] = {0.0, 0.0, 0.0, 0.0};   // second set of chains for the odd passes
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int t = tb + b * G + grp;
    const double at = (t < pos) ? (double)att[t] : 0.0;
    if (b & 1) { e[0] += at * (double)vr[b].x; e[1] += at * (double)vr[b].y; e[2] += at * (double)vr[b].z; e[3] += at * (double)vr[b].w; }
    else { o[0] += at * (double)vr[b].x; o[1] += at * (double)vr[b].y; o[2] += at * (double)vr[b].z; o[3] += at * (double)vr[b].w; }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] += e[j];
}

template <bool VEC>
__device__ __forceinline__ void attn_body(const AttnArgs& a, char* smem, const int h) {
  const int S = a.seq_len, hs = a.head_size, dim = a.dim, hs4 = (hs + 3) & ~3;
  float* att = reinterpret_cast<float*>(smem);                                   // S floats
  float* kn = att + ((S + 3) & ~3);                                               // hs floats
  float* vn = kn + hs4;
  double* qs = reinterpret_cast<double*>(vn + hs4);                               // hs doubles (q widened once)
  double* red = qs + hs4;                                                         // 8 doubles
  double* pacc = red + 8;                                                         // G * hs doubles

  const int tid = threadIdx.x;
  const int pos = a.pos_plus1 ? a.pos_plus1 - 1 : a.tokpos[1];
  constexpr int W = VEC ? 4 : 1;
  const int lpr = a.lpr, sub = tid & (lpr - 1), grp = tid / lpr, G = 256 / lpr;
  const int e0 = sub * W;                       // first element of this lane inside the head
  const bool live = e0 < hs;

  STAMP(0);
  attn_stage(a, h, tid, qs, kn, vn);
  STAMP(1);
  if (a.fused == 1 && tid == 0) __hip_atomic_store(a.head_done + (size_t)h * CTR_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  // ---- scores (llama2.ts:249-254): one thread per timestep, i ascending in fp64 -- the reference's own
  // summation order, no cross-lane reduction (a lanes-per-row layout with DPP reductions measured slower)
  const double rsq = sqrt((double)hs);
  for (int t = tid; t < pos; t += 256)
    att[t] = (float)(head_dot<VEC>(qs, a.kc + (size_t)t * dim + (size_t)h * hs, hs) / rsq);
  if (tid == (pos & 255)) att[pos] = (float)(head_dot<VEC>(qs, kn, hs) / rsq);
  __syncthreads();
  STAMP(2);

  // ---- softmax (llama2.ts:181-194)
  float mx = -INFINITY;
  for (int t = tid; t <= pos; t += 256) mx = fmaxf(mx, att[t]);
  mx = wave_max(mx);
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
    if (a.att) a.att[(size_t)h * S + t] = pr;
  }
  __syncthreads();
  STAMP(3);

  // ---- weighted sum of values (llama2.ts:257-265)
  const float* vbase = a.vc + (size_t)h * hs + e0;
  if (a.exact) {
    // bit-faithful: the accumulator is a Float32Array element, rounded at every timestep, t ascending
    if (grp == 0 && live) {
      float o[W];
#pragma unroll
      for (int j = 0; j < W; ++j) o[j] = 0.0f;
      for (int t = 0; t < pos; ++t) {
        const double at = att[t];
        const float* vp = vbase + (size_t)t * dim;
#pragma unroll
        for (int j = 0; j < W; ++j) o[j] = (float)((double)o[j] + at * (double)vp[j]);
      }
      {
        const double at = att[pos];
#pragma unroll
        for (int j = 0; j < W; ++j) o[j] = (float)((double)o[j] + at * (double)vn[e0 + j]);
      }
#pragma unroll
      for (int j = 0; j < W; ++j) { if (a.fused == 2 || a.xb_sc1) st_sc1(a.xb + (size_t)h * hs + e0 + j, o[j]); else a.xb[(size_t)h * hs + e0 + j] = o[j]; }
    }
  } else {
    double o[W];
#pragma unroll
    for (int j = 0; j < W; ++j) o[j] = 0.0;
    if (live) {
      if (VEC) {
        double o4[4] = {0.0, 0.0, 0.0, 0.0};
        for (int tb = 0; tb < pos;) {
          const int left = (pos - tb + G - 1) / G;
          if (left > 8) { value_batch<16>(vbase, dim, tb, G, grp, pos, att, o4); tb += 16 * G; }
          else if (left > 4) { value_batch<8>(vbase, dim, tb, G, grp, pos, att, o4); tb += 8 * G; }
          else if (left > 2) { value_batch<4>(vbase, dim, tb, G, grp, pos, att, o4); tb += 4 * G; }
          else { value_batch<2>(vbase, dim, tb, G, grp, pos, att, o4); tb += 2 * G; }
        }
#pragma unroll
        for (int j = 0; j < W; ++j) o[j] = o4[j];
      } else {
        for (int t = grp; t < pos; t += G) o[0] += (double)att[t] * (double)vbase[(size_t)t * dim];
      }
      if (grp == (pos % G)) {
        const double at = att[pos];
#pragma unroll
        for (int j = 0; j < W; ++j) o[j] += at * (double)vn[e0 + j];
      }
#pragma unroll
      for (int j = 0; j < W; ++j) pacc[(size_t)grp * hs + e0 + j] = o[j];
    }
    STAMP(4);
    __syncthreads();
    for (int i = tid; i < hs; i += 256) {
      double sacc = 0.0;
      for (int g2 = 0; g2 < G; ++g2) sacc += pacc[(size_t)g2 * hs + i];
      if (a.fused == 2 || a.xb_sc1) st_sc1(a.xb + (size_t)h * hs + i, (float)sacc); else a.xb[(size_t)h * hs + i] = (float)sacc;
    }
    STAMP(5);
  }
}

template <bool VEC>
__global__ void __launch_bounds__(256) attn_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  attn_body<VEC>(a, smem, blockIdx.x);
}

// Attention with every load up front (pos < 256, head_size = 4 NQ in {64, 128}; the default for those shapes, and the
// attention half of the fused attention + wo launch): EVERY cache byte the head needs is requested in the first few
// hundred cycles of the launch -- threads 0..255 their timestep's K row,
// threads 256..511 up to NQ V rows per NQ-lane group, then q / k / v of this position -- because a moment later the
// other workgroups flood HBM with wo and any load issued after that waits behind 67 MB (measured: the ordinary
// body takes 12 us instead of 6 inside the fused launch).  After the requests the head runs from registers and LDS.
// Same per-thread score order as attn_body (bit-identical scores); value sums are fp64 partials as there.
// KPRE = false: the K side keeps attn_body's way (row loads where the dot uses them, overlapping the arithmetic) and only
// the V rows are requested up front -- for 128-wide heads, whose up-front K request phase is texture-bound.
template <int NQ, bool KPRE = true>
__device__ __forceinline__ void attn_pre_body(const AttnArgs& a, char* smem, const int h) {
  constexpr int hs = 4 * NQ, G = 256 / NQ;                    // V side: G groups of NQ lanes, rows t = grp + G b
  const int S = a.seq_len, dim = a.dim;
  float* att = reinterpret_cast<float*>(smem);               // S floats (same layout as attn_body: attn_lds sizes it)
  float* kn = att + ((S + 3) & ~3);
  float* vn = kn + hs;
  double* qs = reinterpret_cast<double*>(vn + hs);
  double* red = qs + hs;
  double* pacc = red + 8;                                    // G * hs doubles
  const int tid = threadIdx.x;
  const int pos = a.tokpos[1];
  const bool kside = tid < 256;
  const int vt = tid - 256, sub = vt & (NQ - 1), grp = vt / NQ;
  const int last = max(pos - 1, 0);
  STAMP(0);
  f4 r[NQ];                                                   // K side: the timestep's row; V side: rows grp + G b, column sub
  if (kside) {
    if (KPRE) {
      const float* kp = a.kc + (size_t)min(tid, last) * dim + (size_t)h * hs;
#pragma unroll
      for (int i = 0; i < NQ; ++i) r[i] = reinterpret_cast<const f4*>(kp)[i];
    }
  } else {
    const float* vp = a.vc + (size_t)h * hs + 4 * sub;
#pragma unroll
    for (int b = 0; b < NQ; ++b) r[b] = *reinterpret_cast<const f4*>(vp + (size_t)min(grp + G * b, last) * dim);
  }
  for (int i = tid; i < hs; i += 512) {
    qs[i] = (double)a.q[(size_t)h * hs + i];
    kn[i] = a.knew[(size_t)h * hs + i];
    vn[i] = a.vnew[(size_t)h * hs + i];
  }
  __syncthreads();
  STAMP(1);
  const double rsq = sqrt((double)hs);
  if (kside) {                                                // scores (llama2.ts:249-254), same chains as head_dot
    const d2* q2 = reinterpret_cast<const d2*>(qs);
    if (tid < pos) {
      if (KPRE) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
          const d2 qa = q2[2 * i], qb = q2[2 * i + 1];
          s0 += qa.x * (double)r[i].x; s1 += qa.y * (double)r[i].y; s2 += qb.x * (double)r[i].z; s3 += qb.y * (double)r[i].w;
        }
        att[tid] = (float)(((s0 + s1) + (s2 + s3)) / rsq);
      } else {
        att[tid] = (float)(head_dot<true>(qs, a.kc + (size_t)tid * dim + (size_t)h * hs, hs) / rsq);
      }
    }
    if (tid == pos) att[pos] = (float)(head_dot<true>(qs, kn, hs) / rsq);
  }
  __syncthreads();
  STAMP(2);
  // softmax (llama2.ts:181-194) by the K side; the V side only keeps the barriers
  float mx = -INFINITY;
  if (kside && tid <= pos) mx = att[tid];
  mx = wave_max(mx);
  float* redf = reinterpret_cast<float*>(red);
  if (kside && (tid & 63) == 0) redf[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
  double lsum = 0.0;
  if (kside && tid <= pos) {
    const float e = (float)exp((double)att[tid] - (double)mx);    // stored to fp32 (llama2.ts:187)
    att[tid] = e;
    lsum = (double)e;
  }
  __syncthreads();                                            // red is reused below
  lsum = wave_sum(lsum);
  if (kside && (tid & 63) == 0) red[tid >> 6] = lsum;
  __syncthreads();
  const double sum = ((red[0] + red[1]) + red[2]) + red[3];
  if (kside && tid <= pos) {
    const float pr = (float)((double)att[tid] / sum);          // llama2.ts:192
    att[tid] = pr;
    if (a.att) a.att[(size_t)h * S + tid] = pr;
  }
  __syncthreads();
  STAMP(3);
  if (!kside) {                                                // weighted sum of values (llama2.ts:257-265), fp64 partial per group
    double o[4] = {0.0, 0.0, 0.0, 0.0}, e[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int b = 0; b < NQ; ++b) {
      const int t = grp + G * b;
      const double at = (t < pos) ? (double)att[t] : 0.0;
      if (b & 1) { e[0] += at * (double)r[b].x; e[1] += at * (double)r[b].y; e[2] += at * (double)r[b].z; e[3] += at * (double)r[b].w; }
      else { o[0] += at * (double)r[b].x; o[1] += at * (double)r[b].y; o[2] += at * (double)r[b].z; o[3] += at * (double)r[b].w; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] += e[j];
    if (grp == (pos % G)) {
      const double at = att[pos];
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] += at * (double)vn[4 * sub + j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) pacc[(size_t)grp * hs + 4 * sub + j] = o[j];
  }
  STAMP(4);
  __syncthreads();
  for (int i = tid; i < hs; i += 512) {
    double sacc = 0.0;
    for (int g2 = 0; g2 < G; ++g2) sacc += pacc[(size_t)g2 * hs + i];
    if (a.xb_sc1) st_sc1(a.xb + (size_t)h * hs + i, (float)sacc); else a.xb[(size_t)h * hs + i] = (float)sacc;
  }
  STAMP(5);
}

// The same body as its own launch: one 512-thread workgroup per head.
template <int NQ, bool KPRE>
__global__ void __launch_bounds__(512) attn_pre_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  attn_pre_body<NQ, KPRE>(a, smem, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// Attention + wo in ONE launch, with wo resident on chip.
//
// Attention keeps one workgroup per head busy (32 CUs on 7B) for ~7 us and reads almost nothing, then the wo GEMV
// needs ~12 us to stream its 67 MB -- HBM idles through the first and the CUs through the second.  The register
// files of the other CUs (512 KB each) hold far more than wo: here workgroups H .. grid-1 each REQUEST their
// ~18 rows of wo into registers (16 float4 per lane per row) the moment the launch starts, while workgroups
// 0 .. H-1 run the attention; when every head has published its slice of xb (write-through stores, drained, then one
// agent-scope add per head on `ready`), the wo workgroups pull xb into LDS with L1-bypassing loads and finish
// their rows from registers in ~1 us.  Waiters only wait on LOWER block ids (dispatched first), so nothing
// depends on co-residency.  The last wo workgroup through the wait zeroes both counters for the next launch.
struct WoRegArgs {
  const float* w;       // wo[l] (rows, n) row-major
  const float* xb;      // attention output (n floats), written by workgroups 0 .. n_attn-1 of this launch
  const float* res;     // x: residual in (llama2.ts:273)
  const float* emb;     // token embedding table when the residual is still the embedding row (layer 0), else null
  float* out;           // x
  float* aux;           // xb2 as the reference stores it (parity reads), or null
  const int* tokpos;
  unsigned* ready;      // attention workgroups done (own 128-B line)
  unsigned* done;       // wo workgroups past the wait (own line)
  int* err;
  int rows, n, dim, n_attn;
  int delay;            // 1024-cycle sleeps before the weight requests (experiments; 0)
  unsigned long long* dbg;  // diagnostic stamps (L2_STAMPS builds), else null
};

template <int NF4, int MAXR, int NQ>
__global__ void __launch_bounds__(512) attn_wo_kernel(const AttnArgs aa, const WoRegArgs wa) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < wa.n_attn) {
    attn_pre_body<NQ>(aa, smem, blockIdx.x);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's write-through stores are out
    __syncthreads();
    // relaxed: the data went out write-through and is drained; a release here would write back the whole L2
    if (tid == 0) __hip_atomic_fetch_add(wa.ready, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const int b = (int)blockIdx.x - wa.n_attn, nb = (int)gridDim.x - wa.n_attn;
#ifdef L2_STAMPS
#define WSTAMP(k) do { if (wa.dbg && tid == 0 && (b == 0 || b == nb / 2 || b == nb - 1)) { unsigned long long t_; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); wa.dbg[((b == 0) ? 0 : (b == nb - 1) ? 2 : 1) * L2_NSTAMP + (k)] = t_; } } while (0)
#else
#define WSTAMP(k) do { } while (0)
#endif
  WSTAMP(0);
  const int lane = tid & 63, wave = tid >> 6;                // 8 waves: wave w owns rows r0 + w, r0 + w + 8, ...
  const int r0 = (int)(((long long)b * wa.rows) / nb), r1 = (int)(((long long)(b + 1) * wa.rows) / nb);
  const int n4 = wa.n >> 2;
  for (int d_ = 0; d_ < wa.delay; ++d_) __builtin_amdgcn_s_sleep(16);      // 16 * 64 = 1024 cycles per step
  f4 w[MAXR][NF4];
#pragma unroll
  for (int q = 0; q < MAXR; ++q) {
    const float* wr = wa.w + (size_t)min(r0 + wave + 8 * q, wa.rows - 1) * wa.n;      // clamped, never predicated
#pragma unroll
    for (int k = 0; k < NF4; ++k) w[q][k] = ldg_nt(wr + 4 * min(lane + 64 * k, n4 - 1));
  }
  const int token = wa.tokpos[0];
  WSTAMP(1);
#ifdef L2_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // diagnostic only: when did this wave's weights land
  WSTAMP(2);
#endif
  if (tid == 0) {
    unsigned spins = 0;
    // relaxed L1-bypassing polls (an acquire per poll would invalidate this XCD's L2 every iteration); xb is read with
    // L1-bypassing loads afterwards and was never touched by this launch before
    while (ld_sc1(wa.ready) < (unsigned)wa.n_attn) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 22)) { *wa.err = 1; break; }      // never hang the GPU: give up, the host reports it
    }
  }
  __syncthreads();
  WSTAMP(3);
  f4* xs = reinterpret_cast<f4*>(smem);                      // xb, zero padded to NF4 * 64 float4
  const f4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int i = tid; i < NF4 * 64; i += 512) xs[i] = (i < n4) ? ld16_sc1(wa.xb, i, n4) : zero;
  __syncthreads();
  WSTAMP(4);
#pragma unroll
  for (int q = 0; q < MAXR; ++q) {
    const int row = r0 + wave + 8 * q;
    if (row < r1) {                                          // wave-uniform
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
      for (int k = 0; k < NF4; ++k) {
        const f4 xv = xs[lane + 64 * k];                     // zero beyond n: the clamped duplicate weights drop out
        s0 += (double)w[q][k].x * (double)xv.x; s1 += (double)w[q][k].y * (double)xv.y;
        s2 += (double)w[q][k].z * (double)xv.z; s3 += (double)w[q][k].w * (double)xv.w;
      }
      const double acc = wave_sum((s0 + s1) + (s2 + s3));
      if (lane == 0) {
        const float xr = wa.emb ? wa.emb[(size_t)token * wa.dim + row] : wa.res[row];
        const float mv = (float)acc;                         // xb2 as the reference stores it (llama2.ts:270)
        wa.out[row] = xr + mv;                               // llama2.ts:273
        if (wa.aux) wa.aux[row] = mv;
      }
    }
  }
  WSTAMP(5);
  if (tid == 0) {
    const unsigned old = __hip_atomic_fetch_add(wa.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == (unsigned)nb - 1u) {                          // everyone is past the wait: re-arm for the next launch
      __hip_atomic_store(wa.ready, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(wa.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Prefill: grid (head, query).  Query p of the chunk sits at position pos0 + p and sees cache rows 0..pos0+p,
// all written by the chunk's QKV GEMM in an earlier launch.
template <bool VEC>
__global__ void __launch_bounds__(256) pf_attn_kernel(const AttnArgs a, int pos0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  AttnArgs b = a;
  const int p = blockIdx.y, pos = pos0 + p;
  b.q = a.q + (size_t)p * a.dim;
  b.xb = a.xb + (size_t)p * a.dim;
  b.knew = a.kc + (size_t)pos * a.dim;
  b.vnew = a.vc + (size_t)pos * a.dim;
  b.att = nullptr;
  b.pos_plus1 = pos + 1;
  attn_body<VEC>(b, smem, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// Split form (flash-decode): split s scores its slice of 0..pos, keeps a local softmax (max m_s,
// e_t = exp(score - m_s), l_s = sum e_t) and the fp64 value partial acc_s = sum e_t * v_t, publishes
// {acc_s, l_s, m_s} write-through and takes a ticket; the workgroup that draws the last ticket of its head
// merges: out = sum_s w_s acc_s / sum_s w_s l_s, w_s = exp(m_s - max m).  Hand-off per the MI355X guide: one
// lane's agent-scope atomic add after every storing wave drained its sc1 stores and the workgroup barrier; the
// last arriver reads with sc1 loads after a barrier.  Probabilities are not rounded to fp32 before the weighted
// sum here (~1e-7 relative vs the reference): default mode only, never with `exact`.
template <bool VEC>
__device__ __forceinline__ void attn_split_body(const AttnArgs& a, char* smem, const int h, const int sp) {
  const int S = a.seq_len, hs = a.head_size, dim = a.dim, NS = a.nsplit, hs4 = (hs + 3) & ~3;
  const int cmax = (S + NS - 1) / NS;
  float* es = reinterpret_cast<float*>(smem);                                     // cmax floats
  float* kn = es + ((cmax + 3) & ~3);                                             // hs floats
  float* vn = kn + hs4;
  double* qs = reinterpret_cast<double*>(vn + hs4);                               // hs doubles
  double* red = qs + hs4;                                                         // 16 doubles
  double* pacc = red + 16;                                                        // G * hs doubles
  unsigned* ticket = reinterpret_cast<unsigned*>(red + 15);   // all LDS in the one dynamic array (16-byte aligned base)

  const int tid = threadIdx.x;
  const int pos = a.tokpos[1], T = pos + 1;
  const int chunk = (T + NS - 1) / NS;
  const int t0 = sp * chunk, t1 = min(T, t0 + chunk);
  const int tg = min(t1, pos);                  // [t0, tg) comes from the cache, t == pos from LDS
  const bool has_new = (t1 == T) && (t0 <= pos);
  constexpr int W = VEC ? 4 : 1;
  const int lpr = a.lpr, sub = tid & (lpr - 1), grp = tid / lpr, G = 256 / lpr;
  const int e0 = sub * W;
  const bool live = e0 < hs;
  const int rec = (hs + 2 + 15) & ~15;
  double* mypart = a.part + ((size_t)h * NS + sp) * rec;

  attn_stage(a, h, tid, qs, kn, vn);

  // scores of this slice (llama2.ts:249-254), one thread per timestep
  const double rsq = sqrt((double)hs);
  float mx = -INFINITY;
  for (int t = t0 + tid; t < tg; t += 256) {
    const float sf = (float)(head_dot<VEC>(qs, a.kc + (size_t)t * dim + (size_t)h * hs, hs) / rsq);
    es[t - t0] = sf;
    mx = fmaxf(mx, sf);
  }
  if (has_new && tid == ((pos - t0) & 255)) {
    const float sf = (float)(head_dot<VEC>(qs, kn, hs) / rsq);
    es[pos - t0] = sf;
    mx = fmaxf(mx, sf);
  }
  mx = wave_max(mx);
  float* redf = reinterpret_cast<float*>(red);
  if ((tid & 63) == 0) redf[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
  double lsum = 0.0;
  for (int t = t0 + tid; t < t1; t += 256) {
    const float e = (float)exp((double)es[t - t0] - (double)mx);
    es[t - t0] = e;
    st_sc1(a.att + (size_t)h * S + t, e);      // rescaled to probabilities by the merging workgroup
    lsum += (double)e;
  }
  const double l = block_sum(lsum, red + 8, tid, 256);   // (has the barriers that publish es[])

  // value partial of this slice (llama2.ts:257-265)
  const float* vbase = a.vc + (size_t)h * hs + e0;
  double o[W];
#pragma unroll
  for (int j = 0; j < W; ++j) o[j] = 0.0;
  if (live) {
#pragma unroll 4
    for (int t = t0 + grp; t < tg; t += G) {
      const double at = es[t - t0];
      const float* vp = vbase + (size_t)t * dim;
      if (VEC) {
        const f4 vv = *reinterpret_cast<const f4*>(vp);
        o[0] += at * (double)vv.x; o[1] += at * (double)vv.y; o[2] += at * (double)vv.z; o[3] += at * (double)vv.w;
      } else {
        o[0] += at * (double)vp[0];
      }
    }
    if (has_new && grp == ((pos - t0) % G)) {
      const double at = es[pos - t0];
#pragma unroll
      for (int j = 0; j < W; ++j) o[j] += at * (double)vn[e0 + j];
    }
#pragma unroll
    for (int j = 0; j < W; ++j) pacc[(size_t)grp * hs + e0 + j] = o[j];
  }
  __syncthreads();
  for (int i = tid; i < hs; i += 256) {
    double sacc = 0.0;
    for (int g2 = 0; g2 < G; ++g2) sacc += pacc[(size_t)g2 * hs + i];
    st_sc1(mypart + i, sacc);
  }
  if (tid == 0) { st_sc1(mypart + hs, l); st_sc1(mypart + hs + 1, (double)mx); }

  // publish: every storing wave drains its write-through stores, barrier, ONE ticket per workgroup
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) *ticket = __hip_atomic_fetch_add(a.counter + (size_t)h * CTR_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (*ticket != (unsigned)(NS - 1)) return;

  // ---- last arriver of this head: merge the NS partials (sc1 loads: they bypass this CU's L1)
  const double* hp = a.part + (size_t)h * NS * rec;
  double M = -INFINITY;
  for (int s2 = 0; s2 < NS; ++s2) M = fmax(M, ld_sc1(hp + (size_t)s2 * rec + hs + 1));
  double Lsum = 0.0;
  for (int s2 = 0; s2 < NS; ++s2) Lsum += exp(ld_sc1(hp + (size_t)s2 * rec + hs + 1) - M) * ld_sc1(hp + (size_t)s2 * rec + hs);
  for (int i = tid; i < hs; i += 256) {
    double num = 0.0;
    for (int s2 = 0; s2 < NS; ++s2) num += exp(ld_sc1(hp + (size_t)s2 * rec + hs + 1) - M) * ld_sc1(hp + (size_t)s2 * rec + i);
    if (a.fused == 2) st_sc1(a.xb + (size_t)h * hs + i, (float)(num / Lsum)); else a.xb[(size_t)h * hs + i] = (float)(num / Lsum);
  }
  for (int t = tid; t < T; t += 256) {          // probabilities for parity reads of RunState.att
    const double ws = exp(ld_sc1(hp + (size_t)(t / chunk) * rec + hs + 1) - M);
    a.att[(size_t)h * S + t] = (float)((double)ld_sc1(a.att + (size_t)h * S + t) * ws / Lsum);
  }
  if (tid == 0) {   // every split of this head has passed its wait and taken its ticket: re-arm both counters
    __hip_atomic_store(a.counter + (size_t)h * CTR_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.fused == 1) __hip_atomic_store(a.head_done + (size_t)h * CTR_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <bool VEC>
__global__ void __launch_bounds__(256) attn_split_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  attn_split_body<VEC>(a, smem, blockIdx.x, blockIdx.y);
}

// ------------------------------------------------------------------------------------------------
// rmsnorm + q,k,v GEMVs + RoPE + KV store AND the attention of the same layer in ONE launch: workgroups
// [0, nattn) are attention workgroups (head = id % H, split = id / H) that wait per head, the rest stream
// the wq/wk/wv rows head-major and tick head_done[] as each head's rows land.  Attention of head h overlaps
// the weight stream of heads h+1.. and one kernel boundary per layer disappears.  Progress does not depend
// on co-residency: the GEMV workgroups never wait, so queued ones always get the slots they free.
template <int U, int PRE, bool SPLIT>
__global__ void __launch_bounds__(256) qkv_attn_kernel(const PhaseArgs pa, const AttnArgs aa) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nattn = aa.n_heads * (SPLIT ? aa.nsplit : 1);
  if ((int)blockIdx.x < nattn) {
    if (SPLIT) attn_split_body<true>(aa, smem, blockIdx.x % aa.n_heads, blockIdx.x / aa.n_heads);
    else attn_body<true>(aa, smem, blockIdx.x);
  } else {
    phase_body<MODE_QKV, 2, U, PRE, false>(pa, smem, blockIdx.x - nattn, gridDim.x - nattn);
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
  if ((V & 3) == 0) {   // 16-byte loads, all issued before the first compare
    const f4* l4 = reinterpret_cast<const f4*>(logits);
#pragma unroll 8
    for (int c = tid; c < V / 4; c += 1024) {
      const f4 v = l4[c];
      if (v.x > bv) { bv = v.x; bi = 4 * c; }
      if (v.y > bv) { bv = v.y; bi = 4 * c + 1; }
      if (v.z > bv) { bv = v.z; bi = 4 * c + 2; }
      if (v.w > bv) { bv = v.w; bi = 4 * c + 3; }
    }
  } else {
    for (int i = tid; i < V; i += 1024) { const float v = logits[i]; if (v > bv) { bv = v; bi = i; } }
  }
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
// Chain launch: ONE kernel per token.  Workgroups are grouped into phases by block id -- per layer
// [QKV | attention | WO | W13 | W2], then the classifier (and the greedy argmax) -- and a workgroup of phase q
// only ever waits on phase q-1, i.e. on LOWER block ids.  The dispatcher hands workgroups out in block-id
// order, so everything a waiting workgroup depends on is already running or done: no grid barrier, no
// co-residency requirement, and a phase's workgroups start (and stream their first two weight batches) in the
// slots the previous phase's tail frees -- the HBM pipe never drains at a phase boundary.  Every hand-off is
// write-through stores + completion counters + L1-bypassing loads (see chain_signal / chain_wait); waits are
// bounded and set `err` instead of hanging, the host then falls back to one launch per phase.
struct ChainPhase {
  PhaseArgs pa;      // GEMV phases
  AttnArgs aa;       // attention phases
  int mode;          // MODE_* or CHAIN_ATTN / CHAIN_ARGMAX
  int U;             // 2 or 4
  int split;         // attention: 1 = split form
  int nblocks;
};
enum { CHAIN_ATTN = 5, CHAIN_ARGMAX = 6 };

struct ChainLaunch {
  const ChainPhase* phases;   // [5*L + 1 (+1)]
  int L;                      // layers
  int off[6];                 // first block of each role inside a layer, off[5] = blocks per layer
  int cls_first, cls_blocks;  // classifier phase
  int has_argmax;
  // argmax role
  const float* logits; int V; int* tokpos; int* tokens_out;
  unsigned long long* tl;     // diagnostic timeline (L2_STAMPS builds): {start, after-wait, end} per workgroup, 100 MHz ticks
};

template <int MODE>
__device__ __forceinline__ void chain_gemv(const ChainPhase& ph, char* smem, int vb) {
  if (ph.U == 4) phase_body<MODE, 2, 4, 4, true>(ph.pa, smem, vb, ph.nblocks);
  else phase_body<MODE, 2, 2, 4, true>(ph.pa, smem, vb, ph.nblocks);
}

__global__ void __launch_bounds__(256) chain_kernel(const ChainLaunch cl) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int bid = blockIdx.x, tid = threadIdx.x;
  int pidx, vb;
  const int per_layer = cl.off[5];
  if (bid < cl.L * per_layer) {
    const int layer = bid / per_layer, rem = bid - layer * per_layer;
    const int role = (rem >= cl.off[1]) + (rem >= cl.off[2]) + (rem >= cl.off[3]) + (rem >= cl.off[4]);
    pidx = layer * 5 + role;
    vb = rem - cl.off[role];
  } else if (bid < cl.cls_first + cl.cls_blocks) {
    pidx = cl.L * 5;
    vb = bid - cl.cls_first;
  } else {
    pidx = cl.L * 5 + 1;
    vb = 0;
  }
  const ChainPhase& ph = cl.phases[pidx];
#ifdef L2_STAMPS
  if (cl.tl && tid == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); cl.tl[(size_t)bid * 2] = t_; }
#endif
  switch (ph.mode) {
    case MODE_QKV: chain_gemv<MODE_QKV>(ph, smem, vb); break;
    case MODE_WO: chain_gemv<MODE_WO>(ph, smem, vb); break;
    case MODE_W13: chain_gemv<MODE_W13>(ph, smem, vb); break;
    case MODE_W2: chain_gemv<MODE_W2>(ph, smem, vb); break;
    case MODE_CLS: chain_gemv<MODE_CLS>(ph, smem, vb); break;
    case CHAIN_ATTN: {
      const AttnArgs& aa = ph.aa;
      if (ph.split) attn_split_body<true>(aa, smem, vb % aa.n_heads, vb / aa.n_heads);
      else attn_body<true>(aa, smem, vb);
      chain_signal(ph.pa.done_shard, vb, tid);
      break;
    }
    default: {   // greedy argmax (llama2.ts:364-366) + advance {token,pos,step}; logits arrive write-through
      chain_wait(ph.pa.wait_shard, ph.pa.wait_blocks, ph.pa.err, tid);
      float* sv = reinterpret_cast<float*>(smem);
      int* si = reinterpret_cast<int*>(smem) + 16;
      float bv = -INFINITY; int bi = 0x7fffffff;
      for (int i = tid; i < cl.V; i += 256) { const float v = ld_sc1(cl.logits + i); if (v > bv) { bv = v; bi = i; } }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(bv, off, 64); const int oi = __shfl_xor(bi, off, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      }
      if ((tid & 63) == 0) { sv[tid >> 6] = bv; si[tid >> 6] = bi; }
      __syncthreads();
      if (tid == 0) {
        for (int w = 1; w < 4; ++w) if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
        if (bi == 0x7fffffff) bi = 0;
        const int step = cl.tokpos[2];
        cl.tokens_out[step] = bi;
        cl.tokpos[0] = bi; cl.tokpos[1] = cl.tokpos[1] + 1; cl.tokpos[2] = step + 1;
      }
      break;
    }
  }
#ifdef L2_STAMPS
  if (cl.tl && tid == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); cl.tl[(size_t)bid * 2 + 1] = t_; }
#endif
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

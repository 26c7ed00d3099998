// prefill.hip.h -- batched prompt ingestion.
//
// The reference feeds a prompt one token per transformer() call and throws the logits away
// (llama2.ts:471-473), i.e. it streams every weight once per prompt token.  Up to 16 prompt positions share
// the weights here: each phase becomes a (16 tokens) x (rows) x (n) GEMM, the first true dense contraction on
// this path, and runs on the matrix cores: v_mfma_f64_16x16x4_f64 with the fp32 operands widened to fp64, so the
// numeric contract is unchanged (exact products, fp64 accumulation, one fp32 rounding per stored element;
// SURVEY.md 8(a-N)).  At 16 tokens the fp64 MFMA rate (16 B of weights per clock per CU) is about the HBM rate,
// so a 16-token chunk costs about one decode step instead of sixteen.
//
// Tile orientation: M = tokens (A operand = activations), N = 16 output rows (B operand = weight rows), K = n.
// Operand lanes (MI355X guide, f64 MFMA): A[i = l&15][k = l>>4], B[k = l>>4][j = l&15]; results
// C[row i = (l>>4) + 4*reg][col j = l&15].  Every lane loads one float4 of its activation row and one of its
// weight row per 16-column block; the four MFMAs of a block take element 0..3 of both, so A and B see the same
// (permuted) k order.  The NW waves of a workgroup take the 16-column blocks round-robin (split K) and their
// partial tiles are added in a fixed order through LDS.  First version: a 16-token chunk of the 7B shape takes
// 8.6 ms (9x faster than sixteen decode steps, about 1.8x one decode step; tools/prefill_bench.py).
#pragma once
#include "kernels.hip.h"

namespace l2k {

typedef double d4 __attribute__((ext_vector_type(4)));

enum { PF_T = 64, PF_S = 4 };   // tokens per chunk: one, two or four MFMA tiles of 16; chunks per launch of the register-blocked GEMMs (blockIdx.y)

struct PfArgs {
  const float* w0;     // QKV: wq  W13: w1  else the matrix
  const float* w1;     // QKV: wk  W13: w3
  const float* w2;     // QKV: wv
  const float* xin;    // [16][n] activations (normed x, attention output, or hb)
  float* x;            // [16][dim] residual stream (WO / W2: updated in place)
  float* out;          // QKV: q [16][dim]   W13: hb [16][rows]
  float* kc; float* vc;        // QKV: cache slabs of this layer
  const float* fr; const float* fi;
  int n, rows, dim, head_size;
  int pos0, nvalid;    // first position of the chunk, tokens in it (<= PF_T)
  // One copy of the weights: when the decode step's repacked copy of this phase's matrices exists (kernels.hip.h, pack_kernel) the
  // row-major tensors are gone (w0 = w1 = w2 = null) and the GEMMs read the repacked layout instead:
  const float* wp;     // [round][column batch][place in the round][row of the group][u][lane] float4 of this layer, or null
  int pk_wstride;      // row groups per round (waves of the decode launch's grid)
  int pk_groups;       // row groups of the phase (2 rows each; w1 / w3: row g of both)
};

// Where float4 `c4` of row `r` of row group `g` lies in the repacked copy, split into the part that depends on the lane's row only
// (two variants: ordinary column batches hold two 64-float4 sub-batches per row, a short last batch `ulast`) and the part that
// depends on the column batch: offset = (last batch ? base_last : base) + nwr256 * ci + (u * 64 + ln), all in float4.
struct PkLane { unsigned base, base_last, nwr256; };
__device__ __forceinline__ PkLane pk_lane(const PfArgs& a, int g, int r) {
  const int n4 = a.n >> 2, ws = a.pk_wstride;
  const int rnd = g / ws, place = g - rnd * ws, nwr = min(ws, a.pk_groups - rnd * ws);
  const int nchunks = (n4 + 127) >> 7, ulast = (n4 - (nchunks - 1) * 128) >> 6;
  PkLane p;
  const unsigned round0 = (unsigned)rnd * (unsigned)ws * 2u * (unsigned)n4;
  p.base = round0 + (unsigned)place * 256u + (unsigned)r * 128u;
  p.base_last = round0 + (unsigned)place * 128u * (unsigned)ulast + (unsigned)r * 64u * (unsigned)ulast;
  p.nwr256 = (unsigned)nwr * 256u;
  return p;
}
// row group and row-in-group of output row `row` (an index into the phase's rows: q | k | v stacked for QKV) of stream `second`
// (w3 of the w1 / w3 phase)
template <int MODE>
__device__ __forceinline__ void pk_group_of(int row, bool second, int& g, int& r) {
  if (MODE == MODE_W13) { g = row; r = second ? 1 : 0; }
  else { g = row >> 1; r = row & 1; }
}

__global__ void __launch_bounds__(256) pf_embed_kernel(float* x, const float* emb, const int* tokens, int dim, int nvalid) {
  const int t = blockIdx.x;
  for (int i = threadIdx.x; i < dim; i += 256) x[(size_t)t * dim + i] = (t < nvalid) ? emb[(size_t)tokens[t] * dim + i] : 0.0f;
}

// rmsnorm of every token row (llama2.ts:172-179): one workgroup per token
__global__ void __launch_bounds__(256) pf_norm_kernel(float* xn, const float* x, const float* w, int dim) {
  __shared__ double red[8];
  const int t = blockIdx.x, tid = threadIdx.x;
  const float* xr = x + (size_t)t * dim;
  constexpr int MAXE = 16;                       // the row stays in registers up to dim = 4096; longer rows re-read the tail
  float v[MAXE], g[MAXE];
  double ss = 0.0;
#pragma unroll
  for (int e = 0; e < MAXE; ++e) {                 // clamped, never predicated: a predicated load serialises the batch
    const int j = min(tid + 256 * e, dim - 1);
    v[e] = xr[j];
    g[e] = w[j];
  }
#pragma unroll
  for (int e = 0; e < MAXE; ++e) ss += (tid + 256 * e < dim) ? (double)v[e] * (double)v[e] : 0.0;
  for (int j = tid + 256 * MAXE; j < dim; j += 256) { const double u = xr[j]; ss += u * u; }
  ss = block_sum(ss, red, tid, 256);
  ss /= (double)dim;
  ss = 1.0 / sqrt(1e-5 + ss);
#pragma unroll
  for (int e = 0; e < MAXE; ++e) {
    const int j = tid + 256 * e;
    if (j < dim) xn[(size_t)t * dim + j] = (float)((double)g[e] * (ss * (double)v[e]));
  }
  for (int j = tid + 256 * MAXE; j < dim; j += 256) xn[(size_t)t * dim + j] = (float)((double)w[j] * (ss * (double)xr[j]));
}

// Epilogue of one 16-row x 16-token tile: the lane holds four tokens of output index i -- t = toff + kq + 4r (r = 0..3) in the f64 MFMA's
// result layout, t = toff + 4 kq + r in the f32 MFMA's (F32: the opt-in fp32-accumulate GEMMs, pf_gemm3_kernel<.., true>).
template <int MODE, bool F32 = false, class V = d4>
__device__ __forceinline__ void pf_emit(const PfArgs& a, const V& av, const V& acc3, int m, int i, int j, int kq, int toff) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int t = F32 ? toff + 4 * kq + r : toff + kq + 4 * r;
    const float sv = (float)av[r];                                  // matmul store (llama2.ts:201)
    if (MODE == MODE_QKV) {
      const int pos = a.pos0 + t;
      if (m == 2) {
        if (t < a.nvalid) a.vc[(size_t)pos * a.dim + i] = sv;         // llama2.ts:240
      } else {
        // RoPE pair (i even, i+1) sits in adjacent lanes (llama2.ts:224-235)
        const float other = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sv), 0xB1, 0xf, 0xf, false));
        const float s0 = (j & 1) ? other : sv, s1 = (j & 1) ? sv : other;
        const int idx = pos * (a.head_size / 2) + (i % a.head_size) / 2;
        const int cidx = (t < a.nvalid) ? idx : 0;
        const double fcr = a.fr[cidx], fci = a.fi[cidx];
        const float o = (j & 1) ? (float)((double)s0 * fci + (double)s1 * fcr) : (float)((double)s0 * fcr - (double)s1 * fci);
        if (t < a.nvalid) {
          if (m == 0) a.out[(size_t)t * a.dim + i] = o;
          else a.kc[(size_t)pos * a.dim + i] = o;                      // llama2.ts:239
        }
      }
    } else if (MODE == MODE_W13) {
      const float h1 = sv, h3 = (float)acc3[r];
      const double v = h1;
      const float sl = (float)(v * (1.0 / (1.0 + exp(-v))));          // llama2.ts:285
      a.out[(size_t)t * a.rows + i] = (float)((double)sl * (double)h3);  // llama2.ts:289
    } else {   // WO / W2: residual accum (llama2.ts:273, 295)
      float* xp = a.x + (size_t)t * a.dim + i;
      *xp = *xp + sv;
    }
  }
}

// Accumulators of one workgroup: TT token tiles x (1 or 2 weight streams) x two chains (even / odd k steps: a
// dependent MFMA cannot issue back to back).
template <int TT, bool DUAL>
struct PfAcc {
  d4 a[TT][2], c[TT][2];                               // a: first stream (w0 / w1), c: second stream (w3)
  __device__ __forceinline__ void clear() {
    const d4 z = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < TT; ++t) { a[t][0] = z; a[t][1] = z; c[t][0] = z; c[t][1] = z; }
  }
  // one 16-column block: weight fragment w (and w3), activation fragments x[t] of the TT token tiles
  __device__ __forceinline__ void block(const f4& w, const f4& w3, const f4 (&x)[TT]) {
    const double w0 = w.x, w1 = w.y, w2 = w.z, w3d = w.w;
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
    if (DUAL) { v0 = w3.x; v1 = w3.y; v2 = w3.z; v3 = w3.w; }
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      const double x0 = x[t].x, x1 = x[t].y, x2 = x[t].z, x3 = x[t].w;
      a[t][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, w0, a[t][0], 0, 0, 0);
      a[t][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, w1, a[t][1], 0, 0, 0);
      if (DUAL) {
        c[t][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, v0, c[t][0], 0, 0, 0);
        c[t][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, v1, c[t][1], 0, 0, 0);
      }
      a[t][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x2, w2, a[t][0], 0, 0, 0);
      a[t][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x3, w3d, a[t][1], 0, 0, 0);
      if (DUAL) {
        c[t][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x2, v2, c[t][0], 0, 0, 0);
        c[t][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x3, v3, c[t][1], 0, 0, 0);
      }
    }
  }
  // chains folded, split-K partials of waves 1..NW-1 added in wave order into wave 0 (part: LDS, [TT][2][NW-1][4][64])
  template <int NW>
  __device__ __forceinline__ bool combine(double* part, int wave, int lane) {
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) { a[t][0][r] += a[t][1][r]; c[t][0][r] += c[t][1][r]; }
    if (NW == 1) return true;
    if (wave > 0) {
#pragma unroll
      for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          part[(((t * 2 + 0) * (NW - 1) + wave - 1) * 4 + r) * 64 + lane] = a[t][0][r];
          if (DUAL) part[(((t * 2 + 1) * (NW - 1) + wave - 1) * 4 + r) * 64 + lane] = c[t][0][r];
        }
    }
    __syncthreads();
    if (wave != 0) return false;
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int w = 0; w < NW - 1; ++w) {
          a[t][0][r] += part[(((t * 2 + 0) * (NW - 1) + w) * 4 + r) * 64 + lane];
          if (DUAL) c[t][0][r] += part[(((t * 2 + 1) * (NW - 1) + w) * 4 + r) * 64 + lane];
        }
    return true;
  }
};

// One 16-row weight tile per workgroup, NW waves split K, TT tiles of 16 tokens share every weight fragment (the
// chunk is 16 or 32 tokens: with 32 each streamed weight byte feeds two MFMAs and is widened once).  Weight loads in
// MFMA operand layout.  Measured (7B shapes): NW = 4 is best everywhere (1-2 waves starve wo / w2, 8-16 lose to the
// combine); two 16-row tiles per workgroup and 8-block register sets were slower (fewer workgroups / occupancy).
template <int MODE, int NW, int TT>
__global__ void __launch_bounds__(64 * NW) pf_gemm_kernel(const PfArgs a) {
  constexpr bool DUAL = (MODE == MODE_W13);          // w1 and w3 on the same activations
  constexpr int UN = (DUAL && TT == 4) ? 2 : 4;      // 16-column blocks per batch (two register sets); W13 x 4 token tiles: 2 (18.3 -> 17.9 ms per 64 tokens; 1: 19.4)
  __shared__ double part[TT * 2 * (NW > 1 ? NW - 1 : 1) * 4 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = a.n, nblk = n >> 4;                 // 16-column blocks
  const int row0 = blockIdx.x * 16;
  int m = 0, i0 = row0;
  const float* wbase = a.w0;
  if (MODE == MODE_QKV) { m = row0 / a.dim; i0 = row0 - m * a.dim; wbase = (m == 0) ? a.w0 : (m == 1) ? a.w1 : a.w2; }
  const int j = lane & 15, kq = lane >> 4;
  const bool pk = a.wp != nullptr;
  const float* wrow = pk ? nullptr : wbase + (size_t)(i0 + j) * n + 4 * kq;
  const float* wrow3 = (DUAL && !pk) ? a.w1 + (size_t)(i0 + j) * n + 4 * kq : nullptr;
  PkLane pl = {0, 0, 0}, pl3 = {0, 0, 0};
  if (pk) {
    int g, r;
    pk_group_of<MODE>(row0 + j, false, g, r); pl = pk_lane(a, g, r);
    if (DUAL) { pk_group_of<MODE>(row0 + j, true, g, r); pl3 = pk_lane(a, g, r); }
  }
  const int pk_lastci = (((n >> 2) + 127) >> 7) - 1;
  const f4* wp4 = reinterpret_cast<const f4*>(a.wp);
  auto pk_at = [&](const PkLane& q, int sb) -> const f4* {      // 16-column block sb of the lane's row: float4 4 sb + kq
    const int ci = sb >> 5;
    return wp4 + ((ci == pk_lastci ? q.base_last : q.base) + q.nwr256 * (unsigned)ci + (unsigned)(((sb >> 4) & 1) * 64 + 4 * (sb & 15) + kq));
  };
  const float* xrow = a.xin + (size_t)j * n + 4 * kq;       // token j (and 16 + j) as the A row
  PfAcc<TT, DUAL> acc;
  acc.clear();
  // a wave takes UN ADJACENT 16-column blocks per batch (256 contiguous bytes of every weight row), batches round-robin
  // over the NW waves; batch b+1 loads while batch b is on the matrix pipe
  const int npair = (nblk + UN - 1) / UN;
  struct Batch { f4 wv[UN], w3[DUAL ? UN : 1], xv[UN][TT]; };
  auto load = [&](Batch& b, int p0) {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int sb = min(p0 * UN + u, nblk - 1);     // clamped (never predicated) loads; masked in mma()
      if (pk) {
        b.wv[u] = __builtin_nontemporal_load(pk_at(pl, sb));
        if (DUAL) b.w3[u] = __builtin_nontemporal_load(pk_at(pl3, sb));
      } else {
        b.wv[u] = ldg_nt(wrow + 16 * sb);
        if (DUAL) b.w3[u] = ldg_nt(wrow3 + 16 * sb);
      }
#pragma unroll
      for (int t = 0; t < TT; ++t) b.xv[u][t] = *reinterpret_cast<const f4*>(xrow + (size_t)16 * t * n + 16 * sb);
    }
  };
  auto mma = [&](const Batch& b, int p0) {
#pragma unroll
    for (int u = 0; u < UN; ++u)
      if (p0 * UN + u < nblk) acc.block(b.wv[u], b.w3[DUAL ? u : 0], b.xv[u]);
  };
  Batch A, B;
  int p0 = wave;
  if (p0 < npair) load(A, p0);
  while (p0 < npair) {
    const int p1 = p0 + NW;
    load(B, p1 < npair ? p1 : p0);                   // unconditional (clamped) prefetch keeps the waits counted
    mma(A, p0);
    if (p1 >= npair) break;
    const int p2 = p1 + NW;
    load(A, p2 < npair ? p2 : p1);
    mma(B, p1);
    p0 = p2;
  }
  if (!acc.template combine<NW>(part, wave, lane)) return;
#pragma unroll
  for (int t = 0; t < TT; ++t) pf_emit<MODE>(a, acc.a[t][0], acc.c[t][0], m, i0 + j, j, kq, 16 * t);
}

// (An LDS-tile variant of this kernel -- the weight chunk requested as 2 rows x 512 contiguous bytes per instruction and re-read in MFMA
// layout from a wave-private tile: QKV 60.6 -> 54.5 us, W2 56.5 -> 49.7 at 7B width -- shipped until round 4 for chunks of at most 32
// tokens over row-major tensors; with the register-blocked form below taking every 64-token chunk and the repacked copies not
// row-shaped, it served short chunks of the small models only and was removed: profiles/r05/pruned_instances.txt.)

// ------------------------------------------------------------------------------------------------
// Register-blocked form.  On gfx950 v_mfma_f64_16x16x4_f64 IS the fp64 vector pipe (fp64 matrix peak = fp64 vector peak, 64
// cycles per instruction and SIMD), so nothing a wave issues on the vector ALU overlaps it: every v_cvt_f64_f32 and every
// address instruction adds to the 64 cycles of each MFMA, and the operand-shaped activation fragments (16 rows x 64 B per load
// instruction, re-read from L2 by every wave) compete for the CU's 64 B / clock vector-memory path.  Per workgroup output block
// of R rows x T tokens the conversions per MFMA are 16 (R + T) / (R T) -- 1.25 for the 16 x 64 block of pf_gemm_kernel, 0.58 for
// 48 x 64 -- and the fragment bytes per MFMA fall the same way.  Here a wave holds RT row tiles x TT token tiles (RT (x 2 for
// w1 / w3) x TT accumulators in AGPRs), takes all operands through buffer loads whose per-block / per-tile offsets are SCALAR
// (no vector address arithmetic, no exec masks: every wave runs whole batches), and the NW waves of a workgroup split K; their
// partial tiles are added in wave order through LDS, one weight stream at a time (64 KB at NW = 8), by NW waves in parallel.
// Requires n % 32 == 0 (whole batches of two blocks) and (rows / 16) % RT == 0; other shapes take pf_gemm_kernel.
// One k-step of a wave's NS x 4 output tiles as ONE asm statement: acc[s][t] += x[t] (A operand, 16 tokens) * w[s] (B operand, 16 rows).
// The accumulators are pinned in AGPRs ("+a"): left to hipcc, the loop-carried tiles live in VGPRs and are copied to AGPRs and
// back around every iteration (64 - 96 v_accvgpr_write per 128 - 192 MFMAs -- on the very pipe the MFMAs need).  What hipcc does
// not do for an asm statement (MI355X guide 5.7) is done by hand: `s_nop 1` in front covers the two wait states between a VALU
// write of an operand (the conversions, the zero-fill of the tiles) and the MFMA that reads it; consecutive MFMAs of a statement
// touch different tiles, and a tile is next written NS x 4 MFMAs later.
#define L2_MF "v_mfma_f64_16x16x4_f64 "
__device__ __forceinline__ void mfma_group(d4 (&c)[1][4], double x0, double x1, double x2, double x3, const double (&w)[1]) {
  asm volatile("s_nop 1\n\t" L2_MF "%0, %4, %8, %0\n\t" L2_MF "%1, %5, %8, %1\n\t" L2_MF "%2, %6, %8, %2\n\t" L2_MF "%3, %7, %8, %3"
               : "+a"(c[0][0]), "+a"(c[0][1]), "+a"(c[0][2]), "+a"(c[0][3]) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(w[0]));
}
__device__ __forceinline__ void mfma_group(d4 (&c)[2][4], double x0, double x1, double x2, double x3, const double (&w)[2]) {
  asm volatile("s_nop 1\n\t" L2_MF "%0, %8, %12, %0\n\t" L2_MF "%1, %9, %12, %1\n\t" L2_MF "%2, %10, %12, %2\n\t" L2_MF "%3, %11, %12, %3\n\t"
               L2_MF "%4, %8, %13, %4\n\t" L2_MF "%5, %9, %13, %5\n\t" L2_MF "%6, %10, %13, %6\n\t" L2_MF "%7, %11, %13, %7"
               : "+a"(c[0][0]), "+a"(c[0][1]), "+a"(c[0][2]), "+a"(c[0][3]), "+a"(c[1][0]), "+a"(c[1][1]), "+a"(c[1][2]), "+a"(c[1][3])
               : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(w[0]), "v"(w[1]));
}
__device__ __forceinline__ void mfma_group(d4 (&c)[3][4], double x0, double x1, double x2, double x3, const double (&w)[3]) {
  asm volatile("s_nop 1\n\t" L2_MF "%0, %12, %16, %0\n\t" L2_MF "%1, %13, %16, %1\n\t" L2_MF "%2, %14, %16, %2\n\t" L2_MF "%3, %15, %16, %3\n\t"
               L2_MF "%4, %12, %17, %4\n\t" L2_MF "%5, %13, %17, %5\n\t" L2_MF "%6, %14, %17, %6\n\t" L2_MF "%7, %15, %17, %7\n\t"
               L2_MF "%8, %12, %18, %8\n\t" L2_MF "%9, %13, %18, %9\n\t" L2_MF "%10, %14, %18, %10\n\t" L2_MF "%11, %15, %18, %11"
               : "+a"(c[0][0]), "+a"(c[0][1]), "+a"(c[0][2]), "+a"(c[0][3]), "+a"(c[1][0]), "+a"(c[1][1]), "+a"(c[1][2]), "+a"(c[1][3]),
                 "+a"(c[2][0]), "+a"(c[2][1]), "+a"(c[2][2]), "+a"(c[2][3])
               : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(w[0]), "v"(w[1]), "v"(w[2]));
}
__device__ __forceinline__ void mfma_group(d4 (&c)[4][4], double x0, double x1, double x2, double x3, const double (&w)[4]) {
  asm volatile("s_nop 1\n\t" L2_MF "%0, %16, %20, %0\n\t" L2_MF "%1, %17, %20, %1\n\t" L2_MF "%2, %18, %20, %2\n\t" L2_MF "%3, %19, %20, %3\n\t"
               L2_MF "%4, %16, %21, %4\n\t" L2_MF "%5, %17, %21, %5\n\t" L2_MF "%6, %18, %21, %6\n\t" L2_MF "%7, %19, %21, %7\n\t"
               L2_MF "%8, %16, %22, %8\n\t" L2_MF "%9, %17, %22, %9\n\t" L2_MF "%10, %18, %22, %10\n\t" L2_MF "%11, %19, %22, %11\n\t"
               L2_MF "%12, %16, %23, %12\n\t" L2_MF "%13, %17, %23, %13\n\t" L2_MF "%14, %18, %23, %14\n\t" L2_MF "%15, %19, %23, %15"
               : "+a"(c[0][0]), "+a"(c[0][1]), "+a"(c[0][2]), "+a"(c[0][3]), "+a"(c[1][0]), "+a"(c[1][1]), "+a"(c[1][2]), "+a"(c[1][3]),
                 "+a"(c[2][0]), "+a"(c[2][1]), "+a"(c[2][2]), "+a"(c[2][3]), "+a"(c[3][0]), "+a"(c[3][1]), "+a"(c[3][2]), "+a"(c[3][3])
               : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]));
}
#undef L2_MF
// The same k-step on v_mfma_f32_16x16x4_f32 (L2_OPT_PREFILL_F32_MFMA, opt-in): fp32 operands as loaded, fp32 accumulate -- bit for bit a
// k-ordered fmaf chain per element (MI355X guide), NOT the reference's fp64 accumulate; 32 cycles per instruction against 64, no
// widening conversions on the vector pipe.  Same A / B operand lanes as the f64 form; results: row = 4 (lane >> 4) + register.
#define L2_MF "v_mfma_f32_16x16x4_f32 "
__device__ __forceinline__ void mfma_group(f4 (&c)[1][4], float x0, float x1, float x2, float x3, const float (&w)[1]) {
  asm volatile("s_nop 1\n\t" L2_MF "%0, %4, %8, %0\n\t" L2_MF "%1, %5, %8, %1\n\t" L2_MF "%2, %6, %8, %2\n\t" L2_MF "%3, %7, %8, %3"
               : "+a"(c[0][0]), "+a"(c[0][1]), "+a"(c[0][2]), "+a"(c[0][3]) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(w[0]));
}
__device__ __forceinline__ void mfma_group(f4 (&c)[2][4], float x0, float x1, float x2, float x3, const float (&w)[2]) {
  asm volatile("s_nop 1\n\t" L2_MF "%0, %8, %12, %0\n\t" L2_MF "%1, %9, %12, %1\n\t" L2_MF "%2, %10, %12, %2\n\t" L2_MF "%3, %11, %12, %3\n\t"
               L2_MF "%4, %8, %13, %4\n\t" L2_MF "%5, %9, %13, %5\n\t" L2_MF "%6, %10, %13, %6\n\t" L2_MF "%7, %11, %13, %7"
               : "+a"(c[0][0]), "+a"(c[0][1]), "+a"(c[0][2]), "+a"(c[0][3]), "+a"(c[1][0]), "+a"(c[1][1]), "+a"(c[1][2]), "+a"(c[1][3])
               : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(w[0]), "v"(w[1]));
}
__device__ __forceinline__ void mfma_group(f4 (&c)[3][4], float x0, float x1, float x2, float x3, const float (&w)[3]) {
  asm volatile("s_nop 1\n\t" L2_MF "%0, %12, %16, %0\n\t" L2_MF "%1, %13, %16, %1\n\t" L2_MF "%2, %14, %16, %2\n\t" L2_MF "%3, %15, %16, %3\n\t"
               L2_MF "%4, %12, %17, %4\n\t" L2_MF "%5, %13, %17, %5\n\t" L2_MF "%6, %14, %17, %6\n\t" L2_MF "%7, %15, %17, %7\n\t"
               L2_MF "%8, %12, %18, %8\n\t" L2_MF "%9, %13, %18, %9\n\t" L2_MF "%10, %14, %18, %10\n\t" L2_MF "%11, %15, %18, %11"
               : "+a"(c[0][0]), "+a"(c[0][1]), "+a"(c[0][2]), "+a"(c[0][3]), "+a"(c[1][0]), "+a"(c[1][1]), "+a"(c[1][2]), "+a"(c[1][3]),
                 "+a"(c[2][0]), "+a"(c[2][1]), "+a"(c[2][2]), "+a"(c[2][3])
               : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(w[0]), "v"(w[1]), "v"(w[2]));
}
__device__ __forceinline__ void mfma_group(f4 (&c)[4][4], float x0, float x1, float x2, float x3, const float (&w)[4]) {
  asm volatile("s_nop 1\n\t" L2_MF "%0, %16, %20, %0\n\t" L2_MF "%1, %17, %20, %1\n\t" L2_MF "%2, %18, %20, %2\n\t" L2_MF "%3, %19, %20, %3\n\t"
               L2_MF "%4, %16, %21, %4\n\t" L2_MF "%5, %17, %21, %5\n\t" L2_MF "%6, %18, %21, %6\n\t" L2_MF "%7, %19, %21, %7\n\t"
               L2_MF "%8, %16, %22, %8\n\t" L2_MF "%9, %17, %22, %9\n\t" L2_MF "%10, %18, %22, %10\n\t" L2_MF "%11, %19, %22, %11\n\t"
               L2_MF "%12, %16, %23, %12\n\t" L2_MF "%13, %17, %23, %13\n\t" L2_MF "%14, %18, %23, %14\n\t" L2_MF "%15, %19, %23, %15"
               : "+a"(c[0][0]), "+a"(c[0][1]), "+a"(c[0][2]), "+a"(c[0][3]), "+a"(c[1][0]), "+a"(c[1][1]), "+a"(c[1][2]), "+a"(c[1][3]),
                 "+a"(c[2][0]), "+a"(c[2][1]), "+a"(c[2][2]), "+a"(c[2][3]), "+a"(c[3][0]), "+a"(c[3][1]), "+a"(c[3][2]), "+a"(c[3][3])
               : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]));
}
#undef L2_MF
// after the last MFMA: its result is read by ordinary vector instructions (v_accvgpr_read) that hipcc pads for its own MFMAs only
template <class V, int NS>
__device__ __forceinline__ void mfma_drain(V (&c)[NS][4]) {
#pragma unroll
  for (int s_ = 0; s_ < NS; ++s_)
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(c[s_][0]), "+a"(c[s_][1]), "+a"(c[s_][2]), "+a"(c[s_][3]));
}

template <int MODE, int NW, int RT, int TT, bool F32 = false>
__global__ void __launch_bounds__(64 * NW) pf_gemm3_kernel(const PfArgs a_in) {
  static_assert(TT == 4, "mfma_group is written for four token tiles (a 64-token chunk)");
  using V = typename std::conditional<F32, f4, d4>::type;      // one result tile per lane: four fp64 (reference arithmetic) or four fp32 (opt-in)
  using E = typename std::conditional<F32, float, double>::type;
  // blockIdx.y: which 64-token chunk of the launch (up to PF_S): the same weight tile against the next 64 activation rows.  More
  // chunks per launch = more workgroups per launch, which is what lets wo / w2 (256 row tiles) take 2 or 4 row tiles per wave
  // and still fill 256 CUs, and what evens out w1 / w3's 688 tiles (2.7 per CU in one chunk, 10.75 in four).
  PfArgs a = a_in;
  {
    const int ch = blockIdx.y;
    a.xin += (size_t)ch * PF_T * a.n;
    if (a.x) a.x += (size_t)ch * PF_T * a.dim;
    if (a.out) a.out += (size_t)ch * PF_T * (MODE == MODE_W13 ? a.rows : a.dim);
    a.pos0 += ch * PF_T;
    a.nvalid = min(max(a.nvalid - ch * PF_T, 0), (int)PF_T);
  }
  constexpr bool DUAL = (MODE == MODE_W13);
  constexpr int NS = DUAL ? 2 * RT : RT;             // weight streams of the wave: row tiles (w1 tile r, w3 tile r, ... when DUAL)
  constexpr int UN = 2;                              // 16-column blocks per batch, two batches in flight (4 blocks: occupancy 2 -> 1 for w1 / w3, 215 -> 232 us)
  extern __shared__ __attribute__((aligned(16))) char part3_raw[];
  E* part3 = reinterpret_cast<E*>(part3_raw);                        // [TT][NW][4][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = a.n, npair = (n >> 4) / UN;
  const int j = lane & 15, kq = lane >> 4;
  const unsigned voff = (unsigned)(((size_t)j * n + 4 * kq) * 4);     // this lane's element of a 16-row tile, bytes
  // one buffer descriptor per weight stream (a tile never straddles wq / wk / wv: dim % 16 == 0), one for the activations
  // Repacked weights (a.wp: the decode step's copy, the only one once the row-major tensors have been given back): ONE descriptor
  // over the layer's slice; a stream's lane offset is base + nwr256 * (column batch) -- one multiply-add per stream and batch of
  // two blocks (32 MFMAs) -- and the block's place inside the batch stays in the scalar offset.
  const bool pk = a.wp != nullptr;
  __amdgpu_buffer_rsrc_t wrs[NS];
  PkLane pkl[NS];
  int tm[RT], ti0[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    const int row0 = (blockIdx.x * RT + r) * 16;
    int m = 0, i0 = row0;
    const float* wb = a.w0;
    if (MODE == MODE_QKV) { m = row0 / a.dim; i0 = row0 - m * a.dim; wb = (m == 0) ? a.w0 : (m == 1) ? a.w1 : a.w2; }
    tm[r] = m; ti0[r] = i0;
    const unsigned bytes = (unsigned)16 * (unsigned)n * 4u;
    if (pk) {
      const unsigned slice = (unsigned)a.pk_groups * 2u * (unsigned)n * 4u;       // bytes of the layer's repacked slice (< 2 GiB)
      int g, rr;
      if (DUAL) {
        wrs[2 * r] = wrs[2 * r + 1] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, slice, 0x00020000);
        pk_group_of<MODE>(row0 + j, false, g, rr); pkl[2 * r] = pk_lane(a, g, rr);
        pk_group_of<MODE>(row0 + j, true, g, rr); pkl[2 * r + 1] = pk_lane(a, g, rr);
      } else {
        wrs[r] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, slice, 0x00020000);
        pk_group_of<MODE>(row0 + j, false, g, rr); pkl[r] = pk_lane(a, g, rr);
      }
    } else if (DUAL) {
      wrs[2 * r] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w0 + (size_t)i0 * n), 0, bytes, 0x00020000);
      wrs[2 * r + 1] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w1 + (size_t)i0 * n), 0, bytes, 0x00020000);
    } else {
      wrs[r] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wb + (size_t)i0 * n), 0, bytes, 0x00020000);
    }
  }
  const int pk_lastci = (((n >> 2) + 127) >> 7) - 1;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.xin), 0, (unsigned)(16 * TT) * (unsigned)n * 4u, 0x00020000);
  const unsigned tstride = 16u * (unsigned)n * 4u;                    // bytes between token tiles
  V acc[NS][TT];
  {
    const V z = {0, 0, 0, 0};
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_)
#pragma unroll
      for (int t = 0; t < TT; ++t) acc[s_][t] = z;
  }
  struct Batch { f4 wv[UN][NS], xv[UN][TT]; };
  // batch i of this wave is batch wave + i * NW of the row (split K, round-robin); a batch past the wave's last one is requested
  // OUT OF RANGE of both descriptors and reads as zeros: the loop below then runs whole pairs of batches for every wave, with
  // one back edge and one exit (with a second exit between the two halves hipcc keeps two copies of the accumulator tiles and
  // moves 96 AGPRs from one to the other in every iteration)
  const int nb = (npair - wave + NW - 1) / NW;
  const unsigned oob = (unsigned)(16 * TT) * (unsigned)n * 4u;
  auto load = [&](Batch& b, int i) {
    const unsigned base = (i < nb) ? (unsigned)((wave + i * NW) * UN) * 64u : oob;     // scalar: 16 columns = 64 bytes per block
    // repacked: blocks 2p, 2p + 1 of a batch share their column batch ci and sub-batch; a batch past the wave's last one is
    // requested 2 GiB out (beyond any slice, no 32-bit wrap)
    const int blk0 = (wave + i * NW) * UN, ci = blk0 >> 5;
    const unsigned pbase = (i < nb) ? (unsigned)(((blk0 >> 4) & 1) * 64 + 4 * (blk0 & 15)) * 16u : 0x80000000u;
    unsigned pvoff[NS];
    if (pk) {
#pragma unroll
      for (int s_ = 0; s_ < NS; ++s_) pvoff[s_] = ((ci == pk_lastci ? pkl[s_].base_last : pkl[s_].base) + pkl[s_].nwr256 * (unsigned)ci + (unsigned)kq) * 16u;
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const unsigned soff = base + (unsigned)u * 64u;
#pragma unroll
      for (int s_ = 0; s_ < NS; ++s_)
        b.wv[u][s_] = pk ? __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(wrs[s_], pvoff[s_], pbase + (unsigned)u * 64u, 2))
                         : __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(wrs[s_], voff, soff, 2));   // nt: each weight byte is read once
#pragma unroll
      for (int t = 0; t < TT; ++t) b.xv[u][t] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(xrs, voff, soff + (unsigned)t * tstride, 0));
    }
  };
  auto mma = [&](const Batch& b) {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      E wd[NS][4], xd[TT][4];      // (fp64: widened here, on the pipe the MFMAs need; fp32: the loaded registers themselves)
#pragma unroll
      for (int s_ = 0; s_ < NS; ++s_) { wd[s_][0] = b.wv[u][s_].x; wd[s_][1] = b.wv[u][s_].y; wd[s_][2] = b.wv[u][s_].z; wd[s_][3] = b.wv[u][s_].w; }
#pragma unroll
      for (int t = 0; t < TT; ++t) { xd[t][0] = b.xv[u][t].x; xd[t][1] = b.xv[u][t].y; xd[t][2] = b.xv[u][t].z; xd[t][3] = b.xv[u][t].w; }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        E wk[NS];
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) wk[s_] = wd[s_][k];
        mfma_group(acc, xd[0][k], xd[1][k], xd[2][k], xd[3][k], wk);
      }
    }
  };
  Batch A, B;
  load(A, 0);
  for (int i = 0; i < nb; i += 2) {
    load(B, i + 1);                                    // batch i + 1 loads while batch i is on the matrix pipe
    mma(A);
    load(A, i + 2);
    mma(B);
  }
  mfma_drain<V, NS>(acc);
  // ---- split-K combine, one weight stream at a time: every wave parks its TT partial tiles, wave t (t < TT) adds token tile t
  // over the waves IN WAVE ORDER (the same sum on every run) and finishes it
  auto park = [&](int s_) {
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) part3[((size_t)(t * NW + wave) * 4 + r) * 64 + lane] = acc[s_][t][r];
  };
  auto gather = [&](int t) {
    V v = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      E sacc = part3[((size_t)(t * NW + 0) * 4 + r) * 64 + lane];
#pragma unroll
      for (int w = 1; w < NW; ++w) sacc += part3[((size_t)(t * NW + w) * 4 + r) * 64 + lane];
      v[r] = sacc;
    }
    return v;
  };
  static_assert(NW >= TT, "one finishing wave per token tile");
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    V first = {0, 0, 0, 0}, second = {0, 0, 0, 0};
    __syncthreads();                                   // the previous stream's partials have been read
    park(DUAL ? 2 * r : r);
    __syncthreads();
    if (wave < TT) first = gather(wave);
    if (DUAL) {
      __syncthreads();
      park(2 * r + 1);
      __syncthreads();
      if (wave < TT) second = gather(wave);
    }
    if (wave < TT) pf_emit<MODE, F32, V>(a, first, second, tm[r], ti0[r] + j, j, kq, 16 * wave);
  }
}

// ------------------------------------------------------------------------------------------------
// Attention of a prompt chunk on the matrix cores (llama2.ts:244-267 for 16 query positions at a time).
// pf_attn_tile_kernel (attention.hip.h) is the decode kernel run once per (head, query): every query re-reads its head's cache
// rows, 23 / 57 / 155 us per layer at 64 / 128 / 256 prompt tokens (7B width).  Here one workgroup takes one head and 16
// consecutive queries: scores = Q (16 x hs) K^T on v_mfma_f64_16x16x4_f64 (Q widened once into registers, the key blocks of 16
// dealt round-robin to the 4 waves), one fp32 rounding of dot x 1/sqrt(hs) per score as the reference has it (:253), causal
// mask, softmax per query row with the reference's roundings (exp stored fp32 :187, fp64 sum of the rounded values :190, quotient
// stored fp32 :192), then xb = P V on the same instruction (each wave owns hs / 64 of the 16-wide output column tiles, fp64
// accumulate over all keys, ONE rounding: the accumulate of L2_OPT_EXACT_ATTENTION = 0).  Scores / probabilities live in LDS
// as 16 rows of fp32 (row stride + 4 floats: conflict-free 16-byte reads of the A fragments).
struct PfAttnArgs {
  const float* q;       // [rows][dim] rotated queries of the chunk
  const float* kc;      // key_cache + l*S*dim (rows pos0 .. pos0+n-1 were written by the chunk's qkv GEMM)
  const float* vc;
  float* xb;            // [rows][dim] out
  int dim, head_size, seq_len, pos0, nvalid;
  double inv_sqrt_hs;
};

// LDS of a launch whose last query sits at position `last_pos`: 16 rows x (keys rounded up to 16, + 4 floats of padding)
__host__ __device__ inline size_t pf_attn_lds(int last_pos) { return (size_t)16 * (size_t)((((last_pos + 16) + 15) & ~15) + 4) * 4 + 64 * 8; }

template <int HS>      // head_size: 64 or 128
__global__ void __launch_bounds__(256) pf_attn_mfma_kernel(const PfAttnArgs a) {
  static_assert(HS % 64 == 0, "four waves x whole 16-wide column tiles");
  constexpr int KB = HS / 16;                          // 16-column blocks of a head row
  constexpr int CT = HS / 64;                          // output column tiles per wave
  extern __shared__ __attribute__((aligned(16))) char pfa_smem[];
  const int h = blockIdx.x, qt = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;
  const int q0 = qt * 16;                              // first query row of the tile (relative to the chunk)
  const int T = a.pos0 + q0 + 16;                      // keys 0 .. T-1 cover every query of the tile (T % 16 == pos0 % 16)
  const int Tb = (T + 15) >> 4;                        // key blocks of 16
  const int ST = Tb * 16 + 4;                          // LDS row stride, floats
  float* sc = reinterpret_cast<float*>(pfa_smem);      // [16][ST]
  double* red = reinterpret_cast<double*>(pfa_smem + (size_t)16 * ST * 4);
  const int dim = a.dim;
  const unsigned slab = (unsigned)a.seq_len * (unsigned)dim * 4u;
  const auto krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.kc), 0, slab, 0x00020000);
  const auto vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.vc), 0, slab, 0x00020000);

  // ---- Q fragments of the tile, widened once: lane (j, kq) holds q[q0 + j][h*HS + 16 b + 4 kq + e], b < KB, e < 4
  double qd[KB][4];
  {
    const float* qrow = a.q + (size_t)(q0 + j) * dim + (size_t)h * HS + 4 * kq;
#pragma unroll
    for (int b = 0; b < KB; ++b) {
      const f4 v = *reinterpret_cast<const f4*>(qrow + 16 * b);
      qd[b][0] = v.x; qd[b][1] = v.y; qd[b][2] = v.z; qd[b][3] = v.w;
    }
  }
  // ---- scores: key block kb (keys 16 kb .. 16 kb + 15) -> D[query kq + 4 r][key j]
  const unsigned kvoff = (unsigned)(((size_t)j * dim + (size_t)h * HS + 4 * kq) * 4);     // this lane's element of a 16-row block
  const unsigned blk = 16u * (unsigned)dim * 4u;                                          // bytes per key block
  const double rsq = a.inv_sqrt_hs;
  for (int kb = wave; kb < Tb; kb += 4) {
    f4 kf[KB];
#pragma unroll
    for (int b = 0; b < KB; ++b) kf[b] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(krs, kvoff, (unsigned)kb * blk + (unsigned)b * 64u, 0));
    d4 acc = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};   // two chains: a dependent MFMA does not issue back to back
#pragma unroll
    for (int b = 0; b < KB; ++b) {
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(qd[b][0], (double)kf[b].x, acc, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(qd[b][1], (double)kf[b].y, acc1, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(qd[b][2], (double)kf[b].z, acc, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(qd[b][3], (double)kf[b].w, acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] += acc1[r];
    const int t = kb * 16 + j;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qi = kq + 4 * r;                        // query row of the tile; its position is pos0 + q0 + qi
      const float s = (float)(acc[r] * rsq);            // one rounding (llama2.ts:253 divides by sqrt(head_size))
      sc[qi * ST + t] = (t <= a.pos0 + q0 + qi) ? s : -INFINITY;
    }
  }
  __syncthreads();
  // ---- softmax per query row (llama2.ts:181-194): wave w owns rows 4w .. 4w + 3
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    float* row = sc + (wave * 4 + rr) * ST;
    const int nk = a.pos0 + q0 + wave * 4 + rr + 1;     // keys this query sees
    float mx = -INFINITY;
    for (int t = lane; t < nk; t += 64) mx = fmaxf(mx, row[t]);
    mx = wave_max(mx);
    double lsum = 0.0;
    for (int t = lane; t < Tb * 16; t += 64) {
      const float e = (t < nk) ? (float)exp_fast((double)row[t] - (double)mx) : 0.0f;    // stored to fp32 (:187); masked keys contribute nothing
      row[t] = e;
      lsum += (double)e;                                                                  // sum of the ROUNDED values (:190)
    }
    lsum = wave_sum(lsum);
    const double rs = rcp_fast(lsum);
    for (int t = lane; t < nk; t += 64) row[t] = (float)((double)row[t] * rs);            // quotient stored fp32 (:192)
  }
  __syncthreads();
  // ---- xb = P V: wave w owns output columns [w * 16 CT, (w + 1) * 16 CT) of the head; D[query kq + 4 r][column j]
  d4 o[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) o[c] = d4{0.0, 0.0, 0.0, 0.0};
  const unsigned vcol = (unsigned)(((size_t)h * HS + (size_t)wave * 16 * CT + j) * 4);    // this lane's column, bytes
  for (int kb = 0; kb < Tb; ++kb) {
    const f4 pf = *reinterpret_cast<const f4*>(sc + j * ST + kb * 16 + 4 * kq);          // P[query j][keys 16 kb + 4 kq .. + 3]
    float vv[CT][4];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e)      // V[key 16 kb + 4 kq + e][column]: the key differs per lane, so the row offset rides in the vector offset
        vv[c][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(vrs, vcol + (unsigned)c * 64u + (unsigned)(kb * 16 + 4 * kq + e) * (unsigned)dim * 4u, 0, 0));
    const double p0 = pf.x, p1 = pf.y, p2 = pf.z, p3 = pf.w;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      o[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(p0, (double)vv[c][0], o[c], 0, 0, 0);
      o[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(p1, (double)vv[c][1], o[c], 0, 0, 0);
      o[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(p2, (double)vv[c][2], o[c], 0, 0, 0);
      o[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(p3, (double)vv[c][3], o[c], 0, 0, 0);
    }
  }
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qi = kq + 4 * r;
      if (q0 + qi < a.nvalid) a.xb[(size_t)(q0 + qi) * dim + (size_t)h * HS + (size_t)wave * 16 * CT + c * 16 + j] = (float)o[c][r];   // ONE rounding of the fp64 sum
    }
  (void)red;
}

}  // namespace l2k

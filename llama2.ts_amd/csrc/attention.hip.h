// attention.hip.h -- multi-head attention over the KV cache for one position (llama2.ts:244-267), gfx950.
//
// One workgroup per (head, split).  The cache rows of a head are `head_size` contiguous floats every `dim`
// floats (cache layout [L][S][d], llama2.ts:160-161), so a row is read by LR = head_size/4 (rounded up to a
// power of two) adjacent lanes, 16 bytes each: one wave instruction covers 64/LR whole rows and touches each
// 128-byte line once.  (The round-1 kernel gave every thread its own row: 64 lines per instruction, and the
// address path, not HBM, set the time -- 11 500 cycles per head at 100 cached positions on stories110M.)
//
//   scores   (:249-254)  every lane multiplies its float4 of the row with its float4 of q in fp64; the LR partial
//                        sums of a row are added through a wave-private LDS transpose (16 tiles at a time), one
//                        rounding to fp32 after the fp64 divide by sqrt(head_size);
//   softmax  (:181-194)  block-wide, with the reference's three roundings (exp stored fp32, sum of the rounded
//                        values in fp64, quotient stored fp32);
//   values   (:257-265)  same tiles, fp64 partial per (row slot, wave), summed through LDS, ONE rounding --
//                        or, with `exact`, the reference's own t-sequential fp32-rounded accumulate.
//
// Every K and V tile of the first 16 x NW x (64/LR) rows is requested before anything is waited for; nothing in
// the kernel depends on q until the first multiply.  Splits (flash-decode) publish {acc, l, m} write-through and
// the last arriver of a head merges, as the MI355X guide's hand-off recipe prescribes (sc1 stores, every storing
// wave drains, barrier, one agent-scope ticket; the merger reads with sc1 loads).
#pragma once
#include "kernels.hip.h"

namespace l2k {

struct AttnArgs {
  Mut<const float> q;    // (dim) rotated q of this position      (Mut: bytes an earlier launch of the run wrote -- no plain load compiles, kernels.hip.h)
  Mut<const float> kc;   // key_cache   + l*S*d   (row `pos` was written by the QKV launch before this one)
  Mut<const float> vc;   // value_cache + l*S*d
  Mut<float> att;        // (H, S) probabilities (RunState.att, parity reads)
  Mut<float> xb;         // (dim) out
  const int* tokpos;
  Mut<double> part;      // split form: [H][nsplit][rec] doubles, rec = round_up(hs + 2, 16)
  unsigned* counter;     // split form: [H] merge tickets (one per 128-byte line), zero between launches
  int dim, head_size, seq_len, n_heads, nsplit;
  int kv_dim, kv_mul;    // floats of a cache row; query heads per cache head (1 unless the context honours n_kv_heads < n_heads)
  double inv_sqrt_hs;    // 1 / sqrt(head_size)
  int exact;             // 1: fp32-rounded t-sequential value accumulate (llama2.ts:263); never with nsplit > 1
  int pos_plus1;         // scalar fallback kernel, prefill: non-zero = the queries are pos0 + blockIdx.y (else tokpos)
  // fused QKV + attention launch (qkv_attn_small_kernel): q, k, v of THIS position arrive as hand-off granules from the workgroups
  // of the same launch that compute them (kernels.hip.h: granule_store)
  const unsigned long long* gran;   // [dim + 2 kv_dim] words, tag = *gran_ep + 1
  unsigned* gran_ep;               // launch counter: read by every workgroup at its start, advanced once per launch by whoever finishes head 0
  int fused_four_waves;            // fused launch: contexts of up to 128 rows on four waves (else eight)
  int* herr;                       // host-mapped: set when a granule wait gave up
  unsigned long long wait_ticks;   // bound of that wait on the 100 MHz clock
  // fused attention + wo launch of a tensor-parallel rank (attn_wo_kernel): the output is ALSO published as granules for the wo
  // workgroups of the same launch; tag = *gout_ep + 1 (advanced by the launch after this one: tp_p2p_combine_kernel)
  unsigned long long* gout;
  const unsigned* gout_ep;
  unsigned long long* dbg;  // diagnostic stamps (L2_STAMPS builds), else null
};

// the attention output, element i of head h: RunState.xb, and the hand-off granule when a workgroup of this launch waits for it
__device__ __forceinline__ void attn_out(const AttnArgs& a, size_t idx, float v, unsigned otag) {
  if (a.gout) granule_store(a.gout + idx, v, otag);      // first: a workgroup of this launch waits for it
  a.xb.st(idx, v);
}

// A wave's wait for its hand-off granules (fused launch): every lane has up to N words to collect; all L1-bypassing loads of a pass
// are issued together (ONE round trip per pass), and the wave leaves when every lane has every value.  Bounded by wall time: the
// producers are workgroups of the same launch with LOWER block ids, already dispatched when this workgroup runs, so the wait can
// only give up if the launch itself is broken -- then the host hears about it (herr) instead of the GPU hanging.
// `dead`: the device-side error word, set by the first wait that gave up and read by every fused launch when it starts: a launch that
// finds it set does not wait again (a systematically broken launch would otherwise spend the full bound in every layer of every token
// before the host hears about it); the host clears it when it reports the error (check_p2p).
template <int N>
__device__ __forceinline__ void granules_wait(const unsigned long long* const (&g)[N], float (&v)[N], unsigned tag, int* herr, unsigned long long wait_ticks, unsigned* dead) {
  unsigned spins = 0;
  unsigned long long t0 = 0;
  for (;;) {
    unsigned long long x[N];
#pragma unroll
    for (int k = 0; k < N; ++k) x[k] = __hip_atomic_load(g[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (the error word rides along with the FIRST sweep: a test in front of it would be a round trip of its own before the wait begins)
    if (spins == 0 && __builtin_amdgcn_readfirstlane(__hip_atomic_load(dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) return;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < N; ++k) { ok = ok && (unsigned)(x[k] >> 32) == tag; v[k] = __uint_as_float((unsigned)x[k]); }
    if (__all(ok)) return;
    if ((++spins & 255u) == 0) {
      const unsigned long long now = __builtin_amdgcn_s_memrealtime();
      if (!t0) t0 = now;
      else if (now - t0 > wait_ticks) { *herr = 1; __hip_atomic_store(dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
    }
  }
}

// Fused launch: every head has a launch counter of its own, gran_ep[h]; the granules of head h carry tag gran_ep[h] + 1.  The
// workgroup that finishes head h (its only one, or the merger of its splits) stores the new number when it is done -- a plain
// store nobody waits for.  Every reader of gran_ep[h] in this launch has read it by then: the attention workgroups of head h read
// it before they take their ticket, and a QKV wave reads it before it stores a granule that head h waits for.
__device__ __forceinline__ void fused_head_done(const AttnArgs& a, int h, unsigned tag, int tid) {
  if (tid == 0) __hip_atomic_store(a.gran_ep + h, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// rows of a split: ceil(T / nsplit).  (A shift for power-of-two split counts behind a run-time test was measured: hipcc then carries
// both forms through all three tile-count bodies, 90 KB more code, and the kernel lost 0.4 us: profiles/r04/attention_prologue_ab.txt.)
__device__ __forceinline__ int attn_chunk(const AttnArgs& a, int T) {
  return (T + a.nsplit - 1) / a.nsplit;
}

constexpr int ATT_PS = 65;       // doubles per tile slot in the transpose buffer (odd: conflict-free reads)

__host__ __device__ inline int attn_rec(int hs) { return (hs + 2 + 15) & ~15; }

// NT = tiles (wave instructions) per wave per round.  LDS bytes: sc[cmax] floats | red[32] doubles | P[NW][NT][65] doubles (reused as the value partials
// [NW * 64/LR][hs], never larger) -- the host sizes the launch with the same function.
__host__ __device__ inline size_t attn_tile_lds(int S, int nsplit, int NW, int NT) {
  const int cmax = (S + nsplit - 1) / nsplit;
  return (size_t)((cmax + 3) & ~3) * 4 + 32 * 8 + (size_t)NW * NT * ATT_PS * 8;
}

// FUSED: this workgroup runs in the launch that computes q, k, v of position `pos` (qkv_attn_small_kernel).  Cache rows 0 .. pos - 1
// are requested at once like always (the descriptor ends at row pos: the row being written reads as zeros and contributes
// nothing); q, then k and v of row pos come as granules: the score of row pos is one wave's dot product of the q and k granules,
// its share of the output att[pos] * v[pos] is added where the partial sums are folded -- fp64, one rounding, like every other row.
// MULTI: the workgroup's rows may take more than one round of NW * NT tiles (A / B register sets, the next round in flight while this
// one is used).  Only the half-size body is ever built that way (attn_tile_dispatch): with both 16-tile sets live ACROSS a loop the
// instance needed 255 VGPRs + 64 AGPRs as spill space (216 register copies in the loop); a single-round body drops the loop code.
template <int LR, int NW, int NT, bool FUSED = false, bool MULTI = true>
__device__ __forceinline__ void attn_tile_body(const AttnArgs& a, char* smem, const int h, const int sp, const int pos) {
  constexpr int RPT = 64 / LR;             // rows per tile
  constexpr int RG = NT * RPT;         // rows per wave per round
  constexpr int RR = NW * RG;              // rows per workgroup per round
  constexpr int NTH = 64 * NW;
  const int S = a.seq_len, hs = a.head_size, dim = a.kv_dim, NS = a.nsplit;   // `dim`: the stride of the cache rows
  const int hk = h / a.kv_mul;                                                // this head's columns of a cache row
  const int cmax = (S + NS - 1) / NS;
  float* sc = reinterpret_cast<float*>(smem);
  double* red = reinterpret_cast<double*>(smem + (size_t)((cmax + 3) & ~3) * 4);
  double* P = red + 32;
  unsigned* ticket = reinterpret_cast<unsigned*>(red + 31);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = pos + 1;
  const int chunk = attn_chunk(a, T);
  const int t0 = sp * chunk, t1 = min(T, t0 + chunk);
  const int n = max(t1 - t0, 0);                       // rows of this workgroup
  const int rounds = MULTI ? (n + RR - 1) / RR : 1;
  const int r = lane / LR, c = lane % LR;
  const bool cl = 4 * c < hs;                           // lanes past the head's width carry zeros
  const int cc4 = cl ? 4 * c : 0;

  STAMP_INIT_SEL(a.dbg, FUSED ? ((h == 0 && sp == 0) ? 2 : -1) : -2);      // fused launch: head 0's workgroup is the stamped one
  STAMP(0);
  // ---- requests first: K tiles of round 0, then q, then (single round) the V tiles of round 0.
  // Buffer loads: ONE 32-bit lane offset for all 16 tiles of a set (the tile stride goes in the scalar offset), and
  // the descriptor ends at row t1, so rows past this workgroup's slice read as zeros -- no clamps, no predicates.
  f4 ra[NT], rb[NT];
  auto row_of = [&](int rd, int j) { return ((rd * NT + j) * NW + wave) * RPT + r; };   // relative to t0
  const bool own_pos = FUSED && t1 == T && n > 0;      // this split ends with the row of this position
  const int n_tile = own_pos ? n - 1 : n;               // rows that come through the cache tiles
  const unsigned slab = (unsigned)max(FUSED ? min(t1, pos) : t1, 0) * (unsigned)dim * 4u;
  const auto krs = a.kc.rsrc_bytes(slab);
  const auto vrs = a.vc.rsrc_bytes(slab);
  const unsigned voff = (unsigned)(((size_t)(t0 + wave * RPT + r) * dim + (size_t)hk * hs + cc4) * 4);
  const unsigned tstride = (unsigned)(NW * RPT) * (unsigned)dim * 4u;                            // bytes between a wave's tiles
  auto issue = [&](f4 (&buf)[NT], bool values, int rd) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
      buf[j] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(values ? vrs : krs, voff, (unsigned)(rd * NT + j) * tstride, L2_SC1_AUX));      // sc1: cache rows are written by launches of this run (coherence rule, kernels.hip.h)
  };
  issue(ra, false, 0);
  f4 q4;
  if (!FUSED) { const auto qrs = a.q.rsrc(a.dim); q4 = L2_ACT_LD4(qrs, ((size_t)h * hs + cc4) >> 2); }
  if (!MULTI || rounds <= 1) issue(rb, true, 0);
  // FUSED: the cache tiles are in flight; now ONE wait for everything of this position the lane will need: its four q values,
  // and -- the wave that scores row pos -- q[i] and k[i], and -- the threads that fold the output -- v[i] (hs <= 64: one each)
  float gk_q = 0.0f, gk_k = 0.0f, gv = 0.0f;
  unsigned ftag = 0;
  if (FUSED) {
    const unsigned tag = ftag = ld_word(const_cast<const unsigned*>(a.gran_ep) + h) + 1u;      // advanced by an EARLIER launch (and again only when this head is done)
    const unsigned long long* gq = a.gran + (size_t)h * hs + cc4;
    const bool need_k = own_pos && wave == NW - 1 && lane < hs, need_v = own_pos && tid < hs;
    const unsigned long long* const gp[7] = {gq, gq + 1, gq + 2, gq + 3,
        need_k ? a.gran + (size_t)h * hs + lane : gq, need_k ? a.gran + (size_t)a.dim + (size_t)hk * hs + lane : gq,
        need_v ? a.gran + (size_t)a.dim + a.kv_dim + (size_t)hk * hs + tid : gq};
    float gvals[7];
    granules_wait<7>(gp, gvals, tag, a.herr, a.wait_ticks, a.gran_ep + a.n_heads);      // (the word behind the heads' counters)
    STAMP(6);
    q4.x = gvals[0]; q4.y = gvals[1]; q4.z = gvals[2]; q4.w = gvals[3];
    gk_q = need_k ? gvals[4] : 0.0f; gk_k = need_k ? gvals[5] : 0.0f; gv = need_v ? gvals[6] : 0.0f;
  }
  const double q0 = cl ? (double)q4.x : 0.0, q1 = cl ? (double)q4.y : 0.0, q2 = cl ? (double)q4.z : 0.0, q3 = cl ? (double)q4.w : 0.0;
  const double rsq = a.inv_sqrt_hs;             // 1 / sqrt(head_size), rounded once by the host (llama2.ts:253 divides)
  // (requested here, used at the very end: the launch counter the output granules' tag comes from)
  const unsigned otag = a.gout ? __hip_atomic_load(a.gout_ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u : 0u;
  double* Pw = P + (size_t)wave * NT * ATT_PS;
  STAMP(1);

  // ---- scores (llama2.ts:249-254)
  auto score_round = [&](const f4 (&buf)[NT], int rd) {
    // every tile, live or not (rows past the slice read as zeros): straight-line code, so the NT independent chains
    // interleave -- a branch per tile left each chain's fp64 latency (~30 cycles an instruction) exposed
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const double s = (q0 * (double)buf[j].x + q1 * (double)buf[j].y) + (q2 * (double)buf[j].z + q3 * (double)buf[j].w);
      Pw[j * ATT_PS + lane] = s;
    }
    __builtin_amdgcn_wave_barrier();
    // wave-private transpose: dot (tile j, row slot rr) = sum over the row's LR lanes (four interleaved chains)
#pragma unroll
    for (int dd = lane; dd < NT * RPT; dd += 64) {
      const int rr = dd / NT, j = dd % NT;
      const int tr = ((rd * NT + j) * NW + wave) * RPT + rr;
      const double* pp = Pw + j * ATT_PS + rr * LR;
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
      for (int k = 0; k < LR; k += 4) { s0 += pp[k]; s1 += pp[k + 1]; s2 += pp[k + 2]; s3 += pp[k + 3]; }
      if (tr < n_tile) sc[tr] = (float)(((s0 + s1) + (s2 + s3)) * rsq);
    }
    __builtin_amdgcn_wave_barrier();
  };
  if (FUSED) {
    if (own_pos && wave == NW - 1) {   // score of row pos from the handed-off q and k (llama2.ts:249-253), by the wave with the fewest tile rows
      const double s = wave_sum((double)gk_q * (double)gk_k);
      if (lane == 0) sc[n - 1] = (float)(s * rsq);
    }
  }
  if (!MULTI || rounds <= 1) {
    score_round(ra, 0);
  } else {
    for (int rd = 0; rd < rounds; rd += 2) {        // A / B register sets: the next round is in flight while this one is used
      if (rd + 1 < rounds) issue(rb, false, rd + 1);
      score_round(ra, rd);
      if (rd + 1 < rounds) {
        if (rd + 2 < rounds) issue(ra, false, rd + 2);
        score_round(rb, rd + 1);
      }
    }
    issue(ra, true, 0);
    if (rounds > 1) issue(rb, true, 1);
  }
  __syncthreads();
  STAMP(2);

  // ---- softmax (llama2.ts:181-194); a split keeps its own max / sum, rescaled at the merge.
  // Every wave finds the maximum of all n scores itself (one LDS read per 64 rows): no cross-wave reduction for it.
  float mx = -INFINITY;
  for (int t = lane; t < n; t += 64) mx = fmaxf(mx, sc[t]);
  mx = wave_max(mx);
  // the exps below overwrite sc[] in place whenever a thread has more than one element or a split keeps them for its
  // value round: no wave may store before EVERY wave has finished its maximum scan of the scores (workgroup-uniform test)
  if (n > NTH || NS != 1) __syncthreads();
  double lsum = 0.0;
  float e_own = 0.0f;                                           // n <= NTH (the usual case): the thread's one element stays in a register
  for (int t = tid; t < n; t += NTH) {
    const float e = (float)exp_fast((double)sc[t] - (double)mx);    // stored to fp32 (llama2.ts:187)
    if (n > NTH || NS != 1) sc[t] = e;
    e_own = e;
    lsum += (double)e;                                          // sum of the ROUNDED values (:190)
  }
  lsum = wave_sum(lsum);
  if (lane == 0) red[8 + wave] = lsum;
  __syncthreads();
  double sum = red[8];
#pragma unroll
  for (int w = 1; w < NW; ++w) sum += red[8 + w];
  if (NS == 1) {
    const double rs = rcp_fast(sum);                            // llama2.ts:192 divides; the quotient is rounded to fp32
    for (int t = tid; t < n; t += NTH) {
      const float pr = (float)((double)(n > NTH ? sc[t] : e_own) * rs);
      sc[t] = pr;
      if (a.att) a.att.st((size_t)h * S + t, pr);
    }
  } else if (a.att) {
    for (int t = tid; t < n; t += NTH) a.att.st_sc1((size_t)h * S + t0 + t, sc[t]);   // rescaled by the merging workgroup
  }
  __syncthreads();
  STAMP(3);

  // ---- weighted sum of values (llama2.ts:257-265)
  if (a.exact && NS == 1) {
    // the reference's own rounding points: the accumulator is a Float32Array element, rounded at every timestep, t ascending
    for (int i = tid; i < hs; i += NTH) {
      const Mut<const float> vp = a.vc + ((size_t)hk * hs + i);
      float o = 0.0f;
      for (int t = 0; t < n; ++t) o = (float)((double)o + (double)sc[t] * (double)vp.ld((size_t)t * dim));
      attn_out(a, (size_t)h * hs + i, o, otag);
    }
    return;
  }
  double o0 = 0.0, o1 = 0.0, o2 = 0.0, o3 = 0.0, e0 = 0.0, e1 = 0.0, e2 = 0.0, e3 = 0.0;   // two sets of chains
  auto value_round = [&](const f4 (&buf)[NT], int rd) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int tr = row_of(rd, j);
      const double at = (tr < n_tile) ? (double)sc[tr] : 0.0;
      if (j & 1) { e0 += at * (double)buf[j].x; e1 += at * (double)buf[j].y; e2 += at * (double)buf[j].z; e3 += at * (double)buf[j].w; }
      else { o0 += at * (double)buf[j].x; o1 += at * (double)buf[j].y; o2 += at * (double)buf[j].z; o3 += at * (double)buf[j].w; }
    }
  };
  if (!MULTI || rounds <= 1) {
    value_round(rb, 0);
  } else {
    for (int rd = 0; rd < rounds; rd += 2) {
      value_round(ra, rd);
      if (rd + 2 < rounds) issue(ra, true, rd + 2);
      if (rd + 1 < rounds) {
        value_round(rb, rd + 1);
        if (rd + 3 < rounds) issue(rb, true, rd + 3);
      }
    }
  }
  o0 += e0; o1 += e1; o2 += e2; o3 += e3;
  double* pacc = P;                                              // [NW * RPT][hs]: the transpose buffer is free now
  if (cl) {
    double* dst = pacc + (size_t)(wave * RPT + r) * hs + 4 * c;
    dst[0] = o0; dst[1] = o1; dst[2] = o2; dst[3] = o3;
  }
  STAMP(4);
  __syncthreads();
  if (NS == 1) {
    for (int i = tid; i < hs; i += NTH) {
      double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
#pragma unroll
      for (int g = 0; g < NW * RPT; g += 4) {
        c0 += pacc[(size_t)g * hs + i]; c1 += pacc[(size_t)(g + 1) * hs + i];
        c2 += pacc[(size_t)(g + 2) * hs + i]; c3 += pacc[(size_t)(g + 3) * hs + i];
      }
      double own = 0.0;
      if (FUSED) { if (own_pos) own = (double)sc[n - 1] * (double)gv; }      // hs <= 64 <= NTH: i == tid
      if (FUSED) a.xb.st((size_t)h * hs + i, (float)(((c0 + c1) + (c2 + c3)) + own));   // ONE rounding of the fp64 sum
      else attn_out(a, (size_t)h * hs + i, (float)((c0 + c1) + (c2 + c3)), otag);
    }
    STAMP(5);
    if (FUSED) fused_head_done(a, h, ftag, tid);
    return;
  }

  // ---- split form: publish {acc, l, m}, last arriver of the head merges
  const int rec = attn_rec(hs);
  const Mut<double> mypart = a.part + ((size_t)h * NS + sp) * rec;
  for (int i = tid; i < hs; i += NTH) {
    double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
#pragma unroll
    for (int g = 0; g < NW * RPT; g += 4) {
      c0 += pacc[(size_t)g * hs + i]; c1 += pacc[(size_t)(g + 1) * hs + i];
      c2 += pacc[(size_t)(g + 2) * hs + i]; c3 += pacc[(size_t)(g + 3) * hs + i];
    }
    double own = 0.0;
    if (FUSED) { if (own_pos) own = (double)sc[n - 1] * (double)gv; }
    if (FUSED) mypart.st_sc1(i, ((c0 + c1) + (c2 + c3)) + own);
    else mypart.st_sc1(i, (c0 + c1) + (c2 + c3));
  }
  if (tid == 0) { mypart.st_sc1(hs, sum); mypart.st_sc1(hs + 1, (double)mx); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave drains its write-through stores
  __syncthreads();
  if (tid == 0) *ticket = __hip_atomic_fetch_add(a.counter + (size_t)h * CTR_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (*ticket != (unsigned)(NS - 1)) return;

  const Mut<double> hp = a.part + (size_t)h * NS * rec;     // (ld: past this CU's L1)
  if (NS <= 8) {
    // ONE round of loads: every thread requests {m, l} of all splits and its element of every partial at once
    // (three dependent rounds of L1-bypassing loads were 3 us of the merge)
    for (int i = tid; i < hs; i += NTH) {
      double mm[8], ll[8], aa[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const Mut<double> rp = hp + (size_t)min(k, NS - 1) * rec;
        mm[k] = rp.ld(hs + 1); ll[k] = rp.ld(hs); aa[k] = rp.ld(i);
      }
      double M = mm[0];
#pragma unroll
      for (int k = 1; k < 8; ++k) M = fmax(M, mm[k]);
      double Lsum = 0.0, num = 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const double w = (k < NS) ? exp_fast(mm[k] - M) : 0.0;
        Lsum += w * ll[k]; num += w * aa[k];
      }
      attn_out(a, (size_t)h * hs + i, (float)(num / Lsum), otag);
    }
  } else {
    double M = -INFINITY;
    for (int s2 = 0; s2 < NS; ++s2) M = fmax(M, hp.ld((size_t)s2 * rec + hs + 1));
    double Lsum = 0.0;
    for (int s2 = 0; s2 < NS; ++s2) Lsum += exp(hp.ld((size_t)s2 * rec + hs + 1) - M) * hp.ld((size_t)s2 * rec + hs);
    for (int i = tid; i < hs; i += NTH) {
      double num = 0.0;
      for (int s2 = 0; s2 < NS; ++s2) num += exp(hp.ld((size_t)s2 * rec + hs + 1) - M) * hp.ld((size_t)s2 * rec + i);
      attn_out(a, (size_t)h * hs + i, (float)(num / Lsum), otag);
    }
  }
  if (a.att) {                                          // probabilities for parity reads of RunState.att (L2_OPT_KEEP_ATT)
    double M = -INFINITY;
    for (int s2 = 0; s2 < NS; ++s2) M = fmax(M, hp.ld((size_t)s2 * rec + hs + 1));
    double Lsum = 0.0;
    for (int s2 = 0; s2 < NS; ++s2) Lsum += exp(hp.ld((size_t)s2 * rec + hs + 1) - M) * hp.ld((size_t)s2 * rec + hs);
    for (int t = tid; t < T; t += NTH) {
      const double ws = exp(hp.ld((size_t)(t / chunk) * rec + hs + 1) - M);
      a.att.st((size_t)h * S + t, (float)((double)a.att.ld((size_t)h * S + t) * ws / Lsum));
    }
  }
  if (tid == 0) __hip_atomic_store(a.counter + (size_t)h * CTR_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // every split has its ticket: re-arm
  if (FUSED) fused_head_done(a, h, ftag, tid);
}

// Tiles per wave per round by the rows this workgroup has: a short context requests (and pays address-path cycles
// for) only the tiles it has -- NT / 4, NT / 2 or NT (one round covers NW * NT * 64 / LR rows).
template <int LR, int NW, int NT, bool FUSED = false>
__device__ __forceinline__ void attn_tile_dispatch(const AttnArgs& a, char* smem, const int h, const int sp, const int pos) {
  const int T = pos + 1;
  const int chunk = attn_chunk(a, T);
  const int n = min(T, sp * chunk + chunk) - sp * chunk;
  constexpr int RQ = NW * (NT / 4) * (64 / LR);
  if (n <= RQ) attn_tile_body<LR, NW, NT / 4, FUSED, false>(a, smem, h, sp, pos);
  else if (n <= 2 * RQ) attn_tile_body<LR, NW, NT / 2, FUSED, false>(a, smem, h, sp, pos);
  else if (FUSED || n <= 4 * RQ) attn_tile_body<LR, NW, NT, FUSED, false>(a, smem, h, sp, pos);      // (the fused launch is only taken while one round covers the rows: launch.hip.h)
  else attn_tile_body<LR, NW, NT / 2, FUSED, true>(a, smem, h, sp, pos);      // several rounds: half-size register sets (see attn_tile_body)
}

// ------------------------------------------------------------------------------------------------
// ONE launch for the two phases whose edge is head-local (llama2.ts:216-240 -> 244-267): the first gridDim.x - H * nsplit
// workgroups are the latency-form QKV phase (rmsnorm + q, k, v GEMVs + RoPE + cache rows), the LAST H * nsplit are the attention
// workgroups of the heads.  An attention workgroup requests its cache rows the moment it starts -- they depend on nothing of
// this position -- and the three vectors that do arrive as hand-off granules straight from the waves that compute them: no
// launch boundary between the phases, and the cache rows' way from HBM is hidden behind the GEMV.  Workgroups are dispatched in
// block order, so every producer is resident (or done) before a consumer starts to wait: nothing here can deadlock, and the
// wait is bounded anyway.  Needs one cache head per query head (no grouped queries) and the latency form's 512-thread workgroups.
template <int XV, int LR, int NT>
__global__ void __launch_bounds__(512) qkv_attn_small_kernel(const PhaseArgs a, const AttnArgs at, const int nattn) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // Every kernel argument either role needs is fetched HERE, in one batch: left alone, hipcc sinks the argument loads into the
  // role branches and fetches them in four or five dependent rounds (the scalar cache is cold at the start of a launch: ~0.2 us a
  // round, ~1 us before the first weight request -- seen with tools/stamps_fused.py).  An empty asm statement that names the
  // values as scalar inputs makes them due before the branch.
#define L2_PIN4(x0, x1, x2, x3) asm volatile("" ::"s"(x0), "s"(x1), "s"(x2), "s"(x3))
  L2_PIN4(a.w0, a.w1, a.w2, a.in.addr()); L2_PIN4(a.emb, a.rmsw, a.out.addr(), a.out_k.addr()); L2_PIN4(a.out_v.addr(), a.fr, a.fi, a.tokpos);
  L2_PIN4(a.n, a.rows, a.dim, a.kv_dim); L2_PIN4(a.head_size, a.gran, a.gran_ep, a.inv_n); asm volatile("" ::"s"(a.gran_hmagic)); asm volatile("" ::"s"(a.aux.addr()), "s"(a.aux2.addr()));
  L2_PIN4(at.kc.addr(), at.vc.addr(), at.att.addr(), at.xb.addr()); L2_PIN4(at.tokpos, at.part.addr(), at.counter, at.dim); L2_PIN4(at.head_size, at.seq_len, at.n_heads, at.nsplit);
  asm volatile("" ::"s"(at.kv_dim), "s"(at.kv_mul)); L2_PIN4(at.inv_sqrt_hs, at.gran, at.gran_ep, at.herr); L2_PIN4(at.wait_ticks, at.exact, nattn, at.dbg);
#undef L2_PIN4
  const int nq = (int)gridDim.x - nattn;
  if ((int)blockIdx.x < nq) { phase_small_body<MODE_QKV, XV, 2>(a, smem, blockIdx.x, nq); return; }
  int sp = 0, h = (int)blockIdx.x - nq;
  while (h >= at.n_heads) { h -= at.n_heads; ++sp; }               // (split, head) without a division: at most nsplit steps
  const int pos = at.tokpos[1];
  // up to 128 rows: FOUR waves (one per SIMD: a dependent fp64 instruction then issues as soon as its operands are there, and a
  // barrier has four waves to collect, not eight), the other four leave at once; beyond: eight waves, one round of 256 rows
  if (at.nsplit == 1 && pos + 1 <= 4 * NT * (64 / LR) && at.fused_four_waves) {
    if (threadIdx.x >= 256) return;
    attn_tile_dispatch<LR, 4, NT, true>(at, smem, h, sp, pos);
    return;
  }
  attn_tile_dispatch<LR, 8, NT, true>(at, smem, h, sp, pos);
}

// ------------------------------------------------------------------------------------------------
// ONE launch for attention and the wo GEMV of a TENSOR-PARALLEL rank (llama2.ts:244-267 -> 270).  wo is column-sharded, so a rank's
// wo reads only the outputs of the rank's own H / G heads: the first `nattn` workgroups are those heads' attention workgroups
// (the plain tile kernel, its output ALSO published as granules), the others are the latency-form wo phase whose x wave gathers the
// d / G granules while its compute waves' weights -- which depend on nothing -- arrive: the rank's whole wo shard (8 MB at 8 ranks)
// streams while the chip would otherwise idle behind four attention workgroups, and one launch boundary goes.  Rows leave through
// tp_push_row like those of the wo launch.  Producers have the lower block ids (dispatched first); the wait is bounded anyway.
template <int XV, int R, int LR, int NW, int NT>
__global__ void __launch_bounds__(512) attn_wo_kernel(const AttnArgs at, const PhaseArgs wo, const int nattn) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((int)blockIdx.x >= nattn) { phase_small_body<MODE_WO, XV, R, true>(wo, smem, (int)blockIdx.x - nattn, (int)gridDim.x - nattn); return; }
  if (NW * 64 < 512 && (int)threadIdx.x >= NW * 64) return;          // (a four-wave attention form in an eight-wave launch)
  int sp = 0, h = (int)blockIdx.x;
  while (h >= at.n_heads) { h -= at.n_heads; ++sp; }
  const int pos = at.tokpos[1];
  attn_tile_dispatch<LR, NW, NT>(at, smem, h, sp, pos);
}

template <int LR, int NW, int NT>
__global__ void __launch_bounds__(64 * NW) attn_tile_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // every argument in ONE fetch round, and the position (device memory: one captured graph serves every position) requested right
  // behind it -- left alone, hipcc fetches the arguments in two dependent rounds before it even asks for pos (see qkv_attn_small_kernel)
  asm volatile("" ::"s"(a.q.addr()), "s"(a.kc.addr()), "s"(a.vc.addr()), "s"(a.att.addr()), "s"(a.xb.addr()), "s"(a.tokpos), "s"(a.part.addr()), "s"(a.counter));
  const int pos = a.tokpos[1];
  asm volatile("" ::"s"(a.dim), "s"(a.head_size), "s"(a.seq_len), "s"(a.n_heads), "s"(a.nsplit), "s"(a.kv_dim), "s"(a.kv_mul), "s"(a.inv_sqrt_hs), "s"(a.gout), "s"(a.gout_ep));
  attn_tile_dispatch<LR, NW, NT>(a, smem, blockIdx.x, blockIdx.y, pos);
}

// Prefill: grid (head, query).  Query p of the chunk sits at position pos0 + p and sees cache rows 0..pos0+p,
// all written by the chunk's QKV GEMM in an earlier launch.
template <int LR, int NW, int NT>
__global__ void __launch_bounds__(64 * NW) pf_attn_tile_kernel(const AttnArgs a, int pos0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  AttnArgs b = a;
  const int p = blockIdx.y;
  b.q = a.q + (size_t)p * a.dim;
  b.xb = a.xb + (size_t)p * a.dim;
  b.att = Mut<float>(nullptr);
  b.nsplit = 1;
  attn_tile_dispatch<LR, NW, NT>(b, smem, blockIdx.x, 0, pos0 + p);
}

#ifndef L2_NO_PLAIN_KERNELS
// Shapes whose head_size or dim is not a multiple of 4 (rows are not 16-byte aligned): one workgroup per head,
// one thread per timestep, scalar loads.  Correctness only; keeps every rounding of the reference.
__global__ void __launch_bounds__(256) attn_scalar_kernel(const AttnArgs a, int pos0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int S = a.seq_len, hs = a.head_size, dim = a.dim, kvd = a.kv_dim, h = blockIdx.x, hk = h / a.kv_mul, tid = threadIdx.x;
  float* att = reinterpret_cast<float*>(smem);                    // S floats
  double* red = reinterpret_cast<double*>(att + ((S + 3) & ~3));  // 8 doubles
  const int pos = a.pos_plus1 ? pos0 + (int)blockIdx.y : a.tokpos[1];
  const Mut<const float> q = a.q + ((a.pos_plus1 ? (size_t)blockIdx.y * dim : 0) + (size_t)h * hs);
  const Mut<float> xb = a.xb + ((a.pos_plus1 ? (size_t)blockIdx.y * dim : 0) + (size_t)h * hs);
  const double rsq = sqrt((double)hs);
  for (int t = tid; t <= pos; t += 256) {
    const Mut<const float> kp = a.kc + ((size_t)t * kvd + (size_t)hk * hs);
    double s = 0.0;
    for (int i = 0; i < hs; ++i) s += (double)q.ld(i) * (double)kp.ld(i);
    att[t] = (float)(s / rsq);
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int t = tid; t <= pos; t += 256) mx = fmaxf(mx, att[t]);
  mx = wave_max(mx);
  float* redf = reinterpret_cast<float*>(red);
  if ((tid & 63) == 0) redf[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
  __syncthreads();
  double lsum = 0.0;
  for (int t = tid; t <= pos; t += 256) { const float e = (float)exp((double)att[t] - (double)mx); att[t] = e; lsum += (double)e; }
  const double sum = block_sum(lsum, red, tid, 256);
  for (int t = tid; t <= pos; t += 256) {
    const float pr = (float)((double)att[t] / sum);
    att[t] = pr;
    if (a.att && !a.pos_plus1) a.att.st((size_t)h * S + t, pr);
  }
  __syncthreads();
  for (int i = tid; i < hs; i += 256) {
    const Mut<const float> vp = a.vc + ((size_t)hk * hs + i);
    if (a.exact) {
      float o = 0.0f;
      for (int t = 0; t <= pos; ++t) o = (float)((double)o + (double)att[t] * (double)vp.ld((size_t)t * kvd));
      xb.st(i, o);
    } else {
      double o = 0.0;
      for (int t = 0; t <= pos; ++t) o += (double)att[t] * (double)vp.ld((size_t)t * kvd);
      xb.st(i, (float)o);
    }
  }
}

#endif  // L2_NO_PLAIN_KERNELS

}  // namespace l2k

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
#include <type_traits>

namespace l2k {

typedef float f4 __attribute__((ext_vector_type(4)));

enum { MODE_QKV = 0, MODE_WO = 1, MODE_W13 = 2, MODE_W2 = 3, MODE_CLS = 4 };

// Hand-off counters sit one per 128-byte line: atomics on one line serialise at ~12 ns each at the memory side
// (6144 adds on one line cost a 7B layer 74 us; spread over 32 lines they overlap).
enum { CTR_STRIDE = 32 };

// Diagnostic build only (-DL2_STAMPS, tools/stamps.py): shader-clock stamps of waves 0, 1 and the last one of the first / middle /
// last workgroup, kept in scalar registers and written out when the wave ends -- a stamp must not put a store in the
// wave's vector-memory queue (hipcc then waits for that store wherever it next waits for a load, and under the
// weight stream a store takes microseconds to complete: the first version of these stamps measured mostly itself).
// The values go to a buffer nothing else reads.  In the product build STAMP() is empty.
#ifdef L2_STAMPS
#define L2_NSTAMP 12
struct Stamps {
  unsigned long long t[L2_NSTAMP];
  unsigned long long* dst;
  unsigned long long* wg;      // every workgroup: {start, end} on the constant 100 MHz clock all XCDs share (s_memrealtime)
  unsigned long long rt0;
  __device__ __forceinline__ Stamps(unsigned long long* dbg, unsigned long long* dbg_wg = nullptr, int force_sel = -2, bool last_wave = false) {
    // (latency form: wave 0 is the x wave and leaves at the barrier -- the workgroup's lifetime is taken from its last wave)
    wg = (dbg_wg && threadIdx.x == (last_wave ? blockDim.x - 64 : 0) && blockIdx.y == 0 && blockIdx.x < 1024) ? dbg_wg + 2 * blockIdx.x : nullptr;
    rt0 = __builtin_readcyclecounter();
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0));
#pragma unroll
    for (int k = 0; k < L2_NSTAMP; ++k) t[k] = 0;
    const int b = blockIdx.x, nb = gridDim.x, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int sel = force_sel != -2 ? force_sel : (b == 0) ? 0 : (b == nb / 2) ? 1 : (b == nb - 1) ? 2 : -1;
    const int ws = (w == 0) ? 0 : (w == 1) ? 1 : (w == nw - 1) ? 2 : -1;     // waves 0, 1 and the last one
    dst = (dbg && sel >= 0 && ws >= 0 && (threadIdx.x & 63) == 0 && blockIdx.y == 0) ? dbg + (sel * 3 + ws) * L2_NSTAMP : nullptr;
  }
  __device__ __forceinline__ ~Stamps() {
    if (wg) {
      unsigned long long rt1;
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1));
      unsigned xcc, hwid;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      wg[0] = rt0; wg[1] = (rt1 - rt0) | ((unsigned long long)(xcc & 0xf) << 32) | ((unsigned long long)(hwid & 0xffff00) << 36);
    }
    if (dst) {
#pragma unroll
      for (int k = 0; k < L2_NSTAMP; ++k) dst[k] = t[k];
      dst[L2_NSTAMP - 1] = rt0;      // the wave's start on the 100 MHz clock all XCDs share (stamp 11 is not used by any kernel)
    }
  }
};
#define STAMP_INIT(dbg) Stamps st_(dbg)
#define STAMP_INIT_WG(dbg, wg) Stamps st_(dbg, wg)
#define STAMP_INIT_WGL(dbg, wg) Stamps st_(dbg, wg, -2, true)
#define STAMP_INIT_SEL(dbg, sel) Stamps st_(dbg, nullptr, sel)
#define STAMP(k) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_.t[k])::"memory")
#else
#define STAMP_INIT(dbg) do { } while (0)
#define STAMP_INIT_WG(dbg, wg) do { } while (0)
#define STAMP_INIT_WGL(dbg, wg) do { } while (0)
#define STAMP_INIT_SEL(dbg, sel) do { } while (0)
#define STAMP(k) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// write-through (sc1) stores / L1-bypassing loads: relaxed agent-scope atomics (split-attention hand-off)
__device__ __forceinline__ void st_sc1(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Diagnostic build ONLY (-DL2_COHERENCE_BREAK, `make coherence_break`: never shipped): every load the coherence rule sends past L1 -- the
// element loads below, the 16-byte buffer loads of activations (L2_ACT_LD4) and of cache rows (attention.hip.h) -- becomes a PLAIN cached
// load.  It exists to prove that tests/test_coherence_gpu.py fails when the rule is broken (L2_TEST_COHERENCE_BREAK=1).
#ifdef L2_COHERENCE_BREAK
#define L2_SC1_AUX 0
#else
#define L2_SC1_AUX 16      // the sc1 bit of a buffer load's cache-policy field
#endif
__device__ __forceinline__ double ld_sc1(const double* p) {
#ifdef L2_COHERENCE_BREAK
  return *(const volatile double*)p;
#else
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#endif
}
__device__ __forceinline__ void st_sc1(float* p, float v) {
  __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_sc1(const float* p) {
#ifdef L2_COHERENCE_BREAK
  return *(const volatile float*)p;
#else
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#endif
}

// The coherence rule (stated in full below, "COHERENCE RULE of the decode step's kernels") as a TYPE.  A pointer to bytes that a launch of
// the run may have written -- activations, cache rows, split-attention partials, probabilities kept for parity reads -- travels in the
// kernel-argument structs as Mut<T>: it has NO operator* and NO operator[], so a plain (L1-cached) load of such a byte does not compile
// in a kernel.  What it hands out instead: ld(i) = one element past L1 (a relaxed agent-scope atomic load), rsrc(n) = the buffer
// descriptor the 16-byte sc1 loads go through (L2_ACT_LD4, the attention tiles), st(i, v) = a PLAIN store (stores stay plain under the
// rule), st4 = 16 bytes of them, st_sc1(i, v) = a write-through store (hand-offs inside one launch), addr() = the raw address -- for the
// host, and in device code only to NAME the value in an `asm volatile("" :: "s"(...))` pin (tests/test_abi_cpu.py fails on any other use).
// Same size and layout as the pointer it replaces (the kernel-argument blocks did not move).
template <class T>
struct Mut {
  T* p;
  Mut() = default;
  __host__ __device__ Mut(T* q) : p(q) {}
  __host__ __device__ explicit operator bool() const { return p != nullptr; }
  __host__ __device__ Mut operator+(size_t i) const { return Mut(p + i); }
  __host__ __device__ T* addr() const { return p; }
  __device__ __forceinline__ auto ld(size_t i = 0) const { return ld_sc1(p + i); }
  __device__ __forceinline__ void st(size_t i, T v) const { p[i] = v; }
  __device__ __forceinline__ void st_sc1(size_t i, T v) const { l2k::st_sc1(p + i, v); }
  __device__ __forceinline__ void st4(size_t i4, f4 v) const { reinterpret_cast<f4*>(p)[i4] = v; }      // 16 bytes, plain (T = float, 16-byte aligned base)
  // buffer descriptor over `bytes` bytes from here (loads past its end read as zeros)
  __device__ __forceinline__ auto rsrc_bytes(unsigned bytes) const { return __builtin_amdgcn_make_buffer_rsrc(const_cast<typename std::remove_const<T>::type*>(p), 0, bytes, 0x00020000); }
  __device__ __forceinline__ auto rsrc(unsigned n_elems) const { return rsrc_bytes(n_elems * (unsigned)sizeof(T)); }
};
static_assert(sizeof(Mut<float>) == sizeof(float*), "Mut<T> is the pointer it wraps");

struct PhaseArgs {
  const float* w0;    // QKV: wq[l]   W13: w1[l]   else: the matrix
  const float* w1;    // QKV: wk[l]   W13: w3[l]
  const float* w2;    // QKV: wv[l]
  Mut<const float> in; // input vector: x (QKV, W13, CLS), xb (WO), hb (W2) -- written by the launch before (Mut: no plain load)
  const float* emb;   // token_embedding_table when the input/residual is the embedding row (layer 0), else null
  const float* rmsw;  // rmsnorm weight (QKV, W13, CLS) or null
  Mut<float> out;      // QKV: q   WO/W2: x   W13: hb   CLS: logits
  Mut<float> out_k;    // QKV: key_cache   + l*S*d
  Mut<float> out_v;    // QKV: value_cache + l*S*d
  Mut<float> aux;      // CLS: final-normed x (llama2.ts:299)  QKV: k  W13: hb2  WO: xb2  W2: xb (parity reads; may be null)
  Mut<float> aux2;     // QKV: v (parity reads; may be null)   CLS: host-mapped pinned logits (zero-copy hand-off) or null
  Mut<const float> res; // WO/W2: residual source x
  const float* fr;    // freq_cis_real
  const float* fi;    // freq_cis_imag
  const int* tokpos;  // {token, pos, step, _}
  int n;              // input length (columns)
  int rows;           // output rows (QKV: dim + 2*kv_dim, W13: hidden (pairs of w1/w3 rows))
  int dim;
  int kv_dim;         // QKV: rows of wk / wv = floats of a cache row (= dim unless the context honours n_kv_heads < n_heads)
  int head_size;
  // tensor parallel: column-sharded WO/W2 write fp64 partials instead of x
  double* partial;    // non-null (WO/W2 of a tensor-parallel rank): out[i] is not written, partial[i] = fp64 sum
  // ... or, with the one-shot peer-to-peer exchange (tp_exchange.hip.h), pushed straight into every peer's inbox by the lanes of the
  // wave that reduced it: `push` = device table of the peers' inboxes, `push_epoch` = the exchange counter the tags come from
  const struct TpPush* push;   // (the table also names the exchange counter the tags come from)
  double inv_n;       // 1.0 / n, correctly rounded by the host (rmsnorm's mean, llama2.ts:174)
  int rot;            // streaming form: row group g starts its rows at column batch (g * rot) % batches and wraps (0: every row from column 0)
  const float* wp;    // streaming form: this launch's matrix (matrices) repacked in the order the chip consumes it (pack_kernel), or null
  unsigned long long* amax;  // CLS of the greedy loop: 8 argmax keys (one per 128-byte line) the workgroups fold their best logit into, or null
  unsigned long long* gran;  // QKV of the fused QKV + attention launch: [dim + 2 kv_dim] hand-off granules {value, tag} for q, k, v of this position, or null
  const unsigned* gran_ep;   // per head: the launch counter the granule tags of that head come from (tag = gran_ep[h] + 1)
  unsigned gran_hmagic;      // ceil(2^20 / head_size): row / head_size without a division
  // fused attention + wo launch of a tensor-parallel rank (attention.hip.h: attn_wo_kernel; WO only): the input vector arrives as granules
  // from the attention workgroups of the SAME launch -- `gran` are those granules, tag = *gran_ep + 1 (the launch after this one advances
  // the counter), and a wait that gives up says so here
  int* gin_herr;
  // greedy loop with the pick folded into the step (no launch of its own): non-null in the three launches that share the work -- layer 0's
  // q / k / v launch takes the token from the classifier's argmax keys (`amax`) and records it here, layer 0's wo launch reads it back from
  // here and re-arms the keys, the classifier launch advances {pos, step} (greedy_* below)
  int* tok_out;
#ifdef L2_STAMPS
  unsigned long long* dbg;  // diagnostic stamps (L2_STAMPS builds only: the argument block of the product build stays within four 64-byte
  unsigned long long* dbg_wg;  // lines -- a fifth cost every launch of the small models ~0.1 us, 2 % of a stories110M token)
#endif
};
static_assert(sizeof(void*) != 8 || sizeof(PhaseArgs) <= 224 + 16, "PhaseArgs: keep the kernel-argument block within four 64-byte lines");

// Tensor-parallel push (tp_exchange.hip.h): where the fp64 partial of a row goes.  gin[r] = rank r's inbox of granule pairs,
// [2 parities][MAXG sources][n] x 16 bytes, mapped into this process (uncached memory); in a shard-timing context every "peer" is this rank.
struct TpPush { unsigned long long* gin[8]; const unsigned long long* epoch; int G, rank, n, solo; };

// One fp64 partial = two hand-off granules {low word, tag}, {high word, tag} written by ONE 16-byte system-scope store (sc0 sc1: past
// L1 and L2, over xGMI for a peer's memory); each 8-byte half is its own flag (MI355X guide, recipe R2: 8-byte halves of a 16-byte
// sc1 store are observed untorn), so the payload needs no flag, no drain and no barrier behind it.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void tp_push_store(unsigned long long* slot, double v, unsigned tag) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const u32x4 w = {(unsigned)b, tag, (unsigned)(b >> 32), tag};
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(slot), "v"(w) : "memory");
}
// What a wave needs to push its rows, fetched ONCE when the kernel starts (the table load must not queue behind the weight stream):
// lane r's target inbox, the group's shape, the number of this exchange (read past L1: the launch before advanced it; a scalar load
// would see a stale copy under graph replay).
struct PushCtx { unsigned long long* gin = nullptr; int G = 0, rank = 0, n = 0, solo = 0; unsigned e = 0; };
__device__ __forceinline__ PushCtx tp_push_ctx(const TpPush* p, int lane) {
  PushCtx c;
  // (uniform values pinned to scalar registers: the context lives through the whole GEMV loop, where vector registers are the budget)
  c.G = __builtin_amdgcn_readfirstlane(p->G); c.rank = __builtin_amdgcn_readfirstlane(p->rank);
  c.n = __builtin_amdgcn_readfirstlane(p->n); c.solo = __builtin_amdgcn_readfirstlane(p->solo);
  c.gin = p->gin[c.solo ? c.rank : min(lane, c.G - 1)];
  c.e = __builtin_amdgcn_readfirstlane((unsigned)__hip_atomic_load(p->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) + 1u;
  return c;
}
// lane r < G of the wave that holds the reduced sum stores it into rank r's inbox, slot [parity of the exchange][this rank][row i]
__device__ __forceinline__ void tp_push_row(const PushCtx& c, int i, double v, int lane) {
  if (lane < c.G) tp_push_store(c.gin + 2 * ((size_t)((c.e & 1u) * 8u + (unsigned)(c.solo ? lane : c.rank)) * (size_t)c.n + (size_t)i), v, c.e);
}

// Hand-off granule of the fused QKV + attention launch (attention.hip.h): ONE naturally aligned 8-byte word {fp32 value, tag},
// written by ONE write-through store (relaxed, agent scope = sc1) and only ever read by L1-bypassing loads: the data is its own
// flag, no fence on either side (MI355X guide, hand-off recipe R2).  tag = number of this launch (a device counter the launch
// itself advances when its last head is done), so a granule of an earlier launch never matches and nothing has to be zeroed.
__device__ __forceinline__ void granule_store(unsigned long long* g, float v, unsigned tag) {
  __hip_atomic_store(g, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Consumer side of a vector handed over as granules: float4 number u * 64 + lane of the vector, u < NU, = four consecutive granules =
// two 16-byte loads past L1 (sc1); ALL loads of a pass are requested together, a pass is repeated until every lane has `tag` on all of
// its granules.  Lanes past the vector's end re-read its last float4 (and see the same tags).  Bounded on the 100 MHz clock:
// false = gave up (*herr is set).  Only ONE wave of a workgroup ever does this (MI355X guide, row polling-cost).
typedef unsigned gu4 __attribute__((ext_vector_type(4)));
template <int NU>
__device__ __forceinline__ bool granules_gather_f4(const unsigned long long* g, int n, int lane, unsigned tag, f4 (&xr)[NU], int* herr, unsigned long long wait_ticks, int nap, unsigned* dead) {
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned long long*>(g), 0, (unsigned)n * 8u, 0x00020000);
  const int n4 = n >> 2;
  unsigned off[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) off[u] = (unsigned)min(u * 64 + lane, n4 - 1) * 32u;
  unsigned spins = 0;
  unsigned long long t0 = 0;
  for (;;) {
    gu4 lo[NU], hi[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      lo[u] = __builtin_bit_cast(gu4, __builtin_amdgcn_raw_buffer_load_b128(rs, off[u], 0, 16));           // aux 16 = sc1
      hi[u] = __builtin_bit_cast(gu4, __builtin_amdgcn_raw_buffer_load_b128(rs, off[u] + 16u, 0, 16));
    }
    // `dead`: device-side error word -- an earlier wait of this context gave up: do not wait again (the host clears it when it reports the
    // error).  It rides along with the first sweep: a test in front of it would be a round trip of its own.
    if (spins == 0 && __builtin_amdgcn_readfirstlane(__hip_atomic_load(dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
#pragma unroll
      for (int u = 0; u < NU; ++u) xr[u] = f4{0.f, 0.f, 0.f, 0.f};
      return false;
    }
    bool ok = true;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      ok = ok & (lo[u].y == tag) & (lo[u].w == tag) & (hi[u].y == tag) & (hi[u].w == tag);      // (bitwise: no branch per granule)
      xr[u].x = __uint_as_float(lo[u].x); xr[u].y = __uint_as_float(lo[u].z); xr[u].z = __uint_as_float(hi[u].x); xr[u].w = __uint_as_float(hi[u].z);
    }
    if (__all(ok)) return true;
    if (nap == 1) __builtin_amdgcn_s_sleep(4); else if (nap == 2) __builtin_amdgcn_s_sleep(16);      // (a long wait: fewer sweeps in the producers' way)
    if ((++spins & 63u) == 0) {
      const unsigned long long now = __builtin_amdgcn_s_memrealtime();
      if (!t0) t0 = now;
      else if (now - t0 > wait_ticks) { if (lane == 0) { *herr = 1; __hip_atomic_store(dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } return false; }
    }
  }
}

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

// Two sums at once: the six steps of each are interleaved so that neither chain waits for the other's DPP hazards.
__device__ __forceinline__ void wave_sum2(double& a, double& b) {
  a += dpp_f64<0xB1, 0xf>(a);  b += dpp_f64<0xB1, 0xf>(b);
  a += dpp_f64<0x4E, 0xf>(a);  b += dpp_f64<0x4E, 0xf>(b);
  a += dpp_f64<0x141, 0xf>(a); b += dpp_f64<0x141, 0xf>(b);
  a += dpp_f64<0x140, 0xf>(a); b += dpp_f64<0x140, 0xf>(b);
  a += dpp_f64<0x142, 0xa>(a); b += dpp_f64<0x142, 0xa>(b);
  a += dpp_f64<0x143, 0xc>(a); b += dpp_f64<0x143, 0xc>(b);
  a = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a), 63), __builtin_amdgcn_readlane(__double2loint(a), 63));
  b = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(b), 63), __builtin_amdgcn_readlane(__double2loint(b), 63));
}

// rmsnorm's scale 1 / sqrt(1e-5 + ss / n) (llama2.ts:174-175).  The reference's three operations (divide, sqrt,
// divide) are each correctly rounded; as library calls they are ~70 DEPENDENT fp64 instructions, 2 500 - 4 000 cycles
// on the one wave every other wave of the workgroup is waiting for (profiles/r02/stamps_phase_110M_xwave.txt).
// Here: one multiply by the host's correctly rounded 1/n, the hardware reciprocal-square-root seed and two Newton
// steps -- within 2 ulp of the same real number, the same distance the tree-ordered sum(x^2) already has from the
// reference's sequential one; every value derived from it is rounded to fp32 before it is used.
__device__ __forceinline__ double rms_scale(double ss, double inv_n) {
  const double y = 1e-5 + ss * inv_n;
  double r = __builtin_amdgcn_rsq(y);
  const double h = 0.5 * y;
  r = fma(r, fma(-h * r, r, 0.5), r);
  r = fma(r, fma(-h * r, r, 0.5), r);
  return r;
}

// 1 / x within 2 ulp: hardware seed + two Newton steps (every quotient formed with it is rounded to fp32 before use)
__device__ __forceinline__ double rcp_fast(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(r, fma(-x, r, 1.0), r);
  r = fma(r, fma(-x, r, 1.0), r);
  return r;
}

// exp(x) within ~2 ulp, as a SHORT dependency chain: the reference's Math.exp result is rounded to fp32 at once
// (softmax :187, SwiGLU :285), and on this chip a dependent fp64 instruction costs ~30 cycles when the wave has
// nothing else to issue -- the library's Horner form is ~30 of them in a row, this one ~10.  x = k ln2 + r,
// |r| <= ln2 / 2; the degree-13 Taylor polynomial (remainder < 6e-18) is evaluated pairwise (Estrin); 2^k by ldexp
// (overflow -> inf, underflow -> denormal / 0, like the library).
__device__ __forceinline__ double exp_fast(double x) {
  x = fmin(fmax(x, -1100.0), 1100.0);                    // keeps k inside ldexp's range; exp is 0 / inf out there
  const double k = rint(x * 1.4426950408889634074);
  double r = fma(-k, 6.93147180369123816490e-01, x);
  r = fma(-k, 1.90821492927058770002e-10, r);
  const double r2 = r * r, r4 = r2 * r2, r8 = r4 * r4;
  const double a0 = fma(r, 1.0, 1.0);
  const double a1 = fma(r, 1.0 / 6.0, 0.5);
  const double a2 = fma(r, 1.0 / 120.0, 1.0 / 24.0);
  const double a3 = fma(r, 1.0 / 5040.0, 1.0 / 720.0);
  const double a4 = fma(r, 1.0 / 362880.0, 1.0 / 40320.0);
  const double a5 = fma(r, 1.0 / 39916800.0, 1.0 / 3628800.0);
  const double a6 = fma(r, 1.0 / 6227020800.0, 1.0 / 479001600.0);
  const double b0 = fma(a1, r2, a0), b1 = fma(a3, r2, a2), b2 = fma(a5, r2, a4);
  const double d0 = fma(b1, r4, b0), d1 = fma(a6, r4, b2);
  return ldexp(fma(d1, r8, d0), (int)k);
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
// COHERENCE RULE of the decode step's kernels.  On the library's own queue (aql_queue.h) the launches of a token carry a RELEASE
// fence only (the command processor writes the L2s' dirty lines back when a launch ends; the chip then keeps its eight L2s
// coherent by itself) and NO acquire: nothing invalidates a CU's vector L1 between two launches.  So every VECTOR load of a byte
// that an earlier launch of the run wrote -- activations, cache rows, launch counters, argmax keys -- goes past L1 (sc1: a buffer
// load with the sc1 bit for 16 bytes, a relaxed agent-scope atomic load for a word).  Stores stay plain.  {token, pos, step} change
// once per token, in its last launch, and are read through the SCALAR cache as before: the first launch of every token acquires at
// agent scope, which refreshes it.  Weights, norm weights, RoPE tables and the embedding table are immutable during a run and keep
// their plain / non-temporal loads.  (Measured on the way, tools/aql/microbench_aql.cpp + profiles/r05/aql_*: write-through stores
// instead of the release fence cost every launch 0.3 - 0.9 us -- a 4-byte sc1 store is a fabric write of its own; {token, pos} as
// vector loads queue behind the weight requests of the latency form and delay its epilogue operands.)
// Round 6, measured: this chip does not NEED the load half of the rule.  With every load below turned into a plain cached one
// (-DL2_COHERENCE_BREAK) and a kernel behind every launch that fills every CU's L1 with the step's mutable lines (l1_pollute_kernel), the
// step is still bit-identical (profiles/r06/coherence_adversary.txt); directly: 2.8e9 plain loads of lines the same CUs had pulled two
// dispatches earlier, no acquire: 0 stale (profiles/r06/l1_across_dispatches.txt) -- on gfx950 / ROCm 7.2 a dispatch does not see
// vector-L1 lines of an earlier dispatch.  The rule stays: it is what the HSA memory model requires of packets without an acquire fence,
// it costs nothing (the sc1 form is 1 - 2 % FASTER at the small models, equal at 7B: profiles/r06/plain_vs_sc1_loads_ab.txt), the Mut<T>
// type below enforces it at compile time and tests/test_coherence_gpu.py would catch a chip or firmware that does carry lines over.
//
// a launch counter of the fused launch (it changes from launch to launch): ONE vector load past L1, broadcast through a scalar register
__device__ __forceinline__ unsigned ld_word(const unsigned* p) {
  return (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// 16 bytes of an activation vector past L1: a buffer load with the sc1 bit (aux 16), tracked by the compiler's wait counts like any
// other load; the descriptor ends at the vector's end (elements past it read as zeros)
#define L2_ACT_RSRC(ptr, n_floats) __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ptr), 0, (unsigned)(n_floats) * 4u, 0x00020000)
#define L2_ACT_LD4(rs, idx4) __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(idx4) * 16u, 0, L2_SC1_AUX))

// QKV row groups: all q rows (dim), then k, then v (kv_dim each).
__device__ __forceinline__ void qkv_group(const PhaseArgs& a, int g, int R, int& m, int& i0) {
  const int row0 = g * R;
  m = (row0 >= a.dim) + (row0 >= a.dim + a.kv_dim);   // compares, not a division: this runs once per batch
  i0 = row0 - (m >= 1 ? a.dim : 0) - (m == 2 ? a.kv_dim : 0);
}

template <int MODE, int R>
__device__ __forceinline__ void row_ptrs(const PhaseArgs& a, int g, int n, const float* (&rp)[R]) {
  if (MODE == MODE_QKV) {
    int m, i0;
    qkv_group(a, g, R, m, i0);
    const float* base = (m == 0) ? a.w0 : (m == 1) ? a.w1 : a.w2;
#pragma unroll
    for (int r = 0; r < R; ++r) rp[r] = base + (size_t)min(i0 + r, (m == 0 ? a.dim : a.kv_dim) - 1) * n;
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

// argmax (llama2.ts:364-366: `arr.reduce((maxIdx, val, idx, array) => (val > array[maxIdx] ? idx : maxIdx), 0)` -- first maximum, strict
// '>'): a logit and its index travel as ONE 64-bit key (order-preserving bits of the value, then ~index), so "largest value, smallest
// index" is an unsigned maximum.  NaN never wins a '>' (key bits 0: below -inf) -- EXCEPT at index 0, where the reduce() starts: no
// `val > NaN` is ever true, so the pick stays 0 whatever the other logits are (the all-ones key: above +inf, and it decodes to 0).
__device__ __forceinline__ unsigned long long argmax_key(float v, int i) {
  v = v + 0.0f;                                          // -0 -> +0: '>' does not tell them apart
  const unsigned u = __float_as_uint(v);
  const unsigned o = (v != v) ? (i == 0 ? 0xffffffffu : 0u) : ((u & 0x80000000u) ? ~u : (u | 0x80000000u));
  return ((unsigned long long)o << 32) | (unsigned)(~i);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_max_u64(unsigned long long v) {   // lanes outside ROW_MASK keep v
  const int lo = (int)(unsigned)v, hi = (int)(unsigned)(v >> 32);
  const unsigned nlo = (unsigned)__builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
  const unsigned nhi = (unsigned)__builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
  const unsigned long long o = ((unsigned long long)nhi << 32) | nlo;
  return o > v ? o : v;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
  v = dpp_max_u64<0xB1, 0xf>(v);
  v = dpp_max_u64<0x4E, 0xf>(v);
  v = dpp_max_u64<0x141, 0xf>(v);
  v = dpp_max_u64<0x140, 0xf>(v);
  v = dpp_max_u64<0x142, 0xa>(v);
  v = dpp_max_u64<0x143, 0xc>(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
  return ((unsigned long long)hi << 32) | lo;
}


// ------------------------------------------------------------------------------------------------
// The greedy pick folded into the step (llama2.ts:477-479, 364-366, 496: next = argmax(logits); token = next; pos++).  The classifier's
// workgroups fold their best (logit, index) into eight keys (below); instead of a one-wave launch that finishes the pick, the NEXT token's
// first launch does: every workgroup of layer 0's q / k / v launch takes the maximum of the eight keys itself (eight loads past L1, one DPP
// reduction -- requested beside the scalar load of {pos, step}, so the embedding row is asked for no later than before) and workgroup 0
// records the token; layer 0's wo launch, which needs the token again for its residual (the embedding row), reads the record and re-arms
// the keys (every reader of them is a launch back by then); the classifier launch, none of whose workgroups reads {pos, step} and behind
// which nothing of this token runs, advances them.  step == 0: the run's first token comes from the host ({token, pos} as uploaded).  One
// launch per token fewer; the run's last pick is finished by a launch of its own (argmax_last_kernel).
__device__ __forceinline__ int greedy_token_from_keys(const PhaseArgs& a, int lane, int step) {
  unsigned long long k = lane < 8 ? __hip_atomic_load(a.amax + (size_t)lane * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
  k = wave_max_u64(k);
  const int from_keys = (k == 0) ? 0 : (int)~(unsigned)k;       // nothing but NaN: reduce() keeps index 0
  return step > 0 ? from_keys : a.tokpos[0];
}

// Epilogue operands that do not depend on the GEMV (RoPE table entries of the row pair, residual value of the row):
// lane p's operands for pair / row p of row group g.  The latency kernel requests them with the weights; the
// streaming kernel loads them in the epilogue (there the extra live registers cost more than the L2 round trip).
struct EpiPre { float e0, e1; unsigned tag; };   // tag: hand-off granules of the fused QKV + attention launch

template <int MODE, int R>
__device__ __forceinline__ EpiPre epi_prefetch(const PhaseArgs& a, int g, int lane, int token, int pos) {
  EpiPre e = {0.0f, 0.0f, 0u};
  if (MODE == MODE_QKV) {
    int m, i0;
    qkv_group(a, g, R, m, i0);
    const int p = min(lane, R / 2 > 0 ? R / 2 - 1 : 0);
    const int i = min(i0 + 2 * p, (m == 0 ? a.dim : a.kv_dim) - 2);
    const int idx = pos * (a.head_size / 2) + (i % a.head_size) / 2;
    e.e0 = a.fr[idx]; e.e1 = a.fi[idx];
  } else if (MODE == MODE_WO || MODE == MODE_W2) {
    if (!a.partial && !a.push) {
      const int i = min(g * R + min(lane, R - 1), a.rows - 1);
      e.e0 = (MODE == MODE_WO && a.emb) ? a.emb[(size_t)token * a.dim + i] : a.res.ld(i);
    }

  }
  return e;
}

// Epilogue of one row group; every lane holds every reduced sum, lane p finishes output / pair p.
template <int MODE, int R, bool PREF>
__device__ __forceinline__ void finish_group(const PhaseArgs& a, int g, const double (&acc)[R], int lane, int token, int pos, const EpiPre& pre,
                                             unsigned long long& best, const PushCtx& pc = PushCtx(), const bool pushing = false) {
  if (MODE == MODE_QKV) {
    int m, i0;
    qkv_group(a, g, R, m, i0);
    static_assert(MODE != MODE_QKV || R % 2 == 0, "RoPE rotates adjacent row pairs");
#pragma unroll
    for (int p = 0; p < R / 2; ++p) {
      if (lane == p && i0 + 2 * p < (m == 0 ? a.dim : a.kv_dim)) {
        const int i = i0 + 2 * p;
        const float s0 = (float)acc[2 * p], s1 = (float)acc[(2 * p + 1) % R];  // matmul store, llama2.ts:201
        if (m == 2) {  // v: straight into the cache row (llama2.ts:240)
          if (a.gran) { unsigned long long* gp = a.gran + a.dim + a.kv_dim + i; granule_store(gp, s0, pre.tag); granule_store(gp + 1, s1, pre.tag); }      // first: a workgroup of this launch waits for them
          const Mut<float> vc = a.out_v + (size_t)pos * a.kv_dim;
          vc.st(i, s0); vc.st(i + 1, s1);
          if (a.aux2) { a.aux2.st(i, s0); a.aux2.st(i + 1, s1); }
        } else {       // RoPE on the adjacent pair (llama2.ts:224-235)
          double fcr, fci;
          if (PREF) { fcr = pre.e0; fci = pre.e1; }
          else { const int idx = pos * (a.head_size / 2) + (i % a.head_size) / 2; fcr = a.fr[idx]; fci = a.fi[idx]; }
          const float o0 = (float)((double)s0 * fcr - (double)s1 * fci);
          const float o1 = (float)((double)s0 * fci + (double)s1 * fcr);
          if (a.gran) { unsigned long long* gp = a.gran + (m == 0 ? 0 : a.dim) + i; granule_store(gp, o0, pre.tag); granule_store(gp + 1, o1, pre.tag); }      // first: a workgroup of this launch waits for them
          if (m == 0) { a.out.st(i, o0); a.out.st(i + 1, o1); }
          else {        // k: cache row (llama2.ts:239)
            const Mut<float> kc = a.out_k + (size_t)pos * a.kv_dim;
            kc.st(i, o0); kc.st(i + 1, o1);
            if (a.aux) { a.aux.st(i, o0); a.aux.st(i + 1, o1); }
          }
        }
      }
    }
  } else if (MODE == MODE_W13) {
    const int row0 = g * (R / 2);
#pragma unroll
    for (int p = 0; p < R / 2; ++p) {
      if (lane == p && row0 + p < a.rows) {
        const float h1 = (float)acc[p], h3 = (float)acc[R / 2 + p];       // llama2.ts:280-281
        const double v = h1;
        const float sl = (float)(v * rcp_fast(1.0 + exp_fast(-v)));       // llama2.ts:285 (store #1)
        const float hv = (float)((double)sl * (double)h3);                  // llama2.ts:289 (store #2)
        a.out.st(row0 + p, hv);
        if (a.aux) a.aux.st(row0 + p, h3);
      }
    }
  } else if (MODE == MODE_CLS) {
    const int row0 = g * R;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (lane == r && row0 + r < a.rows) {
        const float lg = (float)acc[r];                                     // llama2.ts:302
        a.out.st(row0 + r, lg);
        if (a.aux2) a.aux2.st(row0 + r, lg);   // straight into the host's RunState.logits (pinned, mapped)
        const unsigned long long key = argmax_key(lg, row0 + r);
        best = key > best ? key : best;
      }
    }
  } else {  // WO / W2: matmul store then residual accum (llama2.ts:270-273, 292-295)
    const int row0 = g * R;
    if (pushing) {
      // tensor parallel, peer-to-peer exchange: every lane holds every sum; lane r hands row i to rank r
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (row0 + r < a.rows) tp_push_row(pc, row0 + r, acc[r], lane);
      return;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (lane == r && row0 + r < a.rows) {
        const int i = row0 + r;
        if (a.partial) {
          a.partial[i] = acc[r];
        } else {
          const float xr = PREF ? pre.e0 : ((MODE == MODE_WO && a.emb) ? a.emb[(size_t)token * a.dim + i] : a.res.ld(i));
          const float mv = (float)acc[r];   // xb2 (WO) / xb (W2) as the reference stores it
          a.out.st(i, xr + mv);
          if (a.aux) a.aux.st(i, mv);
        }
      }
    }
  }
}

template <int MODE>
__device__ __forceinline__ constexpr bool mode_has_norm() { return MODE == MODE_QKV || MODE == MODE_W13 || MODE == MODE_CLS; }

// ------------------------------------------------------------------------------------------------
// STREAMING form (large matrices: Llama-2-7B's phases, every classifier).
// One dependency phase of a layer = prologue (stage the input vector in LDS, optional rmsnorm) +
// GEMV over this phase's matrix rows + fused epilogue.
//
// Work split: a wave owns R consecutive output rows ("row group") at a time and walks the columns in
// batches of U x 64 float4 per row, so one batch = R*U independent 16-byte non-temporal loads per lane
// (R*U KiB per wave).  Two register sets (A/B) are filled alternately: batch b+1 is always issued
// BEFORE batch b is consumed, across row-group boundaries too, and the first batch is issued before the
// prologue -- the weight stream never depends on the activations, only the FMAs do.  hipcc turns the
// in-order load queue into counted `s_waitcnt vmcnt(R*U)` waits, so ~2*R*U KiB per wave stay in flight.
// Vector path: n % 4 == 0 (every real checkpoint).  LDS: xs[npad4] float4 (zero padded to whole batches),
// ws[n4] float4 (norm weight, norm modes only), 8 doubles of reduction scratch.
//
// PK: the matrix comes from `a.wp`, REPACKED in consumption order (pack_kernel below): [round][column batch][wave][row of the
// group][u][lane] float4, round = the wave's row groups in turn.  All waves of the (co-resident) grid march through their
// batches in step, so what the chip asks for at one moment is ONE contiguous stretch (2048 waves x 4 KB = 8 MB) moving through
// memory, as in a plain copy -- instead of 4096 pieces of 2 KB a whole row apart.  tools/microbench_rows.hip: 55.3 -> 51.6 us
// for the w1/w3 shape of Llama-2-7B (rows + rotation vs packed, FMAs included), 31.9 -> 30.2 wqkv, 29.2 -> 27.7 w2,
// 12.4 -> 11.7 wo; a grid-stride copy takes 53.3 / 30.7 / 27.7 / 11.4.  Same rows to the same waves, same columns to the
// same lanes in the same order: the arithmetic is untouched.  Needs n % 256 == 0 (whole 64-lane sub-batches).
// PUSH (WO / W2 of a tensor-parallel rank with the peer-to-peer exchange): rows leave through tp_push_row.  A template parameter, not a
// run-time test: carrying the push context through the loop as a run-time option cost the one-GPU w2 instance 29 VGPRs (71 -> 100, 7 -> 4 waves per SIMD).
template <int MODE, int R, int U, int PRE, bool PK = false, bool PUSH = false>
__device__ __forceinline__ void phase_body(const PhaseArgs& a, char* smem, const int vblock, const int vgrid) {
  constexpr int CPI = 64 * U;                       // float4 per row per batch
  const int n = a.n, n4 = n >> 2;
  const int nchunks = (n4 + CPI - 1) / CPI;
  const int npad4 = nchunks * CPI;
  // PRE = float4 per thread per staging round, picked by the host so one round covers the vector
  const int tid = threadIdx.x, nthreads = blockDim.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = nthreads >> 6;
  const int nstage4 = ((npad4 + PRE * nthreads - 1) / (PRE * nthreads)) * (PRE * nthreads);  // whole staging rounds
  f4* xs4 = reinterpret_cast<f4*>(smem);
  f4* ws4 = xs4 + nstage4;
  double* red = reinterpret_cast<double*>(smem + (size_t)(nstage4 * (mode_has_norm<MODE>() ? 2 : 1)) * 16);

  const int rows_per_group = (MODE == MODE_W13) ? R / 2 : R;
  const int groups = (a.rows + rows_per_group - 1) / rows_per_group;
  const int wstride = vgrid * nwaves;

  STAMP_INIT_WG(a.dbg, a.dbg_wg);
  STAMP(0);
  PushCtx pctx;
  constexpr bool pushing = PUSH && (MODE == MODE_WO || MODE == MODE_W2);
  if (pushing) pctx = tp_push_ctx(a.push, threadIdx.x & 63);
  f4 bufA[R][U], bufB[R][U];
  const int ulast = (n4 - (nchunks - 1) * CPI) >> 6;   // PK: 64-lane sub-batches of a row's last batch
  // rnd (PK only) = which of its row groups the wave is at: gi = (place in the round) + rnd * wstride
  auto issue = [&](f4 (&buf)[R][U], int gi, int ci, int rnd) {
    if (PK) {
      const int nwr = min(wstride, groups - rnd * wstride);        // row groups of this round (the last one may be short)
      const int uc = (ci == nchunks - 1) ? ulast : U;
      const f4* base = reinterpret_cast<const f4*>(a.wp) + ((size_t)rnd * wstride * R * n4 + (size_t)nwr * (R * CPI) * ci + (size_t)(gi - rnd * wstride) * (R * 64) * uc);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int uu = (u < uc) ? u : 0;                           // past the row's end: re-read the first sub-batch (x there is 0)
#pragma unroll
        for (int r = 0; r < R; ++r) buf[r][u] = __builtin_nontemporal_load(base + (r * uc + uu) * 64 + lane);
      }
      return;
    }
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
  // Every wave starts its rows at another column batch and wraps around: with all waves marching through their rows in step,
  // the chip's requests of one moment sit a whole row (a power of two of bytes) apart and crowd the same HBM channels
  // (tools/microbench_rows.hip: 56.5 -> 54.2 us for the w1/w3 shape).  ci counts batches, col(gi, ci) is where batch ci lies.
  // (PK: no rotation -- the layout is what spreads the requests; `rg` then counts the wave's rounds instead)
  // (never for a classifier: a rotated row adds its column batches in another order than its neighbour, and two rows with the SAME weights
  //  -- an exact tie of the maximum, which the reference resolves to the smaller index, llama2.ts:364-366 -- must round to the same logit)
  auto rot_of = [&](int gi) { return (!PK && MODE != MODE_CLS && a.rot) ? (int)((unsigned)(gi * a.rot) % (unsigned)nchunks) : 0; };
  auto col = [&](int ci, int rg) { if (PK) return ci; const int c = ci + rg; return c >= nchunks ? c - nchunks : c; };
  auto next = [&](int& gi, int& ci, bool& hv, int& rg) {
    if (++ci == nchunks) { ci = 0; gi += wstride; rg = PK ? rg + 1 : rot_of(gi); }
    hv = gi < groups;
  };
  const int g0 = vblock * nwaves + wave, c0 = 0;
  const bool h0 = g0 < groups;
  const int r0 = rot_of(g0);

  // ---- prologue: input vector -> LDS (rmsnorm fused, llama2.ts:172-179).
  // Vector-memory results return in issue order: the activations are requested FIRST and the first weight batch
  // right behind them, so waiting for x costs one L2 round trip while the weight stream (which depends on
  // nothing) is already in flight.
  int token = 0, pos = 0;
  if (MODE == MODE_QKV || MODE == MODE_WO) { token = a.tokpos[0]; pos = a.tokpos[1]; }
  // (the greedy pick folded into the step, greedy_token_from_keys: looked at in layer 0 only -- `emb` is the test every launch makes anyway)
  if ((MODE == MODE_QKV || MODE == MODE_WO) && a.emb && a.tok_out) {
    const int step = a.tokpos[2];
    if (MODE == MODE_QKV) { token = greedy_token_from_keys(a, lane, step); if (vblock == 0 && tid == 0 && step > 0) a.tok_out[step - 1] = token; }
    if (MODE == MODE_WO) { if (step > 0) token = a.tok_out[step - 1]; if (vblock == 0 && tid < 8) a.amax[(size_t)tid * 16] = 0ull; }
  }
  Mut<const float> src = a.in;      // (the embedding row of layer 0 is immutable, but takes the same way)
  if (MODE == MODE_QKV) { if (a.emb) src = Mut<const float>(a.emb + (size_t)token * n); }
  const auto srs = src.rsrc(n);      // (the input vector: past L1 -- the coherence rule above)
  const f4* rw4 = reinterpret_cast<const f4*>(a.rmsw);
  const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
  double ss = 0.0, ss1 = 0.0, ss2 = 0.0, ss3 = 0.0;   // four chains: fp64 FMA latency is not on the path
  auto stage_load = [&](f4 (&xr)[PRE], f4 (&wr)[PRE], int base) {
#pragma unroll
    for (int k = 0; k < PRE; ++k) {
      const int cc = min(base + tid + k * nthreads, n4 - 1);
      xr[k] = L2_ACT_LD4(srs, cc);
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
  {
    f4 xr[PRE], wr[PRE];
    stage_load(xr, wr, 0);
    issue(bufA, h0 ? g0 : groups - 1, h0 ? col(c0, r0) : 0, h0 ? 0 : (groups - 1) / wstride);
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
    ss = rms_scale(ss, a.inv_n);
    for (int c = tid; c < n4; c += nthreads) {   // same thread, same elements as above: no barrier needed in between
      const f4 xv = xs4[c], wv = ws4[c];
      f4 o;
      o.x = (float)((double)wv.x * (ss * (double)xv.x));
      o.y = (float)((double)wv.y * (ss * (double)xv.y));
      o.z = (float)((double)wv.z * (ss * (double)xv.z));
      o.w = (float)((double)wv.w * (ss * (double)xv.w));
      xs4[c] = o;
      if (MODE == MODE_CLS && vblock == 0 && a.aux) a.aux.st4(c, o);  // rmsnorm(x, x, ...) in place, llama2.ts:299
    }
  }
  __syncthreads();
  STAMP(4);

  const EpiPre nopre = {0.0f, 0.0f, 0u};
  unsigned long long best = 0;      // CLS: this lane's best (logit, index) so far
  // The epilogue of a finished row group (reduction across the lanes, RoPE / SwiGLU / residual, stores) runs AFTER the next
  // batch has been requested, from a copy of the sums: both register sets stay in flight while it computes (with the
  // epilogue in front of the request the wave had one batch in flight for ~500 cycles per group, 6 times per wave of w1/w3)
  double pacc[R];
  int pend = -1;
  auto stash = [&](int gi) {
#pragma unroll
    for (int r = 0; r < R; ++r) { pacc[r] = acc[r]; acc[r] = 0.0; }
    pend = gi;
  };
  auto finish = [&]() {
    if (pend < 0) return;
    if (R == 2) wave_sum2(pacc[0], pacc[R - 1]);
    else {
#pragma unroll
      for (int r = 0; r < R; ++r) pacc[r] = wave_sum(pacc[r]);
    }
    STAMP(6);
    finish_group<MODE, R, false>(a, pend, pacc, lane, token, pos, nopre, best, pctx, pushing);
    STAMP(7);
    pend = -1;
  };
  // ---- GEMV, double buffered: batch k+1 is issued, then batch k consumed (A holds batch 0 on entry).
  // (Pre-issuing two batches costs ~20 VGPRs and one wave per SIMD of occupancy: measured slower.)
  int g = g0, ch = c0, rg = r0;
  bool have = h0;
  while (have) {
    int g2 = g, ch2 = ch, rg2 = rg;
    bool have2 = true;
    next(g2, ch2, have2, rg2);
    issue(bufB, have2 ? g2 : g, have2 ? col(ch2, rg2) : col(ch, rg), have2 ? rg2 : rg);   // unconditional: keeps the wait counts uniform
    finish();
    consume(bufA, col(ch, rg));
    STAMP(5);
    if (ch == nchunks - 1) stash(g);
    if (!have2) break;
    int g3 = g2, ch3 = ch2, rg3 = rg2;
    bool have3 = true;
    next(g3, ch3, have3, rg3);
    issue(bufA, have3 ? g3 : g2, have3 ? col(ch3, rg3) : col(ch2, rg2), have3 ? rg3 : rg2);
    finish();
    consume(bufB, col(ch2, rg2));
    if (ch2 == nchunks - 1) stash(g2);
    g = g3; ch = ch3; rg = rg3; have = have3;
  }
  finish();
  if (MODE == MODE_CLS && a.amax) {
    // greedy loop: ONE memory-side maximum per workgroup (no value returned, nothing waits for it); the launch
    // boundary orders it before the one-wave kernel that reads the eight keys (argmax_finish_kernel)
    best = wave_max_u64(best);
    unsigned long long* sk = reinterpret_cast<unsigned long long*>(red);
    __syncthreads();
    if (lane == 0) sk[wave] = best;
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < nwaves; ++w) best = sk[w] > best ? sk[w] : best;
      __hip_atomic_fetch_max(a.amax + (size_t)(vblock & 7) * 16, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // the greedy pick folded into the step: {pos, step} advance here -- no workgroup of this launch reads them, nothing of this token
      // runs behind it, and the next token's first launch takes the token from the keys itself (greedy_token_from_keys)
      if (vblock == 0 && a.tok_out) { int* tp = const_cast<int*>(a.tokpos); const int p1 = tp[1], step = tp[2]; tp[1] = p1 + 1; tp[2] = step + 1; }
    }
  }
}

template <int MODE, int R, int U, int PRE, bool PK = false, bool PUSH = false>
__global__ void __launch_bounds__(256) phase_kernel(const PhaseArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  phase_body<MODE, R, U, PRE, PK, PUSH>(a, smem, blockIdx.x, gridDim.x);
}

// Repack the matrix (matrices) of one launch for phase_body<.., PK = true>: one thread per float4, the same row_ptrs as the
// GEMV, so a row group is whatever the phase says it is (rows 2g, 2g + 1; row g of w1 and of w3; q rows, then k, then v).
//   dst[round][batch ci][place in the round][r][u][lane],  group g = place + round * wstride,  column = ci * cpi + u * 64 + lane
template <int MODE, int R>
__global__ void __launch_bounds__(256) pack_kernel(const PhaseArgs a, f4* dst, int U, int wstride) {
  const int n4 = a.n >> 2, cpi = 64 * U, nchunks = (n4 + cpi - 1) / cpi, ulast = (n4 - (nchunks - 1) * cpi) >> 6;
  const int rows_per_group = (MODE == MODE_W13) ? R / 2 : R;
  const int groups = (a.rows + rows_per_group - 1) / rows_per_group;
  const int gi = blockIdx.x, e = blockIdx.y * 256 + threadIdx.x;     // e = r * n4 + column (row groups in x: a vocabulary has more than 65535 of them)
  if (e >= R * n4) return;
  const int r = e / n4, c4 = e - r * n4;
  const float* rp[R];
  row_ptrs<MODE, R>(a, gi, a.n, rp);
  const float* src = rp[0];
#pragma unroll
  for (int k = 1; k < R; ++k) src = (r == k) ? rp[k] : src;
  const int ci = c4 / cpi, u = (c4 - ci * cpi) >> 6, ln = c4 & 63;
  const int rnd = gi / wstride, place = gi - rnd * wstride, nwr = min(wstride, groups - rnd * wstride), uc = (ci == nchunks - 1) ? ulast : U;
  dst[(size_t)rnd * wstride * R * n4 + (size_t)nwr * (R * cpi) * ci + (size_t)place * (R * 64) * uc + (r * uc + u) * 64 + ln] = reinterpret_cast<const f4*>(src)[c4];
}

// The inverse: the row-major matrix (matrices) of one launch back out of the repacked copy -- what l2_read_tensor and a later
// l2_upload into a packed phase work on once the row-major tensors have been given back (one copy of the weights).
template <int MODE, int R>
__global__ void __launch_bounds__(256) unpack_kernel(const PhaseArgs a, const f4* src, int U, int wstride) {
  const int n4 = a.n >> 2, cpi = 64 * U, nchunks = (n4 + cpi - 1) / cpi, ulast = (n4 - (nchunks - 1) * cpi) >> 6;
  const int rows_per_group = (MODE == MODE_W13) ? R / 2 : R;
  const int groups = (a.rows + rows_per_group - 1) / rows_per_group;
  const int gi = blockIdx.x, e = blockIdx.y * 256 + threadIdx.x;
  if (e >= R * n4) return;
  const int r = e / n4, c4 = e - r * n4;
  // the rows of this group (a clamped duplicate row of a short last group is skipped: its source slot holds a copy of the last row)
  if (MODE == MODE_W13) { if (gi * (R / 2) + (r % (R / 2)) >= a.rows) return; }
  else if (MODE == MODE_QKV) { int m, i0; qkv_group(a, gi, R, m, i0); if (i0 + r >= (m == 0 ? a.dim : a.kv_dim)) return; }
  else if (gi * R + r >= a.rows) return;
  const float* rp[R];
  row_ptrs<MODE, R>(a, gi, a.n, rp);
  const float* dstp = rp[0];
#pragma unroll
  for (int k = 1; k < R; ++k) dstp = (r == k) ? rp[k] : dstp;
  const int ci = c4 / cpi, u = (c4 - ci * cpi) >> 6, ln = c4 & 63;
  const int rnd = gi / wstride, place = gi - rnd * wstride, nwr = min(wstride, groups - rnd * wstride), uc = (ci == nchunks - 1) ? ulast : U;
  reinterpret_cast<f4*>(const_cast<float*>(dstp))[c4] = src[(size_t)rnd * wstride * R * n4 + (size_t)nwr * (R * cpi) * ci + (size_t)place * (R * 64) * uc + (r * uc + u) * 64 + ln];
}

// ------------------------------------------------------------------------------------------------
// LATENCY form (small matrices: every phase of stories15M / stories110M moves 0.3 - 12.6 MB, i.e. 0.05 - 2 us of
// HBM time, so the phase costs what its chain of dependent steps costs).  In-kernel anatomy of the streaming form
// on stories110M (profiles/r02/stamps_phase_110M_before.txt, cycles): requests issued 1000-1500, x landed
// 1400-2400 (its load queues behind the weight requests of the workgroup that started earlier on the same CU:
// vector memory returns in order per CU), block reduction of sum(x^2) 1000, norm + barrier 800, weights consumed
// 1200, two wave reductions 900, epilogue 600-2100 (RoPE table / residual loads issued only after the
// reduction).  Here:
//   * ONE 512-thread workgroup per CU.  Wave 0 only requests x (and the norm weight), normalises it alone (one
//     wave reduction, no block reduction) and publishes it in LDS -- it requests nothing else, so its wait for x is
//     exact; the other seven waves' weight requests are in flight meanwhile and they meet wave 0 at ONE barrier;
//   * every wave then holds x widened to fp64 in registers for all its rows (no LDS reads in the loop);
//   * a wave's first two row groups are requested before the barrier, with the epilogue operands of each group
//     (RoPE entries, residual value) behind them, so nothing is requested after a reduction.
// Numerics are those of the streaming form: fp64 accumulate, one fp32 rounding per stored element.
// GIN: the input vector is handed over as granules by workgroups of the same launch (a.gin): the x wave -- and only it -- polls them,
// behind the compute waves' weight requests (those do not depend on the input: by the time the producers are done they have landed).
template <int MODE, int XV, int R, bool GIN = false>
__device__ __forceinline__ void phase_small_body(const PhaseArgs& a, char* smem, const int vblock, const int vgrid) {
  constexpr int NC = 7;                             // compute waves; wave 0 is the x wave
  f4* xs4 = reinterpret_cast<f4*>(smem);            // 64 * XV float4, zero padded
  const int n = a.n, n4 = n >> 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int RPG = (MODE == MODE_W13) ? R / 2 : R;   // output rows per group
  const int groups = (a.rows + RPG - 1) / RPG;
  const int gstride = vgrid * NC;

  STAMP_INIT_WGL(a.dbg, a.dbg_wg);
  STAMP(0);
  int token = 0, pos = 0;
  PushCtx pctx;
  f4 bufA[R][XV], bufB[R][XV];
  EpiPre preA = {0.0f, 0.0f, 0u}, preB = {0.0f, 0.0f, 0u};
  // the wave's k-th row group (`groups` = none): consecutive row groups go to different CUs
  const int g0 = (wave - 1) * vgrid + vblock;
  auto nth = [&](int k) -> int { const int g = g0 + k * gstride; return g < groups ? g : groups; };
  const int gA0 = nth(0), gB0 = nth(1);
  auto issue = [&](f4 (&buf)[R][XV], int gi) {
    const float* rp[R];
    row_ptrs<MODE, R>(a, gi, n, rp);
#pragma unroll
    for (int u = 0; u < XV; ++u) {
      const int c = min(u * 64 + lane, n4 - 1);
#pragma unroll
      for (int r = 0; r < R; ++r) buf[r][u] = ldg_nt(rp[r] + 4 * c);
    }
  };
  // ---- the x wave requests x (and the norm weight); the other waves request their weights at once (the stream is
  // what bounds the big phases) and need {token, pos} only for the epilogue operands, requested after the x barrier
  f4 xr[XV], wr[XV];
  if (wave == 0) {
    Mut<const float> src = a.in;
    if (MODE == MODE_QKV) {      // only layer 0 waits for the token
      if (a.emb) {
        int tk;
        if (a.tok_out) { const int step = a.tokpos[2]; tk = greedy_token_from_keys(a, lane, step); if (vblock == 0 && lane == 0 && step > 0) a.tok_out[step - 1] = tk; }
        else tk = a.tokpos[0];
        src = Mut<const float>(a.emb + (size_t)tk * n);
      }
    }
    if (GIN) {
      const unsigned gtag_in = __hip_atomic_load(a.gran_ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
      granules_gather_f4<XV>(a.gran, n, lane, gtag_in, xr, a.gin_herr, 200000000ull, 1, const_cast<unsigned*>(a.gran_ep) + 1);      // bounded at 2 s; a short nap between sweeps (the wait is a whole attention long)
    } else {
      const auto srs = src.rsrc(n);
#pragma unroll
      for (int u = 0; u < XV; ++u) xr[u] = L2_ACT_LD4(srs, min(u * 64 + lane, n4 - 1));
    }
    if (mode_has_norm<MODE>()) {
#pragma unroll
      for (int u = 0; u < XV; ++u) wr[u] = reinterpret_cast<const f4*>(a.rmsw)[min(u * 64 + lane, n4 - 1)];
    }
  }
#ifdef L2_XBARRIER
  __syncthreads();
#endif
  if (wave == 0) {
    const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
    if (mode_has_norm<MODE>()) {     // rmsnorm (llama2.ts:172-179): ss = sum x^2 / n; ss = 1/sqrt(1e-5 + ss); o = w * (ss * x)
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
      for (int u = 0; u < XV; ++u) {
        if (u * 64 + lane >= n4) xr[u] = zero4;
        s0 += (double)xr[u].x * (double)xr[u].x; s1 += (double)xr[u].y * (double)xr[u].y;
        s2 += (double)xr[u].z * (double)xr[u].z; s3 += (double)xr[u].w * (double)xr[u].w;
      }
      double ss = wave_sum((s0 + s1) + (s2 + s3));
      STAMP(2);
      ss = rms_scale(ss, a.inv_n);
      STAMP(4);
#pragma unroll
      for (int u = 0; u < XV; ++u) {
        f4 o;
        o.x = (float)((double)wr[u].x * (ss * (double)xr[u].x));
        o.y = (float)((double)wr[u].y * (ss * (double)xr[u].y));
        o.z = (float)((double)wr[u].z * (ss * (double)xr[u].z));
        o.w = (float)((double)wr[u].w * (ss * (double)xr[u].w));
        xs4[u * 64 + lane] = o;
        if (MODE == MODE_CLS && vblock == 0 && a.aux && u * 64 + lane < n4) a.aux.st4(u * 64 + lane, o);   // llama2.ts:299
        if (u == 0) STAMP(8);
      }
      STAMP(9);
    } else {
#pragma unroll
      for (int u = 0; u < XV; ++u) xs4[u * 64 + lane] = (u * 64 + lane < n4) ? xr[u] : zero4;
    }
  } else if (gA0 < groups) {
    // ---- compute waves: the first two row groups + their epilogue operands (wave-uniform branches: a wave without
    // a second group requests nothing for it -- duplicate requests cost address-path cycles, 16 per KiB per CU)
    issue(bufA, gA0);
    if (gB0 < groups) issue(bufB, gB0);
  }
  STAMP(1);
  __syncthreads();
  STAMP(3);
  if (wave == 0 || gA0 >= groups) return;
  if (MODE == MODE_QKV || MODE == MODE_WO) { token = a.tokpos[0]; pos = a.tokpos[1]; }
  if (MODE == MODE_WO && a.emb && a.tok_out) {      // the greedy pick folded into the step (greedy_token_from_keys): layer 0 only
    const int step = a.tokpos[2];
    if (step > 0) token = a.tok_out[step - 1];
    if (vblock == 0 && wave == 1 && lane < 8) a.amax[(size_t)lane * 16] = 0ull;
  }
  // (the tensor-parallel push pointer is looked at HERE, behind the weight requests: hipcc fetches a kernel argument where it is first
  // used, and a test of it in front of them put one more cold scalar-cache round -- 0.4 us -- before the first weight request of EVERY
  // wo launch, tensor parallel or not: stories110M 4 010 -> 3 930 tok/s until it was found by a same-box run of the round-4 library)
  const bool pushing = (MODE == MODE_WO || MODE == MODE_W2) && a.push;
  if (pushing) pctx = tp_push_ctx(a.push, lane);
  // hand-off tag of a row group = its head's launch counter + 1 (advanced by an EARLIER launch: an ordinary load)
  auto gtag = [&](int gi) -> unsigned {
    if (!(MODE == MODE_QKV) || !a.gran) return 0u;
    int m, i0;
    qkv_group(a, gi, R, m, i0);
    return ld_word(a.gran_ep + (((unsigned)i0 * a.gran_hmagic) >> 20)) + 1u;
  };
  preA = epi_prefetch<MODE, R>(a, gA0, lane, token, pos); preA.tag = gtag(gA0);
  if (gB0 < groups) { preB = epi_prefetch<MODE, R>(a, gB0, lane, token, pos); preB.tag = gtag(gB0); }
  double xd[XV][4];
#pragma unroll
  for (int u = 0; u < XV; ++u) {
    const f4 v = xs4[u * 64 + lane];
    xd[u][0] = v.x; xd[u][1] = v.y; xd[u][2] = v.z; xd[u][3] = v.w;
  }

  auto run = [&](const f4 (&buf)[R][XV], int gi, const EpiPre& pre) {
    double acc[R], acb[R];           // two chains per row
#pragma unroll
    for (int r = 0; r < R; ++r) { acc[r] = 0.0; acb[r] = 0.0; }
#pragma unroll
    for (int u = 0; u < XV; ++u) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        acc[r] += (double)buf[r][u].x * xd[u][0];
        acb[r] += (double)buf[r][u].y * xd[u][1];
        acc[r] += (double)buf[r][u].z * xd[u][2];
        acb[r] += (double)buf[r][u].w * xd[u][3];
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] += acb[r];
    STAMP(5);
    if (R == 2) wave_sum2(acc[0], acc[R - 1]);
    else {
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]);
    }
    STAMP(6);
    unsigned long long nobest = 0;
    finish_group<MODE, R, true>(a, gi, acc, lane, token, pos, pre, nobest, pctx, pushing);
    STAMP(7);
  };
  for (int k = 0;; k += 2) {
    const int g = nth(k);
    if (g >= groups) break;
    run(bufA, g, preA);
    const int g2 = nth(k + 2);
    if (g2 < groups) { issue(bufA, g2); preA = epi_prefetch<MODE, R>(a, g2, lane, token, pos); preA.tag = gtag(g2); }
    const int g1 = nth(k + 1);
    if (g1 >= groups) break;
    run(bufB, g1, preB);
    const int g3 = nth(k + 3);
    if (g3 < groups) { issue(bufB, g3); preB = epi_prefetch<MODE, R>(a, g3, lane, token, pos); preB.tag = gtag(g3); }
  }
}

template <int MODE, int XV, int R>
__global__ void __launch_bounds__(512) phase_small_kernel(const PhaseArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  phase_small_body<MODE, XV, R>(a, smem, blockIdx.x, gridDim.x);
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
  Mut<const float> src = a.in;
  if ((MODE == MODE_QKV) && a.emb) src = Mut<const float>(a.emb + (size_t)token * n);
  if (mode_has_norm<MODE>()) {
    double ss = 0.0;
    for (int j = tid; j < n; j += nthreads) { const double v = src.ld(j); ss += v * v; }
    ss = block_sum(ss, red, tid, nthreads);
    ss = rms_scale(ss, a.inv_n);
    for (int j = tid; j < n; j += nthreads) {
      const float o = (float)((double)a.rmsw[j] * (ss * (double)src.ld(j)));
      xs[j] = o;
      if (MODE == MODE_CLS && blockIdx.x == 0 && a.aux) a.aux.st(j, o);
    }
  } else {
    for (int j = tid; j < n; j += nthreads) xs[j] = src.ld(j);
  }
  __syncthreads();
  const int rows_per_group = (MODE == MODE_W13) ? R / 2 : R;
  const int groups = (a.rows + rows_per_group - 1) / rows_per_group;
  unsigned long long best = 0;
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
    { const EpiPre nopre = {0.0f, 0.0f, 0u};
      PushCtx pctx;
      const bool pushing = (MODE == MODE_WO || MODE == MODE_W2) && a.push;
      if (pushing) pctx = tp_push_ctx(a.push, lane);
      finish_group<MODE, R, false>(a, g, acc, lane, token, pos, nopre, best, pctx, pushing); }
  }
  if (MODE == MODE_CLS && a.amax) {   // as in phase_body
    best = wave_max_u64(best);
    unsigned long long* sk = reinterpret_cast<unsigned long long*>(red);
    __syncthreads();
    if (lane == 0) sk[wave] = best;
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < nwaves; ++w) best = sk[w] > best ? sk[w] : best;
      __hip_atomic_fetch_max(a.amax + (size_t)(blockIdx.x & 7) * 16, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// (plain, non-template kernels: defined in ONE translation unit -- attention_inst.hip includes this header for the templates only)
#ifndef L2_NO_PLAIN_KERNELS
// ------------------------------------------------------------------------------------------------
// argmax (llama2.ts:364-366: first maximum, strict '>') + advance {token,pos,step}: keeps the greedy
// loop (llama2.ts:465-508 at -t 0) on the device.  A logit and its index travel as ONE 64-bit key
// (order-preserving bits of the value, then ~index), so "largest value, smallest index" is an unsigned maximum and
// the cross-lane reduction runs on the DPP path; {pos, step} are requested at the top, not after the reduction.
// The greedy loop's pick when the classifier kernel has folded its logits into the eight keys of `amax`: one wave
// takes their maximum, re-arms them, records the token and advances {token, pos, step}.
__global__ void __launch_bounds__(64) argmax_finish_kernel(unsigned long long* amax, int* tokpos, int* tokens_out) {
  const int lane = threadIdx.x;
  const int p1 = tokpos[1], step = tokpos[2];
  unsigned long long k = lane < 8 ? __hip_atomic_load(amax + (size_t)lane * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;      // (past L1: the coherence rule)
  if (lane < 8) amax[(size_t)lane * 16] = 0ull;
  k = wave_max_u64(k);
  if (lane == 0) {
    const int bi = (k == 0) ? 0 : (int)~(unsigned)k;
    tokens_out[step] = bi;
    tokpos[0] = bi; tokpos[1] = p1 + 1; tokpos[2] = step + 1;
  }
}

// TEST HOOK (L2_DEBUG_POLLUTE=1 behind L2_TEST_HOOKS; tests/test_coherence_gpu.py): the adversary of the coherence rule.  Launched behind
// EVERY launch of a recorded step, it makes every CU pull every mutable line of the step -- activations, logits, argmax keys, launch
// counters, hand-off granules, split-attention partials, {token, pos, step}, the token record and the cache rows of the current and the
// next position in every layer -- into ITS vector L1 with PLAIN loads.  Nothing invalidates those lines before a later launch of the token
// rewrites the bytes (on the library's queue no launch but a token's first acquires), so any load of them that does not go past L1 returns
// what the polluter saw.  With the rule kept, tokens and logits are bit for bit those of a run without the polluter.
struct PolluteArgs { const float* buf[20]; unsigned bytes[20]; const float* kc; const float* vc; const int* tokpos; float* sink; int nb, L, S, kvd; };
__global__ void __launch_bounds__(256) l1_pollute_kernel(const PolluteArgs a) {
  const int tid = threadIdx.x;
  float acc = 0.0f;
  auto pull = [&](const float* base, unsigned bytes) {      // one plain dword load per 64-byte half line
    for (unsigned o = (unsigned)tid * 64u; o < bytes; o += 256u * 64u) {
      float v;
      asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(reinterpret_cast<const char*>(base) + o) : "memory");
      acc += v;
    }
  };
  for (int b = 0; b < a.nb; ++b) pull(a.buf[b], a.bytes[b]);
  int pos;
  asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(pos) : "v"(a.tokpos + 1) : "memory");
  pos = pos < 0 ? 0 : (pos >= a.S ? a.S - 1 : pos);
  const int p1 = pos + 1 < a.S ? pos + 1 : pos;
  for (int l = 0; l < a.L; ++l) {      // row pos (written by this token's q / k / v launch of layer l) and row pos + 1 (by the next token's)
    const size_t slab = (size_t)l * a.S * a.kvd;
    pull(a.kc + slab + (size_t)pos * a.kvd, (unsigned)a.kvd * 4u); pull(a.vc + slab + (size_t)pos * a.kvd, (unsigned)a.kvd * 4u);
    pull(a.kc + slab + (size_t)p1 * a.kvd, (unsigned)a.kvd * 4u); pull(a.vc + slab + (size_t)p1 * a.kvd, (unsigned)a.kvd * 4u);
  }
  if (acc == 1.2345e-38f) a.sink[0] = acc;      // (keeps the loads alive; never true in practice)
}

// The blocking call (llama2.ts:468) on the library's own queue: {token, pos} of the call come from pinned host memory -- one load over
// PCIe by one wave in front of the step, instead of a stream copy and its boundary in front of a graph launch.
__global__ void __launch_bounds__(64) set_tokpos_kernel(const int* host_tokpos, int* tokpos) {
  const int lane = threadIdx.x;
  if (lane < 4) tokpos[lane] = __hip_atomic_load(host_tokpos + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ... and with the pick folded into the step (greedy_token_from_keys): the run's LAST pick, once per run -- {pos, step} were advanced by
// the classifier launch already
__global__ void __launch_bounds__(64) argmax_last_kernel(unsigned long long* amax, int* tokpos, int* tokens_out) {
  const int lane = threadIdx.x;
  const int step = tokpos[2];
  unsigned long long k = lane < 8 ? __hip_atomic_load(amax + (size_t)lane * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
  if (lane < 8) amax[(size_t)lane * 16] = 0ull;
  k = wave_max_u64(k);
  if (lane == 0 && step > 0) {
    const int bi = (k == 0) ? 0 : (int)~(unsigned)k;
    tokens_out[step - 1] = bi;
    tokpos[0] = bi;
  }
}

__global__ void __launch_bounds__(1024) argmax_advance_kernel(const float* logits, int V, int* tokpos, int* tokens_out) {
  __shared__ unsigned long long sk[16];
  const int tid = threadIdx.x;
  int p1 = 0, step = 0;
  if (tid == 0) {      // (past L1: the coherence rule)
    p1 = (int)__hip_atomic_load(reinterpret_cast<const unsigned*>(tokpos + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    step = (int)__hip_atomic_load(reinterpret_cast<const unsigned*>(tokpos + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  unsigned long long best = 0;
  if ((V & 3) == 0) {   // 16-byte loads, all issued before the first compare
    const auto lrs = L2_ACT_RSRC(logits, V);
#pragma unroll 8
    for (int c = tid; c < V / 4; c += 1024) {
      const f4 v = L2_ACT_LD4(lrs, c);
      unsigned long long k = argmax_key(v.x, 4 * c); best = k > best ? k : best;
      k = argmax_key(v.y, 4 * c + 1); best = k > best ? k : best;
      k = argmax_key(v.z, 4 * c + 2); best = k > best ? k : best;
      k = argmax_key(v.w, 4 * c + 3); best = k > best ? k : best;
    }
  } else {
    for (int i = tid; i < V; i += 1024) { const unsigned long long k = argmax_key(ld_sc1(logits + i), i); best = k > best ? k : best; }
  }
  best = wave_max_u64(best);
  if ((tid & 63) == 0) sk[tid >> 6] = best;
  __syncthreads();
  if (tid < 64) {
    best = wave_max_u64(tid < 16 ? sk[tid] : 0ull);
    if (tid == 0) {
      const int bi = (best == 0) ? 0 : (int)~(unsigned)best;   // nothing but NaN: reduce() keeps index 0
      tokens_out[step] = bi;
      tokpos[0] = bi; tokpos[1] = p1 + 1; tokpos[2] = step + 1;
    }
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

#endif  // L2_NO_PLAIN_KERNELS

}  // namespace l2k

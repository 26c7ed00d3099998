// Can the NEXT dependent GEMV phase start streaming its weights while the previous one is still finishing?
//
// A Llama-2-7B layer is five dependent launches (qkv 201 MB -> attention -> wo 67 MB -> w1/w3 361 MB -> w2 180 MB).  Between two
// launches of one stream the chip drains: the slowest workgroup of a launch ends 5 - 10 % after the median one, then comes the
// boundary (1.5 - 1.9 us), then the next launch ramps up.  The weight stream of phase k + 1 does not depend on phase k -- only
// its FMAs do (through x).  Here the phases ALTERNATE BETWEEN TWO STREAMS with no stream-level dependency: phase k + 1 is
// dispatched while phase k runs, requests the first PRE column batches of its first row group into registers, and only then
// waits -- in the kernel -- for phase k's completion counter (agent-scope release / acquire as the MI355X guide prescribes), reads
// x and starts its FMAs.  Within a stream the launches stay ordered (k + 2 cannot start before k has ended), so at most two phases
// are ever co-resident.
//
//   mode 0  one stream, plain dependent launches (what libllama2hip does today)
//   mode 1  two streams + in-kernel hand-off, PRE = 2 batches requested before the wait (the register sets the loop has anyway)
//   mode 2  the same, PRE = 6 (most of the wave's first row group is in registers before x arrives)
// Output: us per layer for each mode; every spin is bounded by wall time (err counter printed).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mbo tools/microbench_overlap.hip && /tmp/mbo
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void fill(float* p, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { unsigned x = (unsigned)i * 2654435761u; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; p[i] = ((float)(x & 0xffff) - 32768.0f) * 1e-6f; } }

enum { MAXWG = 512 };
struct Flag { unsigned arrive; unsigned pad0[31]; unsigned done; unsigned pad1[31]; unsigned per_wg[MAXWG * 32]; };   // counters on their own 128-byte lines; per_wg[b * 32]: the consumer workgroup b's private copy of `done`
__device__ int g_sleep = 1, g_private = 0, g_fences = 1;   // g_fences 0: outputs stored write-through (sc1), x read with sc1 loads, no release / acquire fence

__device__ __forceinline__ void st1(float* p, float v) { __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld1(const float* p) { return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// consumer side: one lane polls the predecessor's `done` until it is one ahead of this phase's own count, then acquires
__device__ __forceinline__ void wait_prev(const Flag* prev, const Flag* mine, int* err, int lead, int want_arg = -1) {
  if (prev) {
    if (threadIdx.x == 0) {
      const unsigned want = want_arg >= 0 ? (unsigned)want_arg : __hip_atomic_load(&mine->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + (unsigned)lead;   // lead 0: the first phase of the cycle (its predecessor is the LAST phase of the previous round)
      unsigned spins = 0; unsigned long long t0 = 0;
      const unsigned* word = g_private ? &prev->per_wg[(blockIdx.x % MAXWG) * 32] : &prev->done;
      const int slp = g_sleep;
      while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {
        if (slp <= 1) __builtin_amdgcn_s_sleep(1); else if (slp <= 8) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(32);
        if ((++spins & 255u) == 0) { const unsigned long long now = __builtin_amdgcn_s_memrealtime(); if (!t0) t0 = now; else if (now - t0 > 20000000ull) { atomicAdd(err, 1); break; } }   // 0.2 s
      }
      if (g_fences) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    }
    __syncthreads();
  }
}
// producer side: every wave drains its stores, barrier, one lane releases and takes a ticket; the last workgroup publishes
__device__ __forceinline__ void signal_done(Flag* mine) {
  if (mine) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ unsigned s_last;
    if (threadIdx.x == 0) {
      if (g_fences) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      const unsigned t = __hip_atomic_fetch_add(&mine->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = 0;
      if (t == gridDim.x - 1) {
        __hip_atomic_store(&mine->arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = __hip_atomic_fetch_add(&mine->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
      }
    }
    if (g_private) {      // the last workgroup hands every consumer workgroup its own copy of the count (no shared hot line to poll)
      __syncthreads();
      const unsigned gen = s_last;
      if (gen) for (int b = threadIdx.x; b < MAXWG; b += blockDim.x) __hip_atomic_store(&mine->per_wg[b * 32], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// One GEMV phase: out[g] = f(dot(row a of group g, x), dot(row b of group g, x)); a wave owns a row group at a time, walks it in
// NB batches of 2 x 64 float4 per row with a rotated start, two register sets in flight; PRE batches of the FIRST group are
// requested before the wait for x.  PAIR: rows g and groups + g (w1 / w3), else 2g and 2g + 1.
template <int NB, int PRE, bool PAIR>
__global__ void __launch_bounds__(256) phase(const f4* w, int rows, int n, const float* xin, float* xout, const Flag* prev, Flag* mine, int* err, int lead, int want = -1) {
  constexpr int U = 2;
  extern __shared__ f4 xs[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n4 = n / 4, groups = rows / 2, tw = gridDim.x * 4;
  auto load = [&](f4 (&b)[2][U], int g, int ci) {
    const f4* r0 = w + (size_t)(PAIR ? g : 2 * g) * n4;
    const f4* r1 = w + (size_t)(PAIR ? groups + g : 2 * g + 1) * n4;
    int c0 = ci + (g * 5) % NB; c0 -= c0 >= NB ? NB : 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = min(c0 * 64 * U + u * 64 + lane, n4 - 1);
      b[0][u] = __builtin_nontemporal_load(r0 + c);
      b[1][u] = __builtin_nontemporal_load(r1 + c);
    }
  };
  double d0 = 0.0, d1 = 0.0;
  auto fma2 = [&](const f4 (&b)[2][U], int g, int ci) {
    int c0 = ci + (g * 5) % NB; c0 -= c0 >= NB ? NB : 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 * 64 * U + u * 64 + lane;
      const f4 xv = xs[c];            // zero padded to NB whole batches
      const double x0 = xv.x, x1 = xv.y, x2 = xv.z, x3 = xv.w;
      d0 += (double)b[0][u].x * x0; d0 += (double)b[0][u].y * x1; d0 += (double)b[0][u].z * x2; d0 += (double)b[0][u].w * x3;
      d1 += (double)b[1][u].x * x0; d1 += (double)b[1][u].y * x1; d1 += (double)b[1][u].z * x2; d1 += (double)b[1][u].w * x3;
    }
  };
  auto finish = [&](int g) {
    const double s0 = wave_sum(d0), s1 = wave_sum(d1);
    if (lane == 0) {
      if (g_fences) { if (PAIR) xout[g] = (float)(s0 * 0.5 + s1 * 0.25); else { xout[2 * g] = (float)s0; xout[2 * g + 1] = (float)s1; } }
      else { if (PAIR) st1(xout + g, (float)(s0 * 0.5 + s1 * 0.25)); else { st1(xout + 2 * g, (float)s0); st1(xout + 2 * g + 1, (float)s1); } }
    }
    d0 = 0.0; d1 = 0.0;
  };
  const int g0 = blockIdx.x * 4 + wave;
  const bool have = g0 < groups;
  f4 P[PRE][2][U];
#pragma unroll
  for (int b = 0; b < PRE; ++b) load(P[b], have ? g0 : groups - 1, b);
  wait_prev(prev, mine, err, lead, want);
  if (g_fences) { for (int i = threadIdx.x; i < NB * 64 * U; i += 256) xs[i] = i < n4 ? reinterpret_cast<const f4*>(xin)[i] : f4{0.f, 0.f, 0.f, 0.f}; }
  else { for (int i = threadIdx.x; i < NB * 64 * U * 4; i += 256) reinterpret_cast<float*>(xs)[i] = i < n ? ld1(xin + i) : 0.f; }
  __syncthreads();
  if (have) {
    // first group: the preloaded batches, each replaced by a later batch of the stream as it is consumed
    int g = g0, ci = PRE;             // next batch to REQUEST
    int gq = g0, cq = 0;              // next batch to CONSUME
    // flattened stream position helpers
    auto adv = [&](int& gg, int& cc) { if (++cc == NB) { cc = 0; gg += tw; } };
    if (ci >= NB) { ci -= NB; g += tw; }
#pragma unroll
    for (int b = 0; b < PRE; ++b) {
      // consume P[b], then refill it with the next batch of the stream (if any): PRE batches stay in flight throughout
      fma2(P[b], gq, cq);
      if (cq == NB - 1) finish(gq);
      adv(gq, cq);
      load(P[b], g < groups ? g : groups - 1, ci); adv(g, ci);      // unconditional (clamped): keeps hipcc's waits counted
    }
    // steady state: round-robin over the PRE register sets
    while (gq < groups) {
#pragma unroll
      for (int b = 0; b < PRE; ++b) {
        if (gq < groups) {
          fma2(P[b], gq, cq);
          if (cq == NB - 1) finish(gq);
          adv(gq, cq);
          load(P[b], g < groups ? g : groups - 1, ci); adv(g, ci);
        }
      }
    }
  }
  signal_done(mine);
}

// attention stand-in: 32 workgroups; per head a dependent chain q -> 64 cache rows -> reduce -> out (the cache rows do not depend
// on the previous phase and are requested before the wait)
__global__ void __launch_bounds__(256) attn_like(const float* cache, const float* q, float* out, const Flag* prev, Flag* mine, int* err, int lead, int want = -1) {
  __shared__ float qs[128];
  __shared__ float red[256];
  const int h = blockIdx.x, t = threadIdx.x;
  const f4* rows = reinterpret_cast<const f4*>(cache + (size_t)h * 64 * 128);
  f4 kv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) kv[j] = rows[j * 256 + t];           // 64 rows x 128 floats = 2048 float4
  wait_prev(prev, mine, err, lead, want);
  if (t < 128) qs[t] = g_fences ? q[h * 128 + t] : ld1(q + h * 128 + t);
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) { const int c = ((j * 256 + t) & 31) * 4; s += kv[j].x * qs[c] + kv[j].y * qs[c + 1] + kv[j].z * qs[c + 2] + kv[j].w * qs[c + 3]; }
  red[t] = s;
  __syncthreads();
  if (t < 128) { if (g_fences) out[h * 128 + t] = red[t] + red[t + 128] * 0.5f; else st1(out + h * 128 + t, red[t] + red[t + 128] * 0.5f); }
  signal_done(mine);
}

int main(int argc, char** argv) {
  const int d = 4096, hd = 11008, LAYERS_MEM = 6, LAYERS = 32;
  const size_t per_layer = ((size_t)3 * d * d + (size_t)d * d + (size_t)2 * hd * d + (size_t)d * hd);   // floats
  float* w; (void)hipMalloc(&w, per_layer * 4 * LAYERS_MEM);
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, w, per_layer * LAYERS_MEM);
  float *x, *q, *xb, *hb, *cache; int* err;
  (void)hipMalloc(&x, d * 4); (void)hipMalloc(&q, 3 * d * 4); (void)hipMalloc(&xb, d * 4); (void)hipMalloc(&hb, hd * 4); (void)hipMalloc(&cache, 32 * 64 * 128 * 4);
  (void)hipMalloc(&err, 4); (void)hipMemset(err, 0, 4);
  hipLaunchKernelGGL(fill, dim3(64), dim3(256), 0, 0, x, (size_t)d); hipLaunchKernelGGL(fill, dim3(64), dim3(256), 0, 0, cache, (size_t)32 * 64 * 128);
  Flag* flags; (void)hipMalloc(&flags, sizeof(Flag) * 5); (void)hipMemset(flags, 0, sizeof(Flag) * 5);
  (void)hipDeviceSynchronize();
  hipStream_t sa, sb; (void)hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  hipEvent_t e0, e1, ej; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreateWithFlags(&ej, hipEventDisableTiming);
  const size_t lds4096 = 8 * 128 * 16, lds11008 = 22 * 128 * 16;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&phase<22, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&phase<22, 6, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  printf("us per Llama-2-7B-shaped layer (qkv, attention stand-in, wo, w1/w3, w2), %d layers, best of 5\n", LAYERS);
  // {mode, sleep, private flags, fences}; mode 3: ONE stream, but with the completion signal / wait protocol on every edge; modes 10..14: ONE stream, the
  // protocol on ONE edge only (10: qkv -> attention, 11: attention -> wo, 12: wo -> w1/w3, 13: w1/w3 -> w2, 14: w2 -> next qkv), everything else plain
  const int variants[][4] = {{0, 8, 0, 1}, {1, 8, 0, 1}, {1, 8, 0, 0}, {2, 8, 0, 0}, {3, 8, 0, 1}, {3, 8, 0, 0}, {10, 8, 0, 1}, {11, 8, 0, 1}, {12, 8, 0, 1}, {13, 8, 0, 1}, {14, 8, 0, 1}, {0, 8, 0, 1}};
  for (auto& var : variants) {
    const int mode = var[0];
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sleep), &var[1], 4); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_private), &var[2], 4); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fences), &var[3], 4);
    (void)hipMemset(x, 0, d * 4); hipLaunchKernelGGL(fill, dim3(64), dim3(256), 0, 0, x, (size_t)d); (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      (void)hipMemset(flags, 0, sizeof(Flag) * 5);
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0, sa);
      const bool two = mode == 1 || mode == 2;
      if (two) { (void)hipEventRecord(ej, sa); (void)hipStreamWaitEvent(sb, ej, 0); }
      int k = 0;      // launch index: even -> stream a, odd -> stream b (modes 1, 2)
      for (int l = 0; l < LAYERS; ++l) {
        const float* wl = w + per_layer * (size_t)(l % LAYERS_MEM);
        const f4* wqkv = (const f4*)wl; const f4* wo = (const f4*)(wl + (size_t)3 * d * d); const f4* w13 = (const f4*)(wl + (size_t)4 * d * d);
        const f4* w2 = (const f4*)(wl + (size_t)4 * d * d + (size_t)2 * hd * d);
        auto st = [&]() { hipStream_t s = (two && (k & 1)) ? sb : sa; ++k; return s; };
        Flag* F = mode ? flags : nullptr;
        const int edge = mode >= 10 ? mode - 10 : -1;          // the one edge that carries the protocol (producer phase index)
        auto fl = [&](int i) -> Flag* { return F ? F + i : nullptr; };
        // per phase p (0 qkv, 1 attention, 2 wo, 3 w1/w3, 4 w2): the flag it waits on, the flag it signals, the count it expects
        auto prev_of = [&](int p) -> Flag* { if (edge < 0) return fl((p + 4) % 5); return ((p + 4) % 5 == edge) ? flags + edge : nullptr; };
        auto mine_of = [&](int p) -> Flag* { if (edge < 0) return fl(p); return (p == edge) ? flags + edge : nullptr; };
        auto want_of = [&](int p) { return edge < 0 ? -1 : (p == 0 ? l : l + 1); };   // one signal per layer on that edge (edge 4 is consumed by the NEXT layer's qkv)
#define PH(NB, PAIR, grid, lds, W, rows, n, xin, xout, prev, mine, lead, want) do { hipStream_t s_ = st(); \
          if (mode == 2) hipLaunchKernelGGL((phase<NB, 6, PAIR>), dim3(grid), dim3(256), lds, s_, W, rows, n, xin, xout, prev, mine, err, lead, want); \
          else hipLaunchKernelGGL((phase<NB, 2, PAIR>), dim3(grid), dim3(256), lds, s_, W, rows, n, xin, xout, prev, mine, err, lead, want); } while (0)
        PH(8, false, 512, lds4096, wqkv, 3 * d, d, x, q, prev_of(0), mine_of(0), 0, want_of(0));
        { hipStream_t s_ = st(); hipLaunchKernelGGL(attn_like, dim3(32), dim3(256), 0, s_, cache, q, xb, prev_of(1), mine_of(1), err, 1, want_of(1)); }
        PH(8, false, 512, lds4096, wo, d, d, xb, x, prev_of(2), mine_of(2), 1, want_of(2));
        PH(8, true, 459, lds4096, w13, 2 * hd, d, x, hb, prev_of(3), mine_of(3), 1, want_of(3));
        PH(22, false, 512, lds11008, w2, d, hd, hb, x, prev_of(4), mine_of(4), 1, want_of(4));
      }
      if (two) { (void)hipEventRecord(ej, sb); (void)hipStreamWaitEvent(sa, ej, 0); }
      (void)hipEventRecord(e1, sa); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    int herr = 0; (void)hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
    float hx[4]; (void)hipMemcpy(hx, x + 100, 16, hipMemcpyDeviceToHost);
    printf("mode %d sleep %2d private %d fences %d  %s  %8.2f us per layer   (%.3f ms per 32 layers; bounded waits that gave up: %d; x[0..1] = %g %g)\n", mode, var[1], var[2], var[3],
           mode == 0 ? "one stream, plain launches          " : mode == 1 ? "two streams, hand-off, 2 batches pre" : mode == 2 ? "two streams, hand-off, 6 batches pre" : mode == 3 ? "ONE stream + the hand-off protocol   " :
           mode == 10 ? "protocol on qkv -> attention only   " : mode == 11 ? "protocol on attention -> wo only    " : mode == 12 ? "protocol on wo -> w1/w3 only        " : mode == 13 ? "protocol on w1/w3 -> w2 only        " : "protocol on w2 -> next qkv only     ", best * 1e3 / LAYERS, best, herr, hx[0], hx[1]);
  }
  return 0;
}

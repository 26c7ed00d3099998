// Can the HBM-idle window of the attention launch be used to pull the head of the NEXT matrix (wo) into the Infinity Cache?
//
// A Llama-2-7B layer is  wqkv (201 MB) -> attention (4.9 us, a few MB, HBM idle) -> wo (67 MB) -> ...; the streaming launches
// run at ~6.7 TB/s + ~2.3 us each.  tools/microbench_mall.hip showed that a prefetcher running BESIDE the streams only slows them
// down (it competes for HBM).  This one prefetches only inside the idle window: the stand-in for attention is a launch whose
// first 32 workgroups just wait 4.5 us; the other workgroups of the same launch read the first X MB of the next matrix with
// default-policy loads (they allocate in the memory-side cache) and throw them away.  Then the consumer streams the whole
// matrix with non-temporal loads as the library does.  If cache hits and HBM misses are served side by side the consumer gets
// shorter by about X / 6.7 TB/s; if they share one bottleneck nothing changes.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mbw tools/microbench_window.hip && /tmp/mbw
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) stream_nt(const f4* p, size_t n4, float* sink) {
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += 4 * stride) {
    f4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(p + (i + u * stride < n4 ? i + u * stride : i));
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += v[u];
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = 1.0f;
}

// POLICY 0: default loads, 1: nt loads, 2: sc1 loads
template <int POLICY>
__global__ void __launch_bounds__(256) window(const f4* next, size_t pf4, int idle_wgs, unsigned wait_ticks, float* sink) {
  if ((int)blockIdx.x < idle_wgs) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < wait_ticks) __builtin_amdgcn_s_sleep(2);
    return;
  }
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  const size_t stride = (size_t)(gridDim.x - idle_wgs) * 256;
  for (size_t i = (size_t)(blockIdx.x - idle_wgs) * 256 + threadIdx.x; i < pf4; i += 4 * stride) {
    f4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const f4* q = next + (i + u * stride < pf4 ? i + u * stride : i);
      if (POLICY == 0) v[u] = *q;
      else if (POLICY == 1) v[u] = __builtin_nontemporal_load(q);
      else asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[u]) : "v"(q) : "memory");
    }
    if (POLICY == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += v[u];
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = 1.0f;
}

// The same bytes, but every prefetching workgroup pulls what the consumer workgroups of ITS OWN XCD will ask for first (workgroup b of a
// launch runs on XCD b % 8; the consumer's workgroup b reads float4 b * 256 + tid + m * 512 * 256, m = 0, 1, ...): the lines then also
// sit in the L2 the consumer will look in, not only in the memory-side cache.  `shift` = 1 deliberately picks the WRONG XCD (control).
__global__ void __launch_bounds__(256) window_xcd(const f4* next, int rounds, int idle_wgs, unsigned wait_ticks, int shift, float* sink) {
  if ((int)blockIdx.x < idle_wgs) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < wait_ticks) __builtin_amdgcn_s_sleep(2);
    return;
  }
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  const int p = blockIdx.x - idle_wgs, P = gridDim.x - idle_wgs;       // idle_wgs and P are multiples of 8
  for (int j = p / 8; j < 64; j += P / 8) {
    const int b = ((p + shift) & 7) + 8 * j;                           // a consumer workgroup of this XCD
    for (int m = 0; m < rounds; m += 4) {
      f4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = next[(size_t)b * 256 + threadIdx.x + (size_t)(m + u < rounds ? m + u : m) * 512 * 256];
#pragma unroll
      for (int u = 0; u < 4; ++u) acc += v[u];
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = 1.0f;
}

int main() {
  const size_t POOL = (size_t)6 << 30, QKV = (size_t)201 << 20, WO = (size_t)67 << 20;
  char* pool; float* sink;
  if (hipMalloc(&pool, POOL) != hipSuccess) { printf("pool allocation failed\n"); return 1; }
  (void)hipMalloc(&sink, 64);
  (void)hipMemset(pool, 1, POOL);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int REPS = 20;
  printf("chain per repetition: stream 201 MB -> window launch (32 idle workgroups waiting 4.5 us + prefetchers) -> stream 67 MB; us per repetition\n");
  printf("%-44s %9s %9s\n", "prefetch inside the window", "us/rep", "vs none");
  double base = 0.0;
  struct Var { const char* name; int policy; size_t mb; int wgs; };
  const Var vars[] = {{"none", 0, 0, 0},
                      {"8 MB, default loads, 224 workgroups", 0, 8, 224},   {"16 MB, default loads, 224 workgroups", 0, 16, 224},
                      {"8 MB, default loads, 480 workgroups", 0, 8, 480},   {"16 MB, default loads, 480 workgroups", 0, 16, 480},
                      {"24 MB, default loads, 480 workgroups", 0, 24, 480}, {"32 MB, default loads, 480 workgroups", 0, 32, 480},
                      {"48 MB, default loads, 480 workgroups", 0, 48, 480}, {"67 MB, default loads, 480 workgroups", 0, 67, 480},
                      {"24 MB, nt loads, 480 workgroups", 1, 24, 480},      {"24 MB, sc1 loads, 480 workgroups", 2, 24, 480},
                      {"16 MB, own XCD's lines, 480 workgroups", 3, 16, 480}, {"24 MB, own XCD's lines, 480 workgroups", 3, 24, 480},
                      {"32 MB, own XCD's lines, 480 workgroups", 3, 32, 480}, {"24 MB, another XCD's lines, 480 workgroups", 4, 24, 480},
                      {"24 MB, own XCD's lines, 224 workgroups", 3, 24, 224}};
  for (const Var& v : vars) {
    float best = 1e30f;
    for (int it = 0; it < 5; ++it) {
      size_t off = 0;
      (void)hipEventRecord(e0, 0);
      for (int r = 0; r < REPS; ++r) {
        const f4* a = reinterpret_cast<const f4*>(pool + off); off += QKV;
        const f4* w = reinterpret_cast<const f4*>(pool + off); off += WO;
        if (off + QKV + WO > POOL) off = 0;
        hipLaunchKernelGGL(stream_nt, dim3(512), dim3(256), 0, 0, a, QKV / 16, sink);
        const size_t pf4 = (v.mb << 20) / 16;
        if (v.policy == 0) hipLaunchKernelGGL(window<0>, dim3(32 + v.wgs), dim3(256), 0, 0, w, pf4, 32, 450u, sink);
        else if (v.policy == 1) hipLaunchKernelGGL(window<1>, dim3(32 + v.wgs), dim3(256), 0, 0, w, pf4, 32, 450u, sink);
        else if (v.policy == 2) hipLaunchKernelGGL(window<2>, dim3(32 + v.wgs), dim3(256), 0, 0, w, pf4, 32, 450u, sink);
        else hipLaunchKernelGGL(window_xcd, dim3(32 + v.wgs), dim3(256), 0, 0, w, (int)(v.mb / 2), 32, 450u, v.policy == 4 ? 1 : 0, sink);
        hipLaunchKernelGGL(stream_nt, dim3(512), dim3(256), 0, 0, w, WO / 16, sink);
      }
      (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (it >= 1 && ms < best) best = ms;
    }
    const double us = best * 1e3 / REPS;
    if (v.mb == 0) base = us;
    printf("%-44s %9.2f %+9.2f\n", v.name, us, us - base);
  }
  return 0;
}

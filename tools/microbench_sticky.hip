// Can part of a model that does not fit the 256 MiB Infinity Cache be kept resident in it across tokens?
// stories110M streams 438 MB per token, so with every load allocating the same way the cache thrashes (0 % hits) and each of
// its 61 short launches pays HBM latency / bandwidth for its weights; back to back on ONE layer the same kernels run 20 - 30 %
// faster (profiles/r03: per_kernel vs kernel_stats).  If the loads of "the rest" could be made not to displace a chosen
// ~150 MB, that part would be served on chip every token.
// Here: table A (150 MB) is read every iteration with default-policy loads; between two reads of A, 300 MB of B are streamed
// with one of several policies / allocation types.  Reported: time of the A pass (us) and its rate.  If A stays resident its
// pass runs at cache speed whatever B does.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mbs tools/microbench_sticky.hip && /tmp/mbs
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int POLICY>   // 0 default, 1 nt, 2 sc1 (agent scope), 3 sc0 sc1 (system scope), 4 sc1 nt
__global__ void __launch_bounds__(256) sweep(const f4* p, size_t n4, float* sink) {
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += 4 * stride) {
    f4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const f4* q = p + (i + u * stride < n4 ? i + u * stride : i);
      if (POLICY == 0) v[u] = *q;
      else if (POLICY == 1) v[u] = __builtin_nontemporal_load(q);
      else if (POLICY == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v[u]) : "v"(q) : "memory");
      else if (POLICY == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v[u]) : "v"(q) : "memory");
      else asm volatile("global_load_dwordx4 %0, %1, off sc1 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v[u]) : "v"(q) : "memory");
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += v[u];
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = 1.0f;
}

int main() {
  const size_t A = (size_t)150 << 20, B = (size_t)300 << 20;
  f4 *a, *b, *bu; float* sink;
  (void)hipMalloc(&a, A); (void)hipMalloc(&b, B); (void)hipMalloc(&sink, 64);
  const bool have_uc = hipExtMallocWithFlags((void**)&bu, B, hipDeviceMallocUncached) == hipSuccess;
  (void)hipMemset(a, 1, A); (void)hipMemset(b, 2, B); if (have_uc) (void)hipMemset(bu, 3, B);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const char* names[] = {"B: default-policy loads", "B: nt loads", "B: sc1 loads", "B: sc0 sc1 loads", "B: sc1 nt loads", "B: uncached allocation, default loads", "no B at all (A alone)"};
  printf("A = 150 MB read with default loads every iteration; between two reads, 300 MB of B\n");
  for (int var = 0; var < 7; ++var) {
    if (var == 5 && !have_uc) continue;
    float best_a = 1e30f, best_b = 1e30f;
    for (int it = 0; it < 12; ++it) {
      float ms_b = 0.f;
      (void)hipEventRecord(e0, 0);
      const f4* bp = var == 5 ? bu : b;
      switch (var) {
        case 0: case 5: hipLaunchKernelGGL(sweep<0>, dim3(2048), dim3(256), 0, 0, bp, B / 16, sink); break;
        case 1: hipLaunchKernelGGL(sweep<1>, dim3(2048), dim3(256), 0, 0, bp, B / 16, sink); break;
        case 2: hipLaunchKernelGGL(sweep<2>, dim3(2048), dim3(256), 0, 0, bp, B / 16, sink); break;
        case 3: hipLaunchKernelGGL(sweep<3>, dim3(2048), dim3(256), 0, 0, bp, B / 16, sink); break;
        case 4: hipLaunchKernelGGL(sweep<4>, dim3(2048), dim3(256), 0, 0, bp, B / 16, sink); break;
        default: break;
      }
      (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms_b, e0, e1);
      (void)hipEventRecord(e0, 0);
      hipLaunchKernelGGL(sweep<0>, dim3(2048), dim3(256), 0, 0, a, A / 16, sink);
      (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
      float ms_a; (void)hipEventElapsedTime(&ms_a, e0, e1);
      if (it >= 2) { if (ms_a < best_a) best_a = ms_a; if (ms_b < best_b) best_b = ms_b; }
    }
    printf("%-42s A pass %7.1f us = %5.2f TB/s    B pass %7.1f us = %5.2f TB/s\n", names[var], best_a * 1e3, A / (best_a * 1e-3) / 1e12, best_b * 1e3, var == 6 ? 0.0 : B / (best_b * 1e-3) / 1e12);
  }
  return 0;
}

// Does what a launch pulled into an XCD's L2 survive the launch boundary?  (measurement first; nothing in the library uses this)
//
// A stories110M phase spends ~1.8 us between requesting its first weights and having them (tools/stamps_fused.py: requested 0.45 us,
// dots done 2.7 us) -- longer than its input vector takes (1.1 us).  Workgroup b of a launch lands on XCD b % 8 in every launch, so the
// workgroup that will want a slice of the NEXT phase's matrix could ask for it one launch early, behind its own weight requests, and
// find it in its XCD's 4 MB L2 -- IF the boundary between two dependent launches leaves the L2's clean lines alone.  This measures it:
//   consumer C: workgroup b requests its 48 KB slice of a 12 MB region (all loads in flight), stamps request -> last byte (s_memrealtime);
//   before it, in the same stream:  nothing it touched (cold) | producer P pulled the SAME slices by the same workgroups (same XCDs)
//                                   | P pulled the same region with the slices dealt to OTHER XCDs (b -> slice (b + 3) % grid)
//                                   | the region is small and hot in the Infinity Cache only (touched long ago by another mapping).
// Between trials a 512 MB sweep evicts L2 and Infinity Cache.  Output: mean / median us from request to last byte per case.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mbl2 tools/microbench_l2carry.hip && /tmp/mbl2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int GRID = 256, BLOCK = 512, PER = 6;                    // 512 threads x 6 float4 = 48 KB per workgroup, 12 MB per region

template <bool NT>
__device__ __forceinline__ float pull(const float* region, int slice) {
  const f4* p = reinterpret_cast<const f4*>(region) + (size_t)slice * BLOCK * PER + threadIdx.x;
  f4 v[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) v[k] = NT ? __builtin_nontemporal_load(p + k * BLOCK) : p[k * BLOCK];
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < PER; ++k) s += v[k].x + v[k].y + v[k].z + v[k].w;
  return s;
}

template <bool NT>
__global__ void __launch_bounds__(BLOCK) producer(const float* region, int shift, float* sink) {
  const float s = pull<NT>(region, (blockIdx.x + shift) % GRID);
  if (s == 123.456f) sink[blockIdx.x] = s;
}

template <bool NT>
__global__ void __launch_bounds__(BLOCK) consumer(const float* region, float* sink, unsigned* ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  const float s = pull<NT>(region, blockIdx.x);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (s == 123.456f) sink[blockIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = (unsigned)(t1 - t0);
}

__global__ void sweep(const float* big, size_t n4, float* sink) {
  float s = 0.0f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) { const f4 v = reinterpret_cast<const f4*>(big)[i]; s += v.x + v.w; }
  if (s == 123.456f) sink[0] = s;
}

int main() {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const size_t region_floats = (size_t)GRID * BLOCK * PER * 4, big_floats = (size_t)128 << 20;
  float *regions, *big, *sink;
  unsigned* ticks;
  CK(hipMalloc(&regions, region_floats * 4 * 9));
  CK(hipMalloc(&big, big_floats * 4));
  CK(hipMalloc(&sink, 4096));
  CK(hipMalloc(&ticks, GRID * 4));
  CK(hipMemset(regions, 0, region_floats * 4 * 9));
  CK(hipMemset(big, 0, big_floats * 4));
  struct Case { const char* name; int mode; };
  const Case cases[] = {{"cold (nothing pulled it before)", 0}, {"the launch before pulled the same slices on the same XCDs", 1},
                        {"the launch before pulled the region, slices dealt to other XCDs", 2},
                        {"pulled by the same workgroups two launches earlier (one unrelated launch in between)", 3},
                        {"in the Infinity Cache only (pulled with the other mapping, then 24 MB of other launches)", 4}};
  std::vector<unsigned> h(GRID);
  for (int nt = 0; nt < 4; ++nt) {                  // bit 0: the consumer's loads are non-temporal; bit 1: the producer's
  printf("== consumer loads %s, producer loads %s\n", (nt & 1) ? "non-temporal" : "plain", (nt & 2) ? "non-temporal" : "plain");
#define PROD(...) do { if (nt & 2) hipLaunchKernelGGL(producer<true>, dim3(GRID), dim3(BLOCK), 0, st, __VA_ARGS__); else hipLaunchKernelGGL(producer<false>, dim3(GRID), dim3(BLOCK), 0, st, __VA_ARGS__); } while (0)
  for (const Case& c : cases) {
    std::vector<double> all;
    for (int trial = 0; trial < 12; ++trial) {
      const float* r = regions + (size_t)(trial % 4) * region_floats;
      const float* other = regions + (size_t)(4 + trial % 4) * region_floats;
      hipLaunchKernelGGL(sweep, dim3(2048), dim3(256), 0, st, big, big_floats / 4, sink);
      if (c.mode == 1) PROD(r, 0, sink);
      if (c.mode == 2) PROD(r, 3, sink);
      if (c.mode == 3) { PROD(r, 0, sink); PROD(other, 0, sink); }
      if (c.mode == 4) { PROD(r, 3, sink); PROD(other, 0, sink); PROD(other + region_floats, 0, sink); }
      if (nt & 1) hipLaunchKernelGGL(consumer<true>, dim3(GRID), dim3(BLOCK), 0, st, r, sink, ticks);
      else hipLaunchKernelGGL(consumer<false>, dim3(GRID), dim3(BLOCK), 0, st, r, sink, ticks);
      CK(hipStreamSynchronize(st));
      CK(hipMemcpy(h.data(), ticks, GRID * 4, hipMemcpyDeviceToHost));
      if (trial >= 2) for (unsigned t : h) all.push_back(t * 0.01);          // 100 MHz
    }
    std::sort(all.begin(), all.end());
    double mean = 0.0;
    for (double v : all) mean += v;
    printf("%-95s: mean %5.2f us, median %5.2f, slowest workgroup %5.2f\n", c.name, mean / all.size(), all[all.size() / 2], all.back());
  }
  }
  return 0;
}

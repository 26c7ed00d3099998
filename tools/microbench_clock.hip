// What clock does the chip run at under load?  Shader-clock ticks (s_memtime) per microsecond of the constant 100 MHz
// clock (s_memrealtime), measured inside (a) a kernel that only issues fp64 MFMAs on every SIMD, (b) the same with fp64
// FMAs, (c) a kernel that only streams memory.  Cycle counts in this repository's notes assume ~2.1-2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mbc tools/microbench_clock.hip && /tmp/mbc
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned long long core_clock() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }
__device__ __forceinline__ unsigned long long real_clock() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }

template <int MODE> __global__ void __launch_bounds__(256) load(const f4* w, size_t n4, int iters, float* out, unsigned long long* res) {
  const unsigned long long c0 = core_clock(), r0 = real_clock();
  d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  double x = 1.0 + threadIdx.x, y = 0.5, s = 0;
  f4 acc = {0, 0, 0, 0};
  if (MODE == 0) for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
  }
  if (MODE == 1) for (int i = 0; i < iters * 16; ++i) { s = fma(s, x, y); x = fma(x, 1.0000001, s); }
  if (MODE == 2) { const size_t stride = (size_t)gridDim.x * 256; for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) acc += __builtin_nontemporal_load(w + i); }
  const unsigned long long c1 = core_clock(), r1 = real_clock();
  out[blockIdx.x * 256 + threadIdx.x] = (float)(a0[0] + a1[1] + a2[2] + a3[3] + s + x) + acc.x + acc.y + acc.z + acc.w;
  if (threadIdx.x == 0) { res[2 * blockIdx.x] = c1 - c0; res[2 * blockIdx.x + 1] = r1 - r0; }
}
int main() {
  const size_t bytes = (size_t)4 << 30;
  f4* w; float* out; unsigned long long* res;
  hipMalloc(&w, bytes); hipMemset(w, 1, bytes); hipMalloc(&out, 1 << 22); hipMallocManaged(&res, 8 * 4096);
  const char* names[] = {"fp64 MFMA on every SIMD", "fp64 FMA on every SIMD", "memory stream"};
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 3; ++mode) {
      const int grid = mode == 2 ? 1024 : 512;
      if (mode == 0) hipLaunchKernelGGL(load<0>, dim3(grid), dim3(256), 0, 0, w, bytes / 16, 60000, out, res);
      if (mode == 1) hipLaunchKernelGGL(load<1>, dim3(grid), dim3(256), 0, 0, w, bytes / 16, 60000, out, res);
      if (mode == 2) for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(load<2>, dim3(grid), dim3(256), 0, 0, w, bytes / 16, 0, out, res);
      hipDeviceSynchronize();
      double mn = 1e30, mx = 0, us = 0;
      for (int b = 0; b < grid; ++b) { const double f = (double)res[2 * b] / ((double)res[2 * b + 1] / 100.0); mn = f < mn ? f : mn; mx = f > mx ? f : mx; us = (double)res[2 * b + 1] / 100.0; }
      printf("%-28s shader clock %.0f - %.0f MHz over %.0f us\n", names[mode], mn, mx, us);
    }
  return 0;
}

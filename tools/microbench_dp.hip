// Issue cost of the fp64 building blocks of this path on gfx950: v_cvt_f64_f32, v_fma_f64, v_mfma_f64_16x16x4,
// and an integer-ALU widening of fp32 -> fp64 bits.  One wave per SIMD and four waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define N 4096
__device__ __forceinline__ double widen_int(float f) {   // exact for normal numbers and zero
  const unsigned u = __float_as_uint(f);
  const unsigned e = (u >> 23) & 0xff;
  const unsigned hi = (u & 0x80000000u) | ((e + 896u) << 20) | ((u & 0x7fffffu) >> 3);
  const unsigned lo = u << 29;
  return e ? __hiloint2double((int)hi, (int)lo) : 0.0;
}
template <int MODE> __global__ void k(float* out, const float* in, unsigned long long* cyc) {
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = in[threadIdx.x + 64 * i];
  double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  d4 m0 = {0, 0, 0, 0}, m1 = {0, 0, 0, 0};
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < N; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) a[i] += (double)x[i];                        // cvt + add
      if (MODE == 1) a[i] = fma(a[i], 1.0000001, 0.5);            // fma only
      if (MODE == 2) a[i] += widen_int(x[i]);                     // integer widen + add
      if (MODE == 3) a[i] = fma((double)x[i], 1.5, a[i]);         // cvt + fma (the GEMV inner op)
      x[i] += 1.0f;
    }
    if (MODE == 4) {
      m0 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)x[0], (double)x[1], m0, 0, 0, 0);
      m1 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)x[2], (double)x[3], m1, 0, 0, 0);
    }
    if (MODE == 5) {
      m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], a[1], m0, 0, 0, 0);
      m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], a[3], m1, 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  double s = m0[0] + m1[1];
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  float *in, *out; unsigned long long* cyc;
  hipMalloc(&in, 4096); hipMalloc(&out, 1 << 22); hipMallocManaged(&cyc, 8);
  hipMemset(in, 0, 4096);
  const char* names[] = {"8x (cvt_f64_f32 + add_f64 + add_f32)", "8x (fma_f64 + add_f32)", "8x (int widen + add_f64 + add_f32)",
                         "8x (cvt + fma_f64 + add_f32)", "2x mfma_f64 + 4 cvt + 8 add_f32", "2x mfma_f64 + 8 add_f32"};
  for (int waves = 1; waves <= 4; waves *= 4) {
    for (int mode = 0; mode < 6; ++mode) {
      dim3 g(256), b(256 * waves);
      if (waves == 4 && 256 * waves > 1024) { b = dim3(1024); }
      switch (mode) {
        case 0: hipLaunchKernelGGL(k<0>, g, b, 0, 0, out, in, cyc); break;
        case 1: hipLaunchKernelGGL(k<1>, g, b, 0, 0, out, in, cyc); break;
        case 2: hipLaunchKernelGGL(k<2>, g, b, 0, 0, out, in, cyc); break;
        case 3: hipLaunchKernelGGL(k<3>, g, b, 0, 0, out, in, cyc); break;
        case 4: hipLaunchKernelGGL(k<4>, g, b, 0, 0, out, in, cyc); break;
        default: hipLaunchKernelGGL(k<5>, g, b, 0, 0, out, in, cyc); break;
      }
      hipDeviceSynchronize();
      printf("%d wave(s)/SIMD  %-42s %7.1f cycles per loop iteration\n", waves, names[mode], (double)*cyc / N);
    }
  }
  return 0;
}

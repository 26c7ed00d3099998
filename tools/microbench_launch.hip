// Calibrates the per-kernel floor inside a replayed hipGraph on this chip: chains of N dependent kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty() {}
__global__ void k_touch(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0f; }
// one dependent global load -> store per thread
__global__ void k_chain1(const float* __restrict__ in, float* __restrict__ out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] + 1.0f;
}
// load -> wave reduce (6 bpermute steps on double) -> store
__global__ void k_reduce(const float* __restrict__ in, float* __restrict__ out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  double v = i < n ? (double)in[i] : 0.0;
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  if ((threadIdx.x & 63) == 0) out[i >> 6] = (float)v;
}

template <class F> float time_graph(hipStream_t st, int nk, int reps, F enq) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < nk; ++i) enq(i);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  hipEventRecord(e0, st);
  for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, st);
  hipEventRecord(e1, st); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return ms * 1e3f / (reps * nk);
}

int main() {
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  float *a, *b; CK(hipMalloc(&a, 1 << 22)); CK(hipMalloc(&b, 1 << 22));
  CK(hipMemset(a, 0, 1 << 22)); CK(hipMemset(b, 0, 1 << 22));
  const int nk = 62, reps = 50;
  for (int grid : {1, 256, 768, 2048}) {
    for (int block : {64, 256}) {
      float t0 = time_graph(st, nk, reps, [&](int) { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(block), 0, st); });
      float t1 = time_graph(st, nk, reps, [&](int i) { hipLaunchKernelGGL(k_chain1, dim3(grid), dim3(block), 0, st, (i & 1) ? a : b, (i & 1) ? b : a, grid * block); });
      float t2 = time_graph(st, nk, reps, [&](int i) { hipLaunchKernelGGL(k_reduce, dim3(grid), dim3(block), 0, st, (i & 1) ? a : b, (i & 1) ? b : a, grid * block); });
      float t3 = time_graph(st, nk, reps, [&](int i) { hipLaunchKernelGGL(k_chain1, dim3(grid), dim3(block), 16384, st, (i & 1) ? a : b, (i & 1) ? b : a, grid * block); });
      printf("grid %5d block %3d : empty %.2f us/kernel, load+store %.2f, load+wave-reduce+store %.2f, load+store with 16KB LDS %.2f\n", grid, block, t0, t1, t2, t3);
    }
  }
  return 0;
}

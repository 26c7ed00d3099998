// Does instruction fetch bound a short kernel?  Every kernel of a replayed graph starts with a cold instruction
// cache (the dispatch's acquire invalidates it), and a GEMV phase of a small model executes its few KB of code
// exactly once per wave.  Two kernels execute the SAME number of s_nop instructions: one as straight-line code
// (N x 4 bytes of footprint), one as a loop over a 256-byte body.  us per kernel in a chain of dependent launches.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mbi tools/microbench_icache.hip && /tmp/mbi
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int N>
__global__ void __launch_bounds__(256) straight(float* p) {
  asm volatile(".rept %0\n\ts_nop 0\n\t.endr" ::"i"(N));
  if (threadIdx.x == 0 && p[blockIdx.x] == 12345.f) p[blockIdx.x] = 1.f;
}
template <int N>
__global__ void __launch_bounds__(256) looped(float* p) {
  for (int i = 0; i < N / 64; ++i) asm volatile(".rept 64\n\ts_nop 0\n\t.endr");
  if (threadIdx.x == 0 && p[blockIdx.x] == 12345.f) p[blockIdx.x] = 1.f;
}

template <class K>
static float time_chain(K kernel, int grid, float* buf) {
  hipStream_t st; (void)hipStreamCreate(&st);
  hipGraph_t g; hipGraphExec_t ge;
  (void)hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  const int nk = 200;
  for (int i = 0; i < nk; ++i) hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, st, buf);
  (void)hipStreamEndCapture(st, &g);
  (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) (void)hipGraphLaunch(ge, st);
  (void)hipStreamSynchronize(st);
  (void)hipEventRecord(e0, st);
  const int reps = 20;
  for (int r = 0; r < reps; ++r) (void)hipGraphLaunch(ge, st);
  (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g); (void)hipStreamDestroy(st);
  return ms * 1e3f / (reps * nk);
}

int main() {
  float* buf; (void)hipMalloc(&buf, 1 << 20); (void)hipMemset(buf, 0, 1 << 20);
  for (int grid : {256, 512}) {
    printf("grid %d x 256 threads, us per kernel (straight-line / looped, same instruction count)\n", grid);
#define ROW(N) printf("  %6d s_nop = %5.1f KB of code: %6.2f / %6.2f\n", N, N * 4 / 1024.0, time_chain(straight<N>, grid, buf), time_chain(looped<N>, grid, buf))
    ROW(64); ROW(256); ROW(512); ROW(1024); ROW(2048); ROW(4096); ROW(8192);
  }
  return 0;
}

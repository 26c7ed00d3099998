// Does the shape of a wave's loads matter to HBM?  Streams a rows x n fp32 matrix two ways, no arithmetic to speak of:
//  A ("mfma"): the prefill GEMM's pattern -- one load instruction = 16 rows x 64 B (lane = (row j, 16-byte piece kq)),
//              4 consecutive instructions cover 256 B of each row, two register sets, 4 waves per 16-row tile split K.
//  B ("row") : one load instruction = 1 KB of ONE row (lane = 16-byte piece), the decode GEMV's pattern, 16 rows per tile.
// hipcc --offload-arch=gfx950 -O3 -o tools/microbench_pattern tools/microbench_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4 ldnt(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const f4*>(p)); }

template <int NW, int UN>
__global__ void __launch_bounds__(64 * NW) pat_mfma(const float* __restrict__ w, float* __restrict__ out, int n) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, kq = lane >> 4;
  const float* wrow = w + (size_t)(blockIdx.x * 16 + j) * n + 4 * kq;
  const int nblk = n / 16, npair = nblk / UN;
  f4 acc = {0, 0, 0, 0};
  f4 A[UN], B[UN];
  int p0 = wave;
  auto load = [&](f4 (&b)[UN], int p) { for (int u = 0; u < UN; ++u) b[u] = ldnt(wrow + 16 * (min(p, npair - 1) * UN + u)); };
  auto use = [&](const f4 (&b)[UN]) { for (int u = 0; u < UN; ++u) acc += b[u]; };
  if (p0 < npair) load(A, p0);
  while (p0 < npair) {
    const int p1 = p0 + NW;
    load(B, p1); use(A);
    if (p1 >= npair) break;
    const int p2 = p1 + NW;
    load(A, p2); use(B);
    p0 = p2;
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x * 64 * NW + threadIdx.x] = acc.x;
}

template <int NW, int UN>
__global__ void __launch_bounds__(64 * NW) pat_row(const float* __restrict__ w, float* __restrict__ out, int n) {
  // wave `wave` of the tile takes rows wave, wave + NW, ... of the 16; per row: 1 KB per instruction, UN instructions per set
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f4 acc = {0, 0, 0, 0};
  const int nchunk = n / 256, nb = nchunk / UN;     // 1 KB chunks per row, batches
  for (int r = wave; r < 16; r += NW) {
    const float* wrow = w + (size_t)(blockIdx.x * 16 + r) * n + 4 * lane;
    f4 A[UN], B[UN];
    auto load = [&](f4 (&b)[UN], int p) { for (int u = 0; u < UN; ++u) b[u] = ldnt(wrow + 256 * (min(p, nb - 1) * UN + u)); };
    auto use = [&](const f4 (&b)[UN]) { for (int u = 0; u < UN; ++u) acc += b[u]; };
    load(A, 0);
    for (int p = 0; p < nb; p += 2) { load(B, p + 1); use(A); load(A, p + 2); use(B); }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x * 64 * NW + threadIdx.x] = acc.x;
}

int main() {
  const int rows = 12288, n = 4096;
  const size_t bytes = (size_t)rows * n * 4;
  float *w, *out;
  CK(hipMalloc(&w, bytes * 2)); CK(hipMalloc(&out, 1 << 24));
  CK(hipMemset(w, 0, bytes * 2));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch(i & 1);
    hipEventRecord(e0);
    const int it = 20;
    for (int i = 0; i < it; ++i) launch(i & 1);      // alternate two 201 MB matrices: nothing stays in the 256 MB MALL
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %7.1f us  %6.2f TB/s\n", name, ms / it * 1e3, bytes / (ms / it * 1e-3) / 1e12);
  };
  run("mfma-shaped NW=4 UN=4", [&](int h) { hipLaunchKernelGGL((pat_mfma<4, 4>), dim3(rows / 16), dim3(256), 0, 0, w + (size_t)h * rows * n, out, n); });
  run("mfma-shaped NW=4 UN=8", [&](int h) { hipLaunchKernelGGL((pat_mfma<4, 8>), dim3(rows / 16), dim3(256), 0, 0, w + (size_t)h * rows * n, out, n); });
  run("mfma-shaped NW=8 UN=4", [&](int h) { hipLaunchKernelGGL((pat_mfma<8, 4>), dim3(rows / 16), dim3(512), 0, 0, w + (size_t)h * rows * n, out, n); });
  run("row-shaped  NW=4 UN=4", [&](int h) { hipLaunchKernelGGL((pat_row<4, 4>), dim3(rows / 16), dim3(256), 0, 0, w + (size_t)h * rows * n, out, n); });
  run("row-shaped  NW=4 UN=8", [&](int h) { hipLaunchKernelGGL((pat_row<4, 8>), dim3(rows / 16), dim3(256), 0, 0, w + (size_t)h * rows * n, out, n); });
  run("row-shaped  NW=8 UN=4", [&](int h) { hipLaunchKernelGGL((pat_row<8, 4>), dim3(rows / 16), dim3(512), 0, 0, w + (size_t)h * rows * n, out, n); });
  return 0;
}

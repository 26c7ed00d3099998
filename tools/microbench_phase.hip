// Floor of ONE dependent GEMV phase of a small model, step by step.
//
// A token of stories110M is 61 dependent launches (5 per layer + classifier); each phase is
//   [1] the launch boundary -> [2] x arrives (written by the previous launch) -> [3] rmsnorm: sum(x^2), scale, publish
//   -> [4] the weights arrive from HBM -> [5] fp64 FMAs -> [6] wave reduction + epilogue + store (seen by the next launch).
// This program replays a chain of 200 launches of a stripped phase kernel inside one hipGraph (each launch reads the
// vector the previous one wrote, exactly as the model does) and adds the steps one at a time, for the weight sizes of
// the stories110M phases.  us per launch; the differences are the floors of the steps.  No model code is involved:
// one wave per output row pair, 16-byte non-temporal loads, x staged by wave 0, DPP reductions, fp64 accumulate.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mbp tools/microbench_phase.hip && /tmp/mbp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int CTRL, int RM>
__device__ __forceinline__ double dpp(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  return __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, CTRL, RM, 0xf, false), __builtin_amdgcn_update_dpp(0, lo, CTRL, RM, 0xf, false));
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp<0xB1, 0xf>(v); v += dpp<0x4E, 0xf>(v); v += dpp<0x141, 0xf>(v); v += dpp<0x140, 0xf>(v);
  v += dpp<0x142, 0xa>(v); v += dpp<0x143, 0xc>(v);
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// STEPS: 1 empty | 2 + x in, one value out | 3 + norm published through LDS | 4 + weight stream consumed | 5 + reduction and row stores
template <int STEPS>
__global__ void __launch_bounds__(512) phase(const float* w, const float* xin, float* xout, int n, int rows, unsigned* progress = nullptr, unsigned id = 0) {
  if (progress && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(progress, id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (STEPS < 2) return;
  __shared__ f4 xs[192];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n4 = n / 4;
  // seven compute waves request their two rows first, the x wave requests x (the model's latency-form kernel)
  f4 a[2][2][3];                                       // up to two row pairs per wave, both requested up front
  const int g = (wave - 1) * gridDim.x + blockIdx.x, groups = rows / 2, g2 = g + 7 * gridDim.x;
  if (STEPS >= 4 && wave > 0 && g < groups) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (k == 1 && g2 >= groups) break;
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int u = 0; u < 3; ++u) a[k][r][u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(w + (size_t)(2 * (k ? g2 : g) + r) * n) + min(u * 64 + lane, n4 - 1));
    }
  }
  if (wave == 0) {
    f4 x[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) x[u] = reinterpret_cast<const f4*>(xin)[min(u * 64 + lane, n4 - 1)];
    if (STEPS >= 3) {
      double s = 0.0;
#pragma unroll
      for (int u = 0; u < 3; ++u) s += (double)x[u].x * x[u].x + (double)x[u].y * x[u].y + (double)x[u].z * x[u].z + (double)x[u].w * x[u].w;
      s = wave_sum(s);
      const double y = 1e-5 + s / n;
      double rs = __builtin_amdgcn_rsq(y);
      rs = fma(rs, fma(-0.5 * y * rs, rs, 0.5), rs);
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        f4 o = {(float)(rs * x[u].x), (float)(rs * x[u].y), (float)(rs * x[u].z), (float)(rs * x[u].w)};
        xs[u * 64 + lane] = o;
      }
    } else if (lane == 0 && blockIdx.x == 0) {
      xout[0] = x[0].x + 1.0f;
    }
  }
  if (STEPS < 3) return;
  __syncthreads();
  if (wave == 0 || g >= groups) { if (STEPS == 3 && tid == 0 && blockIdx.x == 0) xout[0] = xs[0].x; return; }
  if (STEPS == 3) return;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int gk = k ? g2 : g;
    if (gk >= groups) break;
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const f4 x = xs[u * 64 + lane];
      acc0 += (double)a[k][0][u].x * x.x + (double)a[k][0][u].y * x.y + (double)a[k][0][u].z * x.z + (double)a[k][0][u].w * x.w;
      acc1 += (double)a[k][1][u].x * x.x + (double)a[k][1][u].y * x.y + (double)a[k][1][u].z * x.z + (double)a[k][1][u].w * x.w;
    }
    if (STEPS == 4) { if (acc0 + acc1 == 12345.678) xout[0] = 1.0f; continue; }   // consumed, nothing reduced or stored
    acc0 = wave_sum(acc0); acc1 = wave_sum(acc1);
    if (lane < 2 && 2 * gk + lane < n) xout[2 * gk + lane] = (float)(lane ? acc1 : acc0) * 1e-3f;
  }
}

// A prefetcher BESIDE the chain (a parallel branch of the same hipGraph): PW workgroups per XCD (workgroup b runs on XCD b % 8), which pull
// the rows that the phase workgroups of their own XCD will read (row group g -> workgroup g % grid -> XCD g % 8) LEAD launches before the
// launch that needs them starts, into that XCD's L2.  Paced by the progress word the phases publish; never waits longer than ~30 us.
__global__ void __launch_bounds__(256) l2_prefetcher(const float* w, int n, int rows, int nk, int lead, const unsigned* progress, float* sink) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3, PW = gridDim.x >> 3, n4 = n / 4, groups = rows / 2;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int s = 1; s < nk; ++s) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((int)__hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + lead < s + 1 && __builtin_amdgcn_s_memrealtime() - t0 < 3000) __builtin_amdgcn_s_sleep(8);
    const f4* m = reinterpret_cast<const f4*>(w + (size_t)(s % 64) * 4096 * n);
    const int per = (groups - xcd + 7) >> 3;
    for (int i = k * 4 + wave; i < per; i += 4 * PW * 4) {
      f4 v[4][2][3];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int g = xcd + 8 * min(i + q * PW * 4, per - 1);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int u = 0; u < 3; ++u) v[q][r][u] = m[(size_t)(2 * g + r) * n4 + min(u * 64 + lane, n4 - 1)];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int u = 0; u < 3; ++u) acc += v[q][r][u];
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = 1.0f;
}

static float chain_prefetched(const float* w, float* xa, float* xb, int n, int rows, int PW, int lead, unsigned* progress) {
  hipStream_t st, s2; (void)hipStreamCreate(&st); (void)hipStreamCreate(&s2);
  hipGraph_t g; hipGraphExec_t ge;
  hipEvent_t fork, join; (void)hipEventCreateWithFlags(&fork, hipEventDisableTiming); (void)hipEventCreateWithFlags(&join, hipEventDisableTiming);
  const int groups = rows / 2, grid = groups / 7 + 1 > 256 ? 256 : groups / 7 + 1, nk = 200;
  (void)hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  (void)hipMemsetAsync(progress, 0, 4, st);
  (void)hipEventRecord(fork, st); (void)hipStreamWaitEvent(s2, fork, 0);
  if (PW > 0) hipLaunchKernelGGL(l2_prefetcher, dim3(8 * PW), dim3(256), 0, s2, w, n, rows, nk, lead, progress, xa + 8192);
  for (int i = 0; i < nk; ++i) hipLaunchKernelGGL(phase<5>, dim3(grid), dim3(512), 0, st, w + (size_t)(i % 64) * 4096 * n, (i & 1) ? xb : xa, (i & 1) ? xa : xb, n, rows, progress, (unsigned)(i + 1));
  (void)hipEventRecord(join, s2); (void)hipStreamWaitEvent(st, join, 0);
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
  (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g); (void)hipStreamDestroy(st); (void)hipStreamDestroy(s2);
  return ms * 1e3f / (reps * nk);
}

__global__ void fill(float* p, size_t n) {   // pseudo-random weights: zero-filled operands let the chip clock higher (MI355X guide, DVFS)
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    p[i] = ((float)(x & 0xffff) - 32768.0f) * 1e-6f;
  }
}

template <int STEPS>
static float chain(const float* w, float* xa, float* xb, int n, int rows) {
  hipStream_t st; (void)hipStreamCreate(&st);
  hipGraph_t g; hipGraphExec_t ge;
  const int groups = rows / 2, grid = groups / 7 + 1 > 256 ? 256 : groups / 7 + 1, nk = 200;
  (void)hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < nk; ++i) hipLaunchKernelGGL(phase<STEPS>, dim3(grid), dim3(512), 0, st, w + (size_t)(i % 64) * 4096 * n, (i & 1) ? xb : xa, (i & 1) ? xa : xb, n, rows);
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
  const int n = 768;
  struct { const char* name; int rows; } shapes[] = {{"wo  (768 x 768, 2.4 MB)", 768}, {"qkv (2304 x 768, 7.1 MB)", 2304}, {"w13 (4096 x 768, 12.6 MB)", 4096}};
  float *w, *xa, *xb;
  (void)hipMalloc(&w, (size_t)4096 * n * 4 * 64);      // 64 copies, one per launch in turn (805 MB): the weights come from HBM, not from the 256 MB Infinity Cache
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, w, (size_t)4096 * n * 64);
  (void)hipDeviceSynchronize();
  (void)hipMalloc(&xa, 65536); (void)hipMalloc(&xb, 65536);
  std::vector<float> h(16384, 0.01f);
  (void)hipMemcpy(xa, h.data(), 65536, hipMemcpyHostToDevice); (void)hipMemcpy(xb, h.data(), 65536, hipMemcpyHostToDevice);
  printf("us per launch in a chain of dependent launches (hipGraph replay), n = %d columns\n", n);
  printf("%-28s %8s %8s %8s %8s %8s\n", "rows of the phase", "empty", "+x", "+norm", "+weights", "+reduce/store");
  for (auto& s : shapes) {
    printf("%-28s %8.2f %8.2f %8.2f %8.2f %8.2f\n", s.name, chain<1>(w, xa, xb, n, s.rows), chain<2>(w, xa, xb, n, s.rows),
           chain<3>(w, xa, xb, n, s.rows), chain<4>(w, xa, xb, n, s.rows), chain<5>(w, xa, xb, n, s.rows));
  }
  unsigned* progress; (void)hipMalloc(&progress, 64);
  printf("\nthe full phase with a prefetcher beside the chain (PW workgroups per XCD pull the rows of launch s into the L2 of the XCD that will read them, LEAD launches ahead)\n");
  printf("%-28s %8s %8s %8s %8s %8s %8s %8s\n", "rows of the phase", "none", "12, L1", "12, L2", "24, L1", "24, L2", "24, L3", "8, L2");
  for (auto& s : shapes) {
    printf("%-28s %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f\n", s.name, chain_prefetched(w, xa, xb, n, s.rows, 0, 0, progress),
           chain_prefetched(w, xa, xb, n, s.rows, 12, 1, progress), chain_prefetched(w, xa, xb, n, s.rows, 12, 2, progress),
           chain_prefetched(w, xa, xb, n, s.rows, 24, 1, progress), chain_prefetched(w, xa, xb, n, s.rows, 24, 2, progress),
           chain_prefetched(w, xa, xb, n, s.rows, 24, 3, progress), chain_prefetched(w, xa, xb, n, s.rows, 8, 2, progress));
  }
  printf("(every launch streams its own copy of the matrix from HBM: bytes / 6.3 TB/s = 0.4 / 1.1 / 2.0 us of the '+weights' step for the three shapes)\n");
  return 0;
}

// How a GEMV should walk a row-major matrix on MI355X: us per link of a chain of dependent kernels that only stream
// (float4 non-temporal loads, summed), for the Llama-2-7B phase shapes and several ways of dealing the bytes to the waves.
//   stream    grid-stride over the whole matrix (what a copy kernel does): the reference point
//   rows      a wave owns R = 2 rows at a time (adjacent rows, or row i of two matrices as w1 / w3) and reads them front to
//             back in batches of U x 1 KB per row, two batches in flight -- libllama2hip's streaming form
//   rows+rot  the same, every row group starting at another batch and wrapping around
//   +stagger  the second row of the group half a row ahead of the first
//   cols      one 512-thread workgroup per CU owns a block of row groups, wave w streams column batch w of each of them
//   +fma      rows+rot with the GEMV's arithmetic (x from LDS, fp64 widening and FMA of every weight)
//   packed    the rows loop, but the matrix REPACKED in the order the chip consumes it: [step][wave][row of the group][u][lane], so that
//             what all waves ask for at one moment is one contiguous stretch (2048 waves x 4 KB = 8 MB), as in `stream`
// With all waves marching through their rows in step, the requests in flight at one moment sit a whole row (16 / 44 KB)
// apart; rotating the start spreads them over the HBM channels.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mbr tools/microbench_rows.hip && /tmp/mbr
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void fill(float* p, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { unsigned x = (unsigned)i * 2654435761u; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; p[i] = ((float)(x & 0xffff) - 32768.0f) * 1e-6f; } }

__global__ void __launch_bounds__(256) stream(const f4* w, size_t n4, const float* carry_in, float* carry_out) {
  const float c = carry_in[0];
  f4 acc = {c, 0.f, 0.f, 0.f};
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    f4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(w + i + u * stride);
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += v[u];
  }
  for (; i < n4; i += stride) acc += w[i];
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) carry_out[1] = 1.0f;
  if (blockIdx.x == 0 && threadIdx.x == 0) carry_out[0] = c * 0.5f + 1.0f;
}

// PAIR: the group's two rows are row g of the first and of the second half of the matrix (w1 / w3); else rows 2g, 2g + 1
// TAILPF: every wave, once its own rows are done, pulls TAILPF KiB of the NEXT link's matrix (plain loads: they allocate in L2 / the
// Infinity Cache) -- in the pattern the next launch's first batches will ask for them -- so that the next launch finds its first bytes on chip
// DRAIN: wait for everything in flight at the top of every iteration, as the library's loop did (hipcc put an s_waitcnt vmcnt(0) in
// front of the first address computation of the iteration: it reuses the other register set's registers for it)
template <bool PAIR, int ROT, bool STAGGER, int WORK = 0, int TAILPF = 0, int PACKED = 0, int DRAIN = 0, int U = 2>
__global__ void __launch_bounds__(256) rows2(const f4* w, int rows, int n, const float* carry_in, float* carry_out, const f4* next_w = nullptr) {
  const float c = carry_in[0];
  f4 acc = {c, 0.f, 0.f, 0.f};
  // WORK 1: the GEMV's arithmetic as well -- x (n floats) staged in LDS, every weight widened to fp64 and multiplied in
  __shared__ f4 xs[WORK ? 2816 : 1];
  double d0 = 0.0, d1 = 0.0;
  if (WORK) {
    for (int i = threadIdx.x; i < n / 4; i += 256) { const float v = 1e-3f * (float)(i & 7); xs[i] = f4{v, v, v, v}; }
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n4 = n / 4, batches = (n4 + 64 * U - 1) / (64 * U), groups = rows / 2, tw = gridDim.x * 4;
  auto load = [&](f4 (&b)[2][U], int g, int ci) {
    if (PACKED) {
      const int gw = blockIdx.x * 4 + wave;
      const size_t k = (size_t)((g - gw) / tw) * batches + ci;
      // PACKED = 1 + m: the waves' places within a step rotate by m workgroups (4 m waves) per step, so that the stretch an XCD reads
      // (workgroup b runs on XCD b % 8) moves through all residues of the address instead of keeping one
      const int place = PACKED > 1 ? (int)(((size_t)gw + (size_t)4 * (PACKED - 1) * k) % (size_t)tw) : gw;
      size_t base = (k * tw + place) * (size_t)(2 * U * 64);
      const size_t lim = (size_t)rows * n4 - 2 * U * 64;
      if (base > lim) base = lim;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        b[0][u] = __builtin_nontemporal_load(w + base + u * 64 + lane);
        b[1][u] = __builtin_nontemporal_load(w + base + (U + u) * 64 + lane);
      }
      return;
    }
    const f4* r0 = w + (size_t)(PAIR ? g : 2 * g) * n4;
    const f4* r1 = w + (size_t)(PAIR ? groups + g : 2 * g + 1) * n4;
    const int rot = ROT ? (g * ROT) % batches : 0;
    int c0 = ci + rot; c0 -= c0 >= batches ? batches : 0;
    int c1 = c0 + (STAGGER ? batches / 2 : 0); c1 -= c1 >= batches ? batches : 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      b[0][u] = __builtin_nontemporal_load(r0 + min(c0 * 64 * U + u * 64 + lane, n4 - 1));
      b[1][u] = __builtin_nontemporal_load(r1 + min(c1 * 64 * U + u * 64 + lane, n4 - 1));
    }
  };
  f4 A[2][U], B[2][U];
  int g = blockIdx.x * 4 + wave, ci = 0;
  if (g < groups) load(A, g, 0);
  while (g < groups) {
    if (DRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int g2 = g, c2 = ci + 1;
    if (c2 == batches) { c2 = 0; g2 += tw; }
    load(B, g2 < groups ? g2 : g, g2 < groups ? c2 : ci);
    if (WORK) {
      const int rot = ROT ? (g * ROT) % batches : 0;
      int c0 = ci + rot; c0 -= c0 >= batches ? batches : 0;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const f4 xv = xs[min(c0 * 64 * U + u * 64 + lane, n4 - 1)];
        const double x0 = xv.x, x1 = xv.y, x2 = xv.z, x3 = xv.w;
        d0 += (double)A[0][u].x * x0; d0 += (double)A[0][u].y * x1; d0 += (double)A[0][u].z * x2; d0 += (double)A[0][u].w * x3;
        d1 += (double)A[1][u].x * x0; d1 += (double)A[1][u].y * x1; d1 += (double)A[1][u].z * x2; d1 += (double)A[1][u].w * x3;
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) { acc += A[0][u]; acc += A[1][u]; }
    }
    if (g2 >= groups) break;
    int g3 = g2, c3 = c2 + 1;
    if (c3 == batches) { c3 = 0; g3 += tw; }
    load(A, g3 < groups ? g3 : g2, g3 < groups ? c3 : c2);
    if (WORK) {
      const int rot = ROT ? (g2 * ROT) % batches : 0;
      int c0 = c2 + rot; c0 -= c0 >= batches ? batches : 0;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const f4 xv = xs[min(c0 * 64 * U + u * 64 + lane, n4 - 1)];
        const double x0 = xv.x, x1 = xv.y, x2 = xv.z, x3 = xv.w;
        d0 += (double)B[0][u].x * x0; d0 += (double)B[0][u].y * x1; d0 += (double)B[0][u].z * x2; d0 += (double)B[0][u].w * x3;
        d1 += (double)B[1][u].x * x0; d1 += (double)B[1][u].y * x1; d1 += (double)B[1][u].z * x2; d1 += (double)B[1][u].w * x3;
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) { acc += B[0][u]; acc += B[1][u]; }
    }
    g = g3; ci = c3;
  }
  if (TAILPF > 0 && next_w) {
    // the next launch's batch 0 (and 1, ...) of this wave's first row group: same rows, same rotated columns
    const int g0 = blockIdx.x * 4 + wave;
    if (g0 < groups) {
#pragma unroll
      for (int b = 0; b < TAILPF / 4; ++b) {
        const f4* r0 = next_w + (size_t)(PAIR ? g0 : 2 * g0) * n4;
        const f4* r1 = next_w + (size_t)(PAIR ? groups + g0 : 2 * g0 + 1) * n4;
        const int rot = ROT ? (g0 * ROT) % batches : 0;
        int c0 = b + rot; c0 -= c0 >= batches ? batches : 0;
#pragma unroll
        for (int u = 0; u < U; ++u) { acc += r0[min(c0 * 64 * U + u * 64 + lane, n4 - 1)]; acc += r1[min(c0 * 64 * U + u * 64 + lane, n4 - 1)]; }
      }
    }
  }
  if (acc.x + acc.y + acc.z + acc.w + (float)(d0 + d1) == 12345.678f) carry_out[1] = 1.0f;
  if (blockIdx.x == 0 && threadIdx.x == 0) carry_out[0] = c * 0.5f + 1.0f;
}

// packed layout, D register sets: D - 1 batches always in flight per wave (rows2 keeps 1 - 2); the GEMV's arithmetic included
template <int D>
__device__ __forceinline__ void packdeep_body(const f4* w, int rows, int n, const f4* xs, float* carry_out, int gw_ = -1, int tw_ = 0);

// Fewer workgroups on the odd XCDs: 8 * S workgroups are launched (workgroup b lands on XCD b % 8, S per XCD), those with slot b / 8 >= n_odd on an
// odd XCD leave at once, the others share the work as 8 * S - 4 * (S - n_odd) dense workgroups.
__global__ void __launch_bounds__(256) packskew(const f4* w, int rows, int n, const float* carry_in, float* carry_out, int n_odd, unsigned* stamp) {
  const float c = carry_in[0];
  __shared__ f4 xs[2816];
  const int S = gridDim.x / 8, x = blockIdx.x & 7, slot = blockIdx.x >> 3;
  if (blockIdx.x == 0 && threadIdx.x == 0) carry_out[0] = c * 0.5f + 1.0f;
  if ((x & 1) && slot >= n_odd) { if (stamp && threadIdx.x == 0) { stamp[2 * blockIdx.x] = 99; stamp[2 * blockIdx.x + 1] = 0; } return; }
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int i = threadIdx.x; i < n / 4; i += 256) { const float v = 1e-3f * (float)(i & 7) + c; xs[i] = f4{v, v, v, v}; }
  __syncthreads();
  const int dense = slot < n_odd ? slot * 8 + x : n_odd * 8 + (slot - n_odd) * 4 + (x >> 1);
  const int working = 8 * S - 4 * (S - n_odd);
  packdeep_body<2>(w, rows, n, xs, carry_out, dense * 4 + (int)(threadIdx.x >> 6), working * 4);
  __syncthreads();
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (stamp && threadIdx.x == 0) { stamp[2 * blockIdx.x] = xcc & 0xf; stamp[2 * blockIdx.x + 1] = (unsigned)(__builtin_amdgcn_s_memrealtime() - t0); }
}

// DS sets for the workgroups on the XCDs of `slow_mask`, DF for the others; `stamp` (or null): per workgroup {XCD, end - start in 10 ns}
template <int DF, int DS>
__global__ void __launch_bounds__(256) packmix(const f4* w, int rows, int n, const float* carry_in, float* carry_out, unsigned slow_mask, unsigned* stamp) {
  const float c = carry_in[0];
  __shared__ f4 xs[2816];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int i = threadIdx.x; i < n / 4; i += 256) { const float v = 1e-3f * (float)(i & 7) + c; xs[i] = f4{v, v, v, v}; }
  __syncthreads();
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 0xf;
  if ((slow_mask >> xcc) & 1) packdeep_body<DS>(w, rows, n, xs, carry_out); else packdeep_body<DF>(w, rows, n, xs, carry_out);
  __syncthreads();
  if (stamp && threadIdx.x == 0) { stamp[2 * blockIdx.x] = xcc; stamp[2 * blockIdx.x + 1] = (unsigned)(__builtin_amdgcn_s_memrealtime() - t0); }
  if (blockIdx.x == 0 && threadIdx.x == 0) carry_out[0] = c * 0.5f + 1.0f;
}

template <int D>
__global__ void __launch_bounds__(256) packdeep(const f4* w, int rows, int n, const float* carry_in, float* carry_out) {
  const float c = carry_in[0];
  __shared__ f4 xs[2816];
  for (int i = threadIdx.x; i < n / 4; i += 256) { const float v = 1e-3f * (float)(i & 7); xs[i] = f4{v, v, v, v}; }
  __syncthreads();
  packdeep_body<D>(w, rows, n, xs, carry_out);
  if (blockIdx.x == 0 && threadIdx.x == 0) carry_out[0] = c * 0.5f + 1.0f;
}

template <int D>
__device__ __forceinline__ void packdeep_body(const f4* w, int rows, int n, const f4* xs, float* carry_out, int gw_, int tw_) {
  constexpr int U = 2;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n4 = n / 4, batches = (n4 + 64 * U - 1) / (64 * U), groups = rows / 2, tw = tw_ ? tw_ : gridDim.x * 4, gw = gw_ >= 0 ? __builtin_amdgcn_readfirstlane(gw_) : blockIdx.x * 4 + wave;
  const int mine = gw < groups ? (groups - gw + tw - 1) / tw : 0, T = mine * batches;       // this wave's steps
  const size_t lim = (size_t)rows * n4 - 2 * U * 64;
  f4 buf[D][2][U];
  auto load = [&](f4 (&b)[2][U], int s) {
    s = s < T ? s : T - 1;
    size_t base = ((size_t)s * tw + gw) * (size_t)(2 * U * 64);
    if (base > lim) base = lim;
#pragma unroll
    for (int u = 0; u < U; ++u) { b[0][u] = __builtin_nontemporal_load(w + base + u * 64 + lane); b[1][u] = __builtin_nontemporal_load(w + base + (U + u) * 64 + lane); }
  };
  double d0 = 0.0, d1 = 0.0;
  auto use = [&](const f4 (&b)[2][U], int s) {
    const int c0 = s % batches;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const f4 xv = xs[min(c0 * 64 * U + u * 64 + lane, n4 - 1)];
      const double x0 = xv.x, x1 = xv.y, x2 = xv.z, x3 = xv.w;
      d0 += (double)b[0][u].x * x0; d0 += (double)b[0][u].y * x1; d0 += (double)b[0][u].z * x2; d0 += (double)b[0][u].w * x3;
      d1 += (double)b[1][u].x * x0; d1 += (double)b[1][u].y * x1; d1 += (double)b[1][u].z * x2; d1 += (double)b[1][u].w * x3;
    }
  };
  if (T > 0) {
#pragma unroll
    for (int d = 0; d < D - 1; ++d) load(buf[d], d);
    for (int s = 0; s < T; s += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        load(buf[(d + D - 1) % D], s + d + D - 1);
        if (s + d < T) use(buf[d], s + d);
      }
    }
  }
  if ((float)(d0 + d1) == 12345.678f) carry_out[1] = 1.0f;
}

// cols: one 512-thread workgroup per CU owns a contiguous block of row groups; wave w streams column batch w (2 KB per row)
// of every group of the block -- every wave moves the same number of bytes, nothing is left to a last round
template <bool PAIR>
__global__ void __launch_bounds__(512) cols8(const f4* w, int rows, int n, const float* carry_in, float* carry_out) {
  constexpr int U = 2;
  const float c = carry_in[0];
  f4 acc = {c, 0.f, 0.f, 0.f};
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n4 = n / 4, groups = rows / 2;
  const int per = (groups + gridDim.x - 1) / gridDim.x, g0 = blockIdx.x * per, g1 = min(groups, g0 + per);
  auto load = [&](f4 (&b)[2][U], int g) {
    const f4* r0 = w + (size_t)(PAIR ? g : 2 * g) * n4 + wave * 64 * U;
    const f4* r1 = w + (size_t)(PAIR ? groups + g : 2 * g + 1) * n4 + wave * 64 * U;
#pragma unroll
    for (int u = 0; u < U; ++u) { b[0][u] = __builtin_nontemporal_load(r0 + u * 64 + lane); b[1][u] = __builtin_nontemporal_load(r1 + u * 64 + lane); }
  };
  f4 A[2][U], B[2][U];
  int g = g0;
  if (g < g1) load(A, g);
  while (g < g1) {
    load(B, g + 1 < g1 ? g + 1 : g);
#pragma unroll
    for (int u = 0; u < U; ++u) { acc += A[0][u]; acc += A[1][u]; }
    if (g + 1 >= g1) break;
    load(A, g + 2 < g1 ? g + 2 : g + 1);
#pragma unroll
    for (int u = 0; u < U; ++u) { acc += B[0][u]; acc += B[1][u]; }
    g += 2;
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) carry_out[1] = 1.0f;
  if (blockIdx.x == 0 && threadIdx.x == 0) carry_out[0] = c * 0.5f + 1.0f;
}

int main(int argc, char** argv) {
  const int GRID = argc > 1 ? atoi(argv[1]) : 512;      // workgroups of the rows / packed variants (the library's balanced w1/w3 grid: 459)
  const size_t total = (size_t)6 << 30;
  f4* w; float* carry;
  (void)hipMalloc(&w, total); (void)hipMalloc(&carry, 1 << 16); (void)hipMemset(carry, 0, 1 << 16);
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (float*)w, total / 4); (void)hipDeviceSynchronize();
  hipStream_t sa; (void)hipStreamCreate(&sa);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  printf("us per link of a chain of dependent streaming kernels (launch boundary included), %d workgroups of 256 threads\n", GRID);
  printf("%-34s %8s %8s %8s %8s %8s %8s %8s %8s %8s %8s %8s %8s %8s %8s %8s %8s %8s %8s %8s %8s\n", "shape", "stream", "rows", "rot 5", "rot 5+st", "rot 3", "rot 7", "rot5+fma", "cols", "fma+pf4K", "fma+pf8K", "packed", "pack+fma", "rot+drn", "pack+drn", "pack U=4", "pack U=1", "deep 2", "deep 3", "deep 4", "deep 6");
  struct { const char* name; int rows, n; bool pair; } shapes[] = {{"w1+w3  2 x 11008 x 4096 (361 MB)", 22016, 4096, true}, {"wqkv   12288 x 4096 (201 MB)", 12288, 4096, false},
                                                                   {"w2     4096 x 11008 (180 MB)", 4096, 11008, false}, {"wo     4096 x 4096 (67 MB)", 4096, 4096, false},
                                                                   {"wcls   32000 x 4096 (524 MB)", 32000, 4096, false}};
  if (argc > 2 && !strcmp(argv[2], "skew")) {   // fewer workgroups on the odd XCDs: 8 * n_even are launched, the odd XCDs keep n_odd each
    unsigned* stamp; (void)hipMalloc(&stamp, 16 * 1024);
    unsigned host[4096];
    const int total_wg = argc > 3 ? atoi(argv[3]) : 512;                  // working workgroups = 4 * (n_even + n_odd)
    for (int si = 0; si < 5; ++si) {
      const auto sh = shapes[si];
      const size_t link4 = (size_t)sh.rows * sh.n / 4; const int nk = (int)(total / 16 / link4);
      for (int d = 0; d <= 8; d += 2) {
        const int n_even = total_wg / 8 + d / 2 + (total_wg % 8 ? 1 : 0), n_odd = total_wg / 4 - n_even;
        if (n_even > 64) break;
        const int S = n_even;
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
          (void)hipEventRecord(e0, sa);
          for (int k = 0; k < nk; ++k) hipLaunchKernelGGL(packskew, dim3(8 * S), dim3(256), 0, sa, w + (size_t)k * link4, sh.rows, sh.n, carry + 16 * k, carry + 16 * (k + 1), n_odd, k == nk - 1 ? stamp : nullptr);
          (void)hipEventRecord(e1, sa); (void)hipEventSynchronize(e1);
          float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        (void)hipMemcpy(host, stamp, 8 * S * 8, hipMemcpyDeviceToHost);
        printf("%-32s %2d workgroups per even XCD, %2d per odd: %6.2f us per link; lifetime by XCD (median):", sh.name, n_even, n_odd, best * 1e3 / nk);
        for (unsigned x = 0; x < 8; ++x) {
          float v[1024]; int m = 0;
          for (int b = 0; b < 8 * S; ++b) if (host[2 * b] == x) v[m++] = host[2 * b + 1] / 100.0f;
          for (int i = 1; i < m; ++i) for (int j = i; j > 0 && v[j] < v[j - 1]; --j) { float t = v[j]; v[j] = v[j - 1]; v[j - 1] = t; }
          printf(" %5.1f", m ? v[m / 2] : 0.f);
        }
        float mx = 0; for (int b = 0; b < 8 * S; ++b) if (host[2 * b] != 99 && host[2 * b + 1] / 100.0f > mx) mx = host[2 * b + 1] / 100.0f;
        printf("  max %5.1f\n", mx);
      }
    }
    return 0;
  }
  if (argc > 2 && !strcmp(argv[2], "xcd")) {   // when do the workgroups of the packed stand-in end, by XCD -- and does a deeper pipeline on the late XCDs even it out?
    unsigned* stamp; (void)hipMalloc(&stamp, 8 * 1024);
    unsigned host[2048];
    const unsigned masks[] = {0u, 0x28u, 0xaau, 0xffu};
    for (int si = 0; si < 4; ++si) {
      const auto sh = shapes[si == 3 ? 1 : 0];
      const size_t link4 = (size_t)sh.rows * sh.n / 4; const int nk = (int)(total / 16 / link4);
      for (unsigned mask : masks) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
          (void)hipEventRecord(e0, sa);
          for (int k = 0; k < nk; ++k) hipLaunchKernelGGL((packmix<2, 3>), dim3(GRID), dim3(256), 0, sa, w + (size_t)k * link4, sh.rows, sh.n, carry + 16 * k, carry + 16 * (k + 1), mask, k == nk - 1 ? stamp : nullptr);
          (void)hipEventRecord(e1, sa); (void)hipEventSynchronize(e1);
          float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        (void)hipMemcpy(host, stamp, GRID * 8, hipMemcpyDeviceToHost);
        printf("%-32s 3 sets on XCD mask 0x%02x: %6.2f us per link; workgroup lifetime by XCD (median):", sh.name, mask, best * 1e3 / nk);
        for (unsigned x = 0; x < 8; ++x) {
          float v[1024]; int m = 0;
          for (int b = 0; b < GRID; ++b) if (host[2 * b] == x) v[m++] = host[2 * b + 1] / 100.0f;
          for (int i = 1; i < m; ++i) for (int j = i; j > 0 && v[j] < v[j - 1]; --j) { float t = v[j]; v[j] = v[j - 1]; v[j - 1] = t; }
          printf(" %5.1f", m ? v[m / 2] : 0.f);
        }
        float mx = 0; for (int b = 0; b < GRID; ++b) if (host[2 * b + 1] / 100.0f > mx) mx = host[2 * b + 1] / 100.0f;
        printf("  max %5.1f\n", mx);
      }
      if (si == 0) si = 2;
    }
    return 0;
  }
  if (argc > 2) {   // soak: the packed w1/w3 stand-in for ~2 s without a pause -- does a link get slower once the chip has been streaming for a while?
    const auto sh = shapes[0];
    const size_t link4 = (size_t)sh.rows * sh.n / 4; const int nk = (int)(total / 16 / link4);
    const int reps = atoi(argv[2]);
    printf("soak, packed + fma, w1+w3: us per link at repetition");
    for (int rep = 0; rep < reps; ++rep) {
      (void)hipEventRecord(e0, sa);
      for (int k = 0; k < nk; ++k)
        hipLaunchKernelGGL((rows2<true, 0, false, 1, 0, 1>), dim3(GRID), dim3(256), 0, sa, w + (size_t)k * link4, sh.rows, sh.n, carry + 16 * k, carry + 16 * (k + 1));
      (void)hipEventRecord(e1, sa);
      if (rep < 4 || rep % (reps / 16 ? reps / 16 : 1) == 0 || rep == reps - 1) {
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("  %d: %.2f", rep, ms * 1e3 / nk);
      }
    }
    printf("\n");
    return 0;
  }
  for (auto sh : shapes) {
    const size_t link4 = (size_t)sh.rows * sh.n / 4; const int nk = (int)(total / 16 / link4);
    printf("%-34s", sh.name);
    for (int var = 0; var < 20; ++var) {
      if (var == 7 && sh.n != 4096) { printf(" %8s", "-"); continue; }            // eight column batches of 2 KB: 4096 columns
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0, sa);
        for (int k = 0; k < nk; ++k) {
          const f4* wk = w + (size_t)k * link4; const float* ci = carry + 16 * k; float* co = carry + 16 * (k + 1);
          const dim3 g(GRID), b(256);
#define ROWSW(R_, S_) do { if (sh.pair) hipLaunchKernelGGL((rows2<true, R_, S_, 1>), g, b, 0, sa, wk, sh.rows, sh.n, ci, co); else hipLaunchKernelGGL((rows2<false, R_, S_, 1>), g, b, 0, sa, wk, sh.rows, sh.n, ci, co); } while (0)
#define ROWS(R_, S_) do { if (sh.pair) hipLaunchKernelGGL((rows2<true, R_, S_>), g, b, 0, sa, wk, sh.rows, sh.n, ci, co); else hipLaunchKernelGGL((rows2<false, R_, S_>), g, b, 0, sa, wk, sh.rows, sh.n, ci, co); } while (0)
          if (var == 0) hipLaunchKernelGGL(stream, dim3(512), b, 0, sa, wk, link4, ci, co);
          else if (var == 1) ROWS(0, false); else if (var == 2) ROWS(5, false); else if (var == 3) ROWS(5, true); else if (var == 4) ROWS(3, false); else if (var == 5) ROWS(7, false); else if (var == 6) ROWSW(5, false);
          else if (var == 10) { if (sh.pair) hipLaunchKernelGGL((rows2<true, 0, false, 0, 0, 1>), g, b, 0, sa, wk, sh.rows, sh.n, ci, co); else hipLaunchKernelGGL((rows2<false, 0, false, 0, 0, 1>), g, b, 0, sa, wk, sh.rows, sh.n, ci, co); }
          else if (var == 11) { if (sh.pair) hipLaunchKernelGGL((rows2<true, 0, false, 1, 0, 1>), g, b, 0, sa, wk, sh.rows, sh.n, ci, co); else hipLaunchKernelGGL((rows2<false, 0, false, 1, 0, 1>), g, b, 0, sa, wk, sh.rows, sh.n, ci, co); }
          else if (var == 12) { if (sh.pair) hipLaunchKernelGGL((rows2<true, 5, false, 1, 0, 0, 1>), g, b, 0, sa, wk, sh.rows, sh.n, ci, co); else hipLaunchKernelGGL((rows2<false, 5, false, 1, 0, 0, 1>), g, b, 0, sa, wk, sh.rows, sh.n, ci, co); }
          else if (var == 13) { if (sh.pair) hipLaunchKernelGGL((rows2<true, 0, false, 1, 0, 1, 1>), g, b, 0, sa, wk, sh.rows, sh.n, ci, co); else hipLaunchKernelGGL((rows2<false, 0, false, 1, 0, 1, 1>), g, b, 0, sa, wk, sh.rows, sh.n, ci, co); }
          else if (var == 14) { if (sh.pair) hipLaunchKernelGGL((rows2<true, 0, false, 1, 0, 1, 0, 4>), g, b, 0, sa, wk, sh.rows, sh.n, ci, co); else hipLaunchKernelGGL((rows2<false, 0, false, 1, 0, 1, 0, 4>), g, b, 0, sa, wk, sh.rows, sh.n, ci, co); }
          else if (var == 15) { if (sh.pair) hipLaunchKernelGGL((rows2<true, 0, false, 1, 0, 1, 0, 1>), g, b, 0, sa, wk, sh.rows, sh.n, ci, co); else hipLaunchKernelGGL((rows2<false, 0, false, 1, 0, 1, 0, 1>), g, b, 0, sa, wk, sh.rows, sh.n, ci, co); }
#define PKROT(M_) do { if (sh.pair) hipLaunchKernelGGL((rows2<true, 0, false, 1, 0, 1 + M_>), g, b, 0, sa, wk, sh.rows, sh.n, ci, co); else hipLaunchKernelGGL((rows2<false, 0, false, 1, 0, 1 + M_>), g, b, 0, sa, wk, sh.rows, sh.n, ci, co); } while (0)
          else if (var == 16) hipLaunchKernelGGL((packdeep<2>), g, b, 0, sa, wk, sh.rows, sh.n, ci, co);
          else if (var == 17) hipLaunchKernelGGL((packdeep<3>), g, b, 0, sa, wk, sh.rows, sh.n, ci, co);
          else if (var == 18) hipLaunchKernelGGL((packdeep<4>), g, b, 0, sa, wk, sh.rows, sh.n, ci, co);
          else if (var == 19) hipLaunchKernelGGL((packdeep<6>), g, b, 0, sa, wk, sh.rows, sh.n, ci, co);
          else if (var >= 8) {
            const f4* nx = (k + 1 < nk) ? wk + link4 : nullptr;
            if (var == 8) { if (sh.pair) hipLaunchKernelGGL((rows2<true, 5, false, 1, 4>), g, b, 0, sa, wk, sh.rows, sh.n, ci, co, nx); else hipLaunchKernelGGL((rows2<false, 5, false, 1, 4>), g, b, 0, sa, wk, sh.rows, sh.n, ci, co, nx); }
            else { if (sh.pair) hipLaunchKernelGGL((rows2<true, 5, false, 1, 8>), g, b, 0, sa, wk, sh.rows, sh.n, ci, co, nx); else hipLaunchKernelGGL((rows2<false, 5, false, 1, 8>), g, b, 0, sa, wk, sh.rows, sh.n, ci, co, nx); }
          }
          else { if (sh.pair) hipLaunchKernelGGL((cols8<true>), dim3(256), dim3(512), 0, sa, wk, sh.rows, sh.n, ci, co); else hipLaunchKernelGGL((cols8<false>), dim3(256), dim3(512), 0, sa, wk, sh.rows, sh.n, ci, co); }
        }
        (void)hipEventRecord(e1, sa); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      printf(" %8.2f", best * 1e3 / nk);
    }
    printf("\n");
  }
  return 0;
}

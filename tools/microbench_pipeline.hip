// What would a PERSISTENT layer pipeline cost per phase at stories110M's shapes?  (measurement first; nothing in the library uses this)
//
// Today a stories110M token is 61 dependent launches: ~3.2 us of constant per launch (1.65 us boundary + argument fetch + the x vector
// + the first weight latency + reduction + store) against ~1.1 us of streaming: 438 MB cost 249 us instead of the 67 us they stream in.
// At this width a workgroup's whole share of a phase's matrix fits in REGISTERS (12.6 MB of w1/w3 over 256 workgroups x 256 threads =
// 48 floats per lane), so a persistent kernel can have a phase's weights on chip BEFORE its input exists and the per-phase critical
// path shrinks to: the producer's outputs becoming visible (granules {fp32 value, tag}, written through, polled past the L1) + staging x
// in LDS + the FMAs + a wave reduction + publishing.  This builds exactly that chain with the real byte counts and dependencies:
//   * NSET sets of NWG workgroups, all co-resident; set s runs phases s, s + NSET, ... of the token's L x NP phase sequence;
//   * phase types (n_in -> n_out): QKV 768 -> 2304, [ATT: 12 workgroups, a dependent cache-row load, 768 outputs], WO 768 -> 768,
//     W13 768 -> 4096 (2048 published), W2 2048 -> 768; every workgroup owns rows w + NWG k, so every workgroup gates the consumer;
//   * a workgroup requests ALL its weights of its next phase right after publishing the previous one, then polls its input granules.
// Output: us per phase and per layer, to hold against 19 us per layer of launches (profiles/r04/stories110M_kernel_stats.csv).
// Every wait is bounded by wall time (0.2 s) and gives up for good once any has (err printed).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mbp tools/microbench_pipeline.hip && /tmp/mbp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int D = 768, H = 2048, L = 12, NWG = 256, BLOCK = 256;
enum { T_QKV = 0, T_WO = 1, T_W13 = 2, T_W2 = 3, T_ATT = 4 };

struct Args {
  const float* w[4];         // per type: L matrices back to back
  u64* out[5];               // per type: the phase's output as granules {value, tag}
  const float* cache;        // the attention stand-in's rows
  int tokens, np, nset, att_rows;
  int* err;
  int order[5];              // phase types in layer order
};

__device__ __forceinline__ u64 granule(float v, unsigned tag) { return ((u64)tag << 32) | __float_as_uint(v); }
__device__ __forceinline__ void publish(u64* g, float v, unsigned tag) { __hip_atomic_store(g, granule(v, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Stage n_in values of `src` (granules tagged `tag`) into xs; every thread polls its own granules, all requested together per pass.
template <int N_IN>
__device__ __forceinline__ bool stage(const u64* src, unsigned tag, float* xs, int* err) {
  constexpr int PER = N_IN / BLOCK;
  static_assert(N_IN % BLOCK == 0, "input length");
  unsigned spins = 0;
  u64 t0 = 0;
  bool ok = true;
  for (;;) {
    u64 g[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) g[k] = __hip_atomic_load(src + threadIdx.x + k * BLOCK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool all = true;
#pragma unroll
    for (int k = 0; k < PER; ++k) all = all && (unsigned)(g[k] >> 32) == tag;
    if (all) {
#pragma unroll
      for (int k = 0; k < PER; ++k) xs[threadIdx.x + k * BLOCK] = __uint_as_float((unsigned)g[k]);
      break;
    }
    if ((++spins & 127u) == 0) {
      const u64 now = __builtin_amdgcn_s_memrealtime();
      if (!t0) t0 = now;
      else if (now - t0 > 20000000ull || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { atomicExch(err, 1); ok = false; break; }
    }
  }
  return __syncthreads_and(ok ? 1 : 0) != 0;
}

// One GEMV phase of one workgroup: rows w + NWG k (k < ROWS), row k on wave k % 4; RPW rows per wave, F4 float4 per lane and row.
template <int N_IN, int ROWS, int PUB>
__device__ __forceinline__ bool gemv_phase(const float* W, const u64* in, unsigned in_tag, u64* out, unsigned out_tag, float* xs, int* err) {
  constexpr int RPW = (ROWS + 3) / 4, F4 = N_IN / 256;          // 64 lanes x 4 floats per step
  const int w = blockIdx.x % NWG, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  f4 wt[RPW][F4];
  // every weight of the phase requested before the input exists
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int k = wave + 4 * r;
    const f4* row = reinterpret_cast<const f4*>(W + (size_t)(w + NWG * (k < ROWS ? k : 0)) * N_IN);
#pragma unroll
    for (int c = 0; c < F4; ++c) wt[r][c] = (k < ROWS) ? __builtin_nontemporal_load(row + lane + 64 * c) : f4{0.f, 0.f, 0.f, 0.f};
  }
  if (!stage<N_IN>(in, in_tag, xs, err)) return false;
  double acc[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r) acc[r] = 0.0;
#pragma unroll
  for (int c = 0; c < F4; ++c) {
    const f4 x = *reinterpret_cast<const f4*>(xs + 4 * (lane + 64 * c));
#pragma unroll
    for (int r = 0; r < RPW; ++r)
      acc[r] += (double)wt[r][c].x * (double)x.x + (double)wt[r][c].y * (double)x.y + (double)wt[r][c].z * (double)x.z + (double)wt[r][c].w * (double)x.w;
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    double v = acc[r];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int k = wave + 4 * r;
    if (lane == 0 && k < ROWS && k < PUB) publish(out + w + NWG * k, (float)v * 1e-3f + 0.5f, out_tag);   // k < PUB: the rows the next phase reads
  }
  __syncthreads();                                                // xs is reused by the workgroup's next phase
  return true;
}

// The attention stand-in: workgroup h < 12 waits for head h's 64 q values, makes one dependent pass over att_rows cache rows, publishes 64 outputs.
__device__ __forceinline__ bool att_phase(const Args& a, const u64* in, unsigned in_tag, u64* out, unsigned out_tag, float* xs) {
  const int h = blockIdx.x % NWG;
  if (h >= 12) return true;
  unsigned spins = 0;
  u64 t0 = 0;
  float q = 0.0f;
  if (threadIdx.x < 64) {
    for (;;) {
      const u64 g = __hip_atomic_load(in + h * 64 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__all((unsigned)(g >> 32) == in_tag)) { q = __uint_as_float((unsigned)g); break; }
      if ((++spins & 127u) == 0) {
        const u64 now = __builtin_amdgcn_s_memrealtime();
        if (!t0) t0 = now;
        else if (now - t0 > 20000000ull || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { atomicExch(a.err, 1); break; }
      }
    }
    xs[threadIdx.x] = q;
  }
  __syncthreads();
  // scores: row t of the head's cache slab (64 floats) . q, rows over the threads; then a value pass that depends on the scores
  float s = 0.0f;
  for (int t = threadIdx.x; t < a.att_rows; t += BLOCK) {
    const f4* row = reinterpret_cast<const f4*>(a.cache + ((size_t)t * 12 + h) * 64);
    float d = 0.0f;
#pragma unroll
    for (int c = 0; c < 16; ++c) { const f4 kv = row[c]; d += kv.x * xs[4 * c] + kv.y * xs[4 * c + 1] + kv.z * xs[4 * c + 2] + kv.w * xs[4 * c + 3]; }
    s += d;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) xs[64 + (threadIdx.x >> 6)] = s;
  __syncthreads();
  const float tot = xs[64] + xs[65] + xs[66] + xs[67];
  float v = 0.0f;
  if (threadIdx.x < 64) {
    for (int t = 0; t < a.att_rows; t += 16) v += a.cache[((size_t)(a.att_rows + t) * 12 + h) * 64 + threadIdx.x] * tot;   // dependent on the scores
    publish(out + h * 64 + threadIdx.x, v * 1e-6f + 0.25f, out_tag);
  }
  __syncthreads();
  return true;
}

__global__ void __launch_bounds__(BLOCK, 4) pipeline_kernel(const Args a) {
  __shared__ __attribute__((aligned(16))) float xs[H];
  const int set = blockIdx.x / NWG, P = L * a.np;
  for (int it = 0; it < a.tokens; ++it) {
    for (int p = set; p < P; p += a.nset) {
      const int layer = p / a.np, type = a.order[p % a.np];
      const int prev_type = a.order[(p + a.np - 1) % a.np];
      const unsigned in_tag = (unsigned)(it * P + p), out_tag = in_tag + 1;     // tag 0 = the seed written by the host side kernel
      const u64* in = a.out[prev_type];
      bool ok = true;
      switch (type) {
        case T_QKV: ok = gemv_phase<D, 9, 3>(a.w[0] + (size_t)layer * 2304 * D, in, in_tag, a.out[T_QKV], out_tag, xs, a.err); break;
        case T_WO: ok = gemv_phase<D, 3, 3>(a.w[1] + (size_t)layer * D * D, in, in_tag, a.out[T_WO], out_tag, xs, a.err); break;
        case T_W13: ok = gemv_phase<D, 16, 8>(a.w[2] + (size_t)layer * 4096 * D, in, in_tag, a.out[T_W13], out_tag, xs, a.err); break;
        case T_W2: ok = gemv_phase<H, 3, 3>(a.w[3] + (size_t)layer * D * H, in, in_tag, a.out[T_W2], out_tag, xs, a.err); break;
        default: ok = att_phase(a, in, in_tag, a.out[T_ATT], out_tag, xs); break;
      }
      if (!ok) return;
    }
  }
}

__global__ void seed_kernel(u64* g, int n, unsigned tag) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) g[i] = granule(0.5f, tag); }
__global__ void fill_kernel(float* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { unsigned x = (unsigned)i * 2654435761u; x ^= x >> 13; p[i] = (float)(x & 0xffff) * (1.0f / 65536.0f) - 0.5f; }
}

int main(int argc, char** argv) {
  const int tokens = argc > 1 ? atoi(argv[1]) : 64;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const size_t wn[4] = {(size_t)L * 2304 * D, (size_t)L * D * D, (size_t)L * 4096 * D, (size_t)L * D * H};
  Args a = {};
  for (int t = 0; t < 4; ++t) { float* p; CK(hipMalloc(&p, wn[t] * 4)); hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, st, p, wn[t]); a.w[t] = p; }
  for (int t = 0; t < 5; ++t) { CK(hipMalloc(&a.out[t], 4096 * 8)); CK(hipMemset(a.out[t], 0xff, 4096 * 8)); }
  { float* c; const size_t cn = (size_t)2 * 1024 * 12 * 64; CK(hipMalloc(&c, cn * 4)); hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, st, c, cn); a.cache = c; }
  CK(hipMalloc(&a.err, 4));
  int occ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, pipeline_kernel, BLOCK, 0));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("pipeline_kernel: %d workgroups of %d threads per CU, %d CUs\n", occ, BLOCK, prop.multiProcessorCount);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct Cfg { const char* name; int np; int order[5]; int nset; int att_rows; };
  const Cfg cfgs[] = {
      {"4 GEMV phases per layer, 2 sets", 4, {T_QKV, T_WO, T_W13, T_W2, 0}, 2, 0},
      {"4 GEMV phases per layer, 3 sets", 4, {T_QKV, T_WO, T_W13, T_W2, 0}, 3, 0},
      {"4 GEMV phases per layer, 4 sets", 4, {T_QKV, T_WO, T_W13, T_W2, 0}, 4, 0},
      {"4 GEMV phases per layer, 1 set (no phase's weights requested early)", 4, {T_QKV, T_WO, T_W13, T_W2, 0}, 1, 0},
      {"with the attention stand-in (128 rows), 2 sets", 5, {T_QKV, T_ATT, T_WO, T_W13, T_W2}, 2, 128},
      {"with the attention stand-in (128 rows), 3 sets", 5, {T_QKV, T_ATT, T_WO, T_W13, T_W2}, 3, 128},
      {"with the attention stand-in (512 rows), 3 sets", 5, {T_QKV, T_ATT, T_WO, T_W13, T_W2}, 3, 512},
  };
  for (const Cfg& c : cfgs) {
    if (c.nset * NWG > occ * prop.multiProcessorCount) { printf("%s: %d workgroups would not be resident together, skipped\n", c.name, c.nset * NWG); continue; }
    a.np = c.np; a.nset = c.nset; a.att_rows = c.att_rows; a.tokens = tokens;
    for (int k = 0; k < 5; ++k) a.order[k] = c.order[k];
    float best = 1e30f;
    int err = 0;
    for (int rep = 0; rep < 3 && !err; ++rep) {
      CK(hipMemsetAsync(a.err, 0, 4, st));
      for (int t = 0; t < 5; ++t) CK(hipMemsetAsync(a.out[t], 0xff, 4096 * 8, st));
      // the first phase of the first token reads the last type of the order, tagged 0
      hipLaunchKernelGGL(seed_kernel, dim3(16), dim3(256), 0, st, a.out[c.order[c.np - 1]], 4096, 0u);
      CK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(pipeline_kernel, dim3(c.nset * NWG), dim3(BLOCK), 0, st, a);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipMemcpy(&err, a.err, 4, hipMemcpyDeviceToHost));
      if (ms < best) best = ms;
    }
    const double per_phase = best * 1e3 / ((double)tokens * L * c.np);
    printf("%-72s: %7.2f us per phase, %7.2f us per layer, %8.1f us per %d-layer token%s\n", c.name, per_phase, per_phase * c.np, per_phase * c.np * L, L, err ? "  (A WAIT GAVE UP: invalid)" : "");
  }
  return 0;
}

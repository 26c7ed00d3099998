// sampler_sort.hip.h -- top-p order without a library sort: bitonic tile sort in registers + rank merge out of LDS
// Part of sampler.hip (included there inside namespace l2s, in order); not a stand-alone header.
#pragma once

// ---- top-p: descending stable order = ascending order of the key (~probability bits, id)
typedef unsigned long long u64;

__device__ __forceinline__ void order_pair(u64& a, u64& b, bool up) {
  const bool sw = (a > b) == up;
  const u64 lo = sw ? b : a, hi = sw ? a : b;
  a = lo; b = hi;
}

// Bitonic sort of one sort tile: SIT consecutive positions per thread, so the small strides stay inside a thread, the
// middle ones are lane exchanges inside a wave, and only the strides >= 64 * SIT go through LDS.  Measured: 8 keys per
// thread (2048-key tiles, half as many for the rank merge to search) take 12 us longer here and save 4 us there.
constexpr int SIT = 4, STILE = TN * SIT;
template <bool FUSED>
__global__ void __launch_bounds__(TN) sort_tile_kernel(ChainArgs a, const float* probs, int V, float* run_p, int* run_id) {
  __shared__ union { ChainShared sh; u64 xch[STILE]; } lds;      // the chain is over before the first exchange (a barrier in between)
  ChainShared& sh = lds.sh;
  u64* xch = lds.xch;
  const int tid = threadIdx.x, base = blockIdx.x * STILE, p0 = tid * SIT;
  double s = 1.0;
  if (FUSED) {                                                 // probs holds the exps: every workgroup derives their exact total itself
    bool in_lds;
    chain_total(a, sh, &in_lds);
    s = sh.val;
  }
  u64 v[SIT];
#pragma unroll
  for (int k = 0; k < SIT; ++k) {
    const int i = base + p0 + k;
    const float e = (i < V) ? probs[i] : 0.0f;
    const float p = FUSED ? (float)((double)e / s) : e;                      // :192
    v[k] = (i < V) ? (((u64)(0xffffffffu - __float_as_uint(p)) << 32) | (unsigned)i) : ~0ull;
  }
#pragma unroll
  for (int k2 = 2; k2 <= STILE; k2 <<= 1) {
#pragma unroll
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      if (j < SIT) {
#pragma unroll
        for (int k = 0; k < SIT; ++k) if ((k & j) == 0) order_pair(v[k], v[k | j], ((p0 + k) & k2) == 0);
      } else {
        const bool keep_min = ((p0 & j) == 0) == ((p0 & k2) == 0);
        u64 o[SIT];
        if (j < 64 * SIT) {
#pragma unroll
          for (int k = 0; k < SIT; ++k) o[k] = __shfl_xor(v[k], j / SIT, 64);
        } else {
          __syncthreads();
#pragma unroll
          for (int k = 0; k < SIT; ++k) xch[p0 + k] = v[k];
          __syncthreads();
#pragma unroll
          for (int k = 0; k < SIT; ++k) o[k] = xch[(p0 ^ j) + k];
        }
#pragma unroll
        for (int k = 0; k < SIT; ++k) v[k] = keep_min ? (v[k] < o[k] ? v[k] : o[k]) : (v[k] > o[k] ? v[k] : o[k]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < SIT; ++k) {
    const bool pad = v[k] == ~0ull;
    run_p[base + p0 + k] = pad ? -1.0f : __uint_as_float(0xffffffffu - (unsigned)(v[k] >> 32));
    run_id[base + p0 + k] = pad ? -1 : (int)(unsigned)v[k];
  }
}

// The same tile sort by 1024 threads with one key each and the exact total handed in (runs_total_kernel): four waves per SIMD hide the
// lane exchanges one wave per SIMD waits for, and no thread carries four 64-bit keys through 55 stages.
constexpr int WT = 1024;
static_assert(WT == STILE, "one key per thread");
__global__ void __launch_bounds__(WT) sort_tile_wide_kernel(const float* exps, const double* total, int V, float* run_p, int* run_id) {
  __shared__ u64 xch[STILE];
  const int tid = threadIdx.x, i = blockIdx.x * STILE + tid;
  const double s = *total;
  const float p = (i < V) ? (float)((double)exps[i] / s) : 0.0f;                  // :192
  u64 v = (i < V) ? (((u64)(0xffffffffu - __float_as_uint(p)) << 32) | (unsigned)i) : ~0ull;
#pragma unroll
  for (int k2 = 2; k2 <= STILE; k2 <<= 1) {
#pragma unroll
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      const bool keep_min = ((tid & j) == 0) == ((tid & k2) == 0);
      u64 o;
      if (j < 64) {
        o = __shfl_xor(v, j, 64);
      } else {
        __syncthreads();
        xch[tid] = v;
        __syncthreads();
        o = xch[tid ^ j];
      }
      v = keep_min ? (v < o ? v : o) : (v > o ? v : o);
    }
  }
  const bool pad = v == ~0ull;
  run_p[i] = pad ? -1.0f : __uint_as_float(0xffffffffu - (unsigned)(v >> 32));
  run_id[i] = pad ? -1 : (int)(unsigned)v;
}

// Every element's place in the merged order: its place in its own tile + the number of elements of every other tile in front of it
// (ties: the tile with the smaller ids first), by binary search in sorted tiles held in LDS.  The searches are what the kernel costs
// (measured at 32 tiles on 64 workgroups: 4 us filling LDS, 8 us searching), so they are spread over the chip: workgroup (x, y) ranks
// RT consecutive elements against the RANK_TQ tiles of group y only and adds its count to acc[element]; the top byte of acc counts the
// groups that have reported, and the thread whose add completes an element knows its rank, stores it and puts acc back to zero.
constexpr int RT = 512;                                           // threads = elements per workgroup of the rank merge
constexpr int RANK_TQ = 8;                                        // sorted tiles a workgroup searches (32 KB of LDS)
// Also adds every element to the sum of the tile of the merged order it lands in (part[], zero on entry): the approximate
// prefix of the next stage.  fp64 atomics in no fixed order -- the prefix only has to be approximate.
__global__ void __launch_bounds__(RT) sort_rank_kernel(const float* run_p, const int* run_id, int GS, int G, unsigned* acc, float* sorted, int* ids, double* part) {
  __shared__ int lds_p[RANK_TQ * STILE];                        // probability bit patterns of the group's tiles (pads: negative)
  __shared__ double lpart[MAX_VOCAB / TILE];
  const int tid = threadIdx.x, n = GS * STILE;
  for (int j = tid; j < G; j += RT) lpart[j] = 0.0;
  const int e = blockIdx.x * RT + tid;
  const int mine = e < n ? reinterpret_cast<const int*>(run_p)[e] : -1;
  const int own = e / STILE;
  const int g0 = blockIdx.y * RANK_TQ, gt = min(RANK_TQ, GS - g0), gn = gt * STILE, groups = gridDim.y;
  {
    // one 16-byte load in flight per thread (deeper queues measured slower), every workgroup starting at another tile
    const int* src = reinterpret_cast<const int*>(run_p) + (size_t)g0 * STILE;
    const int rot = (int)(blockIdx.x % (unsigned)gt) * STILE;
    for (int j = tid * 4; j < gn; j += RT * 4) { int jj = j + rot; if (jj >= gn) jj -= gn; *reinterpret_cast<int4*>(lds_p + jj) = *reinterpret_cast<const int4*>(src + jj); }
  }
  __syncthreads();
  if (mine >= 0) {                                              // not a pad
    int lo[RANK_TQ], thr[RANK_TQ];
#pragma unroll
    for (int u = 0; u < RANK_TQ; ++u) {
      const int b = min(u, gt - 1);
      lo[u] = b * STILE;
      thr[u] = mine - (g0 + b < own ? 1 : 0);                   // earlier tile: elements >= mine come first; later tile: only > mine
    }
#pragma unroll
    for (int s = STILE / 2; s > 0; s >>= 1) {
#pragma unroll
      for (int u = 0; u < RANK_TQ; ++u) if (lds_p[lo[u] + s - 1] > thr[u]) lo[u] += s;
    }
    unsigned count = (own >= g0 && own < g0 + gt) ? (unsigned)(e - own * STILE) : 0u;   // its place in its own tile, counted once
#pragma unroll
    for (int u = 0; u < RANK_TQ; ++u) {
      const int b = min(u, gt - 1);
      int cnt = lo[u] - b * STILE;
      if (cnt == STILE - 1 && lds_p[lo[u]] > thr[u]) cnt = STILE;
      if (u < gt && g0 + b != own) count += (unsigned)cnt;
    }
    const unsigned before = groups > 1 ? atomicAdd(acc + e, count + (1u << 24)) : 0u;
    if ((int)(before >> 24) == groups - 1) {                     // every other group has reported: the rank is complete
      const int rank = (int)((before & 0xffffffu) + count);
      if (groups > 1) __hip_atomic_store(acc + e, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // past the L2, like the adds
      sorted[rank] = __int_as_float(mine);
      ids[rank] = run_id[e];
      atomicAdd(lpart + rank / TILE, (double)__int_as_float(mine));
    }
  }
  __syncthreads();
  for (int j = tid; j < G; j += RT) if (lpart[j] != 0.0) atomicAdd(part + j, lpart[j]);
}

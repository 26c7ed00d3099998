// Kernels of tools/aql/microbench_aql.cpp (built to a code object: hipcc --genco --offload-arch=gfx950).
#include <hip/hip_runtime.h>
extern "C" __global__ void k_empty() {}
// every thread reads an element another WORKGROUP wrote in the launch before (64 elements on) and stores it + 1
extern "C" __global__ void k_rot_plain(const float* in, float* out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;      // (256: the block width -- blockDim would come from the hidden arguments this test does not fill)
  int j = i + 64; if (j >= n) j -= n;
  out[i] = in[j] + 1.0f;
}
// the same with agent-scope (sc1) accesses: store written through, load past L1
extern "C" __global__ void k_rot_sc1(const float* in, float* out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;      // (256: the block width -- blockDim would come from the hidden arguments this test does not fill)
  int j = i + 64; if (j >= n) j -= n;
  const float v = __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(in + j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  __hip_atomic_store(reinterpret_cast<unsigned*>(out + i), __float_as_uint(v + 1.0f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ... and without the wave's own wait for its write-through store before it ends
extern "C" __global__ void k_rot_sc1_nowait(const float* in, float* out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  int j = i + 64; if (j >= n) j -= n;
  const float v = __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(in + j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  __hip_atomic_store(reinterpret_cast<unsigned*>(out + i), __float_as_uint(v + 1.0f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// a reader whose L1 and L2 hold the line through a PLAIN load of the launch before (the value is thrown away), then the sc1 load
extern "C" __global__ void k_rot_mixed(const float* in, float* out, int n, float* sink) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  int j = i + 64; if (j >= n) j -= n;
  const float stale = *reinterpret_cast<const volatile float*>(out + j);      // plain: brings the line this launch's peers are about to overwrite into L1 / L2
  if (stale == -1.0f) sink[i] = stale;
  const float v = __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(in + j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  __hip_atomic_store(reinterpret_cast<unsigned*>(out + i), __float_as_uint(v + 1.0f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// which half of the rule carries it: plain store + sc1 load, sc1 store + plain load
extern "C" __global__ void k_rot_pst_sld(const float* in, float* out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  int j = i + 64; if (j >= n) j -= n;
  const float v = __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(in + j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  out[i] = v + 1.0f;
}
extern "C" __global__ void k_rot_sst_pld(const float* in, float* out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  int j = i + 64; if (j >= n) j -= n;
  const float v = in[j];
  __hip_atomic_store(reinterpret_cast<unsigned*>(out + i), __float_as_uint(v + 1.0f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- does a CU's vector L1 carry a line from one dispatch into a later one?  (round 6: tests/test_coherence_gpu.py's adversary found no
// stale read even in a build whose activation loads are plain.)  k_pull_all: EVERY workgroup reads the WHOLE small buffer with plain
// loads and counts the elements that are not `expect`; k_set_all: one launch rewrites the buffer (plain or sc1 stores).  The chain
// alternates set(v) / pull(v) / pull(v): a pull that finds the value of the set BEFORE the last one read a line its CU kept from the
// pull two dispatches earlier.
extern "C" __global__ void k_pull_all(const float* buf, float* unused, int n, int expect, float* stale_count) {
  int bad = 0;
  for (int i = threadIdx.x; i < n; i += 256) {
    float v;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(buf + i) : "memory");
    bad += (v != (float)expect);
  }
  if (bad) atomicAdd(stale_count, (float)bad);
}
extern "C" __global__ void k_set_all(const float* unused, float* buf, int n, int value, float* sink) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) buf[i] = (float)value;
}
extern "C" __global__ void k_set_all_sc1(const float* unused, float* buf, int n, int value, float* sink) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) __hip_atomic_store(reinterpret_cast<unsigned*>(buf + i), __float_as_uint((float)value), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

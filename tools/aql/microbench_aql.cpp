// What does a dependent kernel node cost when the AQL packets are written by hand?  Chains of N dependent dispatches on a user-mode
// HSA queue, barrier bit set, with the acquire / release fence scopes of the packet header NONE / AGENT / SYSTEM -- against the
// 1.65 us per node of a replayed hipGraph (tools/microbench_launch.hip).  Every chain is checked: element i must equal the
// number of launches (each launch reads what ANOTHER workgroup wrote in the launch before).
//   build: hipcc --offload-arch=gfx950 --cuda-device-only --no-gpu-bundle-output -O3 tools/aql/microbench_aql_kernels.hip -o tools/aql/microbench_aql_kernels.hsaco
//          g++ -O2 -std=c++17 -I/opt/rocm/include tools/aql/microbench_aql.cpp -o tools/aql/microbench_aql -L/opt/rocm/lib -lhsa-runtime64
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#define CK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char* m = nullptr; hsa_status_string(s_, &m); printf("%s failed: %s\n", #x, m ? m : "?"); exit(1); } } while (0)

static hsa_agent_t g_gpu, g_cpu; static bool have_gpu = false, have_cpu = false;
static hsa_amd_memory_pool_t g_dev_pool, g_kernarg_pool; static bool have_dev = false, have_ka = false;
static hsa_status_t on_agent(hsa_agent_t a, void*) {
  hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
  if (t == HSA_DEVICE_TYPE_GPU && !have_gpu) { g_gpu = a; have_gpu = true; }
  if (t == HSA_DEVICE_TYPE_CPU && !have_cpu) { g_cpu = a; have_cpu = true; }
  return HSA_STATUS_SUCCESS;
}
static hsa_status_t on_gpu_pool(hsa_amd_memory_pool_t p, void*) {
  hsa_amd_segment_t seg; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
  uint32_t fl; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &fl);
  bool alloc; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
  size_t psz = 0; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SIZE, &psz);
  if (seg == HSA_AMD_SEGMENT_GLOBAL && alloc && (fl & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !have_dev) { g_dev_pool = p; have_dev = true; }
  return HSA_STATUS_SUCCESS;
}
static hsa_status_t on_cpu_pool(hsa_amd_memory_pool_t p, void*) {
  hsa_amd_segment_t seg; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
  uint32_t fl; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &fl);
  if (seg == HSA_AMD_SEGMENT_GLOBAL && (fl & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !have_ka) { g_kernarg_pool = p; have_ka = true; }
  return HSA_STATUS_SUCCESS;
}
struct Kern { uint64_t object; uint32_t kernarg, group, priv; };
static Kern get_kernel(hsa_executable_t ex, const char* name) {
  hsa_executable_symbol_t sym; std::string n = std::string(name) + ".kd";
  CK(hsa_executable_get_symbol_by_name(ex, n.c_str(), &g_gpu, &sym));
  Kern k;
  CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object));
  CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg));
  CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group));
  CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.priv));
  return k;
}
struct RotArgs { const float* in; float* out; int n; int pad; float* sink; };

int main(int argc, char** argv) {
  const char* path = argc > 1 ? argv[1] : "tools/aql/microbench_aql_kernels.hsaco";
  setvbuf(stdout, nullptr, _IONBF, 0);
  CK(hsa_init());
  CK(hsa_iterate_agents(on_agent, nullptr));
  if (!have_gpu || !have_cpu) { printf("no GPU / CPU agent\n"); return 1; }
  CK(hsa_amd_agent_iterate_memory_pools(g_gpu, on_gpu_pool, nullptr));
  CK(hsa_amd_agent_iterate_memory_pools(g_cpu, on_cpu_pool, nullptr));
  if (!have_dev || !have_ka) { printf("no device / kernarg pool\n"); return 1; }
  FILE* f = fopen(path, "rb"); if (!f) { printf("cannot open %s\n", path); return 1; }
  fseek(f, 0, SEEK_END); const long sz = ftell(f); fseek(f, 0, SEEK_SET);
  std::vector<char> blob(sz); if (fread(blob.data(), 1, sz, f) != (size_t)sz) return 1; fclose(f);
  hsa_code_object_reader_t rd; CK(hsa_code_object_reader_create_from_memory(blob.data(), sz, &rd));
  hsa_executable_t ex; CK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &ex));
  CK(hsa_executable_load_agent_code_object(ex, g_gpu, rd, nullptr, nullptr));
  CK(hsa_executable_freeze(ex, nullptr));
  const Kern kE = get_kernel(ex, "k_empty"), kP = get_kernel(ex, "k_rot_plain"), kS = get_kernel(ex, "k_rot_sc1"), kN = get_kernel(ex, "k_rot_sc1_nowait"), kM = get_kernel(ex, "k_rot_mixed"), kPS = get_kernel(ex, "k_rot_pst_sld"), kSP = get_kernel(ex, "k_rot_sst_pld");
  const int QSZ = 4096;
  hsa_queue_t* q; CK(hsa_queue_create(g_gpu, QSZ, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
  hsa_signal_t done; CK(hsa_signal_create(1, 0, nullptr, &done));
  const int NMAX = 2048 * 256;
  float *a, *b; CK(hsa_amd_memory_pool_allocate(g_dev_pool, NMAX * 4, 0, (void**)&a)); CK(hsa_amd_memory_pool_allocate(g_dev_pool, NMAX * 4, 0, (void**)&b));
  CK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, a)); CK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, b));
  const int N = 1000;
  RotArgs* ka; CK(hsa_amd_memory_pool_allocate(g_kernarg_pool, (size_t)N * sizeof(RotArgs), 0, (void**)&ka));
  CK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, ka));
  float* sink; CK(hsa_amd_memory_pool_allocate(g_dev_pool, NMAX * 4, 0, (void**)&sink));
  float* host; CK(hsa_amd_memory_pool_allocate(g_kernarg_pool, NMAX * 4, 0, (void**)&host));
  CK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, host));
  // kernel arguments in DEVICE memory (as HIP places them on this chip): from host memory every CU's scalar cache fetches them over
  // PCIe -- 26 us per node with 256 workgroups.  argv[2] = "host" keeps them in host memory to show that.
  RotArgs* ka_dev; CK(hsa_amd_memory_pool_allocate(g_dev_pool, (size_t)N * sizeof(RotArgs), 0, (void**)&ka_dev));
  const bool ka_on_host = argc > 2 && !strcmp(argv[2], "host");
  printf("kernel arguments in %s memory\n", ka_on_host ? "host" : "device");

  auto run_chain = [&](const Kern& k, int grid_wgs, int block, int acq, int rel, bool check, const char* label) {
    const int n = grid_wgs * block;
    memset(host, 0, (size_t)n * 4);
    CK(hsa_memory_copy(a, host, (size_t)n * 4)); CK(hsa_memory_copy(b, host, (size_t)n * 4));
    for (int i = 0; i < N; ++i) { ka[i].in = (i & 1) ? b : a; ka[i].out = (i & 1) ? a : b; ka[i].n = n; ka[i].pad = 0; ka[i].sink = sink; }
    if (!ka_on_host) CK(hsa_memory_copy(ka_dev, ka, (size_t)N * sizeof(RotArgs)));
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
      hsa_signal_store_relaxed(done, 1);
      const uint64_t base = hsa_queue_add_write_index_relaxed(q, N);
      while (base + N - hsa_queue_load_read_index_scacquire(q) > (uint64_t)QSZ) {}
      for (int i = 0; i < N; ++i) {
        hsa_kernel_dispatch_packet_t* p = (hsa_kernel_dispatch_packet_t*)q->base_address + ((base + i) & (QSZ - 1));
        const bool first = i == 0, last = i == N - 1;
        p->workgroup_size_x = block; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
        p->grid_size_x = n; p->grid_size_y = 1; p->grid_size_z = 1;
        p->private_segment_size = k.priv; p->group_segment_size = k.group;
        p->kernel_object = k.object; p->kernarg_address = ka_on_host ? &ka[i] : &ka_dev[i]; p->reserved2 = 0;
        p->completion_signal.handle = last ? done.handle : 0;
        const int A = first ? HSA_FENCE_SCOPE_SYSTEM : acq, R = last ? HSA_FENCE_SCOPE_SYSTEM : rel;
        const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                (A << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (R << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
        const uint16_t setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
        __atomic_store_n((uint32_t*)p, (uint32_t)header | ((uint32_t)setup << 16), __ATOMIC_RELEASE);
      }
      const auto t0 = std::chrono::steady_clock::now();
      hsa_signal_store_screlease(q->doorbell_signal, base + N - 1);
      if (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, 5000000000ull, HSA_WAIT_STATE_ACTIVE) != 0) { printf("%s: TIMEOUT\n", label); exit(2); }
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      if (us < best) best = us;
    }
    int bad = 0;
    if (check) {
      CK(hsa_memory_copy(host, (N & 1) ? b : a, (size_t)n * 4));      // the last launch (index N - 1) wrote: odd index -> a
      // 5 repetitions of N launches each: every element has been incremented 5 N times
      for (int i = 0; i < n; ++i) if (host[i] != (float)(5 * N)) ++bad;
    }
    printf("%-62s grid %4d x %3d : %.3f us per node%s\n", label, grid_wgs, block, best / N, check ? (bad ? "   WRONG VALUES" : "   values ok") : "");
    if (check && bad) printf("      %d of %d elements differ (first: %g, expected %d)\n", bad, n, host[0], 5 * N);
  };
  const char* sn[3] = {"none", "agent", "system"};
  for (int grid : {1, 256, 2048}) {
    for (int sc = 0; sc < 3; ++sc) {
      char lab[96];
      snprintf(lab, sizeof lab, "empty kernel, acquire/release %s", sn[sc]); run_chain(kE, grid, 256, sc, sc, false, lab);
    }
    for (int sc = 0; sc < 3; ++sc) {
      char lab[96];
      snprintf(lab, sizeof lab, "load+store (sc1 accesses), fences %s", sn[sc]); run_chain(kS, grid, 256, sc, sc, true, lab);
      snprintf(lab, sizeof lab, "load+store (plain accesses), fences %s", sn[sc]); run_chain(kP, grid, 256, sc, sc, true, lab);
    }
    { char lab[96];
      snprintf(lab, sizeof lab, "load+store (sc1, no wait before the wave ends), fences none"); run_chain(kN, grid, 256, 0, 0, true, lab);
      snprintf(lab, sizeof lab, "plain load of the old line, then sc1 load+store, none"); run_chain(kM, grid, 256, 0, 0, true, lab);
      snprintf(lab, sizeof lab, "PLAIN store + sc1 load, fences none"); run_chain(kPS, grid, 256, 0, 0, true, lab);
      snprintf(lab, sizeof lab, "sc1 store + PLAIN load, fences none"); run_chain(kSP, grid, 256, 0, 0, true, lab); }
    { char lab[96]; snprintf(lab, sizeof lab, "load+store (sc1), acquire none / release agent"); run_chain(kS, grid, 256, 0, 1, true, lab);
      snprintf(lab, sizeof lab, "load+store (sc1), acquire agent / release none"); run_chain(kS, grid, 256, 1, 0, true, lab); }
  }
  // ---- does a CU's vector L1 carry a line from one dispatch into a later one?  chain: set(v), pull(v), pull(v), set(v + 1), pull(v + 1), ...
  {
    const Kern kPull = get_kernel(ex, "k_pull_all"), kSet = get_kernel(ex, "k_set_all"), kSetS = get_kernel(ex, "k_set_all_sc1");
    const int n = 4096;      // 16 KB: every CU's L1 (32 KB) can hold the whole buffer
    const int M = 999;       // 333 x {set, pull, pull}
    for (int variant = 0; variant < 6; ++variant) {
      const bool sc1_store = variant & 1;
      const int acq = 0, rel = (variant >> 1) == 0 ? 0 : ((variant >> 1) == 1 ? 1 : 2);      // release none / agent / system; NO acquire
      if (!sc1_store && rel == 0) continue;      // plain stores without a release never reach the other XCDs' L2s: not what is asked here
      memset(host, 0, (size_t)n * 4);
      CK(hsa_memory_copy(a, host, (size_t)n * 4)); CK(hsa_memory_copy(sink, host, 64));
      for (int i = 0; i < M; ++i) { ka[i].in = a; ka[i].out = a; ka[i].n = n; ka[i].pad = i / 3 + 1; ka[i].sink = sink; }
      CK(hsa_memory_copy(ka_dev, ka, (size_t)M * sizeof(RotArgs)));
      hsa_signal_store_relaxed(done, 1);
      const uint64_t base = hsa_queue_add_write_index_relaxed(q, M);
      while (base + M - hsa_queue_load_read_index_scacquire(q) > (uint64_t)QSZ) {}
      for (int i = 0; i < M; ++i) {
        hsa_kernel_dispatch_packet_t* p = (hsa_kernel_dispatch_packet_t*)q->base_address + ((base + i) & (QSZ - 1));
        const bool first = i == 0, last = i == M - 1, is_set = (i % 3) == 0;
        const Kern& k = is_set ? (sc1_store ? kSetS : kSet) : kPull;
        const int wgs = is_set ? n / 256 : 1024;      // pull: four workgroups per CU, every one reads everything
        p->workgroup_size_x = 256; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
        p->grid_size_x = wgs * 256; p->grid_size_y = 1; p->grid_size_z = 1;
        p->private_segment_size = k.priv; p->group_segment_size = k.group;
        p->kernel_object = k.object; p->kernarg_address = &ka_dev[i]; p->reserved2 = 0;
        p->completion_signal.handle = last ? done.handle : 0;
        const int A = first ? HSA_FENCE_SCOPE_SYSTEM : acq, R = last ? HSA_FENCE_SCOPE_SYSTEM : rel;
        const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                (A << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (R << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
        __atomic_store_n((uint32_t*)p, (uint32_t)header | ((uint32_t)(1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS) << 16), __ATOMIC_RELEASE);
      }
      hsa_signal_store_screlease(q->doorbell_signal, base + M - 1);
      if (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, 5000000000ull, HSA_WAIT_STATE_ACTIVE) != 0) { printf("L1 carry-over: TIMEOUT\n"); exit(2); }
      CK(hsa_memory_copy(host, sink, 64));
      printf("L1 across dispatches: %s stores, NO acquire, release %-6s: 666 pulls x 1024 workgroups x %d plain loads: %.0f stale\n",
             sc1_store ? "sc1  " : "plain", sn[rel], n, host[0]);
    }
  }
  hsa_queue_destroy(q);
  hsa_shut_down();
  return 0;
}

// aql_queue.cpp -- see aql_queue.h.  Host-only C++ over the HSA runtime (libhsa-runtime64, the layer HIP itself sits on).
#include "aql_queue.h"

#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <map>
#include <string>
#include <vector>

namespace {

struct Kernel { uint64_t object; uint32_t kernarg_size, group_static, priv; };

struct Bundle { size_t off, size; };

}  // namespace

struct AqlProgram {
  std::vector<hsa_kernel_dispatch_packet_t> pk;      // header = 0 here; written last at submission
  std::vector<uint8_t> flags;                        // AQL_LAUNCH_* per packet
};

struct AqlQueue {
  hsa_agent_t gpu{};
  hsa_queue_t* q = nullptr;
  hsa_signal_t done{};
  hsa_executable_t exe{};
  std::vector<hsa_code_object_reader_t> readers;
  std::vector<char> file;                             // the shared library's bytes (the code object readers point into it)
  std::map<std::string, Kernel> kernels;
  hsa_amd_memory_pool_t dev_pool{};
  char* karg_dev = nullptr;                           // kernel-argument arena in device memory (where HIP keeps them on this chip too:
  std::vector<char> karg_host;                        //   from host memory every CU's scalar cache fetches them over PCIe -- 26 us a launch)
  size_t karg_cap = 0, karg_uploaded = 0;
  uint64_t widx = 0;                                  // next packet id (this queue has one producer)
  bool inited = false;
  std::atomic<bool> queue_error{false};               // set by the runtime's error callback (its own thread), AFTER cb_err is written
  char cb_err[256] = "";                              // ... which owns this text; err belongs to the calling thread
  bool dead = false;                                  // a run was given up with packets still in the ring: never submit to this queue again
  int wait_s = 300;                                   // L2_QUEUE_WAIT_S: a run that makes no progress for this long is given up
  char err[512] = "";
};

static int qfail(AqlQueue* q, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(q->err, sizeof(q->err), fmt, ap);
  va_end(ap);
  return -1;
}
const char* aql_last_error(const AqlQueue* q) { return q ? q->err : "no queue"; }

static const char* hsa_str(hsa_status_t s) {
  const char* m = nullptr;
  hsa_status_string(s, &m);
  return m ? m : "?";
}

struct FindAgent { int domain, bdf; hsa_agent_t out; bool found; };
static hsa_status_t on_agent(hsa_agent_t a, void* p) {
  FindAgent* f = (FindAgent*)p;
  hsa_device_type_t t;
  if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS || t != HSA_DEVICE_TYPE_GPU) return HSA_STATUS_SUCCESS;
  uint32_t bdf = 0, dom = 0;
  hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf);
  hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &dom);
  if ((int)(bdf & 0xffff) == f->bdf && (int)dom == f->domain) { f->out = a; f->found = true; return HSA_STATUS_INFO_BREAK; }
  return HSA_STATUS_SUCCESS;
}
struct FindPool { hsa_amd_memory_pool_t out; bool found; };
static hsa_status_t on_pool(hsa_amd_memory_pool_t p, void* d) {
  FindPool* f = (FindPool*)d;
  hsa_amd_segment_t seg;
  uint32_t fl = 0;
  bool alloc = false;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &fl);
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
  if (seg == HSA_AMD_SEGMENT_GLOBAL && alloc && (fl & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED)) { f->out = p; f->found = true; return HSA_STATUS_INFO_BREAK; }
  return HSA_STATUS_SUCCESS;
}

static void on_queue_error(hsa_status_t s, hsa_queue_t*, void* d) {
  AqlQueue* q = (AqlQueue*)d;
  snprintf(q->cb_err, sizeof(q->cb_err), "HSA queue error: %s", hsa_str(s));
  q->queue_error.store(true, std::memory_order_release);
}
// the calling thread's view of it: the text is complete once the flag is seen
static int queue_failed(AqlQueue* q) {
  if (!q->queue_error.load(std::memory_order_acquire)) return 0;
  q->dead = true;
  return qfail(q, "%s", q->cb_err);
}

// gfx950 code objects inside the library: every clang offload bundle ("__CLANG_OFFLOAD_BUNDLE__", u64 entries, then per entry
// {u64 offset, u64 size, u64 triple length, triple}; offsets are relative to the bundle's start)
static std::vector<Bundle> find_code_objects(const std::vector<char>& f) {
  std::vector<Bundle> out;
  static const char magic[] = "__CLANG_OFFLOAD_BUNDLE__";
  const size_t ml = sizeof(magic) - 1;
  for (size_t i = 0; i + ml + 8 <= f.size();) {
    const void* hit = memmem(f.data() + i, f.size() - i, magic, ml);
    if (!hit) break;
    const size_t b = (const char*)hit - f.data();
    uint64_t n = 0;
    memcpy(&n, f.data() + b + ml, 8);
    size_t o = b + ml + 8;
    for (uint64_t k = 0; k < n && k < 64 && o + 24 <= f.size(); ++k) {
      uint64_t off, size, tl;
      memcpy(&off, f.data() + o, 8); memcpy(&size, f.data() + o + 8, 8); memcpy(&tl, f.data() + o + 16, 8);
      o += 24;
      if (o + tl > f.size()) break;
      const std::string triple(f.data() + o, tl);
      o += tl;
      if (size && triple.find("gfx950") != std::string::npos && b + off + size <= f.size() && !memcmp(f.data() + b + off, "\x7f" "ELF", 4)) out.push_back({b + off, (size_t)size});
    }
    i = b + ml;
  }
  return out;
}

static bool read_file(const char* path, std::vector<char>& out) {
  FILE* f = path ? fopen(path, "rb") : nullptr;
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  const long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  out.resize(sz > 0 ? (size_t)sz : 0);
  const size_t got = out.empty() ? 0 : fread(out.data(), 1, out.size(), f);
  fclose(f);
  return got == out.size() && got > 0;
}

int aql_count_code_objects(const char* so_path) {
  std::vector<char> f;
  if (!read_file(so_path, f)) return -1;
  return (int)find_code_objects(f).size();
}

AqlQueue* aql_create(int pci_domain, int pci_bus, int pci_device, int pci_function, const char* so_path, char* err, size_t errlen) {
  AqlQueue* q = new AqlQueue();
  auto bail = [&](const char* what, hsa_status_t s) -> AqlQueue* {
    if (err) snprintf(err, errlen, "%s: %s", what, s == HSA_STATUS_SUCCESS ? q->err : hsa_str(s));
    aql_destroy(q);
    return nullptr;
  };
  hsa_status_t s = hsa_init();
  if (s != HSA_STATUS_SUCCESS) return bail("hsa_init", s);
  q->inited = true;
  FindAgent fa = {pci_domain, ((pci_bus & 0xff) << 8) | ((pci_device & 0x1f) << 3) | (pci_function & 7), {}, false};
  hsa_iterate_agents(on_agent, &fa);
  if (!fa.found) { snprintf(q->err, sizeof(q->err), "no HSA GPU agent at PCI %04x:%02x:%02x.%d", pci_domain, pci_bus, pci_device, pci_function); return bail("agent", HSA_STATUS_SUCCESS); }
  q->gpu = fa.out;
  FindPool fp = {{}, false};
  hsa_amd_agent_iterate_memory_pools(q->gpu, on_pool, &fp);
  if (!fp.found) { snprintf(q->err, sizeof(q->err), "the GPU agent has no coarse-grained pool"); return bail("pool", HSA_STATUS_SUCCESS); }
  q->dev_pool = fp.out;
  // ---- the kernels: the library's own gfx950 code objects
  if (!read_file(so_path, q->file)) { snprintf(q->err, sizeof(q->err), "cannot read %s", so_path ? so_path : "(null)"); return bail("library", HSA_STATUS_SUCCESS); }
  const std::vector<Bundle> cos = find_code_objects(q->file);
  if (cos.empty()) { snprintf(q->err, sizeof(q->err), "no gfx950 code object in %s", so_path); return bail("library", HSA_STATUS_SUCCESS); }
  s = hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &q->exe);
  if (s != HSA_STATUS_SUCCESS) return bail("hsa_executable_create_alt", s);
  for (const Bundle& b : cos) {
    hsa_code_object_reader_t rd;
    s = hsa_code_object_reader_create_from_memory(q->file.data() + b.off, b.size, &rd);
    if (s != HSA_STATUS_SUCCESS) return bail("hsa_code_object_reader_create_from_memory", s);
    q->readers.push_back(rd);
    s = hsa_executable_load_agent_code_object(q->exe, q->gpu, rd, nullptr, nullptr);
    if (s != HSA_STATUS_SUCCESS) return bail("hsa_executable_load_agent_code_object", s);
  }
  s = hsa_executable_freeze(q->exe, nullptr);
  if (s != HSA_STATUS_SUCCESS) return bail("hsa_executable_freeze", s);
  s = hsa_queue_create(q->gpu, 16384, HSA_QUEUE_TYPE_SINGLE, on_queue_error, q, UINT32_MAX, UINT32_MAX, &q->q);
  if (s != HSA_STATUS_SUCCESS) return bail("hsa_queue_create", s);
  s = hsa_signal_create(1, 0, nullptr, &q->done);
  if (s != HSA_STATUS_SUCCESS) return bail("hsa_signal_create", s);
  q->karg_cap = 8u << 20;      // (a 7B-shaped tensor-parallel step is ~230 launches x 512 bytes per level and loop)
  s = hsa_amd_memory_pool_allocate(q->dev_pool, q->karg_cap, 0, (void**)&q->karg_dev);
  if (s != HSA_STATUS_SUCCESS) return bail("hsa_amd_memory_pool_allocate (kernel arguments)", s);
  q->karg_host.reserve(q->karg_cap);
  q->widx = hsa_queue_load_write_index_relaxed(q->q);
  { const char* w = getenv("L2_QUEUE_WAIT_S"); const int v = (w && *w) ? atoi(w) : 0; if (v > 0) q->wait_s = v; }
  return q;
}

void aql_destroy(AqlQueue* q) {
  if (!q) return;
  if (q->q) hsa_queue_destroy(q->q);
  if (q->done.handle) hsa_signal_destroy(q->done);
  if (q->karg_dev) hsa_amd_memory_pool_free(q->karg_dev);
  if (q->exe.handle) hsa_executable_destroy(q->exe);
  for (auto& r : q->readers) hsa_code_object_reader_destroy(r);
  if (q->inited) hsa_shut_down();
  delete q;
}

void aql_reset(AqlQueue* q) { if (q) { q->karg_host.clear(); q->karg_uploaded = 0; } }
AqlProgram* aql_program_new(AqlQueue*) { return new AqlProgram(); }
void aql_program_free(AqlProgram* p) { delete p; }
int aql_program_launches(const AqlProgram* p) { return p ? (int)p->pk.size() : 0; }

static const Kernel* find_kernel(AqlQueue* q, const char* name) {
  auto it = q->kernels.find(name);
  if (it != q->kernels.end()) return &it->second;
  hsa_executable_symbol_t sym;
  const std::string kd = std::string(name) + ".kd";
  hsa_status_t s = hsa_executable_get_symbol_by_name(q->exe, kd.c_str(), &q->gpu, &sym);
  if (s != HSA_STATUS_SUCCESS) { qfail(q, "kernel %s is not in the library's code objects: %s", name, hsa_str(s)); return nullptr; }
  Kernel k;
  if (hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object) != HSA_STATUS_SUCCESS ||
      hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg_size) != HSA_STATUS_SUCCESS ||
      hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group_static) != HSA_STATUS_SUCCESS ||
      hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.priv) != HSA_STATUS_SUCCESS) {
    qfail(q, "kernel %s: symbol information unavailable", name);
    return nullptr;
  }
  return &(q->kernels[name] = k);
}

int aql_record(AqlQueue* q, AqlProgram* p, const char* kernel_name, const unsigned grid[3], const unsigned block[3], unsigned lds_dynamic,
               const void* args, size_t arg_bytes, unsigned flags) {
  if (!q || !p || !kernel_name) return -1;
  const Kernel* k = find_kernel(q, kernel_name);
  if (!k) return -1;
  if (k->priv) return qfail(q, "kernel %s needs scratch memory (%u bytes per lane): not dispatched by hand", kernel_name, k->priv);
  const size_t hid = (arg_bytes + 7) & ~(size_t)7;      // the hidden arguments follow the explicit ones, 8-byte aligned
  if (arg_bytes > k->kernarg_size) return qfail(q, "kernel %s: %zu bytes of arguments, its segment holds %u", kernel_name, arg_bytes, k->kernarg_size);
  const size_t slot = ((size_t)k->kernarg_size + 63) & ~(size_t)63;      // a launch's arguments on cache lines of their own
  const size_t at = q->karg_host.size();
  if (at + slot > q->karg_cap) return qfail(q, "kernel-argument arena full (%zu bytes)", q->karg_cap);
  q->karg_host.resize(at + slot, 0);
  char* ka = q->karg_host.data() + at;
  if (arg_bytes) memcpy(ka, args, arg_bytes);
  // code-object-v5 hidden arguments (offsets from `hid`): block counts (3 x u32), group sizes (3 x u16), remainders (3 x u16, 0: whole
  // workgroups only), global offsets (+40, 3 x u64, 0), grid dimensions (+64, u16), dynamic LDS bytes (+120, u32)
  const int dims = (grid[2] > 1 || block[2] > 1) ? 3 : ((grid[1] > 1 || block[1] > 1) ? 2 : 1);
  if (hid + 24 <= k->kernarg_size) {
    const uint32_t bc[3] = {grid[0], grid[1], grid[2]};
    const uint16_t gs[3] = {(uint16_t)block[0], (uint16_t)block[1], (uint16_t)block[2]};
    memcpy(ka + hid, bc, 12);
    memcpy(ka + hid + 12, gs, 6);
  }
  if (hid + 66 <= k->kernarg_size) { const uint16_t gd = (uint16_t)dims; memcpy(ka + hid + 64, &gd, 2); }
  if (hid + 124 <= k->kernarg_size) { const uint32_t dl = lds_dynamic; memcpy(ka + hid + 120, &dl, 4); }
  hsa_kernel_dispatch_packet_t pk;
  memset(&pk, 0, sizeof(pk));
  pk.setup = (uint16_t)(dims << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS);
  pk.workgroup_size_x = (uint16_t)block[0]; pk.workgroup_size_y = (uint16_t)block[1]; pk.workgroup_size_z = (uint16_t)block[2];
  pk.grid_size_x = grid[0] * block[0]; pk.grid_size_y = grid[1] * block[1]; pk.grid_size_z = grid[2] * block[2];
  pk.private_segment_size = 0;
  pk.group_segment_size = k->group_static + lds_dynamic;
  pk.kernel_object = k->object;
  pk.kernarg_address = q->karg_dev + at;
  p->pk.push_back(pk);
  p->flags.push_back((uint8_t)flags);
  return 0;
}

int aql_upload(AqlQueue* q) {
  if (!q) return -1;
  if (q->karg_uploaded == q->karg_host.size()) return 0;
  const hsa_status_t s = hsa_memory_copy(q->karg_dev + q->karg_uploaded, q->karg_host.data() + q->karg_uploaded, q->karg_host.size() - q->karg_uploaded);
  if (s != HSA_STATUS_SUCCESS) return qfail(q, "hsa_memory_copy (kernel arguments): %s", hsa_str(s));
  q->karg_uploaded = q->karg_host.size();
  return 0;
}

int aql_queue_dead(const AqlQueue* q) { return q ? (q->dead ? 1 : 0) : 1; }

int aql_run(AqlQueue* q, int ntok, AqlProgram* const* per_token, int fence, double* elapsed_us) {
  if (!q || ntok < 0 || (ntok && !per_token)) return -1;
  if (q->dead) return qfail(q, "the queue was given up by an earlier run (%s)", q->cb_err[0] ? q->cb_err : "no progress within L2_QUEUE_WAIT_S");
  if (queue_failed(q)) return -1;
  if (q->karg_uploaded != q->karg_host.size()) return qfail(q, "kernel arguments recorded but not uploaded");
  size_t total = 0;
  for (int t = 0; t < ntok; ++t) total += per_token[t] ? per_token[t]->pk.size() : 0;
  if (!total) { if (elapsed_us) *elapsed_us = 0; return 0; }
  const uint32_t size = q->q->size;
  hsa_kernel_dispatch_packet_t* ring = (hsa_kernel_dispatch_packet_t*)q->q->base_address;
  hsa_signal_store_relaxed(q->done, 1);
  // fence = acquire scope + 4 * release scope (0 none, 1 agent, 2 system): the two differ only in A/B runs
  const int scope_a = (fence & 3) == AQL_FENCE_SYSTEM ? HSA_FENCE_SCOPE_SYSTEM : ((fence & 3) == AQL_FENCE_AGENT ? HSA_FENCE_SCOPE_AGENT : HSA_FENCE_SCOPE_NONE);
  const int scope_r = ((fence >> 2) & 3) == AQL_FENCE_SYSTEM ? HSA_FENCE_SCOPE_SYSTEM : (((fence >> 2) & 3) == AQL_FENCE_AGENT ? HSA_FENCE_SCOPE_AGENT : HSA_FENCE_SCOPE_NONE);
  size_t n = 0;
  std::chrono::steady_clock::time_point t0;
  bool started = false;
  // a run that makes no progress for wait_s seconds (L2_QUEUE_WAIT_S, default 300) is given up (the read index is what moves) -- and the
  // queue with it: its ring still holds this run's packets, `done` would be re-armed under them by the next run
  const auto bound = std::chrono::seconds(q->wait_s);
  auto deadline = std::chrono::steady_clock::now() + bound;
  uint64_t seen = hsa_queue_load_read_index_relaxed(q->q);
  auto progressed = [&]() {
    const uint64_t r = hsa_queue_load_read_index_relaxed(q->q);
    if (r != seen) { seen = r; deadline = std::chrono::steady_clock::now() + bound; }
  };
  auto give_up = [&]() { q->dead = true; return qfail(q, "the run made no progress for %d s (L2_QUEUE_WAIT_S); the queue is given up", q->wait_s); };
  for (int t = 0; t < ntok; ++t) {
    const AqlProgram* p = per_token[t];
    if (!p || p->pk.empty()) continue;
    // room for the token's packets (the queue holds a few tens of tokens: the host runs ahead of the chip and then keeps pace)
    while (q->widx + p->pk.size() - hsa_queue_load_read_index_scacquire(q->q) > size) {
      if (queue_failed(q)) return -1;
      progressed();
      if (std::chrono::steady_clock::now() > deadline) return give_up();
    }
    for (size_t i = 0; i < p->pk.size(); ++i, ++n) {
      hsa_kernel_dispatch_packet_t* dst = ring + ((q->widx + i) & (size - 1));
      const hsa_kernel_dispatch_packet_t& src = p->pk[i];
      const bool first = n == 0, last = n == total - 1;
      // body first, the header -- which turns the slot from INVALID into a dispatch -- last, as one 32-bit release store
      dst->workgroup_size_x = src.workgroup_size_x; dst->workgroup_size_y = src.workgroup_size_y; dst->workgroup_size_z = src.workgroup_size_z;
      dst->reserved0 = 0;
      dst->grid_size_x = src.grid_size_x; dst->grid_size_y = src.grid_size_y; dst->grid_size_z = src.grid_size_z;
      dst->private_segment_size = 0; dst->group_segment_size = src.group_segment_size;
      dst->kernel_object = src.kernel_object; dst->kernarg_address = src.kernarg_address; dst->reserved2 = 0;
      dst->completion_signal.handle = last ? q->done.handle : 0;
      // (fence bit 16: the FIRST launch of every token acquires at agent scope -- it refreshes the scalar caches, through which the
      // kernels read {token, pos}: those words change once per token, in its last launch)
      const bool wants_acq = ((fence & 16) && i == 0) || (p->flags[i] & AQL_LAUNCH_ACQUIRES);
      const int acq_tok = (wants_acq && scope_a == HSA_FENCE_SCOPE_NONE) ? HSA_FENCE_SCOPE_AGENT : scope_a;
      const int acq = first ? HSA_FENCE_SCOPE_SYSTEM : acq_tok, rel = last ? HSA_FENCE_SCOPE_SYSTEM : scope_r;
      const uint16_t header = (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                         (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
      __atomic_store_n((uint32_t*)dst, (uint32_t)header | ((uint32_t)src.setup << 16), __ATOMIC_RELEASE);
    }
    q->widx += p->pk.size();
    hsa_queue_store_write_index_release(q->q, q->widx);
    if (!started) { t0 = std::chrono::steady_clock::now(); started = true; }
    hsa_signal_store_screlease(q->q->doorbell_signal, (hsa_signal_value_t)(q->widx - 1));
  }
  // completion: spin briefly (a short run ends within microseconds), then sleep on the signal -- only a TIMED run (elapsed_us given) spins
  // to its end (the wake-up from a sleep is tens of microseconds late); an untimed one leaves the host core alone
  hsa_signal_value_t v = hsa_signal_wait_scacquire(q->done, HSA_SIGNAL_CONDITION_LT, 1, 2000000ull, HSA_WAIT_STATE_ACTIVE);
  while (v >= 1) {
    if (queue_failed(q)) return -1;
    progressed();
    if (std::chrono::steady_clock::now() > deadline) return give_up();
    v = hsa_signal_wait_scacquire(q->done, HSA_SIGNAL_CONDITION_LT, 1, 100000000ull, elapsed_us ? HSA_WAIT_STATE_ACTIVE : HSA_WAIT_STATE_BLOCKED);
  }
  if (elapsed_us) *elapsed_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  return queue_failed(q) ? -1 : 0;
}

// l2_napi.cc -- N-API binding of include/llama2_hip.h for Node (>= 12, N-API 6+).
//
// This is the reference-side stub INTEGRATION.md describes: the JS host keeps llama2.ts's main loop
// (llama2.ts:399-512) and calls these instead of its own transformer() (llama2.ts:468).  The addon has
// no link-time dependency on HIP: it dlopen()s libllama2hip.so when required and throws a JS Error
// if the library (or a GPU) is missing -- there is no CPU fallback.
//
// Typed arrays: napi_get_typedarray_info returns the data pointer ALREADY offset by byteOffset
// (FileHandleReader.getF32Array views may have a non-zero byteOffset, llama2.ts:56).
#include <dlfcn.h>
#include <node_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>

#include "../../include/llama2_hip.h"

namespace {

struct Api {
  void* so = nullptr;
  decltype(&l2_abi_version) abi_version;
  decltype(&l2_device_count) device_count;
  decltype(&l2_last_error) last_error;
  decltype(&l2_create) create;
  decltype(&l2_destroy) destroy;
  decltype(&l2_upload) upload;
  decltype(&l2_synth_fill) synth_fill;
  decltype(&l2_forward) forward;
  decltype(&l2_logits_host) logits_host;
  decltype(&l2_decode_greedy) decode_greedy;
  decltype(&l2_decode_sample) decode_sample;
  decltype(&l2_read_state) read_state;
  decltype(&l2_set_option) set_option;
  decltype(&l2_get_option) get_option;
  decltype(&l2_load_checkpoint) load_checkpoint;
  decltype(&l2_get_header) get_header;
  decltype(&l2_prefill) prefill;
  decltype(&l2_read_tensor) read_tensor;
} api;

std::string g_load_error;

bool load_library(const char* hint) {
  if (api.so) return true;
  const char* env = getenv("L2_LIB_PATH");
  const char* cands[] = {env, hint, "libllama2hip.so"};
  for (const char* c : cands) {
    if (!c || !*c) continue;
    api.so = dlopen(c, RTLD_NOW | RTLD_LOCAL);
    if (api.so) break;
    g_load_error = dlerror();
  }
  if (!api.so) return false;
#define BIND(name)                                                         \
  api.name = (decltype(api.name))dlsym(api.so, "l2_" #name);               \
  if (!api.name) { g_load_error = "missing symbol l2_" #name; dlclose(api.so); api.so = nullptr; return false; }
  BIND(abi_version) BIND(device_count) BIND(last_error) BIND(create) BIND(destroy) BIND(upload) BIND(synth_fill)
  BIND(forward) BIND(logits_host) BIND(decode_greedy) BIND(decode_sample) BIND(read_state) BIND(set_option)
  BIND(get_option) BIND(load_checkpoint) BIND(get_header) BIND(prefill) BIND(read_tensor)
#undef BIND
  if (api.abi_version() != L2_ABI_VERSION) { g_load_error = "ABI version mismatch"; dlclose(api.so); api.so = nullptr; return false; }
  return true;
}

napi_value throw_err(napi_env env, const char* what) {
  napi_throw_error(env, "L2", what);
  return nullptr;
}

napi_value throw_l2(napi_env env, int code) {
  char buf[640];
  snprintf(buf, sizeof(buf), "libllama2hip: %s (code %d)", api.last_error ? api.last_error() : "?", code);
  napi_throw_error(env, "L2", buf);
  return nullptr;
}

#define ARGS(n)                                                       \
  size_t argc = n;                                                    \
  napi_value argv[n];                                                 \
  if (napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr) != napi_ok || argc < n) return throw_err(env, "wrong number of arguments");

// A context stays alive until the JS handle is destroyed (or collected) AND every ArrayBuffer handed out
// over its pinned logits has been collected: a Float32Array must never outlive the memory under it.
struct Slot { l2_ctx* ctx; int views; bool closed; bool handle_gone; };

void slot_settle(Slot* s) {
  if (s->closed && s->views == 0 && s->ctx) { if (api.destroy) api.destroy(s->ctx); s->ctx = nullptr; }
  if (s->handle_gone && s->views == 0) delete s;
}

bool get_slot(napi_env env, napi_value v, Slot** out) {
  void* p = nullptr;
  if (napi_get_value_external(env, v, &p) != napi_ok || !p) { throw_err(env, "expected a context handle"); return false; }
  *out = (Slot*)p;
  if ((*out)->closed) { throw_err(env, "context already destroyed"); return false; }
  return true;
}

bool get_ctx(napi_env env, napi_value v, l2_ctx** out) {
  Slot* s;
  if (!get_slot(env, v, &s)) return false;
  *out = s->ctx;
  return true;
}

bool get_i32(napi_env env, napi_value v, int32_t* out) {
  if (napi_get_value_int32(env, v, out) != napi_ok) { throw_err(env, "expected an integer"); return false; }
  return true;
}

bool get_f32_array(napi_env env, napi_value v, float** data, size_t* len, bool allow_null) {
  napi_valuetype t;
  napi_typeof(env, v, &t);
  if (allow_null && (t == napi_null || t == napi_undefined)) { *data = nullptr; *len = 0; return true; }
  napi_typedarray_type tt;
  void* p = nullptr;
  napi_value ab;
  size_t off;
  if (napi_get_typedarray_info(env, v, &tt, len, &p, &ab, &off) != napi_ok || tt != napi_float32_array) {
    throw_err(env, "expected a Float32Array");
    return false;
  }
  *data = (float*)p;  // already offset by byteOffset
  return true;
}

void finalize_ctx(napi_env, void* data, void*) {   // the JS handle was collected
  Slot* s = (Slot*)data;
  s->closed = true;
  s->handle_gone = true;
  slot_settle(s);
}

// NOTE: no finalizer is attached to the exported ArrayBuffer.  Node 12's N-API aborts at environment
// teardown ("ArrayBufferReference::Finalize: Assertion !obj.IsEmpty()") when an external ArrayBuffer with a
// finalize callback is still alive, and freeing the pinned memory under a live view segfaults at exit.  So a
// context whose logits were exported stays allocated until the process ends (one per process in practice).

// open(libPath) -> true ; throws if the HIP library cannot be loaded
napi_value Open(napi_env env, napi_callback_info info) {
  ARGS(1)
  char path[4096] = "";
  size_t n = 0;
  napi_get_value_string_utf8(env, argv[0], path, sizeof(path), &n);
  if (!load_library(path)) {
    std::string m = "HIP library not loadable (" + g_load_error + "): build it with __graft_entry__.build(); there is no CPU fallback";
    return throw_err(env, m.c_str());
  }
  napi_value r;
  napi_get_boolean(env, true, &r);
  return r;
}

// create(Int32Array(7) header, device) -> handle            (readConfig + newRunState, llama2.ts:80-93, 147-163)
napi_value Create(napi_env env, napi_callback_info info) {
  ARGS(2)
  if (!api.so) return throw_err(env, "call open() first");
  napi_typedarray_type tt;
  size_t len;
  void* p;
  napi_value ab;
  size_t off;
  if (napi_get_typedarray_info(env, argv[0], &tt, &len, &p, &ab, &off) != napi_ok || tt != napi_int32_array || len != 7)
    return throw_err(env, "expected Int32Array(7) header");
  int32_t dev;
  if (!get_i32(env, argv[1], &dev)) return nullptr;
  l2_ctx* c = nullptr;
  int rc = api.create((const int32_t*)p, dev, &c);
  if (rc) return throw_l2(env, rc);
  Slot* slot = new Slot{c, 0, false, false};
  napi_value ext;
  napi_create_external(env, slot, finalize_ctx, nullptr, &ext);
  return ext;
}

// loadCheckpoint(path, device) -> { ctx, header: Int32Array(7) }   (native readWeights)
napi_value LoadCheckpoint(napi_env env, napi_callback_info info) {
  ARGS(2)
  if (!api.so) return throw_err(env, "call open() first");
  char path[4096] = "";
  size_t n = 0;
  napi_get_value_string_utf8(env, argv[0], path, sizeof(path), &n);
  int32_t dev;
  if (!get_i32(env, argv[1], &dev)) return nullptr;
  l2_ctx* c = nullptr;
  uint64_t bytes = 0;
  int rc = api.load_checkpoint(path, dev, 0, 1, nullptr, &c, &bytes);
  if (rc) return throw_l2(env, rc);
  Slot* slot = new Slot{c, 0, false, false};
  napi_value ext, ab, ta, obj;
  napi_create_external(env, slot, finalize_ctx, nullptr, &ext);
  void* data;
  napi_create_arraybuffer(env, 28, &data, &ab);
  api.get_header(c, (int32_t*)data);
  napi_create_typedarray(env, napi_int32_array, 7, ab, 0, &ta);
  napi_create_object(env, &obj);
  napi_set_named_property(env, obj, "ctx", ext);
  napi_set_named_property(env, obj, "header", ta);
  return obj;
}

napi_value Destroy(napi_env env, napi_callback_info info) {
  ARGS(1)
  void* p = nullptr;
  if (napi_get_value_external(env, argv[0], &p) == napi_ok && p) {
    Slot* s = (Slot*)p;
    s->closed = true;   // the context is freed now unless a logits view is still alive
    slot_settle(s);
  }
  return nullptr;
}

// upload(handle, kind, layer, Float32Array)                  (one array of readWeights, llama2.ts:112-129)
napi_value Upload(napi_env env, napi_callback_info info) {
  ARGS(4)
  l2_ctx* c;
  int32_t kind, layer;
  float* data;
  size_t n;
  if (!get_ctx(env, argv[0], &c) || !get_i32(env, argv[1], &kind) || !get_i32(env, argv[2], &layer) ||
      !get_f32_array(env, argv[3], &data, &n, false))
    return nullptr;
  int rc = api.upload(c, kind, layer, data, n);
  if (rc) return throw_l2(env, rc);
  return nullptr;
}

napi_value SynthFill(napi_env env, napi_callback_info info) {
  ARGS(2)
  l2_ctx* c;
  int32_t seed;
  if (!get_ctx(env, argv[0], &c) || !get_i32(env, argv[1], &seed)) return nullptr;
  int rc = api.synth_fill(c, (uint32_t)seed);
  if (rc) return throw_l2(env, rc);
  return nullptr;
}

// l2_forward / l2_prefill write vocab_size floats: a shorter Float32Array would be a heap overflow in the host
static bool logits_fit(napi_env env, l2_ctx* c, const float* out, size_t n) {
  if (!out) return true;
  int32_t hdr[7];
  if (api.get_header(c, hdr) != 0) { throw_err(env, "cannot read the context's header"); return false; }
  const size_t V = (size_t)(hdr[5] < 0 ? -hdr[5] : hdr[5]);
  if (n < V) { throw_err(env, "logits array too small (needs vocab_size floats)"); return false; }
  return true;
}

// forward(handle, token, pos, Float32Array|null)             (transformer(), llama2.ts:205-303 / :468)
napi_value Forward(napi_env env, napi_callback_info info) {
  ARGS(4)
  l2_ctx* c;
  int32_t token, pos;
  float* out;
  size_t n;
  if (!get_ctx(env, argv[0], &c) || !get_i32(env, argv[1], &token) || !get_i32(env, argv[2], &pos) ||
      !get_f32_array(env, argv[3], &out, &n, true))
    return nullptr;
  if (!logits_fit(env, c, out, n)) return nullptr;
  int rc = api.forward(c, token, pos, out);
  if (rc) return throw_l2(env, rc);
  return nullptr;
}

// prefill(handle, Int32Array tokens, pos0, Float32Array|null)   (prompt ingestion)
napi_value Prefill(napi_env env, napi_callback_info info) {
  ARGS(4)
  l2_ctx* c;
  int32_t pos0;
  float* out;
  size_t n;
  if (!get_ctx(env, argv[0], &c) || !get_i32(env, argv[2], &pos0) || !get_f32_array(env, argv[3], &out, &n, true)) return nullptr;
  napi_typedarray_type tt;
  size_t len;
  void* p;
  napi_value ab;
  size_t off;
  if (napi_get_typedarray_info(env, argv[1], &tt, &len, &p, &ab, &off) != napi_ok || tt != napi_int32_array)
    return throw_err(env, "expected Int32Array tokens");
  if (!logits_fit(env, c, out, n)) return nullptr;
  int rc = api.prefill(c, (const int32_t*)p, (int)len, pos0, out);
  if (rc) return throw_l2(env, rc);
  return nullptr;
}

// logitsBuffer(handle, vocab_size) -> ArrayBuffer over the pinned host logits (zero copy RunState.logits)
napi_value LogitsBuffer(napi_env env, napi_callback_info info) {
  ARGS(2)
  Slot* s;
  int32_t V;
  if (!get_slot(env, argv[0], &s) || !get_i32(env, argv[1], &V)) return nullptr;
  float* p = api.logits_host(s->ctx);
  if (!p) return throw_err(env, "no logits buffer");
  napi_value ab;
  if (napi_create_external_arraybuffer(env, p, (size_t)V * 4, nullptr, nullptr, &ab) != napi_ok)
    return throw_err(env, "external ArrayBuffer not supported by this runtime");
  s->views++;
  return ab;
}

// decodeGreedy(handle, firstToken, pos0, steps) -> Int32Array  (llama2.ts:465-508 at -t 0, on the device)
napi_value DecodeGreedy(napi_env env, napi_callback_info info) {
  ARGS(4)
  l2_ctx* c;
  int32_t first, pos0, steps;
  if (!get_ctx(env, argv[0], &c) || !get_i32(env, argv[1], &first) || !get_i32(env, argv[2], &pos0) || !get_i32(env, argv[3], &steps))
    return nullptr;
  if (steps < 0) return throw_err(env, "steps < 0");
  napi_value ab, ta;
  void* data;
  napi_create_arraybuffer(env, (size_t)steps * 4, &data, &ab);
  int rc = api.decode_greedy(c, first, pos0, steps, (int32_t*)data);
  if (rc) return throw_l2(env, rc);
  napi_create_typedarray(env, napi_int32_array, (size_t)steps, ab, 0, &ta);
  return ta;
}

// decodeSample(handle, firstToken, pos0, steps, temperature, topp, Uint32Array[2] rng {lo, hi}) -> Int32Array
// The sampled branch (llama2.ts:480-493) on the device.  The 64-bit xorshift* state travels as two uint32 halves
// (updated in place) so that the binding needs no BigInt support from the N-API level.
napi_value DecodeSample(napi_env env, napi_callback_info info) {
  ARGS(7)
  l2_ctx* c;
  int32_t first, pos0, steps;
  double temperature, topp;
  if (!get_ctx(env, argv[0], &c) || !get_i32(env, argv[1], &first) || !get_i32(env, argv[2], &pos0) || !get_i32(env, argv[3], &steps))
    return nullptr;
  if (napi_get_value_double(env, argv[4], &temperature) != napi_ok || napi_get_value_double(env, argv[5], &topp) != napi_ok)
    return throw_err(env, "temperature and topp must be numbers");
  napi_typedarray_type tt;
  size_t len = 0;
  void* rdata = nullptr;
  if (napi_get_typedarray_info(env, argv[6], &tt, &len, &rdata, nullptr, nullptr) != napi_ok || tt != napi_uint32_array || len != 2)
    return throw_err(env, "rng must be a Uint32Array of length 2 {lo, hi}");
  if (steps < 0) return throw_err(env, "steps < 0");
  uint32_t* halves = (uint32_t*)rdata;
  uint64_t state = ((uint64_t)halves[1] << 32) | halves[0];
  napi_value ab, ta;
  void* data;
  napi_create_arraybuffer(env, (size_t)steps * 4, &data, &ab);
  int rc = api.decode_sample(c, first, pos0, steps, temperature, topp, &state, (int32_t*)data);
  if (rc) return throw_l2(env, rc);
  halves[0] = (uint32_t)state; halves[1] = (uint32_t)(state >> 32);
  napi_create_typedarray(env, napi_int32_array, (size_t)steps, ab, 0, &ta);
  return ta;
}

// readState(handle, which, layer, Float32Array)
napi_value ReadState(napi_env env, napi_callback_info info) {
  ARGS(4)
  l2_ctx* c;
  int32_t which, layer;
  float* out;
  size_t n;
  if (!get_ctx(env, argv[0], &c) || !get_i32(env, argv[1], &which) || !get_i32(env, argv[2], &layer) ||
      !get_f32_array(env, argv[3], &out, &n, false))
    return nullptr;
  int rc = api.read_state(c, which, layer, out, n);
  if (rc) return throw_l2(env, rc);
  return nullptr;
}

// readTensor(handle, kind, layer, offset, Float32Array)      (what l2_upload stored: tests of the hand-over of llama2.ts:51-59's views)
napi_value ReadTensor(napi_env env, napi_callback_info info) {
  ARGS(5)
  l2_ctx* c;
  int32_t kind, layer;
  int64_t offset;
  float* out;
  size_t n;
  if (!get_ctx(env, argv[0], &c) || !get_i32(env, argv[1], &kind) || !get_i32(env, argv[2], &layer)) return nullptr;
  if (napi_get_value_int64(env, argv[3], &offset) != napi_ok || offset < 0) return throw_err(env, "offset must be a non-negative integer");
  if (!get_f32_array(env, argv[4], &out, &n, false)) return nullptr;
  int rc = api.read_tensor(c, kind, layer, (size_t)offset, out, n);
  if (rc) return throw_l2(env, rc);
  return nullptr;
}

napi_value SetOption(napi_env env, napi_callback_info info) {
  ARGS(3)
  l2_ctx* c;
  int32_t key, value;
  if (!get_ctx(env, argv[0], &c) || !get_i32(env, argv[1], &key) || !get_i32(env, argv[2], &value)) return nullptr;
  int rc = api.set_option(c, key, value);
  if (rc) return throw_l2(env, rc);
  return nullptr;
}

napi_value GetOption(napi_env env, napi_callback_info info) {
  ARGS(2)
  l2_ctx* c;
  int32_t key;
  if (!get_ctx(env, argv[0], &c) || !get_i32(env, argv[1], &key)) return nullptr;
  int value = 0;
  int rc = api.get_option(c, key, &value);
  if (rc) return throw_l2(env, rc);
  napi_value r;
  napi_create_int32(env, value, &r);
  return r;
}

napi_value DeviceCount(napi_env env, napi_callback_info) {
  if (!api.so) return throw_err(env, "call open() first");
  napi_value r;
  napi_create_int32(env, api.device_count(), &r);
  return r;
}

napi_value Init(napi_env env, napi_value exports) {
  struct { const char* name; napi_callback fn; } fns[] = {
      {"open", Open}, {"create", Create}, {"destroy", Destroy}, {"upload", Upload}, {"synthFill", SynthFill},
      {"forward", Forward}, {"logitsBuffer", LogitsBuffer}, {"decodeGreedy", DecodeGreedy}, {"decodeSample", DecodeSample}, {"readState", ReadState},
      {"setOption", SetOption}, {"getOption", GetOption}, {"deviceCount", DeviceCount}, {"loadCheckpoint", LoadCheckpoint}, {"prefill", Prefill}, {"readTensor", ReadTensor}};
  for (auto& f : fns) {
    napi_value v;
    napi_create_function(env, f.name, NAPI_AUTO_LENGTH, f.fn, nullptr, &v);
    napi_set_named_property(env, exports, f.name, v);
  }
  return exports;
}

}  // namespace

NAPI_MODULE(l2_napi, Init)

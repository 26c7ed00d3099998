// Device-side temperature / top-p sampling (llama2.ts:348-394 and 476-493): the part of the
// reference's decode loop that sits between transformer() and the next token, kept on the GPU so that a sampled
// run needs no 128 KB logits hand-off and no host sort of 32 000 objects per token.  sampler.hip holds the
// kernels; this is the interface llama2_hip.hip uses.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace l2s {

struct Sampler {
  int V = 0;
  float* probs = nullptr;        // (V) exps of the scaled logits (the serial form: -> probabilities, in place like state.logits)
  float* probs_n = nullptr;      // (V) probabilities
  float* probs_sorted = nullptr; // (V) descending (top-p)
  int* idx = nullptr;            // (V) 0..V-1
  int* idx_sorted = nullptr;     // (V) token ids in descending-probability order, ties by id (stable sort)
  double* params = nullptr;      // device {temperature, topp}
  unsigned long long* rng = nullptr;   // device xorshift* state (the reference's BigInt rng_seed)
  // whole-chip form: tiles of 1024 elements (sampler.hip)
  int G = 0;                     // tiles
  float* run_p = nullptr;        // (G * 1024) tiles sorted one by one (top-p)
  double* part = nullptr;        // (G) approximate tile sums
  double* part_sorted = nullptr; // (G) the same for the sorted order (accumulated by the rank merge, zero between tokens)
  void* recs = nullptr;          // (G * 1025) xs::Run records of the exps' running sum
  int* cnt = nullptr;            // (G) records per tile
  void* recs2 = nullptr;         // the same for the probabilities (in index or sorted order)
  int* cnt2 = nullptr;
  int* off = nullptr;            // (G + 1)
  double* runS = nullptr;        // per run: exact running sum after it
  int* runEnd = nullptr;
  int* runBad = nullptr;
  unsigned long long* cq = nullptr;   // (G * 1024) per element: grid composite since the start of its run
  int* cm = nullptr;
  unsigned* mxkey = nullptr;     // max of the scaled logits (order-preserving key), zero between tokens
  // margin form (sampler_margin.hip.h): tile sums of the probabilities, float spacings of the ambiguous quotients, arrivals, counters
  double* part2 = nullptr;       // (G)
  double* amb = nullptr;         // (G)
  unsigned* ticket = nullptr;    // zero between tokens
  double* total = nullptr;       // top-p: the exact total of the exps (runs_total_kernel -> the tile sort)
  unsigned long long* stats = nullptr;   // {tokens picked, of those by the serial loop}
  bool chain = false;            // L2_SAMPLER_CHAIN=1: every running sum exact on the whole chip (round 2-3 default), kept for A/B
  bool force_serial = false;     // L2_SAMPLER_FORCE_SERIAL=1: the margin form treats every token as undecided
  unsigned* rank_acc = nullptr;  // (G * 1024) the rank merge's per-element accumulators {groups reported : 8, elements in front : 24}, zero between tokens
  bool serial = false;           // L2_SAMPLER_SERIAL=1: one lane accumulates (the straightforward form, kept for A/B)
};

enum { MAX_VOCAB = 256 * 1024 };   // one chain thread per 1024-element tile

// Launch recorder: while one is set (thread-local), every launch of enqueue() is handed to it -- host function, geometry, dynamic LDS,
// the explicit arguments packed as the kernel-argument segment lays them out -- instead of to HIP: the library's own AQL queue
// records the sampled step that way (aql_queue.h).  Returns false when the launch could not be recorded.
typedef bool (*LaunchRecorder)(void* user, const void* host_fn, dim3 grid, dim3 block, size_t lds, hipStream_t st, const void* args, size_t arg_bytes);
void set_recorder(LaunchRecorder r, void* user);
bool recorder_failed();      // since the last set_recorder

hipError_t create(Sampler* s, int V);
void destroy(Sampler* s);
// Enqueue one sampled step after the classifier: reads `logits` (V floats, left untouched), picks the next token
// exactly as llama2.ts:480-493 does, then advances {token, pos, step} in `tokpos` and stores the token in
// tokens_out[step] -- the same protocol as argmax_advance_kernel.  `topp_mode`: the sample_topp branch
// (0 < topp < 1); temperature and topp themselves are read from s.params at run time.  `amax` (may be null): the 8
// argmax keys the classifier folded max(logits) into (one per 128-byte line); usable only for temperature > 0, saves
// the sampler's own maximum pass; the sampler zeroes them for the next token.
hipError_t enqueue(const Sampler& s, const float* logits, bool topp_mode, int* tokpos, int* tokens_out, unsigned long long* amax, hipStream_t st);

// {tokens the margin form picked, of those by its serial loop} since create(); synchronous.
hipError_t read_stats(const Sampler& s, unsigned long long out[2], hipStream_t st);

// Diagnostic: running sums S_i = fl(S_{i-1} + x_i) of n <= MAX_VOCAB non-negative fp32 values, by the exact parallel
// algorithm (synchronous).
hipError_t running_sums(const float* x_dev, int n, double* prefix_dev, hipStream_t st);

}  // namespace l2s

// Device-side sampling for libllama2hip.so (gfx950): temperature scaling, softmax, sample / sample_topp and the
// xorshift* RNG of wizzard0/llama2.ts (llama2.ts:348-394, 476-493), restated so that the SAME token comes out.
//
// What makes that non-trivial: the reference's sums are sequential fp64 accumulations of fp32 values
// (softmax :189, sample :369/:373, sample_topp :385/:391) and the chosen index depends on comparing a random
// threshold against those running sums, so a tree sum (different in the last bits) can flip a token.
//
// Default form (sampler_margin.hip.h; +11 us per token for sample, +39 us for top-p at stories110M): the function returns an INDEX, so the
// running sums only have to be known well enough to decide every comparison the loop makes.  Tree sums over the whole chip plus a proven
// margin (margin_rule.h: n 2^-53 per summation order, the float spacing of every probability a tree total could round differently) decide
// it; a token whose running sum comes within the margin of its threshold (a few in a million) is picked by the reference's loop run as
// written by one lane of the same workgroup.  Launches: sample 2 (exps + tile sums -> probabilities' tile sums, the last workgroup picks);
// top-p 5 (exps; runs of the exps + their exact total by the last workgroup; probabilities + tile sort; rank merge; pick) -- the descending
// order of sample_topp needs the exact probabilities, hence the exact total there.
// Exact-chain form (L2_SAMPLER_CHAIN=1 behind L2_TEST_HOOKS; the default of rounds 2-3; 4 launches for sample, 6 for top-p; +31 / +59 us):
//   exp + tile sums (the maximum comes from the classifier's argmax keys) -> runs of the exps
//   sample:  [exact total -> probabilities -> their runs] -> chain: exact running sums, threshold, search, advance
//   top-p:   [exact total -> probabilities -> sorted tiles] -> rank merge (+ tile sums) -> runs -> chain
// "runs" / "chain" are exact_sum.h: every 1024-element tile turns its elements into integer increments on the grid its
// approximate prefix predicts, one wave walks the ~50 runs of a 32 000-element vector with exact fp64 state (stretches of
// runs on one grid composed by a scan first), every prediction is checked, and the searched index is evaluated inside
// the one run that contains it.  The bracketed steps share a launch: each of their workgroups repeats the walk for the
// total instead of waiting for a launch that would hand it over.  Bit-identical to the serial loop by construction
// (tests/test_exact_sum_cpu.py on the host, l2_debug_running_sums on the GPU).  The default form keeps its first half for top-p.
// The descending stable sort of sample_topp (Array.prototype.sort is stable in V8 >= 7.0) is a bitonic sort of
// (probability, id) keys per 1024-element tile followed by one rank-by-binary-search merge of the sorted tiles out of LDS.
//
// L2_SAMPLER_SERIAL=1 keeps the straightforward form for A/B: ONE lane adds in index order (~10 cycles per element,
// ~130 us per pass over 32 000 values) inside a single 1024-thread workgroup (top-p: behind the same tile sort + rank merge).
#include "sampler.h"
#include <string.h>
#include "exact_sum.h"
#include "margin_rule.h"

#include <stdlib.h>

namespace l2s {

static thread_local LaunchRecorder g_rec = nullptr;
static thread_local void* g_rec_user = nullptr;
static thread_local bool g_rec_failed = false;
void set_recorder(LaunchRecorder r, void* user) { g_rec = r; g_rec_user = user; g_rec_failed = false; }
bool recorder_failed() { return g_rec_failed; }

template <class T>
static void pack_arg(char* buf, size_t& off, const T& v) {
  off = (off + alignof(T) - 1) & ~(alignof(T) - 1);
  memcpy(buf + off, &v, sizeof(T));
  off += sizeof(T);
}
// every launch of the sampled step: HIP, or the recorder (sampler.h)
template <class... KA, class... A>
static void s_launch(void (*kernel)(KA...), dim3 grid, dim3 block, size_t lds, hipStream_t st, const A&... a) {
  static_assert(sizeof...(KA) == sizeof...(A), "argument count");
  if (g_rec) {
    char buf[1024];
    size_t off = 0;
    static_assert((sizeof(KA) + ... + 0) + 8 * sizeof...(KA) <= sizeof(buf), "kernel arguments exceed the packing buffer");
    (pack_arg<KA>(buf, off, static_cast<KA>(a)), ...);
    if (!g_rec(g_rec_user, reinterpret_cast<const void*>(kernel), grid, block, lds, st, buf, off)) g_rec_failed = true;
    return;
  }
  hipLaunchKernelGGL(kernel, grid, block, lds, st, a...);
}

#pragma clang fp contract(off)

constexpr int NT = 1024;     // threads of the one workgroup of the serial form
constexpr int CH = 4096;     // values staged in LDS per chunk, widened to fp64 (32 KB)
constexpr int SEG = 512;     // spacing of recorded running sums
constexpr int MAXSEG = MAX_VOCAB / SEG;

__device__ __forceinline__ float random_f32(unsigned long long* rng) {   // llama2.ts:349-360
  unsigned long long s = *rng;
  s ^= s >> 12;
  s ^= s << 25;
  s ^= s >> 27;
  *rng = s;
  const unsigned u = (unsigned)((s * 0x2545F4914F6CDD1Dull) >> 32);
  return (float)(((double)u / 256.0) / 16777216.0);      // one rounding at the Float32Array store (:358)
}

__device__ __forceinline__ void advance(int* tokpos, int* tokens_out, int next) {
  const int step = tokpos[2];
  tokens_out[step] = next;
  tokpos[0] = next; tokpos[1] = tokpos[1] + 1; tokpos[2] = step + 1;
}

#include "sampler_serial.hip.h"
#include "sampler_chain.hip.h"
#include "sampler_sort.hip.h"
#include "sampler_margin.hip.h"

// Stage 1 = the exps (recs / cnt), stage 2 = the probabilities, in index or in sorted order (recs2 / cnt2, cq / cm): two
// sets of run records because the fused kernels write stage 2 while other workgroups still read stage 1.
static ChainArgs chain_args(const Sampler& s, const float* x, const double* part, bool stage2) {
  ChainArgs a = {};
  a.x = x; a.V = s.V; a.G = s.G; a.part = part;
  a.recs = (const Run*)(stage2 ? s.recs2 : s.recs); a.cnt = stage2 ? s.cnt2 : s.cnt;
  a.off = s.off; a.S = s.runS; a.End = s.runEnd; a.Bad = s.runBad; a.cq = s.cq; a.cm = s.cm;
  a.params = s.params; a.rng = s.rng; a.mxkey = s.mxkey;
  return a;
}

static hipError_t launch_chain(const ChainArgs& a, int mode, hipStream_t st) {
  if (mode == CHAIN_SAMPLE) s_launch(chain_kernel<CHAIN_SAMPLE>, dim3(1), dim3(TN), 0, st, a);
  else if (mode == CHAIN_TOPP) s_launch(chain_kernel<CHAIN_TOPP>, dim3(1), dim3(TN), 0, st, a);
  else s_launch(chain_kernel<CHAIN_DEBUG>, dim3(1), dim3(TN), 0, st, a);
  return hipGetLastError();
}

hipError_t running_sums(const float* x_dev, int n, double* prefix_dev, hipStream_t st) {
  Sampler s;
  hipError_t e = create(&s, n);
  if (e != hipSuccess) return e;
  s_launch(tile_sums_kernel, dim3(s.G), dim3(TN), 0, st, x_dev, n, s.part);
  s_launch(runs_kernel<false>, dim3(s.G), dim3(TN), 0, st, x_dev, n, s.part, (Run*)s.recs, s.cnt, s.cq, s.cm);
  e = launch_chain(chain_args(s, x_dev, s.part, false), CHAIN_DEBUG, st);
  if (e == hipSuccess) {
    s_launch(prefix_kernel, dim3(s.G), dim3(TN), 0, st, x_dev, n, s.part, s.off, s.runS, s.runBad, s.runEnd, prefix_dev);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  destroy(&s);
  return e;
}

hipError_t read_stats(const Sampler& s, unsigned long long out[2], hipStream_t st) {
  if (!s.stats) return hipErrorInvalidValue;
  hipError_t e = hipMemcpyAsync(out, s.stats, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st);
  return e == hipSuccess ? hipStreamSynchronize(st) : e;
}

hipError_t create(Sampler* s, int V) {
  if (V <= 0 || V > MAX_VOCAB) return hipErrorInvalidValue;
  s->V = V;
  s->G = (V + TILE - 1) / TILE;
  hipError_t e;
#define L2S(x) do { e = (x); if (e != hipSuccess) { destroy(s); return e; } } while (0)
  const size_t padded = (size_t)((V + STILE - 1) / STILE) * STILE, max_runs = (size_t)s->G * (TILE + 1);
  L2S(hipMalloc(&s->probs, padded * 4));
  L2S(hipMalloc(&s->probs_n, padded * 4));
  L2S(hipMalloc(&s->probs_sorted, padded * 4));
  L2S(hipMalloc(&s->idx, padded * 4));
  L2S(hipMalloc(&s->idx_sorted, padded * 4));
  L2S(hipMalloc(&s->run_p, padded * 4));
  L2S(hipMalloc(&s->params, 2 * sizeof(double)));
  L2S(hipMalloc(&s->rng, sizeof(unsigned long long)));
  L2S(hipMalloc(&s->part, (size_t)s->G * sizeof(double)));
  L2S(hipMalloc(&s->part_sorted, (size_t)s->G * sizeof(double)));
  L2S(hipMemset(s->part_sorted, 0, (size_t)s->G * sizeof(double)));
  L2S(hipMalloc(&s->recs, max_runs * sizeof(Run)));
  L2S(hipMalloc(&s->recs2, max_runs * sizeof(Run)));
  L2S(hipMalloc(&s->cnt, (size_t)s->G * sizeof(int)));
  L2S(hipMalloc(&s->cnt2, (size_t)s->G * sizeof(int)));
  L2S(hipMalloc(&s->off, (size_t)(s->G + 1) * sizeof(int)));
  L2S(hipMalloc(&s->runS, max_runs * sizeof(double)));
  L2S(hipMalloc(&s->runEnd, max_runs * sizeof(int)));
  L2S(hipMalloc(&s->runBad, max_runs * sizeof(int)));
  L2S(hipMalloc(&s->cq, padded * sizeof(unsigned long long)));
  L2S(hipMalloc(&s->cm, padded * sizeof(int)));
  L2S(hipMalloc(&s->mxkey, sizeof(unsigned)));
  L2S(hipMemset(s->mxkey, 0, sizeof(unsigned)));
  L2S(hipMalloc(&s->part2, (size_t)s->G * sizeof(double)));
  L2S(hipMalloc(&s->amb, (size_t)s->G * sizeof(double)));
  L2S(hipMalloc(&s->ticket, sizeof(unsigned)));
  L2S(hipMemset(s->ticket, 0, sizeof(unsigned)));
  L2S(hipMalloc(&s->total, sizeof(double)));
  L2S(hipMalloc(&s->stats, 2 * sizeof(unsigned long long)));
  L2S(hipMemset(s->stats, 0, 2 * sizeof(unsigned long long)));
  {                                                            // A/B forms, development gate
    const char* g_ = getenv("L2_TEST_HOOKS");
    auto hook = [&](const char* name) { const char* e_ = getenv(name); return g_ && atoi(g_) != 0 && e_ && atoi(e_) != 0; };
    s->serial = hook("L2_SAMPLER_SERIAL"); s->chain = hook("L2_SAMPLER_CHAIN"); s->force_serial = hook("L2_SAMPLER_FORCE_SERIAL");
  }
  L2S(hipMalloc(&s->rank_acc, padded * sizeof(unsigned)));       // the rank merge's per-element accumulators: zero between tokens
  L2S(hipMemset(s->rank_acc, 0, padded * sizeof(unsigned)));
#undef L2S
  return hipSuccess;
}

void destroy(Sampler* s) {
  void* bufs[] = {s->probs, s->probs_n, s->probs_sorted, s->idx, s->idx_sorted, s->run_p, s->params, s->rng, s->part, s->part_sorted,
                  s->recs, s->recs2, s->cnt, s->cnt2, s->off, s->runS, s->runEnd, s->runBad, s->cq, s->cm, s->mxkey,
                  s->part2, s->amb, s->ticket, s->stats, s->rank_acc, s->total};
  for (void* b : bufs) if (b) (void)hipFree(b);
  *s = Sampler();
}

// Descending stable order of V values (probabilities, or exps still to be divided by their exact total when `fused`): sorted tiles, then the
// rank merge writes probs_sorted / idx_sorted and adds every value to the sum of the 1024-element tile it lands in (part_sorted).
static hipError_t rank_merge(const Sampler& s, int gs, hipStream_t st) {
  const int n = gs * STILE;
  s_launch(sort_rank_kernel, dim3((n + RT - 1) / RT, (gs + RANK_TQ - 1) / RANK_TQ), dim3(RT), 0, st, s.run_p, s.idx, gs, s.G, s.rank_acc,
                     s.probs_sorted, s.idx_sorted, s.part_sorted);
  return hipGetLastError();
}

static hipError_t sort_descending(const Sampler& s, const float* values, const ChainArgs& exps, bool fused, hipStream_t st) {
  const int gs = (s.V + STILE - 1) / STILE;
  if (fused) s_launch(sort_tile_kernel<true>, dim3(gs), dim3(TN), 0, st, exps, values, s.V, s.run_p, s.idx);
  else s_launch(sort_tile_kernel<false>, dim3(gs), dim3(TN), 0, st, exps, values, s.V, s.run_p, s.idx);
  return rank_merge(s, gs, st);
}

hipError_t enqueue(const Sampler& s, const float* logits, bool topp_mode, int* tokpos, int* tokens_out, unsigned long long* amax, hipStream_t st) {
  hipError_t e;
  if (s.serial) {
    if (!topp_mode) {
      s_launch(sample_kernel, dim3(1), dim3(NT), 0, st, logits, s.V, s.params, s.probs, s.rng, tokpos, tokens_out);
      return hipGetLastError();
    }
    s_launch(softmax_kernel, dim3(1), dim3(NT), 0, st, logits, s.V, s.params, s.probs, (int*)nullptr);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if ((e = sort_descending(s, s.probs, ChainArgs(), false, st)) != hipSuccess) return e;
    s_launch(topp_kernel, dim3(1), dim3(NT), 0, st, s.probs_sorted, s.idx_sorted, s.V, s.params, s.rng, tokpos, tokens_out);
    return hipGetLastError();
  }
  // temperature + exp (:481-483, :183-188), runs of the exps' running sum
  if (!amax) s_launch(scaled_max_kernel, dim3(s.G), dim3(TN), 0, st, logits, s.V, s.params, s.mxkey);
  s_launch(exp_kernel, dim3(s.G), dim3(TN), 0, st, logits, s.V, s.params, s.mxkey, amax, s.probs, s.part);
  if (!s.chain) {
    MarginArgs m = {};
    m.exps = s.probs; m.part = s.part; m.V = s.V; m.G = s.G; m.part2 = s.part2; m.amb = s.amb; m.ticket = s.ticket;
    m.sorted = s.probs_sorted; m.ids = s.idx_sorted; m.part_sorted = s.part_sorted; m.params = s.params; m.rng = s.rng;
    m.tokpos = tokpos; m.tokens_out = tokens_out; m.mxkey = s.mxkey; m.amax = amax; m.stats = s.stats; m.force_serial = s.force_serial ? 1 : 0;
    if (!topp_mode) {
      s_launch(sample_margin_kernel, dim3(s.G), dim3(TN), 0, st, m);
      return hipGetLastError();
    }
    // the descending order needs the exact probabilities: runs of the exps, [exact total -> probabilities -> sorted tiles], rank merge
    s_launch(runs_total_kernel, dim3(s.G), dim3(TN), 0, st, chain_args(s, s.probs, s.part, false), (Run*)s.recs, s.cnt, s.ticket, s.total);
    const int gs = (s.V + STILE - 1) / STILE;
    s_launch(sort_tile_wide_kernel, dim3(gs), dim3(WT), 0, st, s.probs, s.total, s.V, s.run_p, s.idx);
    if ((e = rank_merge(s, gs, st)) != hipSuccess) return e;
    s_launch(topp_margin_kernel, dim3(1), dim3(TN), 0, st, m);
    return hipGetLastError();
  }
  s_launch(runs_kernel<false>, dim3(s.G), dim3(TN), 0, st, s.probs, s.V, s.part, (Run*)s.recs, s.cnt, s.cq, s.cm);
  const ChainArgs exps = chain_args(s, s.probs, s.part, false);
  ChainArgs pick;
  if (!topp_mode) {
    // exact total -> probabilities -> their runs, in one launch; then sample (:368-376)
    s_launch(normalise_runs_kernel, dim3(s.G), dim3(TN), 0, st, exps, s.probs_n, (Run*)s.recs2, s.cnt2, s.cq, s.cm);
    pick = chain_args(s, s.probs_n, s.part, true);
  } else {
    // sample_topp (:378-394): exact total -> probabilities -> sorted tiles in one launch, rank merge, runs of the sorted order
    if ((e = sort_descending(s, s.probs, exps, true, st)) != hipSuccess) return e;
    s_launch(runs_kernel<true>, dim3(s.G), dim3(TN), 0, st, s.probs_sorted, s.V, s.part_sorted, (Run*)s.recs2, s.cnt2, s.cq, s.cm);
    pick = chain_args(s, s.probs_sorted, s.part_sorted, true);
    pick.part_sorted = s.part_sorted;
  }
  if ((e = hipGetLastError()) != hipSuccess) return e;
  pick.tokpos = tokpos; pick.tokens_out = tokens_out; pick.amax = amax;
  pick.ids = topp_mode ? s.idx_sorted : nullptr;
  return launch_chain(pick, topp_mode ? CHAIN_TOPP : CHAIN_SAMPLE, st);
}

}  // namespace l2s

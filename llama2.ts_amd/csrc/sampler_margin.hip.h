// sampler_margin.hip.h -- the default form of the device sampler: parallel sums with a PROVEN margin, the serial loop only inside it
// Part of sampler.hip (included there inside namespace l2s, in order); not a stand-alone header.
#pragma once

// What the reference's sample() / sample_topp() return (llama2.ts:368-394) is an INDEX: the first i whose sequential fp64 running
// sum cum_i passes a threshold.  The running sums themselves never leave the function, so they only have to be known well enough
// to decide every comparison the loop makes:
//   * any summation order of n non-negative values is within n * 2^-53 (relative) of the true sum, the reference's sequential
//     order included, so a tree sum Q_i is within 2 n 2^-53 Q_n of cum_i;
//   * plain sample(): the probabilities are fl32(e_i / total) with the sequential total of the exps (softmax :189-192).  With a tree
//     total instead, a quotient can only round to another float when it sits within (4 n + 8) fp64 steps of the midpoint of two floats;
//     every such element ("ambiguous": ~16 of 32 000) adds its float spacing to A, which bounds what all of them together can move
//     any running sum;
//   * with M = 8 (n + 64) 2^-53 Q_n + 4 A (twice what the bounds need): `threshold < cum_i` is TRUE if Q_i > threshold~ + M and FALSE if
//     Q_i < threshold~ - M, threshold~ being the same product / constant formed from the tree sums.  cum is monotone, so one element
//     known FALSE directly in front of one known TRUE pins the index the serial loop returns.
// When the two neighbours are not both decided (a running sum within ~2^-33 of the threshold: a few tokens in a million) the SAME
// workgroup runs the reference's loop as written -- one lane, index order -- so the token never rests on anything but a proof or
// the loop itself.  L2_SAMPLER_FORCE_SERIAL=1 (behind L2_TEST_HOOKS) declares every token undecided: the tests run both branches.
//
// Launches per token: plain sample 2 (exps + tile sums -> probabilities' tile sums, the LAST workgroup to add its ticket picks);
// top-p 5 (exps, runs of the exps, [exact total -> probabilities -> sorted tiles], rank merge, pick) -- the descending order
// needs the exact probabilities, so its first half stays the exact chain of sampler_chain.hip.h.

struct MarginArgs {
  const float* exps;             // (V) fp32 exps of the scaled logits (exp_kernel)
  const double* part;            // (G) their tile sums
  int V, G;
  double* part2;                 // (G) tile sums of the probabilities (plain sample), written and read inside one launch
  double* amb;                   // (G) per tile: float spacings of its ambiguous quotients
  unsigned* ticket;              // arrivals of the launch; the last one picks and puts it back to 0
  const float* sorted;           // top-p: (V) probabilities in descending order, ids beside them, tile sums of that order
  const int* ids;
  double* part_sorted;           //   zero between tokens
  const double* params;          // {temperature, topp}
  unsigned long long* rng;
  int* tokpos;
  int* tokens_out;
  unsigned* mxkey;               // reset for the next token
  unsigned long long* amax;      // or the classifier's 8 argmax keys
  unsigned long long* stats;     // {tokens, tokens that took the serial loop}
  int force_serial;
};

struct MarginShared {
  double wsum[NWV], wamb[NWV];
  double val[4];
  int slot[4];
  __attribute__((aligned(16))) double buf[CH];      // the serial loop's staging (values widened to fp64 by all threads)
  double bound[MAX_VOCAB / SEG];                    //   and its recorded running sums
};

// inclusive scan of one value per thread over the workgroup; *total = the sum of all of them (the same bits in every thread)
__device__ __forceinline__ double block_scan(double x, double* wsum, double* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double incl = wave_scan_f64(x);
  __syncthreads();
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  double wb = 0.0;
  for (int w = 0; w < wave; ++w) wb += wsum[w];
  *total = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
  return wb + incl;
}

__device__ __forceinline__ int block_first(bool mine, int index, int* slot) {       // smallest `index` among the threads with `mine`, or INT_MAX
  __syncthreads();
  if (threadIdx.x == 0) *slot = 0x7fffffff;
  __syncthreads();
  if (mine) atomicMin(slot, index);
  __syncthreads();
  return *slot;
}

// Tree running sums of one tile of `V` values produced by value(i), in front of which sits `base`: Q[k] for the thread's IT elements,
// front = the sum in front of the thread's first element.
template <class F>
__device__ __forceinline__ void tile_sums(F value, int tile, int V, double base, double* wsum, double (&Q)[IT], double* front) {
  const int i0 = tile * TILE + threadIdx.x * IT;
  double a[IT];
#pragma unroll
  for (int k = 0; k < IT; ++k) a[k] = (k ? a[k - 1] : 0.0) + ((i0 + k < V) ? (double)value(i0 + k) : 0.0);
  double all;
  const double incl = block_scan(a[IT - 1], wsum, &all);
  *front = base + (incl - a[IT - 1]);
#pragma unroll
  for (int k = 0; k < IT; ++k) Q[k] = *front + a[k];
}

// The tile whose tree sums reach past `thr` first (the last one if none does) and the sum in front of it; incl = this thread's
// inclusive sum over the tile sums (threads >= G hold the total).
__device__ __forceinline__ int tile_of(double thr, double incl, double own, int G, int* slot, double* shv) {
  const int t = block_first((int)threadIdx.x < G && incl > thr, threadIdx.x, slot);
  const int tt = t == 0x7fffffff ? G - 1 : t;
  if ((int)threadIdx.x == tt) *shv = incl - own;
  __syncthreads();
  return tt;
}

// What the loop `for (i = 0; i < limit; ++i) if (thr < cum_i) return i; return -1` returns, decided by the margin rule; -2 when the rule
// does not decide.  *qhit = the tree sum at the first TRUE element (also when that element is >= limit and -1 is returned).
// value(i): the i-th value of the accumulation order; incl / own: this thread's inclusive scan over the tile sums and its own tile sum.
template <class F>
__device__ __forceinline__ int decide_first(F value, int V, int G, double thr, double M, int limit, double incl, double own, MarginShared& sh, double* qhit) {
  const int tid = threadIdx.x;
  if (limit <= 0) return -1;
  const int tt = tile_of(thr, incl, own, G, &sh.slot[0], &sh.val[2]);
  const double base = sh.val[2];
  double Q[IT], front;
  tile_sums(value, tt, V, base, sh.wsum, Q, &front);
  const int i0 = tt * TILE + tid * IT;
  int mine = 0x7fffffff;
  double q_mine = 0.0, q_before = 0.0;
#pragma unroll
  for (int k = IT - 1; k >= 0; --k) if (mr::known_true(Q[k], thr, M)) { mine = i0 + k; q_mine = Q[k]; q_before = k ? Q[k - 1] : front; }
  const int j = block_first(mine != 0x7fffffff, mine, &sh.slot[1]);       // the first element of the tile known TRUE
  if (tid == 0) sh.slot[2] = -2;
  __syncthreads();
  // j is the loop's first TRUE element when the element in front of it is known FALSE (cum is monotone: so is everything before that)
  if (mine == j && j < V && (j == 0 || mr::known_false(q_before, thr, M))) { sh.slot[2] = j < limit ? j : -1; sh.val[3] = q_mine; }
  __syncthreads();
  if (sh.slot[2] == -2) {
    // or the loop ends in front of anything TRUE: its last element, limit - 1, is known FALSE
    const int l = limit - 1;
#pragma unroll
    for (int k = 0; k < IT; ++k) if (i0 + k == l && mr::known_false(Q[k], thr, M)) sh.slot[2] = -1;
    if (tid == 0 && limit == tt * TILE && mr::known_false(base, thr, M)) sh.slot[2] = -1;
  }
  __syncthreads();
  *qhit = sh.val[3];
  return sh.slot[2];
}

// ---- the reference's loops as written, by ONE lane (the undecided tokens) -------------------------------------------------
// Sequential fp64 sum of value(0 .. n) in index order, staged through LDS by all threads; records the running sum every SEG elements
// when `bound`; stops at the first element whose running sum exceeds `limit` (returns its index, else -1).
template <class F>
__device__ __forceinline__ int serial_sum(F value, int n, MarginShared& sh, double* sum, bool record, double limit) {
  const int tid = threadIdx.x;
  if (tid == 0) { sh.val[0] = 0.0; sh.slot[0] = -1; }
  __syncthreads();
  for (int c0 = 0; c0 < n; c0 += CH) {
    const int m = min(CH, n - c0);
    for (int i = tid; i < m; i += TN) sh.buf[i] = (double)value(c0 + i);
    __syncthreads();
    if (tid == 0) {
      double acc = sh.val[0];
      const int at = seq_accumulate(sh.buf, m, c0, acc, record ? sh.bound : nullptr, limit);
      sh.val[0] = acc;
      if (at >= 0) sh.slot[0] = c0 + at;
    }
    __syncthreads();
    if (sh.slot[0] >= 0) break;
  }
  if (tid == 0 && record && (n & (SEG - 1)) && sh.slot[0] < 0) sh.bound[n / SEG] = sh.val[0];   // close the last, partial segment
  __syncthreads();
  *sum = sh.val[0];
  return sh.slot[0];
}

// First i < limit_idx with r < (running sum through i), from the recorded segment sums; -1 if none.
template <class F>
__device__ __forceinline__ int serial_first(F value, int limit_idx, double r, MarginShared& sh) {
  const int tid = threadIdx.x;
  if (tid == 0) {
    int seg = -1;
    for (int s = 0; s * SEG < limit_idx; ++s) {
      const bool complete = (s + 1) * SEG <= limit_idx;
      if (!complete || r < sh.bound[s]) { seg = s; break; }
    }
    sh.slot[0] = seg;
  }
  __syncthreads();
  const int seg = sh.slot[0];
  if (seg < 0) return -1;
  const int g0 = seg * SEG, n = min(SEG, limit_idx - g0);
  for (int i = tid; i < n; i += TN) sh.buf[i] = (double)value(g0 + i);
  __syncthreads();
  if (tid == 0) {
    double a = seg ? sh.bound[seg - 1] : 0.0;
    int hit = -1;
    for (int i = 0; i < n; ++i) { a += sh.buf[i]; if (r < a) { hit = g0 + i; break; } }
    sh.slot[1] = hit;
  }
  __syncthreads();
  return sh.slot[1];
}

__device__ __forceinline__ void pick_done(const MarginArgs& a, int token, bool serial) {
  const int tid = threadIdx.x;
  if (tid == 0) {
    advance(a.tokpos, a.tokens_out, token);
    *a.mxkey = 0;
    a.stats[0] += 1; if (serial) a.stats[1] += 1;
  }
  if (a.amax && tid < 8) a.amax[(size_t)tid * 16] = 0ull;
}

// ---- plain sample(): probabilities' tile sums by every workgroup, the pick by the last one to arrive ----------------------------
__global__ void __launch_bounds__(TN) sample_margin_kernel(const MarginArgs a) {
  __shared__ MarginShared sh;
  const int tid = threadIdx.x, tile = blockIdx.x, n = a.V;
  const double T = tile_base(a.part, a.G);                     // tree total of the exps: the same bits in every lane of every workgroup
  const int win = mr::window(n);
  float v[IT];
  load_tile(a.exps, n, tile, v);
  double amb = 0.0;
#pragma unroll
  for (int k = 0; k < IT; ++k) v[k] = mr::quotient_checked(v[k], T, win, &amb);      // padding: e = 0 -> p = 0
  const double t = tile_total(v, sh.wsum);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) amb += __shfl_xor(amb, off, 64);
  if ((tid & 63) == 0) sh.wamb[tid >> 6] = amb;
  __syncthreads();
  if (tid == 0) {
    // write-through stores another CU's L1-bypassing loads see, drained before the ticket is taken (MI355X_MICROARCH.md, hand-off forms)
    __hip_atomic_store(a.part2 + tile, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(a.amb + tile, (sh.wamb[0] + sh.wamb[1]) + (sh.wamb[2] + sh.wamb[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    sh.slot[3] = (int)__hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (sh.slot[3] != a.G - 1) return;
  // ---- the last workgroup: every tile's sums are in memory
  const double own = tid < a.G ? __hip_atomic_load(a.part2 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
  const double own_amb = tid < a.G ? __hip_atomic_load(a.amb + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
  double Qn, A;
  const double incl = block_scan(own, sh.wsum, &Qn);
  block_scan(own_amb, sh.wamb, &A);
  if (tid == 0) { sh.val[1] = (double)random_f32(a.rng); __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // past the L2, where the adds are made
  __syncthreads();
  const double u = sh.val[1];
  const double M = mr::margin(n, Qn, A);
  auto prob = [&](int i) { return (float)((double)a.exps[i] / T); };
  int hit = -2;
  double qhit;
  if (!a.force_serial && Qn > 0.0 && Qn <= 1.7976931348623157e308) hit = decide_first(prob, n, a.G, u * Qn, M, n, incl, own, sh, &qhit);   // randValue = random_f32() * sum (:370)
  const bool serial = hit == -2;
  if (serial) {                                                // llama2.ts:189-192, :368-376 as written
    double total, sum;
    serial_sum([&](int i) { return a.exps[i]; }, n, sh, &total, false, INFINITY);
    auto p = [&](int i) { return (float)((double)a.exps[i] / total); };
    serial_sum(p, n, sh, &sum, true, INFINITY);
    hit = serial_first(p, n, u * sum, sh);
  }
  pick_done(a, hit < 0 ? 0 : hit, serial);                     // fall-through returns 0 (:375)
}

// ---- sample_topp() behind the sort: one workgroup ------------------------------------------------------------------------------
__global__ void __launch_bounds__(TN) topp_margin_kernel(const MarginArgs a) {
  __shared__ MarginShared sh;
  const int tid = threadIdx.x, n = a.V;
  const double topp = a.params[1];
  const double own = tid < a.G ? a.part_sorted[tid] : 0.0;
  double Qn;
  const double incl = block_scan(own, sh.wsum, &Qn);
  if (tid == 0) sh.val[1] = (double)random_f32(a.rng);
  __syncthreads();
  if (tid < a.G) a.part_sorted[tid] = 0.0;
  const double u = sh.val[1];
  const double M = mr::margin(n, Qn, 0.0);
  auto sorted = [&](int i) { return a.sorted[i]; };
  int token = 0;
  bool serial = a.force_serial || !(Qn > 0.0 && Qn <= 1.7976931348623157e308);
  if (!serial) {
    // cumProb > topp (:385): `topp < cum_i` with an exact constant
    double qc;
    const int c = decide_first(sorted, n, a.G, topp, M, n, incl, own, sh, &qc);
    if (c == -2) serial = true;
    else if (c <= 0) token = 0;                                // never crossed (lastIdx stays 0, :383) or crossed by the first: the second loop is empty
    else {
      __syncthreads();
      double qh;
      const int hit = decide_first(sorted, n, a.G, u * qc, 2.0 * M, c, incl, own, sh, &qh);   // cumProb as the loop left it (:388), i < lastIdx only (:390)
      if (hit == -2) serial = true;
      else token = hit < 0 ? 0 : a.ids[hit];
    }
  }
  if (serial) {                                                // llama2.ts:382-393 as written
    double cum;
    const int at = serial_sum(sorted, n, sh, &cum, true, topp);
    const int last = at < 0 ? 0 : at;
    const int hit = serial_first(sorted, last, u * cum, sh);
    token = hit < 0 ? 0 : a.ids[hit];
  }
  pick_done(a, token, serial);
}

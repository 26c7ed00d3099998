// Host harness for llama2.ts_amd/csrc/margin_rule.h: the device sampler's margin rule against the reference's loops run as written
// (tests/test_margin_rule_cpu.py).  usage: margin_rule_host <exps.f32> <draws> <seed> <skew>
// The exps are what softmax stores before it divides (llama2.ts:187).  The harness
//   1. forms the reference's probabilities with the SEQUENTIAL total and the rule's with another total: the pairwise tree sum, pushed a
//      further `skew` x (n 2^-53) away (skew = 0: the tree sum itself; +-1: the edge of what the rule allows for), and checks that every
//      probability that came out different was seen by quotient_checked (their differences add up to at most A);
//   2. for `draws` thresholds (random 24-bit fractions, and fractions aimed at a running sum) runs sample()'s loop (llama2.ts:368-376) on
//      the reference's probabilities and the rule on tree sums of the rule's probabilities: a DECIDED index must be the loop's index;
//   3. the same for sample_topp's two loops (:382-393) on the descending order (no A: the order needs the exact probabilities), topp
//      values random and aimed at a running sum.
// prints: n mismatching_probabilities A decided undecided wrong  topp_decided topp_undecided topp_wrong  undecided_random topp_undecided_random
// (the last two: among the thresholds that were NOT aimed at a running sum)
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "../llama2.ts_amd/csrc/margin_rule.h"

static double pairwise(const double* v, long n) {
  if (n <= 8) { double s = 0.0; for (long i = 0; i < n; ++i) s += v[i]; return s; }
  return pairwise(v, n / 2) + pairwise(v + n / 2, n - n / 2);
}

// tree running sums: pairwise tile sums in front, pairs inside the tile -- an order the reference never uses
static std::vector<double> tree_prefix(const std::vector<float>& p) {
  const long n = (long)p.size(), tile = 1024;
  std::vector<double> Q(n), w(p.begin(), p.end());
  std::vector<double> parts;
  for (long t0 = 0; t0 < n; t0 += tile) parts.push_back(pairwise(w.data() + t0, std::min(tile, n - t0)));
  for (long t0 = 0, t = 0; t0 < n; t0 += tile, ++t) {
    const double base = pairwise(parts.data(), t);
    for (long i = t0; i < n && i < t0 + tile; ++i) {
      const long k = i - t0 + 1;                         // the first k of the tile: whole blocks of 4, then the rest
      double s = 0.0;
      long j = 0;
      for (; j + 4 <= k; j += 4) s += (w[t0 + j] + w[t0 + j + 1]) + (w[t0 + j + 2] + w[t0 + j + 3]);
      for (; j < k; ++j) s += w[t0 + j];
      Q[i] = base + s;
    }
  }
  return Q;
}

// the loop `for i < limit: if (thr < cum_i) return i; return -1` decided by the rule, -2 when it does not decide
static long decide(const std::vector<double>& Q, double thr, double M, long limit) {
  const long n = (long)Q.size();
  if (limit <= 0) return -1;
  long j = -1;
  for (long i = 0; i < n; ++i) if (mr::known_true(Q[i], thr, M)) { j = i; break; }
  if (j >= 0 && (j == 0 || mr::known_false(Q[j - 1], thr, M))) return j < limit ? j : -1;
  if (mr::known_false(Q[limit - 1], thr, M)) return -1;
  return -2;
}

static uint64_t rng_state;
static uint32_t next_u32() { rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27; return (uint32_t)((rng_state * 0x2545F4914F6CDD1Dull) >> 32); }

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  const int draws = atoi(argv[2]);
  rng_state = strtoull(argv[3], nullptr, 10) | 1;
  const double skew = atof(argv[4]);
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f) / 4;
  fseek(f, 0, SEEK_SET);
  std::vector<float> e(n);
  if (fread(e.data(), 4, n, f) != (size_t)n) return 2;
  fclose(f);

  // 1. the two sets of probabilities
  double Tseq = 0.0;
  for (long i = 0; i < n; ++i) Tseq += (double)e[i];
  std::vector<double> ed(e.begin(), e.end());
  const double Ttree = pairwise(ed.data(), n) * (1.0 + skew * (double)n * mr::TWO_M53);
  std::vector<float> pref(n), prule(n);
  double A = 0.0, moved = 0.0;
  long mism = 0;
  const int win = mr::window((int)n);
  for (long i = 0; i < n; ++i) {
    pref[i] = (float)((double)e[i] / Tseq);
    prule[i] = mr::quotient_checked(e[i], Ttree, win, &A);
    if (pref[i] != prule[i]) { ++mism; moved += fabs((double)pref[i] - (double)prule[i]); }
  }
  if (moved > A) { printf("FAIL probabilities moved by %g, A = %g\n", moved, A); return 1; }

  // 2. sample()
  std::vector<double> cum(n);
  { double c = 0.0; for (long i = 0; i < n; ++i) { c += (double)pref[i]; cum[i] = c; } }
  const double sum = cum[n - 1];
  const std::vector<double> Q = tree_prefix(prule);
  const double Qn = Q[n - 1], M = mr::margin((int)n, Qn, A);
  long decided = 0, undecided = 0, wrong = 0, und_random = 0, tund_random = 0;
  for (int d = 0; d < draws; ++d) {
    float u = (float)(((double)next_u32() / 256.0) / 16777216.0);
    if (d & 1) {                                       // aimed: the 24-bit fraction nearest to a running sum
      const long i = next_u32() % n;
      u = (float)(floor(cum[i] / sum * 16777216.0 + ((d & 2) ? 0.5 : 0.0)) / 16777216.0);
      if (!(u < 1.0f)) u = 0.5f;
    }
    const double r = (double)u * sum;
    long want = 0;
    { long i = 0; for (; i < n; ++i) if (r < cum[i]) break; want = i < n ? i : 0; }
    const long got = decide(Q, (double)u * Qn, M, n);
    if (got == -2) { ++undecided; if (!(d & 1)) ++und_random; } else { ++decided; if ((got < 0 ? 0 : got) != want) ++wrong; }
  }

  // 3. sample_topp() on the descending order of the reference's probabilities
  std::vector<long> order(n);
  for (long i = 0; i < n; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](long a, long b) { return pref[a] > pref[b]; });
  std::vector<float> sorted(n);
  for (long i = 0; i < n; ++i) sorted[i] = pref[order[i]];
  std::vector<double> scum(n);
  { double c = 0.0; for (long i = 0; i < n; ++i) { c += (double)sorted[i]; scum[i] = c; } }
  const std::vector<double> SQ = tree_prefix(sorted);
  const double M0 = mr::margin((int)n, SQ[n - 1], 0.0);
  long tdec = 0, tund = 0, twrong = 0;
  for (int d = 0; d < draws; ++d) {
    const float u = (float)(((double)next_u32() / 256.0) / 16777216.0);
    double topp = (double)next_u32() / 4294967296.0;
    if (d & 1) { const long i = next_u32() % n; topp = (d & 2) ? scum[i] : nextafter(scum[i], (d & 4) ? 2.0 : 0.0); }
    if (!(topp > 0.0 && topp < 1.0)) topp = 0.9;
    // the reference (:382-393)
    double c = 0.0;
    long last = 0;
    for (long i = 0; i < n; ++i) { c += (double)sorted[i]; if (c > topp) { last = i; break; } }
    const double r = (double)u * c;
    long want = 0;
    { double c2 = 0.0; for (long i = 0; i < last; ++i) { c2 += (double)sorted[i]; if (r < c2) { want = order[i]; break; } } }
    // the rule
    long got = -2;
    const long cr = decide(SQ, topp, M0, n);
    if (cr != -2) {
      if (cr <= 0) got = 0;
      else {
        const long h = decide(SQ, (double)u * SQ[cr], 2.0 * M0, cr);
        if (h != -2) got = h < 0 ? 0 : order[h];
      }
    }
    if (got == -2) { ++tund; if (!(d & 1)) ++tund_random; } else { ++tdec; if (got != want) ++twrong; }
  }
  printf("%ld %ld %.3g %ld %ld %ld %ld %ld %ld %ld %ld\n", n, mism, A, decided, undecided, wrong, tdec, tund, twrong, und_random, tund_random);
  return (wrong || twrong) ? 1 : 0;
}

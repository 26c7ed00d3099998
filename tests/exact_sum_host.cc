// Host harness for llama2.ts_amd/csrc/exact_sum.h: the run / composite / chain arithmetic of the device sampler,
// driven by plain loops instead of workgroup scans, so the CPU suite can check it against the serial fp64 loop
// (tests/test_exact_sum_cpu.py).  usage: exact_sum_host <in.f32> <out.f64> [tile] [noise_seed] [sabotage] [margin_bits]
// `noise_seed` != 0 perturbs the approximate prefix by up to 2^-40 relative (2^-25 when margin_bits = 20, the derived
// prefix of the fused normalise step), standing in for another summation order;
// `sabotage` = n falsifies the predicted grid of every n-th run, which the chain's check must catch (element-wise re-add).
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../llama2.ts_amd/csrc/exact_sum.h"

static const uint64_t TWO53_ = 1ull << 53;

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const int tile = argc > 3 ? atoi(argv[3]) : 1024;
  uint64_t noise = argc > 4 ? strtoull(argv[4], nullptr, 10) : 0;
  const int sabotage = argc > 5 ? atoi(argv[5]) : 0;
  const int mb = argc > 6 ? atoi(argv[6]) : 32;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f) / 4;
  fseek(f, 0, SEEK_SET);
  std::vector<float> x(n);
  if (fread(x.data(), 4, n, f) != (size_t)n) return 2;
  fclose(f);

  // approximate prefix: per-tile sums, then base + running value inside the tile
  std::vector<double> A(n);
  double base = 0.0;
  for (long t0 = 0; t0 < n; t0 += tile) {
    double part = 0.0, run = 0.0;
    for (long i = t0; i < n && i < t0 + tile; ++i) {
      run += (double)x[i];
      double a = base + run;
      if (noise) {
        noise ^= noise >> 12; noise ^= noise << 25; noise ^= noise >> 27;
        a *= 1.0 + ((double)(int64_t)(noise >> 40) - 8388608.0) * (mb == 32 ? 0x1p-64 : 0x1p-49);
      }
      A[i] = a;
      part = run;
    }
    base += part;
  }

  // runs
  struct Elem { bool serial; int E; };
  std::vector<Elem> el(n);
  std::vector<xs::Run> runs;
  std::vector<long> run_start;
  {
    xs::Comp acc = xs::identity();
    int accE = xs::E_NONE;
    long start = 0;
    for (long i = 0; i < n; ++i) {
      int E;
      const bool serial = xs::classify(i ? A[i - 1] : 0.0, A[i], x[i], &E, mb);
      el[i].serial = serial; el[i].E = E;
      if (!serial) { acc = xs::compose(acc, E == xs::E_NONE ? xs::identity() : xs::on_grid(x[i], E)); if (E != xs::E_NONE) accE = E; }
      const bool tile_end = (i + 1) % tile == 0 || i + 1 == n;
      if (serial || tile_end) {
        xs::Run r; r.q0 = acc.q0; r.d = acc.d; r.E = accE; r.x = serial ? x[i] : 0.0f; r.end = (int)i;
        if (sabotage && runs.size() % sabotage == 0 && (r.q0 || r.d)) { if (runs.size() % (2 * sabotage)) r.E += 1; else r.q0 += TWO53_; }
        runs.push_back(r); run_start.push_back(start);
        acc = xs::identity(); accE = xs::E_NONE; start = i + 1;
      }
    }
  }

  // chain + per-element values
  std::vector<double> out(n);
  double S = 0.0;
  long bad = 0;
  for (size_t k = 0; k < runs.size(); ++k) {
    bool ok;
    const double S2 = xs::chain_step(S, runs[k], &ok);
    if (!ok) {
      ++bad;
      for (long i = run_start[k]; i <= runs[k].end; ++i) { S += (double)x[i]; out[i] = S; }
      continue;
    }
    xs::Comp acc = xs::identity();
    for (long i = run_start[k]; i <= runs[k].end; ++i) {
      if (el[i].serial) { out[i] = S2; break; }
      acc = xs::compose(acc, el[i].E == xs::E_NONE ? xs::identity() : xs::on_grid(x[i], el[i].E));
      out[i] = xs::value_at(S, acc, runs[k].E);
    }
    S = S2;
  }
  f = fopen(argv[2], "wb");
  fwrite(out.data(), 8, n, f);
  fclose(f);
  long nserial = 0;
  for (long i = 0; i < n; ++i) nserial += el[i].serial;
  printf("%ld elements, %zu runs, %ld serial, %ld failed checks\n", n, runs.size(), nserial, bad);
  return 0;
}

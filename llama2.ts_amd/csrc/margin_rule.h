// The arithmetic of the device sampler's margin rule (sampler_margin.hip.h), kept free of HIP so that the same functions are compiled
// into the kernels and into the host harness tests/margin_rule_host.cc.
//
// Setting: n non-negative fp32 values; the reference adds them one by one in fp64 (llama2.ts:189, :369-373, :384-391) and returns the
// first index whose running sum cum_i passes a threshold.  Any summation order of n non-negative values is within n 2^-53 (relative) of
// the true sum, so a tree sum Q_i and the reference's cum_i differ by at most 2 n 2^-53 of the total -- plus, for plain sample(), what
// the probabilities themselves can differ by when they were formed with a tree total instead of the sequential one (`A` below).
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define MR_HD __host__ __device__ __forceinline__
#else
#define MR_HD inline
#endif

namespace mr {

constexpr double TWO_M53 = 1.1102230246251565e-16;     // 2^-53

MR_HD uint64_t bits_of(double v) { uint64_t u; memcpy(&u, &v, 8); return u; }
MR_HD double double_of(uint64_t u) { double v; memcpy(&v, &u, 8); return v; }

// How far (in steps of the last place of an fp64 quotient e / T) the quotient by the reference's sequential total can sit from the
// quotient by a tree total of the same n values: the totals differ by <= 2 (n + 64) 2^-53 relative (64: the tree's own depth, generous),
// each division rounds once, one step of a quotient q is >= q 2^-53: 2 (n + 64) + 2 steps.  Twice that.
MR_HD int window(int n) { return 4 * (n + 64) + 8; }

// p = fl32(fl64(e / T)) as softmax stores it (llama2.ts:192), and into *amb the spacing of the floats around p when a total within
// `win` steps (window()) could have rounded the quotient to the neighbouring float: the 29 bits a float drops then sit within `win` of
// their midpoint 2^28.  Quotients below 2^-100 (float subnormals and their neighbourhood, spacing <= 2^-123) count 2^-99 unseen.
MR_HD float quotient_checked(float e, double T, int win, double* amb) {
  const double q = (double)e / T;
  if (e != 0.0f) {
    const uint64_t b = bits_of(q);
    const int ex = (int)(b >> 52) & 0x7ff, low = (int)(b & 0x1fffffffu);
    const int off = low - 0x10000000;
    if (ex < 1023 - 100) *amb += 0x1p-99;
    else if ((off < 0 ? -off : off) <= win) *amb += double_of((uint64_t)(ex - 22) << 52);   // 2 ^ (binade of q - 22): the float spacing of the binade above
  }
  return (float)q;
}

// Half-width of the undecided band around a threshold: sums known to Qn's 2 (n + 64) 2^-53 on either side of the comparison, the
// threshold itself a product with another such sum, one rounding of that product; A: quotient_checked's total.  Twice what that needs.
MR_HD double margin(int n, double Qn, double A) { return 8.0 * (double)(n + 64) * TWO_M53 * Qn + 4.0 * A; }

MR_HD bool known_true(double Q, double thr, double M) { return Q > thr + M; }      // then thr < cum
MR_HD bool known_false(double Q, double thr, double M) { return Q < thr - M; }     // then !(thr < cum)

}  // namespace mr

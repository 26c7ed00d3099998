// The arithmetic behind the device sampler's running sums (sampler.hip), kept free of HIP so that the same functions
// are compiled into the kernels and into the host harness tests/exact_sum_host.cc.
//
// The reference accumulates its probabilities one by one in fp64 (`cumProb += prob`, llama2.ts:189, :369-373,
// :384-391) and compares a threshold against the running value, so the sampled token depends on every rounding of
//     S_i = fl(S_{i-1} + x_i),   x_i >= 0 fp32,  S fp64.
// While the exponent E of S stays put, S = M * g with g = 2^(E-52) and 2^52 <= M < 2^53, and the rounded add is the
// integer add M += rn(x / g) -- associative -- except for exact ties (x / g = k + 1/2), which go to the even
// neighbour and so depend on the parity of M in front of them.  A stretch of elements on one grid is therefore a
// function "parity of M at its start -> integer increment", and two such functions compose:
//     Comp {q0, d}: the increment is q0 for even M, q0 + d for odd M, d in {-1, 0, +1}.
// Which grid an element sits on is predicted from an APPROXIMATE prefix sum A (any parallel summation order): the
// exact S differs from A by < 2^-34 relative (n <= 2^18 roundings of 2^-53), so whenever [A_{i-1}, A_i] lies inside one binade with a 2^-32 margin the
// exponent of S_{i-1} is known and the add cannot leave the binade ("regular" element; classify()'s `mb` widens the margin
// for a cruder A).  Every other element --
// a power of two crossed, reached or nearly reached, the first non-zero value -- is "serial": it ends a run, and one
// lane adds it the ordinary way when it walks the runs in order (chain_step).  That walk re-derives M and E from the
// exact S at the start of each run and CHECKS them against the prediction (exponent equal, no carry out of 2^53); a
// run that fails the check is re-added element by element, so the result never rests on the error analysis above.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define XS_HD __host__ __device__ __forceinline__
#else
#define XS_HD inline
#endif

namespace xs {

constexpr int E_NONE = -32768;                 // "no grid": the composite is the identity
constexpr uint64_t TWO52 = 1ull << 52, TWO53 = 1ull << 53;

XS_HD uint64_t bits_of(double v) { uint64_t u; memcpy(&u, &v, 8); return u; }
XS_HD double double_of(uint64_t u) { double v; memcpy(&v, &u, 8); return v; }
XS_HD uint32_t bits_of(float v) { uint32_t u; memcpy(&u, &v, 4); return u; }
XS_HD int expo(double a) { return (int)((bits_of(a) >> 52) & 0x7ff) - 1023; }
XS_HD double grid(int E) { return double_of((uint64_t)(E - 52 + 1023) << 52); }

struct Comp { uint64_t q0; int d; };           // increment: q0 (even M in front) / q0 + d (odd M)

XS_HD Comp identity() { Comp c; c.q0 = 0; c.d = 0; return c; }
XS_HD uint64_t apply(const Comp& c, uint64_t M) { return M + c.q0 + (uint64_t)(int64_t)((M & 1) ? c.d : 0); }

// `a` first, then `b`.  With p the parity behind `a` for an even M in front of it: if a.d = 0 an odd M flips p, so
// b's parity term changes sign; if a.d != 0 both cases leave the same parity behind `a` and only a.d survives.
// (No overflow: an element is at most 2^53, a tile has 1024 of them, and the chain rejects any run total >= 2^53.)
XS_HD Comp compose(const Comp& a, const Comp& b) {
  const bool p = (a.q0 & 1) != 0;
  Comp c;
  c.q0 = a.q0 + b.q0 + (uint64_t)(int64_t)(p ? b.d : 0);
  c.d = a.d ? a.d : (p ? -b.d : b.d);
  return c;
}

// One non-negative fp32 value on the grid 2^(E-52).
XS_HD Comp on_grid(float x, int E) {
  const uint32_t xb = bits_of(x);
  const int ef = (int)((xb >> 23) & 0xff);
  const uint32_t m = (xb & 0x7fffffu) | (ef ? 0x800000u : 0u);
  Comp c = identity();
  if (m == 0) return c;
  const int shift = (ef ? ef - 127 : -126) - 23 - (E - 52);
  if (shift >= 0) { c.q0 = shift > 29 ? TWO53 : ((uint64_t)m << shift); return c; }   // > 29: not on this grid, the chain rejects the run
  if (shift <= -26) return c;                                                // below half a grid step: S + x = S
  const int sft = -shift;
  const uint32_t rem = m & ((1u << sft) - 1u), half = 1u << (sft - 1);
  const uint64_t k = (uint64_t)(m >> sft);
  if (rem != half) { c.q0 = k + (rem > half ? 1u : 0u); return c; }
  c.q0 = k + (k & 1);                                                         // tie: to the even neighbour of M + k + 1/2
  c.d = (int)((k + 1) & 1) - (int)(k & 1);
  return c;
}

// Binade of a(1 - 2^-mb) and of a(1 + 2^-mb), a > 0 finite.  mb: how good the approximate prefix is -- 32 when it is a
// sum of the very values being accumulated, 20 when it was derived (tile sums of the exps divided by their total
// standing in for the tile sums of the fp32-rounded quotients: 2^-24 relative).
XS_HD int low_e(double a, int mb) { return expo(a) - ((bits_of(a) & (TWO52 - 1)) < (1ull << (52 - mb)) ? 1 : 0); }
XS_HD int high_e(double a, int mb) { return expo(a) + ((bits_of(a) & (TWO52 - 1)) >= TWO52 - (1ull << (53 - mb)) ? 1 : 0); }

// Element x with approximate prefix `aprev` in front of it and `acur` including it.  Returns true for a serial element;
// otherwise *E is its grid (E_NONE for a zero, which changes nothing on any grid -- also the padding of the last tile).
XS_HD bool classify(double aprev, double acur, float x, int* E, int mb = 32) {
  *E = E_NONE;
  if (x == 0.0f) return false;                                                // S + 0 = S
  if (aprev == 0.0) return true;                                              // first non-zero value: 0 + x = x
  if (!(acur <= 1.7976931348623157e308)) return true;                          // inf / nan: ordinary adds all the way
  const int lo = low_e(aprev, mb), hi = high_e(acur, mb);
  if (lo != hi) return true;
  *E = lo;
  return false;
}

struct Run {                 // maximal stretch of regular elements inside one tile + the serial element that ends it
  uint64_t q0;               // composite of the regular elements
  int d;
  int E;                     // their grid (E_NONE: identity)
  float x;                   // the serial element (+0 when the run ends with the tile instead)
  int end;                   // global index of the last element of the run
};

// S in front of the run -> S after it.  *ok = false: the prediction did not hold, add the run's elements one by one.
XS_HD double chain_step(double S, const Run& r, bool* ok) {
  *ok = true;
  if (r.q0 != 0 || r.d != 0) {
    const uint64_t sb = bits_of(S);
    Comp c; c.q0 = r.q0; c.d = r.d;
    const uint64_t M2 = apply(c, (sb & (TWO52 - 1)) | TWO52);
    // S > 0 has a clear sign bit, so sb >> 52 is its biased exponent (0 for S = 0, which matches no grid)
    if ((int64_t)(sb >> 52) != (int64_t)r.E + 1023 || r.q0 >= TWO53 || M2 >= TWO53) { *ok = false; return S; }
    S = double_of(((uint64_t)(r.E + 1023) << 52) | (M2 & (TWO52 - 1)));       // M2 * 2^(E-52), 2^52 <= M2 < 2^53
  }
  return S + (double)r.x;
}

// S after a regular element whose inclusive composite since the start of its run is `c`; Sstart is the exact sum in
// front of the run (the run passed chain_step's check, so the value stays inside the binade E).
XS_HD double value_at(double Sstart, const Comp& c, int E) {
  if (c.q0 == 0 && c.d == 0) return Sstart;
  const uint64_t M2 = apply(c, (bits_of(Sstart) & (TWO52 - 1)) | TWO52);
  return double_of(((uint64_t)(E + 1023) << 52) | (M2 & (TWO52 - 1)));
}

}  // namespace xs

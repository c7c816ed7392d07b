// Poseidon2 t=3 permutation over BN254 Fr on gfx950, one state per lane, state in Montgomery form.
//
// Follows the reference's in-tree specification:
//   reference/haskell/src/Poseidon2/Permutation.hs:14-45   (permutation, rounds, linear layers)
//   reference/haskell/src/Poseidon2/RoundConsts.hs:30-128  (constants; table generated into p2_consts_dev.inc)
//   circuit/poseidon2/poseidon2_perm.circom:10-198         (same structure, consumer side)
//
// Round constants sit in __constant__ memory and are fetched with scalar loads (the round index is
// wave-uniform), so they cost no VGPRs and no LDS traffic; the lazy-reduction table (fr::QTab) is in LDS.
#pragma once
#include "fr_gfx950.hpp"

namespace p2 {
using fr::Fe;

struct State {
  Fe x, y, z;
};

// The S-box (fr_gfx950.hpp) keeps its output below 12.7 N (unmasked quotient digits) or 2.4 N (masked) for inputs
// below 60 N (top limbs against multiples of N >> 232 = 0x30644e.7): asserted in the host check build.
template <bool MASKM = false>
__device__ __forceinline__ Fe sbox_checked(const Fe& x) {
  CP2_BOUND(x.l[fr::NL - 1] < 60u * 3171407u, "sbox input >= 60N");
  Fe r = fr::sbox<MASKM>(x);
  CP2_BOUND(r.l[fr::NL - 1] < (MASKM ? 7611377u : 40276869u), "sbox output >= 2.4N / 12.7N");
  return r;
}

__device__ __forceinline__ Fe rc(int idx) {
  Fe r;
#pragma unroll
  for (int i = 0; i < fr::NL; ++i) r.l[i] = fr::P2_RC_MONT[idx][i];
  return r;
}

// Permutation.hs:28-33.  in: limbs < U+8, values < 51 N;  out: limbs < U+8, values < 4 * 12.7 N = 51 N, or
// < 4 * 2.4 N = 9.6 N when MASKM (the last round of each group of four: what it hands to the internal rounds and to the
// caller must be small, because the next step multiplies it by up to 4 again before any S-box sees it).
template <bool MASKM>
__device__ __forceinline__ void external_round(State& s, int rc_base) {
  Fe x = sbox_checked<MASKM>(fr::add_lazy(s.x, rc(rc_base + 0)));
  Fe y = sbox_checked<MASKM>(fr::add_lazy(s.y, rc(rc_base + 1)));
  Fe z = sbox_checked<MASKM>(fr::add_lazy(s.z, rc(rc_base + 2)));
  Fe sum = fr::add_lazy(fr::add_lazy(x, y), z);
  s.x = fr::norm(fr::add_lazy(x, sum));
  s.y = fr::norm(fr::add_lazy(y, sum));
  s.z = fr::norm(fr::add_lazy(z, sum));
}

// Permutation.hs:19-26.  Only x passes through an S-box in the 56 internal rounds; y and z live in five 58-bit limbs
// (fr::Wide) for all of them and are reduced once per PAIR of rounds, between its two halves:
//   half A:  x' = sbox(xin);  S = x'+Y+Z;  Y1 = Y+S;  Z1 = 2Z+S;  xin <- x'+S+c
//   reduce:  Y1, Z1 -> below 2N, limbs below W = 2^58
//   half B:  the same on (xin, Y1r, Z1r), no reduction after it
// in : xin = S-box input INCLUDING its round constant, limbs < 2U + 64 (inside mont_sqr's 2.47U)
// Worst-case bounds (reduce_wide leaves < 2N; q is the exact quotient or one less; S-box output < f(input) with
// f(I) = ((I^2/R + 8.01N)^2/R + 8.01N) * I/R + 8.01N, R = 169.3 N; W = 2^58).  Write a, b for the S-box outputs of halves A, B:
//   half B:  S = b + Y1r + Z1r < b + 4N,  Y2 < b + 6N (limbs < 4W),  Z2 < b + 8N (< 5W),  xin_A = 2b + 5N
//   half A:  S < a + 2b + 14N (< 10W),  Y1 < a + 3b + 20N (< 14W),  Z1 < a + 4b + 30N (< 20W < 2^63),  xin_B = 2a + 2b + 15N,  T < 11W
//   fixed point of (a, b) = (f(2b + 5N), f(2a + 2b + 15N)):  a < 9.7N, b < 12.6N, so xin_A < 30.2N, xin_B < 59.6N (< 60N),
//   Y1 < 67.5N, Z1 < 90.1N: q <= 90 < 96 table rows and every value stays below R = 169N (nine limbs).
//   First pair: x, Y, Z < 9.6N from the masked fourth external round (xin < 10.6N), inside the figures above (b + 8N with b = 12.6N).
__device__ __forceinline__ void wide_half_round(fr::Fe& xin, fr::Wide& Y, fr::Wide& Z, const uint64_t (&rc_next)[fr::NW]) {
  using namespace fr;
  const Fe x = sbox_checked(xin);                                   // x' (normalised: limbs 0..7 < U)
  const uint32_t two29 = FR_TWO29;
  Wide S;
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    uint64_t s = Y.w[j] + Z.w[j];
    s += x.l[2 * j];
    if (2 * j + 1 < NL) s += (uint64_t)x.l[2 * j + 1] * two29;      // one v_mad_u64_u32
    S.w[j] = s;
  }
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    CP2_BOUND(Y.w[j] < 4 * ((uint64_t)1 << 58) + 64 && Z.w[j] < 5 * ((uint64_t)1 << 58) + 64 && S.w[j] < 10 * ((uint64_t)1 << 58) + 64,
              "wide limb beyond its documented bound");
    Y.w[j] += S.w[j];                                                // x' + 2y + z
    Z.w[j] = (Z.w[j] << 1) + S.w[j];                                 // x' + y + 3z
  }
  // next S-box input: x' + S + rc_next, cut into 29-bit pieces
  uint64_t T[NW];
#pragma unroll
  for (int j = 0; j < NW; ++j) T[j] = S.w[j] + rc_next[j];
  Fe r;
#pragma unroll
  for (int j = 0; j < NW - 1; ++j) {
    uint32_t lo = (uint32_t)T[j] & MASK;
    uint32_t mid = (uint32_t)(T[j] >> 29) & MASK;
    r.l[2 * j] = lo + x.l[2 * j] + (j ? (uint32_t)(T[j - 1] >> 58) : 0u);
    r.l[2 * j + 1] = mid + x.l[2 * j + 1];
  }
  r.l[NL - 1] = (uint32_t)T[NW - 1] + x.l[NL - 1] + (uint32_t)(T[NW - 2] >> 58);
#pragma unroll
  for (int i = 0; i < NL - 1; ++i) CP2_BOUND(r.l[i] < 2 * U29 + 64, "internal round: S-box input limb >= 2U + 64");
  xin = r;
}

__device__ __forceinline__ void internal_round_pair(fr::Fe& xin, fr::Wide& Y, fr::Wide& Z, int r, const fr::QTab& qtab) {
  wide_half_round(xin, Y, Z, fr::P2_RCW_MONT[r + 1]);
  Y = fr::reduce_wide(Y, qtab);
  Z = fr::reduce_wide(Z, qtab);
  wide_half_round(xin, Y, Z, fr::P2_RCW_MONT[r + 2]);
}

// Permutation.hs:40-45.  in: limbs < U+16, values < 12N (a state this function returned, plus an absorbed element < 2N);
// out: limbs < U+8, values < 9.6 N.  The linear layer below then feeds the first S-boxes at most 4 * 12 + 1 = 49 N.
__device__ __forceinline__ void permute(State& s, const fr::QTab& qtab) {
  {  // linearLayer, Permutation.hs:35-36
    Fe sum = fr::add_lazy(fr::add_lazy(s.x, s.y), s.z);
    s.x = fr::norm(fr::add_lazy(s.x, sum));
    s.y = fr::norm(fr::add_lazy(s.y, sum));
    s.z = fr::norm(fr::add_lazy(s.z, sum));
  }
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
#pragma unroll 1
    for (int r = 0; r < 3; ++r) external_round<false>(s, (half ? 68 : 0) + 3 * r);
    external_round<true>(s, (half ? 68 : 0) + 9);
    if (half == 0) {
      fr::Wide Y = fr::to_wide(s.y), Z = fr::to_wide(s.z);
      Fe xin = fr::add_lazy(s.x, rc(12));                           // limbs < 2U + 8
#pragma unroll 1
      for (int r = 0; r < 56; r += 2) internal_round_pair(xin, Y, Z, r, qtab);
      s.x = fr::norm(xin);                                           // 2x' + y + z of the last round (its constant row is zero)
      s.y = fr::from_wide(fr::reduce_wide(Y, qtab));                 // back to nine normalised limbs, values < 2N
      s.z = fr::from_wide(fr::reduce_wide(Z, qtab));
    }
  }
}

}  // namespace p2

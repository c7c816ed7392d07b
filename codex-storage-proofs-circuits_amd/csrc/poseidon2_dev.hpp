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

// the S-box keeps its output below 1.3 N for inputs below 30 N (top limb < 1.3 * 0x30644f): asserted in the
// host check build.  (Inputs reach ~24 N in the first external round of a sponge step: state < 4.3 N plus an
// absorbed element < 2 N per lane, times 4 in the linear layer, plus the round constant.)
__device__ __forceinline__ Fe sbox_checked(const Fe& x) {
  Fe r = fr::sbox(x);
  CP2_BOUND(r.l[fr::NL - 1] < 4122830u, "sbox output >= 1.3N");
  return r;
}

__device__ __forceinline__ Fe rc(int idx) {
  Fe r;
#pragma unroll
  for (int i = 0; i < fr::NL; ++i) r.l[i] = fr::P2_RC_MONT[idx][i];
  return r;
}

// Permutation.hs:28-33.  in: limbs < U+8;  out: limbs < U+8, values < 4.3 N
__device__ __forceinline__ void external_round(State& s, int rc_base) {
  Fe x = sbox_checked(fr::add_lazy(s.x, rc(rc_base + 0)));
  Fe y = sbox_checked(fr::add_lazy(s.y, rc(rc_base + 1)));
  Fe z = sbox_checked(fr::add_lazy(s.z, rc(rc_base + 2)));
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
// Worst-case bounds (reduce_wide leaves < 2N; q is the exact quotient or one less):
//   entering a pair  Y < 7.1N (limbs < 4W),  Z < 9.1N (< 5W)   [first pair: < 4.3N, limbs < W, from the external rounds]
//   half A:  S < 17.3N (< 10W),  Y1 < 24.4N (< 14W),  Z1 < 35.5N (< 20W < 2^63),  xin < 19.4N,  T = S + c < 11W
//   half B:  S < 5.1N (< 3W),   Y2 < 7.1N (< 4W),   Z2 < 9.1N (< 5W),   xin < 7.2N
//   S-box inputs stay below 30N, where its output is below 1.3N (asserted); q <= 35 < 64 table rows; 35.5N < R = 169N.
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

// Permutation.hs:40-45.  in: limbs < U+16, values < 8N;  out: limbs < U+8, values < 4.3 N
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
    for (int r = 0; r < 4; ++r) external_round(s, (half ? 68 : 0) + 3 * r);
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

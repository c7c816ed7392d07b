// Poseidon2 t=3 permutation over BN254 Fr on gfx950, one state per lane, state in Montgomery form.
//
// Follows the reference's in-tree specification:
//   reference/haskell/src/Poseidon2/Permutation.hs:14-45   (permutation, rounds, linear layers)
//   reference/haskell/src/Poseidon2/RoundConsts.hs:30-128  (constants; table generated into p2_consts_dev.inc)
//   circuit/poseidon2/poseidon2_perm.circom:10-198         (same structure, consumer side)
//
// Round constants sit in __constant__ memory and are fetched with scalar loads (the round index is
// wave-uniform), so they cost no VGPRs and no LDS traffic; the lazy-reduction table is in LDS.
#pragma once
#include "fr_gfx950.hpp"

namespace p2 {
using fr::Fe;

struct State {
  Fe x, y, z;
};

// the S-box keeps its output below 1.1 N (top limb < 1.1 * 0x30644f): asserted in the host check build
__device__ __forceinline__ Fe sbox_checked(const Fe& x) {
  Fe r = fr::sbox(x);
  CP2_BOUND(r.l[fr::NL - 1] < 3488548u, "sbox output >= 1.1N");
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

// Permutation.hs:19-26.  y and z never pass through an S-box in these 56 rounds, so they are
// brought back below 2N every round by reduce_lazy (values grow ~4x per round otherwise).
__device__ __forceinline__ void internal_round(State& s, int rc_idx, const uint32_t* qtab) {
  Fe x = sbox_checked(fr::add_lazy(s.x, rc(rc_idx)));
  Fe sum = fr::add_lazy(fr::add_lazy(x, s.y), s.z);              // x' + y + z        limbs < 3U
  s.x = fr::norm(fr::add_lazy(x, sum));                           // 2x' + y + z       < 6.2 N
  s.y = fr::reduce_lazy(fr::add_lazy(s.y, sum), qtab);            // x' + 2y + z       < 7.1 N -> < 2N
  s.z = fr::reduce_lazy(fr::add_lazy(fr::add_lazy(s.z, s.z), sum), qtab);  // x' + y + 3z < 9.1 N -> < 2N
}

// Permutation.hs:40-45.  in: limbs < U+16, values < 8N;  out: limbs < U+8, values < 4.3 N
__device__ __forceinline__ void permute(State& s, const uint32_t* qtab) {
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
#pragma unroll 1
      for (int r = 0; r < 56; ++r) internal_round(s, 12 + r, qtab);
    }
  }
}

}  // namespace p2

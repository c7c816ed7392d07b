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

// Permutation.hs:19-26.  y and z never pass through an S-box in these 56 rounds and grow ~4x per round, so
// they are brought back below 2N by reduce_lazy every OTHER round (REDUCE = true); in between one parallel
// carry step is enough (measured -1 % kernel time, bounds machine-checked by tests/host_check); bounds over a (reduce, norm-only) pair, starting from
// y, z < 4.3N (first pair) or < 1.01N (later):
//   reduce round:      y1 = x'+2y+z < 14N,  z1 = x'+y+3z < 18.3N  -> both < 1.01N after reduce_lazy (q < 32)
//   norm-only round:   x_in = 2x'+y1+z1+c < 5.2N;  y2 < 4.1N, z2 < 5.1N, limbs < U+8
//   next reduce round: x_in = 2x'+y2+z2+c < 12.3N (< 13N, so the S-box output stays < 1.1N);
//                      y3 < 14.4N, z3 < 20.5N -> reduced again.
template <bool REDUCE>
__device__ __forceinline__ void internal_round(State& s, int rc_idx, const uint32_t* qtab) {
  Fe x = sbox_checked(fr::add_lazy(s.x, rc(rc_idx)));
  Fe sum = fr::add_lazy(fr::add_lazy(x, s.y), s.z);              // x' + y + z        limbs < 3U+16
  s.x = fr::norm(fr::add_lazy(x, sum));                           // 2x' + y + z
  Fe y = fr::add_lazy(s.y, sum);                                  // x' + 2y + z       limbs < 4U+24
  Fe z = fr::add_lazy(fr::add_lazy(s.z, s.z), sum);               // x' + y + 3z       limbs < 5U+32
  if constexpr (REDUCE) {
    s.y = fr::reduce_lazy(y, qtab);
    s.z = fr::reduce_lazy(z, qtab);
  } else {
    s.y = fr::norm(y);
    s.z = fr::norm(z);
  }
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
      for (int r = 0; r < 56; r += 2) {
        internal_round<true>(s, 12 + r, qtab);
        internal_round<false>(s, 13 + r, qtab);
      }
    }
  }
}

}  // namespace p2

// BN254 scalar-field (Fr) arithmetic for gfx950 (MI355X), one field element per lane.
//
// Representation: 9 limbs x 29 bits in 32-bit VGPRs ("redundant radix"), Montgomery radix R = 2^261.
// Why this and not 8 x 32-bit limbs: tools/ubench_valu.hip measured v_mad_u64_u32 on gfx950 at about the
// same issue cost as any other VOP3 instruction (it is NOT quarter rate, and v_fma_f64 is no faster), so
// the kernel is bound by the NUMBER of VALU instructions.  With 29-bit limbs a whole column of partial
// products (9 of a*b plus 9 of m*N) fits a 64-bit accumulator, so every partial product is exactly one
// v_mad_u64_u32 with the accumulator as both addend and destination: no carry flags, no v_addc, no moves.
// Saturated 32-bit limbs need a third accumulator word and one v_addc per product (2 instructions each).
//
// Bounds (U = 2^29, N = field modulus, R = 2^261 ~ 169.3 N):
//   * a limb vector is "normalized" when limbs 0..7 < U + 8 (and the top limb is small);
//   * mont_mul / mont_sqr with masked quotient digits accept limbs up to 2.47 U on either operand (column sum
//       2^35 + 9*La*Lb + 9*U*U < 2^64 needs La*Lb < 6.1 U^2) and produce limbs 0..7 < U; value bound out < A*B/R + N;
//   * with UNMASKED quotient digits (the S-box): the m*N part of a column is below 2^32 * 3.4005 U, operands up to
//       2.02 U on a squaring, La*Lb < 4.09 U^2 on a product; value bound out < A*B/R + 8.01 N;
//   * either way no conditional subtraction is ever needed inside a chain (lazy Montgomery): R = 169 N contracts
//     x -> x^2/R + 8 N for everything below 160 N, and the S-box sees inputs below 60 N (poseidon2_dev.hpp);
//   * y and z of the internal rounds are held in five 58-bit limbs (struct Wide) with 6 spare bits each.
//
// What the reference computes with this arithmetic: Permutation.hs:14-45 (field ops of
// zikkurat-algebra / constantine there).  Nothing here is derived from those libraries' code.
#pragma once
#include <stdint.h>
#include <utility>

// CP2_HOST_CHECK: the same source compiled for the HOST by tests/host_check (g++ with sanitizers): the 64-bit
// column accumulator becomes a 128-bit shadow that traps on overflow, and every documented limb bound is
// asserted.  This is how the lazy-reduction bounds below are validated off the GPU; the product build never
// defines it.
#ifdef CP2_HOST_CHECK
#include <cstdio>
#include <cstdlib>
#define __device__
#define __forceinline__ inline
#define __constant__ static const
static inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
#define CP2_BOUND(cond, what)                                                       \
  do {                                                                              \
    if (!(cond)) { std::fprintf(stderr, "BOUND VIOLATION: %s (%s:%d)\n", what, __FILE__, __LINE__); std::abort(); } \
  } while (0)
#else
#include <hip/hip_runtime.h>
#define CP2_BOUND(cond, what) ((void)0)
#endif

namespace fr {

#ifdef CP2_HOST_CHECK
struct acc_t {   // 64-bit accumulator with a 128-bit shadow: any carry out of bit 63 aborts
  unsigned __int128 v = 0;
  acc_t() = default;
  acc_t(uint64_t x) : v(x) {}
  acc_t& operator+=(uint64_t x) { v += x; CP2_BOUND((v >> 64) == 0, "column accumulator overflowed 64 bits"); return *this; }
  acc_t& operator>>=(int s) { v >>= s; return *this; }
  explicit operator uint32_t() const { return (uint32_t)v; }
  explicit operator uint64_t() const { return (uint64_t)v; }
};
#else
using acc_t = uint64_t;
#endif

#include "p2_consts_dev.inc"

constexpr int NL = 9;
constexpr uint32_t U29 = 1u << 29;
constexpr uint32_t MASK = U29 - 1;

struct Fe {
  uint32_t l[NL];
};

__device__ __forceinline__ Fe fe_const(const uint32_t (&c)[9]) {
  Fe r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = c[i];
  return r;
}

__device__ __forceinline__ Fe fe_zero() {
  Fe r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = 0;
  return r;
}

// limb-wise lazy add (no carry propagation): limb bounds add
__device__ __forceinline__ Fe add_lazy(const Fe& a, const Fe& b) {
  Fe r;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    CP2_BOUND((uint64_t)a.l[i] + b.l[i] < ((uint64_t)1 << 32), "add_lazy: limb overflowed 32 bits");
    r.l[i] = a.l[i] + b.l[i];
  }
  return r;
}

// one parallel carry step: limbs (< 2^32) -> limbs 0..7 < U + 8, value unchanged
__device__ __forceinline__ Fe norm(const Fe& a) {
  Fe r;
  r.l[0] = a.l[0] & MASK;
#pragma unroll
  for (int i = 1; i < NL - 1; ++i) r.l[i] = (a.l[i] & MASK) + (a.l[i - 1] >> 29);
  r.l[NL - 1] = a.l[NL - 1] + (a.l[NL - 2] >> 29);
  return r;
}

// full sequential carry propagation: limbs 0..7 < U exactly
__device__ __forceinline__ Fe norm_full(const Fe& a) {
  Fe r;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < NL - 1; ++i) {
    uint32_t t = a.l[i] + c;
    r.l[i] = t & MASK;
    c = t >> 29;
  }
  r.l[NL - 1] = a.l[NL - 1] + c;
  return r;
}

#ifdef CP2_HOST_CHECK
template <typename... T> inline void tie(acc_t&, T&...) {}
#else
#include "fr_tie.inc"
#endif

// Experiment hook (tools/pad_probe.py; never defined in a product build): CP2_PAD_ASM is one instruction on the dummy
// operands %0, %1, %3 (32-bit VGPRs) and %2, %4 (64-bit VGPR pairs); it is issued CP2_PAD_N times after every column of every
// multiplication.  The slowdown per added instruction is that instruction's MARGINAL cost inside this kernel, which is
// what decides whether trading one instruction for another pays.
#if defined(CP2_PAD_ASM) && !defined(CP2_HOST_CHECK)
#ifndef CP2_PAD_N
#define CP2_PAD_N 2
#endif
#define CP2_PAD_DECL uint32_t pad0__ = a_in.l[0], pad1__ = a_in.l[1], pad3__ = a_in.l[3]; uint64_t pad2__ = a_in.l[2], pad4__ = a_in.l[4];
#define CP2_PAD()                                                                                         \
  do {                                                                                                    \
    _Pragma("unroll") for (int p__ = 0; p__ < CP2_PAD_N; ++p__)                                           \
        asm volatile(CP2_PAD_ASM : "+v"(pad0__), "+v"(pad1__), "+v"(pad2__), "+v"(pad3__), "+v"(pad4__) : : "vcc");  \
  } while (0)
#else
#define CP2_PAD_DECL
#define CP2_PAD() ((void)0)
#endif

template <int A0, int M0, int... I>
__device__ __forceinline__ void tie_cols(acc_t& acc, uint32_t (&a)[NL], uint32_t (&m)[NL], std::integer_sequence<int, I...>,
                                         std::integer_sequence<int>) {
  tie(acc, a[A0 + I]...);
}
template <int A0, int M0, int... I, int J0, int... J>
__device__ __forceinline__ void tie_cols(acc_t& acc, uint32_t (&a)[NL], uint32_t (&m)[NL], std::integer_sequence<int, I...>,
                                         std::integer_sequence<int, J0, J...>) {
  tie(acc, a[A0 + I]..., m[M0 + J0], m[M0 + J]...);
}

// Montgomery product a*b/R (finely integrated product scanning, 17 columns).
// MASKM = false: the quotient digits m_k are used as the full 32-bit products t_k * N' instead of their low 29 bits.
// They are still correct modulo 2^29 (the column still clears), the extra bits only add multiples of N further up, so
// the result is congruent and every output limb is still exact; what changes is its SIZE, out < A*B/R + 8.01 N instead of
// + N, and the column budget: sum_j m_k N_j < 2^32 * (N_0 + .. + N_8) = 2^32 * 3.4005 U = 2^62.77, which leaves
// 2^63.2 for the a*b products: limbs up to 2.02 U on a squaring, La * Lb < 4.09 U^2 on a product.  One v_and_b32 less
// per low column: 27 instructions per S-box, the only user (its values are re-bounded in poseidon2_dev.hpp).
template <bool MASKM = true>
__device__ __forceinline__ Fe mont_mul(const Fe& a_in, const Fe& b) {
  Fe r;
  uint32_t a[NL], m[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) a[i] = a_in.l[i];
#pragma unroll
  for (int i = 0; i < NL; ++i)   // La * Lb < 6.1 U^2 (column sum < 2^64); the 128-bit shadow accumulator of the host check is the real guard
    CP2_BOUND((uint64_t)a_in.l[i] < ((uint64_t)5 << 29) && (uint64_t)b.l[i] < ((uint64_t)5 << 29), "mont_mul operand limb >= 5U");
  acc_t acc = 0;
  CP2_PAD_DECL
  auto lo_col = [&](auto kc) {
    constexpr int k = decltype(kc)::value;
    if constexpr (k > 0) tie_cols<0, 0>(acc, a, m, std::make_integer_sequence<int, k + 1>{}, std::make_integer_sequence<int, k>{});
#pragma unroll
    for (int i = 0; i <= k; ++i) acc += (uint64_t)a[i] * b.l[k - i];
#pragma unroll
    for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * FR_N[k - i];
    m[k] = MASKM ? (((uint32_t)acc * FR_NPRIME) & MASK) : ((uint32_t)acc * FR_NPRIME);
    acc += (uint64_t)m[k] * FR_N[0];
    acc >>= 29;
    CP2_PAD();
  };
  auto hi_col = [&](auto kc) {
    constexpr int k = decltype(kc)::value;
    constexpr int lo = k - NL + 1, cnt = NL - lo;
    tie_cols<lo, lo>(acc, a, m, std::make_integer_sequence<int, cnt>{}, std::make_integer_sequence<int, cnt>{});
#pragma unroll
    for (int i = lo; i < NL; ++i) acc += (uint64_t)a[i] * b.l[k - i];
#pragma unroll
    for (int i = lo; i < NL; ++i) acc += (uint64_t)m[i] * FR_N[k - i];
    r.l[k - NL] = (uint32_t)acc & MASK;
    acc >>= 29;
    CP2_PAD();
  };
  lo_col(std::integral_constant<int, 0>{}); lo_col(std::integral_constant<int, 1>{}); lo_col(std::integral_constant<int, 2>{});
  lo_col(std::integral_constant<int, 3>{}); lo_col(std::integral_constant<int, 4>{}); lo_col(std::integral_constant<int, 5>{});
  lo_col(std::integral_constant<int, 6>{}); lo_col(std::integral_constant<int, 7>{}); lo_col(std::integral_constant<int, 8>{});
  hi_col(std::integral_constant<int, 9>{}); hi_col(std::integral_constant<int, 10>{}); hi_col(std::integral_constant<int, 11>{});
  hi_col(std::integral_constant<int, 12>{}); hi_col(std::integral_constant<int, 13>{}); hi_col(std::integral_constant<int, 14>{});
  hi_col(std::integral_constant<int, 15>{}); hi_col(std::integral_constant<int, 16>{});
  r.l[NL - 1] = (uint32_t)acc;
  return r;
}

// d[0] takes no part in any product (cross terms are a[i] * d[j] with i < j), so it is never tied and never computed
template <int LO, int DLO, int... I, int... D, int... J>
__device__ __forceinline__ void tie_sq(acc_t& acc, uint32_t (&a)[NL], uint32_t (&d)[NL], uint32_t (&m)[NL],
                                       std::integer_sequence<int, I...>, std::integer_sequence<int, D...>,
                                       std::integer_sequence<int, J...>) {
  tie(acc, a[LO + I]..., d[DLO + D]..., m[LO + J]...);
}

// Montgomery square a*a/R: 45 products instead of 81 (cross terms against the doubled operand).
template <bool MASKM = true>
__device__ __forceinline__ Fe mont_sqr(const Fe& a_in) {
  Fe r;
  uint32_t a[NL], m[NL], d[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) a[i] = a_in.l[i];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    CP2_BOUND((uint64_t)a[i] * 100 < (uint64_t)(MASKM ? 247 : 202) << 29, "mont_sqr operand limb >= 2.47U (2.02U with unmasked quotient digits)");
    d[i] = i ? a[i] << 1 : 0;   // limbs < 2.47 U  =>  doubled < 2^32
  }
  acc_t acc = 0;
  CP2_PAD_DECL
  auto lo_col = [&](auto kc) {
    constexpr int k = decltype(kc)::value;
    if constexpr (k > 0)
      tie_sq<0, 1>(acc, a, d, m, std::make_integer_sequence<int, k + 1>{}, std::make_integer_sequence<int, k>{},
                   std::make_integer_sequence<int, k>{});
#pragma unroll
    for (int i = 0; 2 * i < k; ++i) acc += (uint64_t)a[i] * d[k - i];
    if constexpr ((k & 1) == 0) acc += (uint64_t)a[k / 2] * a[k / 2];
#pragma unroll
    for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * FR_N[k - i];
    m[k] = MASKM ? (((uint32_t)acc * FR_NPRIME) & MASK) : ((uint32_t)acc * FR_NPRIME);
    acc += (uint64_t)m[k] * FR_N[0];
    acc >>= 29;
    CP2_PAD();
  };
  auto hi_col = [&](auto kc) {
    constexpr int k = decltype(kc)::value;
    constexpr int lo = k - NL + 1, cnt = NL - lo;
    tie_sq<lo, lo>(acc, a, d, m, std::make_integer_sequence<int, cnt>{}, std::make_integer_sequence<int, cnt>{},
                   std::make_integer_sequence<int, cnt>{});
#pragma unroll
    for (int i = lo; 2 * i < k; ++i) acc += (uint64_t)a[i] * d[k - i];
    if constexpr ((k & 1) == 0) acc += (uint64_t)a[k / 2] * a[k / 2];
#pragma unroll
    for (int i = lo; i < NL; ++i) acc += (uint64_t)m[i] * FR_N[k - i];
    r.l[k - NL] = (uint32_t)acc & MASK;
    acc >>= 29;
    CP2_PAD();
  };
  lo_col(std::integral_constant<int, 0>{}); lo_col(std::integral_constant<int, 1>{}); lo_col(std::integral_constant<int, 2>{});
  lo_col(std::integral_constant<int, 3>{}); lo_col(std::integral_constant<int, 4>{}); lo_col(std::integral_constant<int, 5>{});
  lo_col(std::integral_constant<int, 6>{}); lo_col(std::integral_constant<int, 7>{}); lo_col(std::integral_constant<int, 8>{});
  hi_col(std::integral_constant<int, 9>{}); hi_col(std::integral_constant<int, 10>{}); hi_col(std::integral_constant<int, 11>{});
  hi_col(std::integral_constant<int, 12>{}); hi_col(std::integral_constant<int, 13>{}); hi_col(std::integral_constant<int, 14>{});
  hi_col(std::integral_constant<int, 15>{}); hi_col(std::integral_constant<int, 16>{});
  r.l[NL - 1] = (uint32_t)acc;
  return r;
}

// x^5 (Permutation.hs:14-17).  Each product is below A*B/R + c N with R = 169.3 N and c = 8.01 (unmasked quotient
// digits, MASKM = false: the default everywhere) or c = 1 (masked: used where the state is handed on, so that values
// do not pile up across permutations).  Input limbs < 2.02 U, value A < 60 N; output limbs 0..7 < U and
//   unmasked:  x^2 < 29.3 N,  x^4 < 13.1 N,  x^5 < 13.1 * 60 / 169.3 + 8.01 < 12.7 N
//   masked:    x^2 < 22.3 N,  x^4 <  3.94 N, x^5 <  3.94 * 60 / 169.3 + 1   <  2.4 N
template <bool MASKM = false>
__device__ __forceinline__ Fe sbox(const Fe& x) {
  Fe x2 = mont_sqr<MASKM>(x);
  Fe x4 = mont_sqr<MASKM>(x2);
  return mont_mul<MASKM>(x4, x);
}

// ---- y and z of the internal rounds: five 58-bit limbs in 64-bit registers ----------------------------------
// Only x passes through an S-box in the 56 internal rounds; y and z follow a linear recurrence (Permutation.hs:19-26).
// In 9 x 29-bit limbs every round needs a carry step on each of them (3 spare bits, values grow 4x per round): 3
// instructions per limb per value per round.  In 58-bit limbs there are 6 spare bits, so two rounds go by without any
// carry step, a 64-bit add is one instruction (v_lshl_add_u64) for two 29-bit limbs, and x's next S-box input is cut
// straight out of the 64-bit sums (its limbs stay below 2U + 64, inside mont_sqr's bound): 158 instead of 237
// instructions per pair of rounds.  Every VALU instruction costs one issue slot in this kernel.
constexpr int NW = 5;
constexpr uint64_t MASK58 = ((uint64_t)1 << 58) - 1;
struct Wide {
  uint64_t w[NW];
};

// limbs (< 2^32 each, lazy) -> wide limbs w[j] = l[2j] + l[2j+1] * 2^29 (value unchanged)
__device__ __forceinline__ Wide to_wide(const Fe& a) {
  Wide r;
#pragma unroll
  for (int j = 0; j < NW - 1; ++j) r.w[j] = (uint64_t)a.l[2 * j] + ((uint64_t)a.l[2 * j + 1] << 29);
  r.w[NW - 1] = a.l[NL - 1];
  return r;
}

// wide limbs 0..3 < 2^58 exactly (after reduce_wide) -> limbs 0..7 < U exactly
__device__ __forceinline__ Fe from_wide(const Wide& a) {
  Fe r;
#pragma unroll
  for (int j = 0; j < NW - 1; ++j) {
    CP2_BOUND(a.w[j] <= MASK58, "from_wide: limb not normalised");
    r.l[2 * j] = (uint32_t)a.w[j] & MASK;
    r.l[2 * j + 1] = (uint32_t)(a.w[j] >> 29);
  }
  r.l[NL - 1] = (uint32_t)a.w[NW - 1];
  return r;
}

// ---- lazy modular reduction (wide) -----------------------------------------------------------------
// Table of (bias - q*N) rows lives in LDS (filled by qtab_fill): row q, 5 x 64 bits, bias = {2^58, 2^58-1, 2^58-1,
// 2^58-1, -1} (sums to zero as a number, so  v + row(q) == v - q*N  with every limb 0..3 non-negative; the last
// limb wraps modulo 2^64 and comes out right because the true result is non-negative).
constexpr int QTAB_ROWS = 96;                // q <= 90 in the worst case of the internal rounds (poseidon2_dev.hpp); 96 rows = 3 840 B, which
                                             // with the 47-word line ring puts k_hash_cells at 51 968 B of LDS per block: three blocks per CU
constexpr int QTAB_WORDS = QTAB_ROWS * NW;   // 64-bit words
struct alignas(8) QTab {
  uint64_t row[QTAB_ROWS][NW];
};

__device__ __forceinline__ void qtab_fill(QTab& tab, int tid, int nthreads) {
  for (int idx = tid; idx < QTAB_WORDS; idx += nthreads) {
    int q = idx / NW, j = idx % NW;
    uint64_t bias = (j == 0) ? ((uint64_t)1 << 58) : (j == NW - 1 ? ~(uint64_t)0 : MASK58);
    tab.row[q][j] = bias - FR_QN_TABW[q][j];
  }
}

// Reduce a lazily-accumulated wide value: input limbs < 2^63, value < 96 N; output limbs 0..3 < 2^58 exactly, top limb
// small, value < 2 N (q is floor(v/N) or one less).
__device__ __forceinline__ Wide reduce_wide(const Wide& a, const QTab& tab) {
  // t ~ floor(v / 2^232), never an over-estimate; N / 2^232 = 0x30644e.72e1...
  uint32_t t = (uint32_t)a.w[NW - 1] + (uint32_t)(a.w[NW - 2] >> 58);
  CP2_BOUND(a.w[NW - 1] < ((uint64_t)1 << 31), "reduce_wide: top limb too large");
  // q = floor(t / (0x30644e + 1)) via 2^32 / 3171407 = 1354.27...; under-estimates only
  uint32_t q = __umulhi(t, 1354u);
  CP2_BOUND(q < (uint32_t)QTAB_ROWS, "reduce_wide: q outside the table (value >= 96N)");
#pragma unroll
  for (int j = 0; j < NW - 1; ++j) CP2_BOUND(a.w[j] < ((uint64_t)1 << 63), "reduce_wide input limb >= 2^63");
  const uint64_t* row = tab.row[q];
  Wide r;
  uint64_t c = 0;
#pragma unroll
  for (int j = 0; j < NW - 1; ++j) {
    uint64_t d = a.w[j] + row[j] + c;
    r.w[j] = d & MASK58;
    c = d >> 58;
  }
  r.w[NW - 1] = a.w[NW - 1] + row[NW - 1] + c;
  CP2_BOUND(r.w[NW - 1] < (1u << 24), "reduce_wide: result not below 2N (top limb)");
  return r;
}

// ---- conversions ------------------------------------------------------------------------------
// 8 little-endian dwords (a 256-bit integer) -> 9 x 29-bit limbs (raw value, NOT Montgomery form)
__device__ __forceinline__ Fe from_words(const uint32_t (&w)[8]) {
  Fe r;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int bit = 29 * i, k = bit / 32, s = bit % 32;
    uint32_t lo = w[k] >> s;
    if (s + 29 > 32 && k + 1 < 8) lo |= w[k + 1] << (32 - s);
    r.l[i] = (i == NL - 1) ? lo : (lo & MASK);   // top limb keeps all remaining 24 bits
  }
  return r;
}

// raw 256-bit integer -> Montgomery form (value < 2N, normalized): a * R^2 / R
__device__ __forceinline__ Fe to_mont(const Fe& raw) { return mont_mul<true>(raw, fe_const(FR_R2)); }

// Montgomery form (limbs < 2.47 U, value < R) -> canonical integer in [0, N) as 8 dwords
__device__ __forceinline__ void to_canonical_words(const Fe& a, uint32_t (&w)[8]) {
  Fe one = fe_zero();
  one.l[0] = 1;
  Fe c = mont_mul<true>(a, one);      // masked quotient digits: (a + m N)/R <= N, limbs 0..7 < U
  uint32_t diff = 0;                  // c == N  <=>  every limb equal (limbs 0..7 < U, so the form is unique)
#pragma unroll
  for (int i = 0; i < NL; ++i) diff |= c.l[i] ^ FR_N[i];
  const uint32_t keep = 0u - (uint32_t)(diff != 0);   // and-mask instead of 9 v_cndmask_b32 (~23 cycles each on gfx950)
#pragma unroll
  for (int i = 0; i < NL; ++i) c.l[i] &= keep;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int bit = 32 * j, i = bit / 29, s = bit % 29;   // word j starts inside limb i at bit s
    uint32_t v = c.l[i] >> s;
    if (i + 1 < NL) v |= c.l[i + 1] << (29 - s);
    if (29 - s + 29 < 32 && i + 2 < NL) v |= c.l[i + 2] << (58 - s);
    w[j] = v;
  }
}

}  // namespace fr

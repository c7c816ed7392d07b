// Host build of the DEVICE arithmetic (csrc/fr_gfx950.hpp + csrc/poseidon2_dev.hpp with CP2_HOST_CHECK):
// runs the very same source on the CPU with a 128-bit shadow accumulator and asserted limb bounds, under
// -fsanitize=address,undefined, on random and adversarial states, and compares every result with the C oracle.
// Usage: host_check <n_random> ; exits non-zero on any mismatch or bound violation.
#define CP2_HOST_CHECK 1
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../codex-storage-proofs-circuits_amd/csrc/poseidon2_dev.hpp"
extern "C" {
#include "../../oracle/p2_oracle.h"
}

using fr::Fe;

static fr::QTab g_qtab;

static Fe load_canonical(const uint8_t* p) {
  uint32_t w[8];
  std::memcpy(w, p, 32);
  return fr::to_mont(fr::from_words(w));
}
static void store_canonical(uint8_t* p, const Fe& v) {
  uint32_t w[8];
  fr::to_canonical_words(v, w);
  std::memcpy(p, w, 32);
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ULL;
static uint64_t rnd() {   // splitmix64
  uint64_t z = (rng_state += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

static int check_state(const uint8_t in[96]) {
  p2::State s;
  s.x = load_canonical(in);
  s.y = load_canonical(in + 32);
  s.z = load_canonical(in + 64);
  p2::permute(s, g_qtab);
  uint8_t got[96], want[96];
  store_canonical(got, s.x);
  store_canonical(got + 32, s.y);
  store_canonical(got + 64, s.z);
  p2o_permute(in, want);
  if (std::memcmp(got, want, 96) != 0) {
    std::fprintf(stderr, "MISMATCH for input ");
    for (int i = 0; i < 96; ++i) std::fprintf(stderr, "%02x", in[i]);
    std::fprintf(stderr, "\n");
    return 1;
  }
  return 0;
}

// the sponge regime: the state is NOT re-canonicalised between permutations (k_hash_cells / k_sponge2_felts):
// absorb two elements into the lazily-reduced state, permute, repeat; compare the digest with the oracle
static int check_sponge(const std::vector<std::vector<uint8_t>>& felts) {
  p2::State s;
  s.x = fr::fe_zero();
  s.y = fr::fe_zero();
  s.z = fr::fe_const(fr::FR_CIV_RATE2_MONT);
  const Fe one = fr::fe_const(fr::FR_R1);
  size_t nf = felts.size(), padded = (nf + 2) & ~(size_t)1;
  for (size_t k = 0; k < padded; k += 2) {
    Fe a = (k < nf) ? load_canonical(felts[k].data()) : (k == nf ? one : fr::fe_zero());
    Fe b = (k + 1 < nf) ? load_canonical(felts[k + 1].data()) : (k + 1 == nf ? one : fr::fe_zero());
    s.x = fr::norm(fr::add_lazy(s.x, a));
    s.y = fr::norm(fr::add_lazy(s.y, b));
    p2::permute(s, g_qtab);
  }
  uint8_t got[32], want[32];
  store_canonical(got, s.x);
  std::vector<uint8_t> flat(nf * 32);
  for (size_t i = 0; i < nf; ++i) std::memcpy(&flat[32 * i], felts[i].data(), 32);
  p2o_sponge2_felts(flat.data(), nf, want);
  if (std::memcmp(got, want, 32) != 0) { std::fprintf(stderr, "SPONGE MISMATCH (nf=%zu)\n", nf); return 1; }
  return 0;
}

int main(int argc, char** argv) {
  size_t n_random = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 2000;
  for (int i = 0; i < fr::QTAB_WORDS; ++i) fr::qtab_fill(g_qtab, i, fr::QTAB_WORDS);
  int bad = 0;
  size_t count = 0;
  // adversarial field elements: 0, 1, r-1, r, r+1, 2^256-1, all-ones limbs, powers of two around limb borders
  std::vector<std::vector<uint8_t>> special;
  auto push_int = [&](std::initializer_list<uint64_t> limbs) {
    std::vector<uint8_t> v(32, 0);
    int k = 0;
    for (uint64_t l : limbs) { std::memcpy(&v[8 * k], &l, 8); ++k; }
    special.push_back(v);
  };
  push_int({0, 0, 0, 0});
  push_int({1, 0, 0, 0});
  push_int({0x43e1f593f0000000ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL});   // r-1
  push_int({0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL});   // r
  push_int({0x43e1f593f0000002ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL});   // r+1
  push_int({~0ULL, ~0ULL, ~0ULL, ~0ULL});
  push_int({~0ULL, ~0ULL, ~0ULL, 0x3fffffffffffffffULL});
  push_int({0x1fffffff1fffffffULL, 0x1fffffff1fffffffULL, 0x1fffffff1fffffffULL, 0x1fffffff1fffffffULL});
  for (int b = 28; b < 256; b += 29) { std::vector<uint8_t> v(32, 0); v[b / 8] = (uint8_t)(1u << (b % 8)); special.push_back(v); }
  for (size_t a = 0; a < special.size(); ++a)
    for (size_t b = 0; b < special.size(); ++b)
      for (size_t c = 0; c < special.size(); c += 3) {
        uint8_t in[96];
        std::memcpy(in, special[a].data(), 32);
        std::memcpy(in + 32, special[b].data(), 32);
        std::memcpy(in + 64, special[(c + a + b) % special.size()].data(), 32);
        bad += check_state(in);
        ++count;
      }
  for (size_t i = 0; i < n_random; ++i) {
    uint8_t in[96];
    for (int k = 0; k < 12; ++k) { uint64_t r = rnd(); std::memcpy(in + 8 * k, &r, 8); }
    if (i & 1) { in[31] &= 0x1f; in[63] &= 0x1f; in[95] &= 0x1f; }   // half canonical, half arbitrary 256-bit
    bad += check_state(in);
    ++count;
  }
  // sponges: adversarial (all elements r-1 / 2^256-1 / 0) and random, lengths 0..70
  size_t sponges = 0;
  for (size_t which : {(size_t)2, (size_t)5, (size_t)0}) {
    for (size_t nf : {(size_t)0, (size_t)1, (size_t)2, (size_t)3, (size_t)67, (size_t)68}) {
      std::vector<std::vector<uint8_t>> f(nf, special[which]);
      bad += check_sponge(f);
      ++sponges;
    }
  }
  for (size_t i = 0; i < n_random / 40; ++i) {
    size_t nf = rnd() % 71;
    std::vector<std::vector<uint8_t>> f(nf, std::vector<uint8_t>(32));
    for (auto& v : f) {
      for (int k = 0; k < 4; ++k) { uint64_t r = rnd(); std::memcpy(&v[8 * k], &r, 8); }
      if (rnd() & 1) v[31] &= 0x1f;
    }
    bad += check_sponge(f);
    ++sponges;
  }
  std::printf("host_check: %zu states + %zu sponges, %d mismatches, no bound violations\n", count, sponges, bad);
  return bad ? 1 : 0;
}

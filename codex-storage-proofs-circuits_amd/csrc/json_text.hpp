// Byte-exact text of a proof input (include/codex_p2.h: cp2_proof_input_json and the streamed bodies), host only.
// Not installed; included by proof_input.cpp alone.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/codex_p2.h"

// ---- JSON (json/bn254.nim:57-74, json/shared.nim:17-25, types/bn254.nim:29-43) -----------------
// The text is written through a raw cursor into a buffer sized up front (90 bytes bound every line): with ~10^4
// 77-digit numbers per witness and thousands of witnesses per second the formatter is the host-side hot loop.
namespace cp2text {


const char DIGIT_PAIRS[201] =
    "0001020304050607080910111213141516171819202122232425262728293031323334353637383940414243444546474849"
    "5051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";

// (u1:u0) / 10^19 with u1 < 10^19, by multiplication with the precomputed reciprocal (Moeller-Granlund, "Improved
// division by invariant integers", algorithm 4; 10^19 has its top bit set, so it is already normalised): two multiplies
// instead of a 128-by-64 DIV (or a __udivti3 call).
constexpr uint64_t CH19 = 10000000000000000000ULL;
constexpr uint64_t CH19_INV = (uint64_t)((~(unsigned __int128)0) / CH19 - ((unsigned __int128)1 << 64));
inline uint64_t div_by_ch19(uint64_t u1, uint64_t u0, uint64_t* rem) {
  unsigned __int128 q = (unsigned __int128)CH19_INV * u1 + (((unsigned __int128)u1 << 64) | u0);
  uint64_t q1 = (uint64_t)(q >> 64) + 1, q0 = (uint64_t)q;
  uint64_t r = u0 - q1 * CH19;
  if (r > q0) { --q1; r += CH19; }
  if (r >= CH19) { ++q1; r -= CH19; }
  *rem = r;
  return q1;
}

inline void put2(char* p, uint32_t v) { std::memcpy(p, DIGIT_PAIRS + 2 * v, 2); }
inline void put8(char* p, uint32_t v) {   // exactly 8 digits of v < 10^8: three independent short division chains
  const uint32_t a = v / 10000, b = v % 10000;
  put2(p, a / 100); put2(p + 2, a % 100); put2(p + 4, b / 100); put2(p + 6, b % 100);
}
// exactly 19 digits of v < 10^19 (zero padded)
inline void put19(char* p, uint64_t v) {
  const uint64_t top = v / 10000000000000000ULL, rest = v % 10000000000000000ULL;   // 3 + 16 digits
  const uint32_t hi = (uint32_t)(rest / 100000000ULL), lo = (uint32_t)(rest % 100000000ULL);
  p[0] = (char)('0' + top / 100);
  put2(p + 1, (uint32_t)(top % 100));
  put8(p + 3, hi);
  put8(p + 11, lo);
}

inline char* put_digits(char* p, uint64_t v) {   // no padding; v == 0 prints "0"
  char tmp[20];
  int n = 0;
  do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
  while (n) *p++ = tmp[--n];
  return p;
}

// A 256-bit integer as five base-10^19 chunks, c[0] least significant (c[4] < 12).  W numbers are converted side by
// side: the remainder chain of one number is a serial dependency (each step waits for the previous remainder), so
// interleaving independent numbers is what keeps the multipliers busy.  Limbs per pass: the quotient after each
// division by 10^19 (63.1 bits) is below 2^192.9, 2^129.8, 2^66.7, 2^3.6.
struct DecChunks { uint64_t c[5]; };
template <int W>
inline void to_chunks(const uint8_t* felts, DecChunks (&out)[W]) {
  uint64_t w[W][4];
  for (int k = 0; k < W; ++k) std::memcpy(w[k], felts + 32 * k, 32);
  constexpr int LIMBS[4] = {4, 4, 3, 2};
  for (int pass = 0; pass < 4; ++pass) {
    uint64_t rem[W];
    for (int k = 0; k < W; ++k) rem[k] = 0;
    for (int i = LIMBS[pass] - 1; i >= 0; --i)
      for (int k = 0; k < W; ++k) w[k][i] = div_by_ch19(rem[k], w[k][i], &rem[k]);
    for (int k = 0; k < W; ++k) out[k].c[pass] = rem[k];
  }
  for (int k = 0; k < W; ++k) out[k].c[4] = w[k][0];
}

// canonical decimal in quotes: no leading zeros, "0" for zero (toDecimalF, types/bn254.nim:29-37)
inline char* put_quoted_chunks(char* p, const DecChunks& d) {
  int h = 4;
  while (h > 0 && d.c[h] == 0) --h;
  *p++ = '"';
  char lead[19];
  put19(lead, d.c[h]);                                // c[4] < 12 also fits
  int z = 0;
  while (z < 18 && lead[z] == '0') ++z;               // keeps one digit: "0" for zero
  std::memcpy(p, lead + z, (size_t)(19 - z));
  p += 19 - z;
  for (int c = h - 1; c >= 0; --c) { put19(p, d.c[c]); p += 19; }
  *p++ = '"';
  return p;
}

inline char* put_quoted_decimal(char* p, const uint8_t* le32) {
  DecChunks d[1];
  to_chunks<1>(le32, d);
  return put_quoted_chunks(p, d[0]);
}

inline char* put_str(char* p, const char* s, size_t n) {
  std::memcpy(p, s, n);
  return p + n;
}
#define PUT_LIT(p, lit) put_str(p, lit, sizeof(lit) - 1)

// writeList specialised to field elements: first "<prefix>[ x", then "<indent>, x", close "<indent>]" where the indent
// is as many spaces as the prefix is long (json/shared.nim:9-25)
inline char* put_felt_list(char* p, const char* prefix, size_t plen, const uint8_t* felts, size_t n) {
  constexpr int W = 4;
  DecChunks d[W];
  for (size_t i0 = 0; i0 < n; i0 += W) {
    const size_t m = std::min<size_t>(W, n - i0);
    if (m == W) {
      to_chunks<W>(felts + 32 * i0, d);
    } else {
      for (size_t k = 0; k < m; ++k) { DecChunks one[1]; to_chunks<1>(felts + 32 * (i0 + k), one); d[k] = one[0]; }
    }
    for (size_t k = 0; k < m; ++k) {
      if (i0 + k == 0) {
        p = put_str(p, prefix, plen);
        p = PUT_LIT(p, "[ ");
      } else {
        std::memset(p, ' ', plen);
        p += plen;
        p = PUT_LIT(p, ", ");
      }
      p = put_quoted_chunks(p, d[k]);
      *p++ = '\n';
    }
  }
  std::memset(p, ' ', plen);
  p += plen;
  return PUT_LIT(p, "]\n");
}

constexpr size_t LINE_BOUND = 96;   // longest line: 6 + 2 prefix characters, two quotes, 78 digits, newline

inline size_t head_bound(const cp2_config& cfg) { return (12 + (size_t)cfg.max_log2_nslots) * LINE_BOUND + 256; }
inline size_t body_bound(const cp2_config& cfg, size_t ns) {
  return (ns * (cp2_felts_per_bytes(cfg.cell_size) + (size_t)cfg.max_depth + 2) + 8) * LINE_BOUND;
}

// "{" .. the slotProof list: everything that needs the dataset tree (json/bn254.nim:59-66); appended to s
inline void text_head(std::string& s, const cp2_config& cfg, uint64_t slot_idx, const uint8_t* dataset_root, const uint8_t* entropy,
               const uint8_t* slot_root, const uint8_t* slot_proof) {
  const size_t at = s.size();
  s.resize(at + head_bound(cfg));
  char* p = &s[at];
  p = PUT_LIT(p, "{\n");
  p = PUT_LIT(p, "  \"dataSetRoot\":      "); p = put_quoted_decimal(p, dataset_root); *p++ = '\n';
  p = PUT_LIT(p, ", \"entropy\":          "); p = put_quoted_decimal(p, entropy); *p++ = '\n';
  p = PUT_LIT(p, ", \"nCellsPerSlot\":    "); p = put_digits(p, cfg.n_cells); *p++ = '\n';
  p = PUT_LIT(p, ", \"nSlotsPerDataSet\": "); p = put_digits(p, cfg.n_slots); *p++ = '\n';
  p = PUT_LIT(p, ", \"slotIndex\":        "); p = put_digits(p, slot_idx); *p++ = '\n';
  p = PUT_LIT(p, ", \"slotRoot\":         "); p = put_quoted_decimal(p, slot_root); *p++ = '\n';
  p = PUT_LIT(p, ", \"slotProof\":\n");
  p = put_felt_list(p, "    ", 4, slot_proof, (size_t)cfg.max_log2_nslots);
  s.resize((size_t)(p - s.data()));
}

// ", \"cellData\":" .. "}": the bulk, a function of the slot's own tree and cells only (json/bn254.nim:67-73); appended to s
inline void text_body(std::string& s, const cp2_config& cfg, size_t ns, const uint8_t* cell_data, const uint8_t* paths) {
  const size_t at = s.size();
  s.resize(at + body_bound(cfg, ns));
  char* p = &s[at];
  const size_t nf = cp2_felts_per_bytes(cfg.cell_size), md = (size_t)cfg.max_depth;
  std::vector<uint8_t> felts(nf * 32);
  p = PUT_LIT(p, ", \"cellData\":\n");
  for (size_t i = 0; i < ns; ++i) {
    cp2_bytes_to_felts(cell_data + i * cfg.cell_size, cfg.cell_size, felts.data());   // json/bn254.nim:25
    p = put_felt_list(p, i == 0 ? "    [ " : "    , ", 6, felts.data(), nf);
  }
  p = PUT_LIT(p, "    ]\n");
  p = PUT_LIT(p, ", \"merklePaths\":\n");
  for (size_t i = 0; i < ns; ++i) p = put_felt_list(p, i == 0 ? "    [ " : "    , ", 6, paths + i * md * 32, md);
  p = PUT_LIT(p, "    ]\n");
  p = PUT_LIT(p, "}\n");
  s.resize((size_t)(p - s.data()));
}

inline size_t body_reserve(const cp2_config& cfg, size_t ns) { return body_bound(cfg, ns); }

}  // namespace cp2text

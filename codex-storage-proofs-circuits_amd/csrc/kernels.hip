// HIP kernels (gfx950 only) for the Codex storage-proof hot path.  One field element per lane.
//
//   k_permute_batch   a1  Permutation.hs:40-45                       192 algorithmic B / permutation
//   k_hash_cells      a5  blocks/bn254.nim:23-29 (= Slot.hs:222-270 + Sponge.hs:30-43)   cellSize+32 B / cell
//   k_compress_layer  a7  merkle/bn254.nim:24-58 (one tree layer, many trees at once)    96 B / node
//   k_sponge2_felts   a3  Sponge.hs:30-43 over field elements (sampling, generic byte strings)
//   k_gen_fake_cells  a10 slot.nim:22-32
//   k_gather_rows         path / cell gather for proof inputs (merkle.nim:21-42 does this on the host)
//
// All global-memory field elements are 32-byte little-endian canonical integers (the ABI format).
#include "kernels.hpp"

#include <algorithm>
#include <cstdlib>
#include <string>

#include "poseidon2_dev.hpp"

namespace cp2k {
using fr::Fe;
using p2::State;

constexpr int TPB = 256;
// minimum waves per SIMD the register allocator must leave room for (tuning knobs)
#ifndef CP2_PERM_WAVES
#define CP2_PERM_WAVES 1
#endif
#ifndef CP2_HASH_WAVES
#define CP2_HASH_WAVES 3
#endif

__device__ __forceinline__ Fe load_fe_canonical(const uint4* p) {
  uint4 a = p[0], b = p[1];
  uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  return fr::to_mont(fr::from_words(w));
}

__device__ __forceinline__ void store_fe_canonical(uint4* p, const Fe& v) {
  uint32_t w[8];
  fr::to_canonical_words(v, w);
  p[0] = make_uint4(w[0], w[1], w[2], w[3]);
  p[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

// nodeKey (Merkle.hs:162-165) in Montgomery form, lane-varying key in {0,1,2,3}: mont(1)*bit0 + mont(2)*bit1
// with and-masks (v_cndmask_b32 costs ~23 cycles on gfx950).  The sum for key 3 is a
// lazy value < 2N with limbs < 2U, inside permute()'s input bounds.
__device__ __forceinline__ Fe key_fe(uint32_t key) {
  const Fe k1 = fr::fe_const(fr::FR_KEY1_MONT), k2 = fr::fe_const(fr::FR_KEY2_MONT);
  const uint32_t m1 = 0u - (key & 1u), m2 = 0u - ((key >> 1) & 1u);
  Fe r;
#pragma unroll
  for (int i = 0; i < fr::NL; ++i) r.l[i] = (k1.l[i] & m1) + (k2.l[i] & m2);
  return r;
}

// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TPB, CP2_PERM_WAVES) k_permute_batch(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
  __shared__ fr::QTab qtab;
  fr::qtab_fill(qtab, threadIdx.x, TPB);
  __syncthreads();
  size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  State s;
  s.x = load_fe_canonical(in + 6 * i);
  s.y = load_fe_canonical(in + 6 * i + 2);
  s.z = load_fe_canonical(in + 6 * i + 4);
  p2::permute(s, qtab);
  store_fe_canonical(out + 6 * i, s.x);
  store_fe_canonical(out + 6 * i + 2, s.y);
  store_fe_canonical(out + 6 * i + 4, s.z);
}

// ------------------------------------------------------------------------------------------------
// One Merkle layer for nseg independent trees: out[seg][j] = compress(in[seg][2j], in[seg][2j+1], key)
// with the odd-tail rule of merkle/bn254.nim:47-53 (compress(last, 0) with key+2).
__global__ void __launch_bounds__(TPB) k_compress_layer(const uint4* __restrict__ in, uint4* __restrict__ out,
                                                          size_t m_in, size_t m_out, size_t nseg, uint32_t bottom,
                                                          size_t in_seg_stride, size_t out_seg_stride) {
  __shared__ fr::QTab qtab;
  fr::qtab_fill(qtab, threadIdx.x, TPB);
  __syncthreads();
  size_t t = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (t >= m_out * nseg) return;
  size_t seg = t / m_out, j = t - seg * m_out;
  const uint4* src = in + 2 * (seg * in_seg_stride + 2 * j);
  bool pair = (2 * j + 1 < m_in);
  State s;
  s.x = load_fe_canonical(src);
  s.y = pair ? load_fe_canonical(src + 2) : fr::fe_zero();
  s.z = key_fe((bottom ? 1u : 0u) + (pair ? 0u : 2u));
  p2::permute(s, qtab);
  store_fe_canonical(out + 2 * (seg * out_seg_stride + j), s.x);
}

// ------------------------------------------------------------------------------------------------
// compress(x, y, key) for n independent pairs (merkle/bn254.nim:18): the seam call `compressWithKey`, batched.
__global__ void __launch_bounds__(TPB) k_compress_pairs(const uint4* __restrict__ xy, uint32_t key, uint4* __restrict__ out, size_t n) {
  __shared__ fr::QTab qtab;
  fr::qtab_fill(qtab, threadIdx.x, TPB);
  __syncthreads();
  size_t t = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (t >= n) return;
  State s;
  s.x = load_fe_canonical(xy + 4 * t);
  s.y = load_fe_canonical(xy + 4 * t + 2);
  s.z = key_fe(key);
  p2::permute(s, qtab);
  store_fe_canonical(out + 2 * t, s.x);
}

// ------------------------------------------------------------------------------------------------
// Batched rate-2 sponge over field elements (Sponge.hs:30-43): item i hashes felts[i*nf .. i*nf+nf).
__global__ void __launch_bounds__(TPB) k_sponge2_felts(const uint4* __restrict__ felts, size_t nf, size_t nitems,
                                                         uint4* __restrict__ out) {
  __shared__ fr::QTab qtab;
  fr::qtab_fill(qtab, threadIdx.x, TPB);
  __syncthreads();
  size_t t = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (t >= nitems) return;
  const uint4* src = felts + 2 * t * nf;
  State s;
  s.x = fr::fe_zero();
  s.y = fr::fe_zero();
  s.z = fr::fe_const(fr::FR_CIV_RATE2_MONT);
  const Fe one = fr::fe_const(fr::FR_R1);
  size_t padded = (nf + 2) & ~(size_t)1;   // nf odd: +1 ("1"); nf even: +2 ("1","0")
#pragma unroll 1
  for (size_t k = 0; k < padded; k += 2) {
    Fe a = (k < nf) ? load_fe_canonical(src + 2 * k) : (k == nf ? one : fr::fe_zero());
    Fe b = (k + 1 < nf) ? load_fe_canonical(src + 2 * (k + 1)) : (k + 1 == nf ? one : fr::fe_zero());
    s.x = fr::norm(fr::add_lazy(s.x, a));
    s.y = fr::norm(fr::add_lazy(s.y, b));
    p2::permute(s, qtab);
  }
  store_fe_canonical(out + 2 * t, s.x);
}

// ------------------------------------------------------------------------------------------------
// hashCell for a contiguous array of cells: one cell per lane.
//
// The byte stream a lane absorbs is  cell || 0x01 || 0-pad to 31*nfelts || sponge pad, where the
// sponge's "1" pad element is itself the chunk {0x01,0,...} and the optional "0" element a zero
// chunk (Slot.hs:243-250 then Sponge.hs:36-39).  So the whole padded input is one byte stream cut
// into 31-byte little-endian chunks, two per permutation (62 bytes).
//
// Staging: the stream is fetched in whole 128-byte lines, each exactly once.  Per stage a wave copies one
// line of each of its 64 cells into a per-cell LDS ring: lane-linear dword loads, 32 consecutive lanes read
// one full line (perfectly coalesced), and the ring row stride of 47 dwords is odd, so the per-lane reads
// are bank-conflict free.  The sponge consumes 62 bytes per permutation, the producer adds 128 per stage,
// so at most 60 bytes wait in the ring when the next line lands (what is left is even and below 62): 188 bytes, the ring's size.
// A lane's 17-dword read window (68 bytes for 62) may run a few bytes past the valid data; chunk_pair uses 512 bits of it.  (The first
// version staged 124-byte tiles, which touches most lines twice: FETCH_SIZE showed 2x the algorithmic
// bytes, calibrated with tools/fetch_calib.hip.)
#ifndef CP2_RING_WORDS
#define CP2_RING_WORDS 47
#define CP2_RING_STRIDE 47
#endif
#ifndef CP2_HASH_BT
#define CP2_HASH_BT 256
#endif
// 47 words = 188 bytes per cell: exactly the 60 bytes that can still wait plus the 128 of a new line.  Round 5: with 48 words and a
// row stride of 49 (round 1) a 256-thread workgroup held 54 016 B of LDS and only TWO of them fitted a CU -- SQ_WAVE_CYCLES showed
// 1.94 waves per SIMD where the registers allow 3; at 47 / 47 it holds 51 968 B, three fit, and the kernel is 1.1 % faster
// (tools/ab_hash_kernel.py, profiles/r05_ab_hash_ring.txt).  The stride stays odd: the per-lane reads are bank-conflict free.
constexpr int RING_WORDS = CP2_RING_WORDS;
constexpr int RING_STRIDE = CP2_RING_STRIDE;
constexpr int LINE_WORDS = 32;             // 128 bytes

// two 31-byte chunks from a 17-dword window whose first chunk starts SH bits into w[0] (SH = 0 or 16)
template <int SH>
__device__ __forceinline__ void chunk_pair(const uint32_t (&w)[17], Fe& a, Fe& b) {
#pragma unroll
  for (int i = 0; i < fr::NL; ++i) {
    const int width = (i == fr::NL - 1) ? 16 : 29;
    {
      const int bit = SH + 29 * i, k = bit / 32, sh = bit % 32;
      uint32_t v = w[k] >> sh;
      if (sh + width > 32) v |= w[k + 1] << (32 - sh);
      a.l[i] = v & ((1u << width) - 1);
    }
    {
      const int bit = SH + 248 + 29 * i, k = bit / 32, sh = bit % 32;
      uint32_t v = w[k] >> sh;
      if (sh + width > 32) v |= w[k + 1] << (32 - sh);
      b.l[i] = v & ((1u << width) - 1);
    }
  }
}

// LDS allows three workgroups per CU = three waves per SIMD: tell the register allocator that is also the MOST it will
// ever get, so that it uses the registers (up to 168) instead of squeezing the staging loops for an occupancy it cannot have
//
// BT = threads per workgroup: 256 (four waves share one reduction table: 51 968 B of LDS with the 47-word ring, three workgroups
// per CU -- two when a launch leaves room; what the product launches) or 64 (one wave per workgroup: 16 384 B, ten per CU; kept for tools/hash_block_sweep.cpp, which showed
// that the workgroup shape does not matter below 256 MiB and that 64 lanes lose above: profiles/r03_hash_block_sweep.txt).
template <int BT>
__global__ void __launch_bounds__(BT, CP2_HASH_WAVES) __attribute__((amdgpu_waves_per_eu(CP2_HASH_WAVES, CP2_HASH_WAVES))) k_hash_cells(const uint8_t* __restrict__ cells, size_t cell_size,
                                                                      size_t n_cells, uint4* __restrict__ out) {
  __shared__ fr::QTab qtab;
  __shared__ uint32_t ring[BT / 64][64 * RING_STRIDE];
  fr::qtab_fill(qtab, threadIdx.x, BT);

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t cell0 = (size_t)blockIdx.x * BT + (size_t)wave * 64;
  const size_t my_cell = cell0 + lane;
  const size_t nfelts = (cell_size + 31) / 31;            // chunks of cell || 0x01
  const size_t total = (nfelts + 2) & ~(size_t)1;          // + sponge pad, even
  const size_t sponge_pad_pos = 31 * nfelts;               // byte position of the sponge's "1"
  const size_t stream_len = 31 * total;                    // multiple of 62
  const size_t nlines = (stream_len + 127) / 128;
  const bool aligned4 = ((cell_size & 3) == 0) && ((reinterpret_cast<uintptr_t>(cells) & 3) == 0);

  State s;
  s.x = fr::fe_zero();
  s.y = fr::fe_zero();
  s.z = fr::fe_const(fr::FR_CIV_RATE2_MONT);

  uint32_t* my_ring = ring[wave];
  const uint32_t* my_row = my_ring + lane * RING_STRIDE;
  size_t cons = 0;                                          // bytes absorbed so far (wave-uniform)
#pragma unroll 1
  for (size_t line = 0; line < nlines; ++line) {
    __syncthreads();   // every lane is done reading what this stage overwrites (and, first pass, the qtab fill)
    const int ring_base = (int)((line * LINE_WORDS) % RING_WORDS);
    // Word w of line `line` of 32 cells per lane: lane = (cell parity, w), cell = cell0 + 2k + (lane >> 5) for k = 0..31.
    // Everything but the cell is the same for all 32 loads of a lane, so it is worked out once per line: the byte offset
    // p0, whether the dword lies inside the cell, the padding bit it may carry, its slot in the ring.
    {
      const int w = lane & 31, half = lane >> 5;
      const size_t p0 = line * 128 + (size_t)w * 4;
      const uint32_t padmask = (sponge_pad_pos >= p0 && sponge_pad_pos < p0 + 4) ? 1u << (8 * (sponge_pad_pos - p0)) : 0u;
      int slot = ring_base + w;
      if (slot >= RING_WORDS) slot -= RING_WORDS;
      uint32_t* dst = my_ring + half * RING_STRIDE + slot;                     // + 2 * RING_STRIDE per k
      const size_t first = cell0 + (size_t)half;
      const int kmax = first < n_cells ? (int)((n_cells - first + 1) / 2 < 32 ? (n_cells - first + 1) / 2 : 32) : 0;   // cells that exist
      const uint8_t* ptr = cells + first * cell_size + p0;
      const size_t step = 2 * cell_size;
      if (aligned4 && p0 + 4 <= cell_size) {            // a whole dword of cell data: the common case
#pragma unroll 8
        for (int k = 0; k < LINE_WORDS; ++k) {
          uint32_t val = padmask;
          if (k < kmax) val |= *reinterpret_cast<const uint32_t*>(ptr);
          dst[k * 2 * RING_STRIDE] = val;
          ptr += step;
        }
      } else if (p0 <= cell_size) {                      // the dword straddles the end of the cell (or cells are not 4-byte aligned)
#pragma unroll 1
        for (int k = 0; k < LINE_WORDS; ++k) {
          uint32_t val = padmask;
          if (k < kmax) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
              const size_t p = p0 + b;
              const uint32_t byte = (p < cell_size) ? ptr[b] : (p == cell_size ? 1u : 0u);   // 0x01 ends the data (Slot.hs:243-250)
              val |= byte << (8 * b);
            }
          }
          dst[k * 2 * RING_STRIDE] = val;
          ptr += step;
        }
      } else {                                           // past the data: zero padding, possibly the sponge's own "1"
#pragma unroll 8
        for (int k = 0; k < LINE_WORDS; ++k) dst[k * 2 * RING_STRIDE] = padmask;
      }
    }
    __syncthreads();
    const size_t avail = (line + 1) * 128 < stream_len ? (line + 1) * 128 : stream_len;
#pragma unroll 1
    while (cons + 62 <= avail) {
      const int off = (int)(cons % (RING_WORDS * 4));
      int d = off >> 2;
      uint32_t w[17];
#pragma unroll
      for (int j = 0; j < 17; ++j) {
        w[j] = my_row[d];
        d = (d + 1 == RING_WORDS) ? 0 : d + 1;
      }
      Fe a, b;
      if (off & 2) chunk_pair<16>(w, a, b);
      else chunk_pair<0>(w, a, b);
      a = fr::to_mont(a);
      b = fr::to_mont(b);
      s.x = fr::norm(fr::add_lazy(s.x, a));
      s.y = fr::norm(fr::add_lazy(s.y, b));
      p2::permute(s, qtab);
      cons += 62;
    }
  }
  if (my_cell < n_cells) store_fe_canonical(out + 2 * my_cell, s.x);
}

// ------------------------------------------------------------------------------------------------
// genFakeCell (slot.nim:22-32): sequential in a cell, independent across cells; one cell per lane.
// Thread t makes "global cell" g = list ? list[t] : first + t.  With cells_per_slot != 0 the global
// index spans several slots: slot = g / cells_per_slot uses seed0 + 1001*slot (dataset.nim:32, seed0
// already holding the "+72" of the first slot) and the cell index inside the slot is g % cells_per_slot.
// Slots cut into units (units_per_slot > 1: a slot's cells spread over several devices, `cells_per_slot` then counts
// the cells of ONE unit): unit = first_unit + g / cells_per_slot, slot = unit / units_per_slot (seed0 = the seed of
// slot 0 of the dataset) and the cell index inside the slot is (unit % units_per_slot) * cells_per_slot + g % cells_per_slot.
__global__ void __launch_bounds__(TPB) k_gen_fake_cells(uint64_t seed0, uint64_t cells_per_slot, uint64_t first,
                                                          const uint64_t* __restrict__ list, size_t n_cells,
                                                          size_t cell_size, uint8_t* __restrict__ out,
                                                          uint64_t units_per_slot, uint64_t first_unit) {
  __shared__ uint4 gen_stage[TPB / 64][64 * 9];
  size_t t = (size_t)blockIdx.x * TPB + threadIdx.x;
  const bool active = t < n_cells;
  if (!active && ((cell_size & 127) != 0 || (reinterpret_cast<uintptr_t>(out) & 15) != 0)) return;
  const uint64_t g = active ? (list ? list[t] : first + t) : 0;   // idle tail lanes only take part in the write-out
  uint64_t slot = cells_per_slot ? g / cells_per_slot : 0;
  uint64_t idx = cells_per_slot ? g - slot * cells_per_slot : g;
  if (units_per_slot > 1) {          // `slot` so far is the unit's index inside this batch
    const uint64_t unit = first_unit + slot;
    slot = unit / units_per_slot;
    idx += (unit - slot * units_per_slot) * cells_per_slot;
  }
  const uint64_t seed1 = (seed0 + 1001 * slot) + 0xdeadcafeULL;
  const uint64_t seed2 = idx + 0x98765432ULL;
  uint64_t state = 1;
  uint8_t* dst = out + t * cell_size;
  const bool wide = ((cell_size & 127) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  if (wide) {
    // Each lane makes 128 bytes of its cell, parks them in LDS, then the wave writes them out so that 8
    // consecutive lanes store one cell's full 128-byte line (a lane storing 16 bytes at a 2 KiB stride made
    // WRITE_SIZE 2.7x the data: partial-line writes).
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint4* my = gen_stage[wave];
    const size_t wave_cell0 = (size_t)blockIdx.x * TPB + (size_t)wave * 64;
#pragma unroll 1
    for (size_t i = 0; i < cell_size; i += 128) {
#pragma unroll 1
      for (int piece = 0; piece < 8; ++piece) {
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int b = 0; b < 16; ++b) {
          state = state * (state + seed1) * (state + seed2) + state * (state ^ 0x5a5a5a5aULL) + seed1 * state + (seed2 + 17);
          state = state % 1698428844001831ULL;
          w[b >> 2] |= (uint32_t)(state & 0xff) << (8 * (b & 3));
        }
        my[lane * 9 + piece] = make_uint4(w[0], w[1], w[2], w[3]);   // row stride 9 x 16 B: conflict-free both ways
      }
      // wave-private region: order this wave's LDS writes before its cross-lane reads (no instructions on wave64,
      // but the compiler may not move the reads up)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int c = k * 8 + (lane >> 3), piece = lane & 7;
        const size_t cell = wave_cell0 + c;
        uint4 v = my[c * 9 + piece];
        if (cell < n_cells) *reinterpret_cast<uint4*>(out + cell * cell_size + i + 16 * piece) = v;
      }
      // ... and the next round's writes after these reads
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    return;
  }
#pragma unroll 1
  for (size_t i = 0; i < cell_size; ++i) {
    state = state * (state + seed1) * (state + seed2) + state * (state ^ 0x5a5a5a5aULL) + seed1 * state + (seed2 + 17);
    state = state % 1698428844001831ULL;
    dst[i] = (uint8_t)state;
  }
}

// ------------------------------------------------------------------------------------------------
// cellIndex (sample/bn254.nim:16-24) + merged, padded path rows (merkle.nim:21-42,86-100; types.nim:27-37) for
// one (slot, counter) pair per lane.  The sponge input [entropy, slotRoot, counter] pads to 4 elements with the
// "1" (Sponge.hs:36-39): two permutations.
__global__ void __launch_bounds__(TPB) k_sample_paths(TreeGeom g, const uint4* __restrict__ nodes, const uint4* __restrict__ entropy,
                                                        const uint64_t* __restrict__ slots, uint64_t slot0, size_t n_items,
                                                        uint32_t ns, uint32_t md, uint64_t* __restrict__ indices,
                                                        uint64_t* __restrict__ gcell, uint64_t* __restrict__ rows) {
  __shared__ fr::QTab qtab;
  fr::qtab_fill(qtab, threadIdx.x, TPB);
  __syncthreads();
  size_t t = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (t >= n_items * ns) return;
  const size_t item = t / ns;
  const uint32_t counter = (uint32_t)(t - item * ns) + 1;            // sample/bn254.nim:27
  const uint64_t slot = slots ? slots[item] : slot0 + item;
  State s;
  s.x = load_fe_canonical(entropy);
  s.y = load_fe_canonical(nodes + 2 * (g.toff[g.nt - 1] + slot));  // treeRoot(bigTree)
  s.z = fr::fe_const(fr::FR_CIV_RATE2_MONT);
  p2::permute(s, qtab);
  Fe c = fr::fe_zero();
  c.l[0] = counter & fr::MASK;
  c.l[1] = counter >> 29;
  s.x = fr::norm(fr::add_lazy(s.x, fr::to_mont(c)));
  s.y = fr::norm(fr::add_lazy(s.y, fr::fe_const(fr::FR_R1)));
  p2::permute(s, qtab);
  uint32_t w[8];
  fr::to_canonical_words(s.x, w);
  const uint64_t cell = (((uint64_t)w[1] << 32) | w[0]) & (g.n_cells - 1);   // extractLowBits, types/bn254.nim:47-59
  indices[t] = cell;
  gcell[t] = slot * g.n_cells + cell;
  uint64_t* r = rows + t * md;
  uint32_t d = 0;
  const uint64_t b = cell / g.cpb;
  uint64_t j = cell - b * g.cpb, m = g.cpb;
  for (uint32_t k = 0; k + 1 < g.nb && d < md; ++k, ++d) {           // bottom proof inside the block tree
    const uint64_t sib = j ^ 1;
    r[d] = (sib < m) ? g.boff[k] + (slot * g.nblocks + b) * g.bsz[k] + sib : ~0ULL;
    j >>= 1;
    m = (m + 1) >> 1;
  }
  j = b;
  m = g.nblocks;
  for (uint32_t k = 0; k + 1 < g.nt && d < md; ++k, ++d) {           // top proof inside the slot's big tree
    const uint64_t sib = j ^ 1;
    r[d] = (sib < m) ? g.toff[k] + slot * g.tsz[k] + sib : ~0ULL;
    j >>= 1;
    m = (m + 1) >> 1;
  }
  for (; d < md; ++d) r[d] = ~0ULL;                                   // padMerkleProof
}

// ------------------------------------------------------------------------------------------------
// Gather nrows rows of row_bytes bytes: out[r] = src[index[r] * row_bytes ...]; rows whose index is
// ~0 are zero-filled (the reference pads paths with zero, merkle.nim:33-34, types.nim:27-37).
__global__ void __launch_bounds__(TPB) k_gather_rows(const uint8_t* __restrict__ src, const uint64_t* __restrict__ index,
                                                       size_t nrows, size_t row_bytes, uint8_t* __restrict__ out) {
  const size_t words_per_row = row_bytes / 4;
  size_t t = (size_t)blockIdx.x * TPB + threadIdx.x;
  size_t stride = (size_t)gridDim.x * TPB;
  for (; t < nrows * words_per_row; t += stride) {
    size_t r = t / words_per_row, w = t - r * words_per_row;
    uint64_t idx = index[r];
    uint32_t v = 0;
    if (idx != ~0ULL) v = reinterpret_cast<const uint32_t*>(src + idx * row_bytes)[w];
    reinterpret_cast<uint32_t*>(out + r * row_bytes)[w] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// Workgroups for n work items.  A grid holds at most 2^31 - 1 workgroups in x; the per-item kernels are launched in slices of
// at most MAX_ITEMS items (every item is independent and addressed from a base pointer), the layer / sampling kernels, whose
// item index is decomposed inside the kernel, refuse what does not fit one grid (2^38 nodes: far beyond any HBM).
// hipGetLastError() returns (and clears) the last error of ANY earlier runtime call of this thread, e.g. a hipMalloc that failed
// and was already reported to the caller: clear it before a launch so that the status read after the launch is the launch's own.
#define CP2K_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)
constexpr size_t MAX_BLOCKS = (size_t)1 << 30;
constexpr size_t MAX_ITEMS = MAX_BLOCKS * TPB;
static inline bool fits_one_grid(size_t n) { return (n + TPB - 1) / TPB <= MAX_BLOCKS; }
static inline unsigned grid_for(size_t n) { return (unsigned)((n + TPB - 1) / TPB); }   // n <= MAX_ITEMS

hipError_t launch_permute_batch(const void* in, void* out, size_t n, hipStream_t st) {
  for (size_t i0 = 0; i0 < n; i0 += MAX_ITEMS) {
    const size_t m = n - i0 < MAX_ITEMS ? n - i0 : MAX_ITEMS;
    CP2K_LAUNCH(k_permute_batch, dim3(grid_for(m)), dim3(TPB), 0, st, (const uint4*)in + 6 * i0, (uint4*)out + 6 * i0, m);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

hipError_t launch_compress_layer(const void* in, void* out, size_t m_in, size_t nseg, bool bottom,
                                 size_t in_seg_stride, size_t out_seg_stride, hipStream_t st) {
  size_t m_out = (m_in + 1) / 2;
  if (m_out * nseg == 0) return hipSuccess;
  if (!fits_one_grid(m_out * nseg)) return hipErrorInvalidValue;
  CP2K_LAUNCH(k_compress_layer, dim3(grid_for(m_out * nseg)), dim3(TPB), 0, st, (const uint4*)in, (uint4*)out,
                     m_in, m_out, nseg, bottom ? 1u : 0u, in_seg_stride, out_seg_stride);
  return hipGetLastError();
}

hipError_t launch_compress_pairs(const void* xy, uint32_t key, void* out, size_t n, hipStream_t st) {
  for (size_t i0 = 0; i0 < n; i0 += MAX_ITEMS) {
    const size_t m = n - i0 < MAX_ITEMS ? n - i0 : MAX_ITEMS;
    CP2K_LAUNCH(k_compress_pairs, dim3(grid_for(m)), dim3(TPB), 0, st, (const uint4*)xy + 4 * i0, key, (uint4*)out + 2 * i0, m);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

hipError_t launch_sponge2_felts(const void* felts, size_t nf, size_t nitems, void* out, hipStream_t st) {
  for (size_t i0 = 0; i0 < nitems; i0 += MAX_ITEMS) {
    const size_t m = nitems - i0 < MAX_ITEMS ? nitems - i0 : MAX_ITEMS;
    CP2K_LAUNCH(k_sponge2_felts, dim3(grid_for(m)), dim3(TPB), 0, st, (const uint4*)felts + 2 * nf * i0, nf, m, (uint4*)out + 2 * i0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// Workgroup size of k_hash_cells: 256 unless CP2_HASH_BLOCK=64 is in the environment (A/B tooling only).
static int hash_block_override() {
  static const int v = [] {
    const char* e = std::getenv("CP2_HASH_BLOCK");
    const int x = e ? std::atoi(e) : 0;
    return (x == 64 || x == 256) ? x : 0;
  }();
  return v;
}

// leave_room: 28 KiB of dynamic LDS nobody uses, so that TWO workgroups fit a CU instead of three.  The kernel itself loses under 1 %
// (issue bound from two waves per SIMD up), and the third of every CU it no longer holds is where the small dependent kernels of the
// streamed build -- a group's layer passes, its sampling and gathers -- run beside it: next to a launch that holds every workgroup
// slot such a chain finishes only when the launch drains (tools/coresidency_probe.cpp, profiles/r05_coresidency_probe.txt).
// Whether a device takes the larger request is decided ONCE, when a context is made on it (hash_cells_can_leave_room, called by
// cp2_init; the callers pass leave_room only where it said yes): a launch is never retried, so an error at a launch is that launch's.
constexpr unsigned HASH_ROOM_BYTES = 28672;

bool hash_cells_can_leave_room(size_t lds_cap, std::string* why) {
  int dev = 0, max_lds = 0;
  hipFuncAttributes fa;
  hipError_t e = hipGetDevice(&dev);
  if (e == hipSuccess) e = hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev);
  if (e == hipSuccess) e = hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_hash_cells<CP2_HASH_BT>));
  if (e != hipSuccess) {
    (void)hipGetLastError();
    if (why) *why = std::string("the device could not be asked (") + hipGetErrorString(e) + ")";
    return false;
  }
  const size_t limit = lds_cap ? std::min<size_t>(lds_cap, (size_t)max_lds) : (size_t)max_lds;
  if (fa.sharedSizeBytes + HASH_ROOM_BYTES > limit) {
    if (why) *why = "the kernel's " + std::to_string(fa.sharedSizeBytes) + " B of LDS + " + std::to_string(HASH_ROOM_BYTES) + " B exceed the " + std::to_string(limit) + " B a workgroup may hold";
    return false;
  }
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_hash_cells<CP2_HASH_BT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)HASH_ROOM_BYTES);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    if (why) *why = std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize) refused (") + hipGetErrorString(e) + ")";
    return false;
  }
  if (why) *why = std::to_string(fa.sharedSizeBytes) + " + " + std::to_string(HASH_ROOM_BYTES) + " B of " + std::to_string(limit) + " B per workgroup";
  return true;
}

hipError_t launch_hash_cells_block(int block, const void* cells, size_t cell_size, size_t n_cells, void* out, hipStream_t st, bool leave_room) {
  if (block != 64 && block != 256) return hipErrorInvalidValue;
  if (block == 256) block = CP2_HASH_BT;
  const unsigned dyn = (leave_room && block != 64) ? HASH_ROOM_BYTES : 0u;
  const size_t max_items = MAX_BLOCKS * (size_t)block;
  for (size_t i0 = 0; i0 < n_cells; i0 += max_items) {
    const size_t m = n_cells - i0 < max_items ? n_cells - i0 : max_items;
    const unsigned grid = (unsigned)((m + block - 1) / block);
    const uint8_t* src = (const uint8_t*)cells + i0 * cell_size;
    if (block == 64) CP2K_LAUNCH(k_hash_cells<64>, dim3(grid), dim3(64), 0, st, src, cell_size, m, (uint4*)out + 2 * i0);
    else CP2K_LAUNCH(k_hash_cells<CP2_HASH_BT>, dim3(grid), dim3(CP2_HASH_BT), dyn, st, src, cell_size, m, (uint4*)out + 2 * i0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

hipError_t launch_hash_cells(const void* cells, size_t cell_size, size_t n_cells, void* out, hipStream_t st, bool leave_room) {
  const int block = hash_block_override();
  return launch_hash_cells_block(block ? block : 256, cells, cell_size, n_cells, out, st, leave_room);
}

hipError_t launch_gen_fake_cells(uint64_t seed0, uint64_t cells_per_slot, uint64_t first, const uint64_t* list,
                                 size_t n_cells, size_t cell_size, void* out, hipStream_t st, uint64_t units_per_slot, uint64_t first_unit) {
  if (n_cells == 0 || cell_size == 0) return hipSuccess;
  if (!fits_one_grid(n_cells)) return hipErrorInvalidValue;
  CP2K_LAUNCH(k_gen_fake_cells, dim3(grid_for(n_cells)), dim3(TPB), 0, st, seed0, cells_per_slot, first, list,
                     n_cells, cell_size, (uint8_t*)out, units_per_slot, first_unit);
  return hipGetLastError();
}

hipError_t launch_sample_paths(const TreeGeom& g, const void* nodes, const void* d_entropy, const uint64_t* slots, uint64_t slot0,
                               size_t n_items, uint32_t ns, uint32_t md, uint64_t* indices, uint64_t* gcell, uint64_t* rows,
                               hipStream_t st) {
  if (n_items == 0 || ns == 0) return hipSuccess;
  if (!fits_one_grid(n_items * ns)) return hipErrorInvalidValue;
  CP2K_LAUNCH(k_sample_paths, dim3(grid_for(n_items * ns)), dim3(TPB), 0, st, g, (const uint4*)nodes, (const uint4*)d_entropy,
                     slots, slot0, n_items, ns, md, indices, gcell, rows);
  return hipGetLastError();
}

hipError_t launch_gather_rows(const void* src, const uint64_t* index, size_t nrows, size_t row_bytes, void* out, hipStream_t st) {
  if (nrows == 0) return hipSuccess;
  size_t work = nrows * (row_bytes / 4);
  unsigned grid = work > (size_t)4096 * TPB ? 4096u : grid_for(work);
  CP2K_LAUNCH(k_gather_rows, dim3(grid), dim3(TPB), 0, st, (const uint8_t*)src, index, nrows, row_bytes, (uint8_t*)out);
  return hipGetLastError();
}

}  // namespace cp2k

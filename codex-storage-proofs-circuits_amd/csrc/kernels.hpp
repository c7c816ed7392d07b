// Launch wrappers of the HIP kernels in kernels.hip (device pointers, asynchronous on `st`).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <string>

namespace cp2k {

// Kernel-argument copy of the layer-major node layout of a batch of slot trees (proof_input.cpp, trees_layout):
// block-tree layer k of block b of slot s starts at element boff[k] + (s * nblocks + b) * bsz[k]; big-tree layer k
// of slot s at toff[k] + s * tsz[k]; the last big-tree layer holds the slot roots.
struct TreeGeom {
  static constexpr int MAX_LAYERS = 40;
  uint32_t nb, nt;                      // layer counts (leaves included) of a block tree / a big tree
  uint64_t cpb, nblocks, n_cells;
  uint64_t boff[MAX_LAYERS], bsz[MAX_LAYERS], toff[MAX_LAYERS], tsz[MAX_LAYERS];
};

hipError_t launch_permute_batch(const void* in, void* out, size_t n, hipStream_t st);
// one Merkle layer of nseg trees; segment strides are in field elements
hipError_t launch_compress_layer(const void* in, void* out, size_t m_in, size_t nseg, bool bottom,
                                 size_t in_seg_stride, size_t out_seg_stride, hipStream_t st);
// n pairs (x, y) of canonical elements -> compress(x, y, key), key in {0,1,2,3}
hipError_t launch_compress_pairs(const void* xy, uint32_t key, void* out, size_t n, hipStream_t st);
hipError_t launch_sponge2_felts(const void* felts, size_t nf, size_t nitems, void* out, hipStream_t st);
// k_hash_cells runs 256-lane workgroups at every batch size.  The 64-lane instantiation was measured against it from 2 MiB to
// 8 GiB (tools/hash_block_sweep.cpp, profiles/r03_hash_block_sweep.txt): identical up to 256 MiB -- a launch lasts at least
// the lifetime of ONE wave, 34 serial permutations = 3.3 ms, whatever the workgroup shape -- and 5...25 % slower above.
hipError_t launch_hash_cells(const void* cells, size_t cell_size, size_t n_cells, void* out, hipStream_t st, bool leave_room = false);   // leave_room: two workgroups per CU instead of three (kernels.hip)
// Can k_hash_cells be launched with leave_room on the CURRENT device?  Asked once per context (cp2_init): the device's LDS per
// workgroup (never more than lds_cap when that is non-zero: test hook) must hold the kernel's own LDS plus the room, and the kernel's
// dynamic-LDS ceiling is raised to the room.  *why says what was found either way.  A launch itself is never retried.
bool hash_cells_can_leave_room(size_t lds_cap, std::string* why);
// the same with the workgroup size given (64 or 256): measurement tooling only
hipError_t launch_hash_cells_block(int block, const void* cells, size_t cell_size, size_t n_cells, void* out, hipStream_t st, bool leave_room = false);
// cells_per_slot == 0: one slot with seed `seed0`; otherwise global cell g belongs to slot g / cells_per_slot
// whose seed is seed0 + 1001 * slot.  list (device, may be NULL) selects explicit global cells.
// units_per_slot > 1: slots cut into units of `cells_per_slot` cells each (batch-local unit g / cells_per_slot is unit
// first_unit + that of the dataset, unit u lies in slot u / units_per_slot; seed0 = the seed of slot 0 of the dataset).
hipError_t launch_gen_fake_cells(uint64_t seed0, uint64_t cells_per_slot, uint64_t first, const uint64_t* list,
                                 size_t n_cells, size_t cell_size, void* out, hipStream_t st, uint64_t units_per_slot = 1,
                                 uint64_t first_unit = 0);
// Sampling + path lookup for proof inputs, all on the device (sample/bn254.nim:16-27, merkle.nim:21-42,86-100,
// types.nim:27-37): for item i < n_items (slot = slots ? slots[i] : slot0 + i, an index INSIDE the batch) and
// counter c = 1..ns:  cell = low bits of sponge2[entropy, slotRoot, c];  indices[i*ns+c-1] = cell;
// gcell[...] = slot * n_cells + cell;  rows[(i*ns+c-1)*md ..] = node-row index of each sibling on the merged path,
// ~0 where the reference pads with zero.  entropy: 32 bytes in device memory.
hipError_t launch_sample_paths(const TreeGeom& g, const void* nodes, const void* d_entropy, const uint64_t* slots, uint64_t slot0,
                               size_t n_items, uint32_t ns, uint32_t md, uint64_t* indices, uint64_t* gcell, uint64_t* rows,
                               hipStream_t st);
hipError_t launch_gather_rows(const void* src, const uint64_t* index, size_t nrows, size_t row_bytes, void* out,
                              hipStream_t st);

}  // namespace cp2k

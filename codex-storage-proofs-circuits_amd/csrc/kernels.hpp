// Launch wrappers of the HIP kernels in kernels.hip (device pointers, asynchronous on `st`).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace cp2k {

hipError_t launch_permute_batch(const void* in, void* out, size_t n, hipStream_t st);
// one Merkle layer of nseg trees; segment strides are in field elements
hipError_t launch_compress_layer(const void* in, void* out, size_t m_in, size_t nseg, bool bottom,
                                 size_t in_seg_stride, size_t out_seg_stride, hipStream_t st);
hipError_t launch_sponge2_felts(const void* felts, size_t nf, size_t nitems, void* out, hipStream_t st);
hipError_t launch_hash_cells(const void* cells, size_t cell_size, size_t n_cells, void* out, hipStream_t st);
// cells_per_slot == 0: one slot with seed `seed0`; otherwise global cell g belongs to slot g / cells_per_slot
// whose seed is seed0 + 1001 * slot.  list (device, may be NULL) selects explicit global cells.
hipError_t launch_gen_fake_cells(uint64_t seed0, uint64_t cells_per_slot, uint64_t first, const uint64_t* list,
                                 size_t n_cells, size_t cell_size, void* out, hipStream_t st);
hipError_t launch_gather_rows(const void* src, const uint64_t* index, size_t nrows, size_t row_bytes, void* out,
                              hipStream_t st);

}  // namespace cp2k

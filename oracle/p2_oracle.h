/* CPU ORACLE (test infrastructure, NOT product code) -- C restatement of the reference algorithm.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product path (libcodex_p2.so, include/codex_p2.h) never links, loads or calls it.
 *
 * Pinning: the permutation is pinned by the reference's one committed known-answer test
 * (reference/haskell/src/Poseidon2/Example.hs:13-19).  Everything above the permutation has no
 * committed expected values in the reference and the reference cannot be built here (no nim / ghc /
 * circom; arithmetic in un-vendored nim-poseidon2@4e2c6e6 + constantine@bc3845a): PARITY UNPINNED at
 * the nim-poseidon2 boundary for sponge / padding / Merkle keys / sampling / JSON, which follow the
 * in-tree Haskell + circom + README specification (see oracle/poseidon2_ref.py for the citations).
 *
 * Field elements cross this interface as 32-byte little-endian canonical integers in [0, r).
 */
#ifndef P2_ORACLE_H
#define P2_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Permutation.hs:40-45 */
void p2o_permute(const uint8_t in[96], uint8_t out[96]);
void p2o_permute_batch(const uint8_t* in, uint8_t* out, size_t n);
/* same, split over `threads` pthreads (cpu_baseline leg) */
void p2o_permute_batch_mt(const uint8_t* in, uint8_t* out, size_t n, int threads);

/* Merkle.hs:202-203 keyedCompression; key in {0,1,2,3} */
void p2o_compress(const uint8_t x[32], const uint8_t y[32], uint32_t key, uint8_t out[32]);

/* Sponge.hs:14-43 */
void p2o_sponge1_felts(const uint8_t* felts, size_t n, uint8_t out[32]);
void p2o_sponge2_felts(const uint8_t* felts, size_t n, uint8_t out[32]);

/* Slot.hs:243-270 + Sponge.hs:30-43: sponge2 over the 10*-padded 31-byte chunks */
void p2o_hash_bytes(const uint8_t* data, size_t len, uint8_t out[32]);
/* number of field elements a `len`-byte string becomes: (len + 1 + 30) / 31 */
size_t p2o_felts_per_bytes(size_t len);
/* the padded chunks themselves, n = p2o_felts_per_bytes(len) elements of 32 bytes */
void p2o_bytes_to_felts(const uint8_t* data, size_t len, uint8_t* out);

/* blocks/bn254.nim:23-29 over a contiguous array of cells */
void p2o_hash_cells(const uint8_t* cells, size_t cell_size, size_t n_cells, uint8_t* out);
void p2o_hash_cells_mt(const uint8_t* cells, size_t cell_size, size_t n_cells, uint8_t* out, int threads);

/* merkle/bn254.nim:24-63.  layers_out receives all layers bottom-first, concatenated;
 * returns the number of layers; layer_sizes[i] (if non-NULL) receives each layer's element count.
 * Total elements = p2o_merkle_total(n). */
size_t p2o_merkle_total(size_t n);
size_t p2o_merkle_tree(const uint8_t* leaves, size_t n, uint8_t* layers_out, size_t* layer_sizes);
void p2o_merkle_root(const uint8_t* leaves, size_t n, uint8_t out[32]);

/* slot.nim:23-32 genFakeCell; dataset.nim:32 parametricSlotSeed */
void p2o_gen_fake_cell(uint64_t seed, uint64_t idx, size_t cell_size, uint8_t* out);
uint64_t p2o_slot_seed(uint64_t seed, uint64_t slot_idx);

/* gen_input/bn254.nim:21-30: fake-data slot -> slot root (block trees, then the tree over block roots) */
void p2o_fake_slot_root(uint64_t slot_seed, size_t cell_size, size_t block_size, size_t n_cells,
                        uint8_t out[32], int threads);

/* the same slot's block roots (blocks/bn254.nim:60-67), n_cells / cellsPerBlock x 32 bytes */
void p2o_fake_slot_block_roots(uint64_t slot_seed, size_t cell_size, size_t block_size, size_t n_cells,
                               uint8_t* out, int threads);

/* sample/bn254.nim:16-24 */
uint64_t p2o_cell_index(const uint8_t entropy[32], const uint8_t slot_root[32], uint64_t n_cells, uint64_t counter);

#ifdef __cplusplus
}
#endif
#endif

/* codex_p2.h -- C ABI of libcodex_p2.so: the MI355X (gfx950) Poseidon2-BN254 proof-input engine.
 *
 * This is the drop-in boundary for the hot path of codex-storage/codex-storage-proofs-circuits'
 * `reference/nim/proof_input` tool (BN254 / Poseidon2 only).  The reference has no FFI for this path:
 * its seam is the set of nim-poseidon2 / constantine calls made from `reference/nim/proof_input/src`
 * (SURVEY.md section 8b).  Each entry point below names the reference call it replaces
 * (paths relative to the reference repository root).  INTEGRATION.md shows the Nim `importc` shim.
 *
 * Conventions
 *   - A field element (Fr of BN254) crosses the ABI as 32 bytes, little-endian, canonical integer in
 *     [0, r).  Inputs >= r are accepted and taken mod r.  Never Montgomery limbs.
 *   - Every function returns CP2_OK (0) or a negative cp2_status; nothing aborts across the ABI
 *     (the reference's `assert`s map to CP2_ERR_INVALID; the Nim shim turns non-zero into raiseAssert).
 *   - Plain functions take HOST pointers and are synchronous: inputs are copied to the GPU, the HIP
 *     kernels run, outputs are copied back before return.
 *   - `_dev` functions take HIP DEVICE pointers (16-byte aligned), enqueue on the context's stream and
 *     return without synchronising; use cp2_sync().
 *   - One context per host thread; contexts are independent.  There is no CPU fallback: without a
 *     usable gfx950 device cp2_init fails with CP2_ERR_NO_DEVICE.
 */
#ifndef CODEX_P2_H
#define CODEX_P2_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CP2_FELT_BYTES 32

/* ---- version of this boundary --------------------------------------------------------------
 * Callers bind the entry points BY NAME at load time (the Nim `dynlib` binding nim/codex_p2.nim, the ctypes binding, dlopen), so
 * nothing but this number tells a caller built against one header from a library built from another.
 *   MAJOR  changes when an existing entry point, struct or status code changes meaning or layout, or one is removed: a caller built
 *          for another major MUST refuse the library (every binding in this repository does, naming both numbers).  The library's
 *          SONAME carries it: libcodex_p2.so.<MAJOR>.
 *   MINOR  grows with every release that only ADDS entry points; a caller needs library minor >= the minor it was written against.
 * cp2_abi_version() returns what the LIBRARY was built from: (MAJOR << 16) | MINOR.  It touches no device and needs no context.
 * History: 1.0 = the 105 entry points of round 5 + this function + cp2_set_ingest's two rings (round 6).                        */
#define CP2_ABI_VERSION_MAJOR 1
#define CP2_ABI_VERSION_MINOR 0
#define CP2_ABI_VERSION ((CP2_ABI_VERSION_MAJOR << 16) | CP2_ABI_VERSION_MINOR)
int cp2_abi_version(void);

typedef enum cp2_status {
  CP2_OK = 0,
  CP2_ERR_INVALID = -1,    /* bad argument (the reference would fail an assert)            */
  CP2_ERR_NO_DEVICE = -2,  /* no usable HIP device / wrong architecture                    */
  CP2_ERR_HIP = -3,        /* a HIP runtime call failed; see cp2_last_error                */
  CP2_ERR_ALLOC = -4,      /* host or device allocation failed                             */
  CP2_ERR_IO = -5,         /* file could not be read / written                             */
  CP2_ERR_ALIGN = -6       /* a _dev pointer is not 16-byte aligned                        */
} cp2_status;

typedef struct cp2_ctx cp2_ctx;

/* ---- context ------------------------------------------------------------------------------ */
int cp2_init(int device, cp2_ctx** out);
void cp2_free(cp2_ctx* ctx);
/* Use a caller-owned HIP stream (hipStream_t) for all work of this context.  The handle is used as is:
 * NULL is HIP's legacy default stream (what torch.cuda.current_stream().cuda_stream returns by default).
 * cp2_reset_stream goes back to the context's own non-blocking stream. */
int cp2_set_stream(cp2_ctx* ctx, void* hip_stream);
int cp2_reset_stream(cp2_ctx* ctx);
int cp2_sync(cp2_ctx* ctx);
const char* cp2_strerror(int status);
const char* cp2_last_error(const cp2_ctx* ctx);
/* 1 when the library's kernels were built for the device of `ctx` (gfx950). */
int cp2_device_is_native(const cp2_ctx* ctx);
/* The environment variables the library reads are parsed STRICTLY: each holds exactly what it takes, or cp2_init / cp2_multi_init
 * fail with CP2_ERR_INVALID -- a mistyped knob is never read as "automatic".  cp2_check_environment (host only: no device is
 * touched, so it also answers on a box without a GPU) returns CP2_OK, or CP2_ERR_INVALID with the first offending variable, its
 * value and what it takes in `msg` (may be NULL).  The variables:
 *   CODEX_P2_GPUS        "all" | a device count | a comma-separated list of device indices      (cp2_multi_init)
 *   CODEX_P2_GATHER      "auto" | "rccl" | "copy" | "host"                                       (cp2_multi_set_policy)
 *   CODEX_P2_MIN_CELLS   a decimal number                                                        (cp2_multi_set_policy)
 *   CODEX_P2_SPLIT       0, 1 or a power of two                                                  (cp2_multi_set_split)
 *   CODEX_P2_KEEP_TREES  "auto" | "1" | "2" | "0"                                                (cp2_set_keep_trees)
 *   CODEX_P2_EXCHANGE_TIMEOUT_S  seconds the exchange of slot roots may take (default 120; 0 = no limit)
 *   CODEX_P2_STAGE_MB    MiB of device staging per chunk of generated cells (default 2048; cp2_init); a compact / roots-only build
 *                        holds half of it in tree nodes per batch
 *   CODEX_P2_MEM_LIMIT_MB        (tests) a cap, in MiB per device, on the device memory this process may hold through the library:
 *                        the automatic residency choice sees min(free, cap left) and an allocation beyond the cap fails like a real
 *                        out-of-memory, so all residency modes and the fallback between them can be reached on an empty 288 GB device
 * Test hooks the shipped library also reads (they exist so that branches a healthy MI355X never takes can be reached by the suite;
 * none of them can change a result, only which path produces it or turn a run into an error):
 *   CODEX_P2_TEST_LDS_LIMIT      a decimal number of bytes: the LDS per workgroup cp2_init's launch-shape decision sees (65536 makes
 *                        the hash launches of the streamed builds hold every workgroup slot instead of leaving room; parsed strictly)
 *   CODEX_P2_TEST_OPTIMISTIC     "1": the automatic residency choice starts at "every node" without looking at the device, so that
 *                        the step-down chain is what finds the mode that fits (anything else: ignored)
 *   CODEX_P2_TEST_EXCHANGE_FAULT "hang_init" | "hang_collective" | "init" | "collective" | "corrupt": the multi-device exchange of
 *                        slot roots stalls, fails or delivers wrong rows at that point (cp2_multi_*; anything else: ignored)
 * and the A/B knobs of the measurement tools, read leniently (anything but the value named means "off"): CP2_STREAM_SERIAL=1,
 * CP2_STREAM_RAMP=0, CP2_HASH_BLOCK=64, CP2_INGEST_COPY_STREAM=1 (the ingestion pipe's uploads on a stream of their own, as until
 * round 6, instead of on the chunk's own hashing stream), CP2_TRACE (any value: stage timings on stderr). */
int cp2_check_environment(char* msg, size_t msg_len);
/* Tuning of the host -> GPU ingestion pipe used by cp2_slot_trees_build_host, cp2_hash_cells (large inputs) and the
 * SlotFile data source (the reference reads one cell per call, reference/nim/proof_input/src/slot.nim:57-68):
 * host threads filling the pinned ring, ring depth (2..8) and bytes per chunk.  0 = keep the default
 * (environment CP2_INGEST_THREADS / CP2_INGEST_RING / CP2_INGEST_CHUNK_MB, else 8 threads, depth 3 and one full
 * residency of the hash kernel per chunk: 768 x 256 cells, 384 MiB at 2 KiB cells; the streamed builds, whose launches leave room
 * for their small kernels, take 768 MiB: three waves of workgroups at that occupancy).  `ring_depth` pinned host buffers
 * (free again as soon as their upload is done) feed ring_depth + 1 device buffers (one landing, two being hashed, slack).
 * A chunk is a range of the BATCH's cells: it holds many small slot files, or a piece of a large one.  Memory: ring_depth x chunk of
 * pinned host memory + (ring_depth + 1) x chunk of device memory while a build runs (2.25 + 3 GiB at the streamed builds' default),
 * cached by the context afterwards (cp2_trim); when the host or the device cannot give that much the chunk is halved until it fits. */
int cp2_set_ingest(cp2_ctx* ctx, int fill_threads, int ring_depth, size_t chunk_bytes);
/* SlotFile source: read the slot files with O_DIRECT (block-aligned requests straight into the pinned ring, no page-cache copy
 * and no eviction of what the cache holds): for files that are NOT cached -- a cached file reads faster through the cache.
 * on = 1 / 0; -1 = the environment variable CP2_INGEST_DIRECT (default off).  A file system that refuses O_DIRECT is read
 * buffered; results are identical either way. */
int cp2_set_ingest_direct(cp2_ctx* ctx, int on);
/* SlotFile source, opt-in: chunks of a slot file that sit in the PAGE CACHE go to the device without a CPU copy -- the file is
 * mmap'ed read-only, a chunk that starts and ends on page boundaries and whose pages are resident (mincore, sampled) is registered
 * with the runtime (the page-cache pages themselves are pinned) and uploaded straight from the mapping by the copy engine, instead
 * of being pread into the pinned ring first.  Registrations are held until the build's ingestion ends (at most 32 GiB at a time).
 * Any other chunk -- not cached, not on page boundaries, past the end of the file, or the runtime refuses to register file pages
 * (RLIMIT_MEMLOCK and the like) -- goes through the ring as before; the two mix chunk by chunk, results are identical.  On one
 * device the throughput equals the ring's (the hash kernel bounds both: DESIGN.md section 6); what it saves is host threads
 * copying and two thirds of the host memory traffic.  on = 1 / 0; -1 = the environment variable CP2_INGEST_MAPPED (default OFF).
 * Not used together with O_DIRECT. */
int cp2_set_ingest_mapped(cp2_ctx* ctx, int on);
/* Memory a long-lived context holds.  Scratch blocks (device staging, pinned landing zones) are cached per context so that
 * repeated calls stop allocating: up to 6 GiB of device memory and 4 GiB of PINNED host memory stay with the context after
 * the calls that needed them.  cp2_trim waits for the context's streams and gives all cached blocks back to the system
 * (blocks still referenced by live proof inputs return when those are freed); the next call allocates again. */
int cp2_trim(cp2_ctx* ctx);
/* Host memory of the streamed proof-input path (cp2_dataset_build_streamed): the JSON body of every local slot (about 0.7 MB
 * at nSamples = 100, cellSize = 2048) is kept until the dataset is freed.  Bodies beyond `max_resident_bytes` per dataset are
 * written to files instead and read back by cp2_dataset_export_streamed / cp2_dataset_streamed_json.  The files hold sampled
 * cell data: they live in a private directory "<spill_dir>/cp2_bodies_XXXXXX" made by mkdtemp (mode 0700), each created with
 * O_EXCL | O_NOFOLLOW and mode 0600; files and directory are removed by cp2_dataset_free.  Defaults: 4 GiB (environment
 * CP2_BODY_BUDGET_MB), spill_dir NULL = $TMPDIR or /tmp.  max_resident_bytes = 0 keeps the current budget; (size_t)-1 = never
 * spill.  A spill that fails is CP2_ERR_IO with the path in cp2_last_error. */
int cp2_set_body_budget(cp2_ctx* ctx, size_t max_resident_bytes, const char* spill_dir);
/* Device memory of cp2_dataset_build.  Three ways to hold the slot trees of a dataset, same results from each:
 *   1  every node resident (3.1 % of the data: 256 MiB per 8 GiB slot): a proof input for any entropy costs two permutations
 *      per sample plus gathers;
 *   2  COMPACT: of every slot tree only the part from the block roots up stays (8 MiB per 8 GiB slot, 1/32 of the above); the
 *      bottom of each path is recomputed from the <= nSamples touched network blocks (regenerated, or read from the slot file:
 *      100 x 64 KiB), whose rebuilt roots are checked against the stored ones (a mismatch is CP2_ERR_IO: the slot data changed):
 *      a few ms per proof input; 4096 slots of 8 GiB hold 32 GiB of device memory instead of 1 TiB;
 *   0  ROOTS ONLY: the 32-byte slot roots stay; cp2_proof_input_generate rebuilds the whole tree of the slot it proves (0.2 s per
 *      8 GiB slot; the reference rebuilds it once per SAMPLE, gen_input/bn254.nim:57).
 * For 2 and 0 the trees are built batch by batch (about 2 GiB of nodes) in the context's scratch, what is kept is copied out, the
 * rest dropped (the copy-out of one batch overlaps the hashing of the next).  mode -1 (default): the environment variable
 * CODEX_P2_KEEP_TREES ("0" / "1" / "2"; "auto" = unset), else the most that fits what the device has free (1, else 2, else 0) --
 * divided by the number of contexts a cp2_multi has placed on that device.  The figure is a snapshot: another tenant of the
 * device (a second process, a framework's caching allocator) can take the memory between the choice and the allocation.  When
 * the mode was chosen AUTOMATICALLY and the build then fails with CP2_ERR_ALLOC, everything is freed, the context's cached
 * scratch is handed back (cp2_trim) and the build is retried one mode down (1 -> 2 -> 0); CP2_TRACE says so, and
 * cp2_dataset_keeps_trees tells what a built dataset did.  A mode the caller named is never changed.  The streamed build follows
 * the same rule (the bodies of a batch of slots are made while its trees exist: every proof input of 4096 slots of 8 GiB in one
 * pass over the data).  cp2_dataset_build_cached caches what the dataset keeps: every node, or -- 1/32 of that -- the compact
 * layers, or the roots; a later run loads them and, compact, proves from the touched blocks alone.  On a roots-only dataset every
 * cp2_proof_input_generate costs one slot rebuild, and the batch / export calls one per slot: to get the proof inputs of ALL
 * slots use the streamed build. */
int cp2_set_keep_trees(cp2_ctx* ctx, int mode);

/* ---- a1: Poseidon2 t=3 permutation --------------------------------------------------------- */
/* replaces nim-poseidon2 `perm` as specified by reference/haskell/src/Poseidon2/Permutation.hs:40-45.
 * in/out: n states of 3 field elements (n x 96 bytes).  Host arrays of more than 2^20 states stream through the device in
 * chunks (upload, kernel and download overlapped); arrays the caller has pinned (hipHostMalloc, hipHostRegister) are read and
 * written in place by the copy engines, pageable ones pass through a pinned ring filled by host threads. */
int cp2_permute_batch(cp2_ctx* ctx, const uint8_t* in, uint8_t* out, size_t n);
int cp2_permute_batch_dev(cp2_ctx* ctx, const void* d_in, void* d_out, size_t n);

/* ---- a6: keyed compression ----------------------------------------------------------------- */
/* replaces `compress(x, y, key = toF(key))`, reference/nim/proof_input/src/merkle/bn254.nim:18,50,53.
 * xy: n pairs (n x 64 bytes); key in {0,1,2,3}; out: n x 32 bytes. */
int cp2_compress_batch(cp2_ctx* ctx, const uint8_t* xy, uint32_t key, uint8_t* out, size_t n);

/* ---- a3: sponge over field elements --------------------------------------------------------- */
/* replaces `Sponge.digest(seq[F], rate = 2)`, reference/nim/proof_input/src/sample/bn254.nim:23. */
int cp2_sponge2_felts(cp2_ctx* ctx, const uint8_t* felts, size_t n, uint8_t out[32]);
/* batched: nitems inputs of nf elements each -> nitems digests */
int cp2_sponge2_felts_batch(cp2_ctx* ctx, const uint8_t* felts, size_t nf, size_t nitems, uint8_t* out);
int cp2_sponge2_felts_batch_dev(cp2_ctx* ctx, const void* d_felts, size_t nf, size_t nitems, void* d_out);

/* ---- a4: bytes -> field elements (host only, no device work) --------------------------------- */
/* replaces the iterator `elements(bytes, F)`, reference/nim/proof_input/src/json/bn254.nim:11,25
 * (10* byte padding, 31-byte little-endian chunks; reference/haskell/src/Slot.hs:243-270). */
size_t cp2_felts_per_bytes(size_t len);
int cp2_bytes_to_felts(const uint8_t* data, size_t len, uint8_t* out /* cp2_felts_per_bytes(len) x 32 */);

/* ---- a5: hashCell ---------------------------------------------------------------------------- */
/* replaces `Sponge.digest(cellData, rate = 2)` over bytes, reference/nim/proof_input/src/blocks/bn254.nim:27.
 * cells: n_cells contiguous cells of cell_size bytes; out: n_cells x 32 bytes. */
int cp2_hash_cells(cp2_ctx* ctx, const uint8_t* cells, size_t cell_size, size_t n_cells, uint8_t* out);
int cp2_hash_cells_dev(cp2_ctx* ctx, const void* d_cells, size_t cell_size, size_t n_cells, void* d_out);
/* one byte string of any length (same function with n_cells = 1) */
int cp2_hash_bytes(cp2_ctx* ctx, const uint8_t* data, size_t len, uint8_t out[32]);

/* ---- a7: Merkle tree -------------------------------------------------------------------------- */
/* replaces `merkleTreeBN254(xs)`, reference/nim/proof_input/src/merkle/bn254.nim:62-63 (all layers,
 * bottom first; keys 1/0 and odd keys 3/2; a singleton still gets one compression).
 * cp2_merkle_total(n) = number of elements over all layers; layers_out holds that many x 32 bytes.
 * layer_sizes (may be NULL) receives the element count of each layer; *n_layers their number. */
size_t cp2_merkle_total(size_t n);
size_t cp2_merkle_num_layers(size_t n);
int cp2_merkle_tree(cp2_ctx* ctx, const uint8_t* leaves, size_t n, uint8_t* layers_out, size_t* layer_sizes,
                    size_t* n_layers);
/* nseg independent trees of n leaves each (d_leaves: nseg x n elements, tree after tree).
 * d_layers_out: nseg x cp2_merkle_total(n) elements, LAYER-major: layer k of all trees is contiguous
 * (it starts at element nseg * (size_0 + ... + size_{k-1})), tree s at offset s * size_k inside it.
 * d_layers_out may equal d_leaves (the leaves are then layer 0 in place). */
int cp2_merkle_trees_dev(cp2_ctx* ctx, const void* d_leaves, size_t n, size_t nseg, void* d_layers_out);
/* replaces `Merkle.digest(xs)`, reference/nim/proof_input/src/merkle/bn254.nim:20 */
int cp2_merkle_root(cp2_ctx* ctx, const uint8_t* leaves, size_t n, uint8_t out[32]);

/* ---- a10: fake slot data ---------------------------------------------------------------------- */
/* replaces `genFakeCell`, reference/nim/proof_input/src/slot.nim:23-32, for cells first..first+n-1.
 * `seed` is the slot's own seed; cp2_slot_seed = parametricSlotSeed, dataset.nim:32. */
uint64_t cp2_slot_seed(uint64_t dataset_seed, uint64_t slot_idx);
int cp2_gen_fake_cells(cp2_ctx* ctx, uint64_t seed, uint64_t first, size_t n, size_t cell_size, uint8_t* out);
int cp2_gen_fake_cells_dev(cp2_ctx* ctx, uint64_t seed, uint64_t first, size_t n, size_t cell_size, void* d_out);

/* ---- a12: sampling ---------------------------------------------------------------------------- */
/* replaces `cellIndices`, reference/nim/proof_input/src/sample/bn254.nim:16-27 (counters 1..n_samples).
 * n_cells must be a power of two and at least 2 (`extractLowBits` asserts k > 0, types/bn254.nim:48). */
int cp2_cell_indices(cp2_ctx* ctx, const uint8_t entropy[32], const uint8_t slot_root[32], uint64_t n_cells,
                     size_t n_samples, uint64_t* out);

/* ---- a8/a9/a13: slot trees (device resident) --------------------------------------------------- */
/* A batch of `n_slots` slot trees of identical geometry, as built by `buildSlotTreeFull`,
 * reference/nim/proof_input/src/gen_input/bn254.nim:21-30: per block a tree over its cell hashes
 * (bottom key 1), then per slot a tree over the block roots (bottom key 1 again). */
typedef struct cp2_slot_trees cp2_slot_trees;

/* slots first_slot..first_slot+n_slots-1 of a fake-data dataset (cells generated on the device) */
int cp2_slot_trees_build_fake(cp2_ctx* ctx, uint64_t dataset_seed, uint64_t first_slot, size_t n_slots,
                              size_t cell_size, size_t block_size, size_t n_cells, cp2_slot_trees** out);
/* The same for UNITS: every slot cut into `units_per_slot` (a power of two) pieces of `cells_per_unit` cells -- whole blocks,
 * at least two, a power of two of them -- and this batch holding units first_unit .. first_unit + n_units - 1 of the dataset
 * (unit u = cells [(u mod units_per_slot) x cells_per_unit, + cells_per_unit) of slot u / units_per_slot).  The root of a unit is
 * the node of its slot's tree above those cells, so several devices can share ONE slot (section e, "by units"; SURVEY.md 8e:
 * "within one very large slot the same scheme applies one level down").  cp2_slot_trees_roots / _paths address units as
 * slots of cells_per_unit cells.  cp2_slot_trees_save / _load carry unit batches too (the file records units_per_slot). */
int cp2_slot_trees_build_fake_units(cp2_ctx* ctx, uint64_t dataset_seed, uint64_t units_per_slot, uint64_t first_unit, size_t n_units,
                                    size_t cell_size, size_t block_size, size_t cells_per_unit, cp2_slot_trees** out);
/* units of slot files "<file_base><slot>.dat" (dataset.nim:34): unit u is read at byte offset (u mod units_per_slot) x
 * cells_per_unit x cell_size of the file of slot u / units_per_slot */
int cp2_slot_trees_build_file_units(cp2_ctx* ctx, const char* file_base, uint64_t units_per_slot, uint64_t first_unit, size_t n_units,
                                    size_t cell_size, size_t block_size, size_t cells_per_unit, cp2_slot_trees** out);
/* n_slots slots whose cells are already in device memory, slot-major (n_slots x n_cells x cell_size bytes).
 * A `_dev`-style call: the hashing is ENQUEUED on the context's stream and the call returns without synchronising
 * (the cells must stay valid until then).  cp2_sync, cp2_slot_trees_roots and cp2_slot_trees_paths synchronise;
 * a launch failure is reported by whichever of them runs first (cp2_last_error). */
int cp2_slot_trees_build_dev(cp2_ctx* ctx, const void* d_cells, size_t n_slots, size_t cell_size, size_t block_size,
                             size_t n_cells, cp2_slot_trees** out);
/* same from host memory (streamed through a pinned staging buffer) */
int cp2_slot_trees_build_host(cp2_ctx* ctx, const uint8_t* cells, size_t n_slots, size_t cell_size,
                              size_t block_size, size_t n_cells, cp2_slot_trees** out);
void cp2_slot_trees_free(cp2_slot_trees* t);
/* Persisted trees (the reference recomputes every tree on every run and once more per
 * sample, gen_input/bn254.nim:42,57).  The file holds the geometry, the data-source description and every
 * node; loading it skips all cell hashing.  Trees built from caller memory (build_dev / build_host) load
 * without a cell source: paths work, sampled-cell retrieval needs cp2_slot_trees_attach_cells first. */
int cp2_slot_trees_save(cp2_slot_trees* t, const char* path);
int cp2_slot_trees_load(cp2_ctx* ctx, const char* path, cp2_slot_trees** out);
int cp2_slot_trees_attach_cells(cp2_slot_trees* t, const uint8_t* host_cells, const void* dev_cells);
size_t cp2_slot_trees_count(const cp2_slot_trees* t);
size_t cp2_slot_trees_depth(const cp2_slot_trees* t);   /* log2(cellsPerBlock) + log2(nBlocks) */
/* roots of all slots in the batch (n_slots x 32 bytes): `treeRoot(bigTree)`, merkle.nim:14-17 */
int cp2_slot_trees_roots(cp2_slot_trees* t, uint8_t* out);
/* device pointer to the same roots (n_slots x 32 bytes, canonical), valid until free */
const void* cp2_slot_trees_roots_dev(const cp2_slot_trees* t);
/* merged bottom+top Merkle paths (merkleProof + mergeMerkleProofs, merkle.nim:21-42,86-100) of
 * n cells of slot `slot` (index inside the batch), padded with zeros to max_depth (types.nim:27-37).
 * out: n x max_depth x 32 bytes; leaf_hashes (may be NULL): n x 32 bytes. */
int cp2_slot_trees_paths(cp2_slot_trees* t, size_t slot, const uint64_t* cell_idx, size_t n, size_t max_depth,
                         uint8_t* out, uint8_t* leaf_hashes);

/* ---- a14/a15: proof input ----------------------------------------------------------------------- */
/* mirrors GlobalConfig + DataSetConfig, reference/nim/proof_input/src/types.nim:82-101 */
typedef struct cp2_config {
  int32_t max_depth;        /* GlobalConfig.maxDepth                                 */
  int32_t max_log2_nslots;  /* GlobalConfig.maxLog2NSlots                            */
  uint64_t cell_size;       /* GlobalConfig.cellSize                                 */
  uint64_t block_size;      /* GlobalConfig.blockSize                                */
  uint64_t n_slots;         /* DataSetConfig.nSlots                                  */
  uint64_t n_cells;         /* DataSetConfig.nCells (power of two)                   */
  uint64_t n_samples;       /* DataSetConfig.nSamples                                */
  uint64_t seed;            /* DataSource FakeData seed (used when file_base == NULL) */
  const char* file_base;    /* DataSource SlotFile base name: slot k = "<base><k>.dat" (dataset.nim:34) */
} cp2_config;

/* A dataset whose slot trees are built: slot roots + dataset tree (gen_input/bn254.nim:41-51). */
typedef struct cp2_dataset cp2_dataset;
/* Builds the trees of slots [first_slot, first_slot + n_local) on this GPU.  For a single GPU pass
 * first_slot = 0, n_local = cfg->n_slots.  */
int cp2_dataset_build(cp2_ctx* ctx, const cp2_config* cfg, uint64_t first_slot, uint64_t n_local, cp2_dataset** out);
/* cp2_dataset_build with what it keeps of the slot trees (cp2_set_keep_trees: every node, the compact layers, or the roots) cached
 * in `cache_path`: read when present, intact and matching the configuration -- and, for the SlotFile source, slot files of
 * unchanged size and mtime --, else built and written (to a temporary name, then renamed) */
int cp2_dataset_build_cached(cp2_ctx* ctx, const cp2_config* cfg, uint64_t first_slot, uint64_t n_local,
                             const char* cache_path, cp2_dataset** out);
void cp2_dataset_free(cp2_dataset* ds);
/* roots of the local slots (n_local x 32 bytes) */
int cp2_dataset_local_roots(cp2_dataset* ds, uint8_t* out);
/* Supply the roots of ALL n_slots slots (after the multi-GPU gather; single GPU: pass NULL to use the
 * local ones) and build the dataset-level tree on the GPU. */
int cp2_dataset_set_roots(cp2_dataset* ds, const uint8_t* all_roots);
/* The same exchange without host copies of the roots (the gather of SURVEY.md 8e is device to device: RCCL, a torch tensor):
 * cp2_dataset_local_roots_dev = device pointer to the n_local x 32 bytes of local roots (valid until the dataset is freed);
 * cp2_dataset_copy_local_roots_dev ENQUEUES a device-to-device copy of them into the caller's buffer on the context's stream
 * (cp2_sync before another stream reads it); cp2_dataset_set_roots_dev takes ALL n_slots roots in device memory (16-byte
 * aligned, any device of the node), builds the dataset tree from them and synchronises. */
const void* cp2_dataset_local_roots_dev(const cp2_dataset* ds);
int cp2_dataset_copy_local_roots_dev(cp2_dataset* ds, void* d_out);
int cp2_dataset_set_roots_dev(cp2_dataset* ds, const void* d_all_roots);
int cp2_dataset_root(cp2_dataset* ds, uint8_t out[32]);
int cp2_dataset_keeps_trees(const cp2_dataset* ds);
/* the slot range and the context a dataset was built with */
int cp2_dataset_range(const cp2_dataset* ds, uint64_t* first_slot, uint64_t* n_local);
cp2_ctx* cp2_dataset_ctx(const cp2_dataset* ds);

/* SlotProofInput, reference/nim/proof_input/src/types.nim:52-60 */
typedef struct cp2_proof_input cp2_proof_input;
/* replaces `generateProofInputBN254`, reference/nim/proof_input/src/gen_input/bn254.nim:35-79, for a
 * slot that is local to `ds`. */
int cp2_proof_input_generate(cp2_dataset* ds, uint64_t slot_idx, const uint8_t entropy[32], cp2_proof_input** out);
/* The same for n slots of `ds` at once (config 4: thousands of slots sharing one dataset tree): one
 * sampling launch, one path gather and one cell fetch for all of them.  out[0..n) receive the objects. */
int cp2_proof_inputs_generate_batch(cp2_dataset* ds, const uint64_t* slot_idx, size_t n, const uint8_t entropy[32],
                                    cp2_proof_input** out);
void cp2_proof_input_free(cp2_proof_input* p);
/* accessors (all field elements canonical 32-byte LE) */
int cp2_proof_input_roots(const cp2_proof_input* p, uint8_t dataset_root[32], uint8_t slot_root[32], uint8_t entropy[32]);
size_t cp2_proof_input_nsamples(const cp2_proof_input* p);
const uint64_t* cp2_proof_input_cell_indices(const cp2_proof_input* p);
const uint8_t* cp2_proof_input_cell_data(const cp2_proof_input* p);     /* nSamples x cellSize bytes      */
const uint8_t* cp2_proof_input_merkle_paths(const cp2_proof_input* p);  /* nSamples x maxDepth x 32 bytes */
const uint8_t* cp2_proof_input_slot_proof(const cp2_proof_input* p);    /* maxLog2NSlots x 32 bytes       */
/* hash of each sampled cell = `leafValue` of its merged proof (merkle.nim:86-100); nSamples x 32 bytes, or NULL for an
 * object made by cp2_proof_input_create without them */
const uint8_t* cp2_proof_input_leaf_hashes(const cp2_proof_input* p);
/* A proof input assembled from caller arrays -- what a `SlotProofInput[Hash]` value holds (types.nim:52-60) -- so that
 * `exportProofInputBN254(hashcfg, fname, prfInput)` (json/bn254.nim:77) can hand any such value to the byte-exact writer.
 * cfg supplies maxDepth, maxLog2NSlots, cellSize, nCells, nSlots; arrays as the accessors above return them; cell_indices
 * and leaf_hashes may be NULL (neither is printed).  Everything is copied; free with cp2_proof_input_free. */
int cp2_proof_input_create(const cp2_config* cfg, uint64_t slot_idx, const uint8_t dataset_root[32], const uint8_t entropy[32],
                           const uint8_t slot_root[32], const uint8_t* slot_proof, size_t n_samples, const uint64_t* cell_indices,
                           const uint8_t* cell_data, const uint8_t* merkle_paths, const uint8_t* leaf_hashes,
                           cp2_proof_input** out);
/* replaces `exportProofInputBN254`, reference/nim/proof_input/src/json/bn254.nim:57-78: byte-exact JSON */
int cp2_proof_input_write_json(const cp2_proof_input* p, const char* path);
/* the same text into a malloc'ed buffer (caller frees with cp2_free_buffer) */
int cp2_proof_input_json(const cp2_proof_input* p, char** text, size_t* len);
void cp2_free_buffer(void* p);
/* Serialise n proof inputs on `threads` host threads and write paths[i] (paths == NULL or paths[i] == NULL:
 * serialise only).  total_bytes (may be NULL) receives the summed text length. */
int cp2_proof_inputs_write_json_batch(const cp2_proof_input* const* ps, size_t n, const char* const* paths, int threads,
                                      uint64_t* total_bytes);
/* All of it for many slots as a two-stage pipeline (GPU: sampling + gathers of batch k+1; host threads: JSON of
 * batch k).  dir == NULL: serialise only; else "<dir>/input_<slot>.json" is written per slot.  batch == 0: 512. */
int cp2_dataset_export_proof_inputs(cp2_dataset* ds, const uint64_t* slot_idx, size_t n, const uint8_t entropy[32],
                                    const char* dir, int threads, size_t batch, uint64_t* total_bytes);
/* The whole of `generateProofInputBN254` + `exportProofInputBN254` for EVERY local slot with the entropy known up front
 * (reference/nim/proof_input/src/gen_input/bn254.nim:35-79, json/bn254.nim:57-78), as one overlapped pipeline: slot trees
 * are built `group_slots` at a time (0: what fills the 2 GiB staging chunk; the last groups shrink so that little formatting is
 * left un-overlapped; with the SlotFile source the slots are read in ring turns of many small files or pieces of large ones, and
 * the groups are what the layer passes and the sampling are done by), and while later slots are still hashing the
 * finished ones are sampled (sample/bn254.nim:16-27), their paths and cells gathered on the device, downloaded into pinned
 * memory and formatted (cellData + merklePaths) on `threads` host threads.  Only the lines that need all slot roots
 * (dataSetRoot, slotProof: gen_input/bn254.nim:49-51,72) are left for cp2_dataset_export_streamed.  The returned dataset
 * is a normal cp2_dataset (roots, set_roots, proof inputs for other entropies all work).
 * Host memory: the formatted bodies stay with the dataset until cp2_dataset_free, in memory up to the context's body budget
 * and in spill files beyond it (cp2_set_body_budget above). */
int cp2_dataset_build_streamed(cp2_ctx* ctx, const cp2_config* cfg, uint64_t first_slot, uint64_t n_local,
                               const uint8_t entropy[32], int threads, size_t group_slots, cp2_dataset** out);
/* Finish what cp2_dataset_build_streamed prepared: needs the dataset tree (cp2_dataset_set_roots; implied when all slots
 * are local).  dir == NULL: format only; else "<dir>/input_<slot>.json" per local slot.  Text identical to
 * cp2_proof_input_json of the same slot and entropy. */
int cp2_dataset_export_streamed(cp2_dataset* ds, const char* dir, int threads, uint64_t* total_bytes);
/* the finished text of one prepared slot in a malloc'ed buffer (cp2_free_buffer) */
int cp2_dataset_streamed_json(cp2_dataset* ds, uint64_t slot_idx, char** text, size_t* len);
/* replaces `writeCircomMainComponent`, reference/nim/proof_input/src/cli.nim:186-204 */
int cp2_write_circom_main(const cp2_config* cfg, const char* path);

/* ---- e: every GPU of the node behind one handle ------------------------------------------------------ */
/* `generateProofInput` hashes every slot of the dataset before it proves one (reference/nim/proof_input/src/gen_input/
 * bn254.nim:41-42), and slots are independent until the dataset tree (:49-51).  A cp2_multi holds one context per device;
 * a dataset built through it is cut into contiguous slot ranges (cp2_shard_range: the first n mod world ranges hold one
 * slot more), each device builds its range on its own host thread with no communication, then ONE exchange -- an all-gather
 * of the 32-byte slot roots, device to device (RCCL over xGMI: in-place ncclAllGather on every context's stream) -- and
 * every device builds the identical dataset tree and serves the proof inputs of its own slots.  One process, no launcher:
 * the caller makes the same calls on a one-GPU and on an eight-GPU node.
 *   - librccl is opened at run time, and only when at least two distinct devices hold a shard.  Without it, or when a device
 *     index repeats (two contexts on one device), the roots are gathered through host memory instead (1 MiB at 32 768 slots);
 *     cp2_multi_gather_mode names what the last build did ("rccl (...)", "host (<why>)", "copy (...)", "none (one shard ...)").
 *   - The exchange is VERIFIED, whatever carried it: every device must find its own slot roots at its own rows of the list it
 *     received, and all devices must compute the same dataset root (one 32-byte-per-slot download per shard) -- a wrong
 *     rank-to-device mapping or a misplaced block is CP2_ERR_HIP ("exchange verification failed ..."), never a wrong
 *     dataSetRoot.  It is also BOUNDED: communicator creation and the collective itself are given CODEX_P2_EXCHANGE_TIMEOUT_S
 *     seconds (default 120); one that does not return is an error that says so (its buffers are abandoned, RCCL is not used again
 *     in this process).  In the automatic mode a device path that fails either way is retried once through host memory, and
 *     cp2_multi_gather_mode says that it was; a way asked for by name is never replaced.
 *   - Small datasets use fewer devices: a device gets a shard only when there is at least `min_cells_per_device` cells of
 *     hashing for it (default: one residency of the hash kernel, 768 x 256 cells -- a device with less finishes no sooner),
 *     so the reference's default run (11 slots x 512 cells, workflow/params.sh) stays on one GPU and pays one context.
 *   - Contexts are created on first use; cp2_multi_ctx(m, i) hands one out for the seam calls and the tuning knobs.
 *   - One handle per host thread (like a context): calls on one cp2_multi / cp2_multi_dataset are not to be made concurrently;
 *     the library starts and joins its per-device threads inside each call.
 *   - Datasets of FEW, LARGE slots are cut BY UNITS instead of by whole slots: when whole slots would leave the busiest device
 *     more than 6 % above its share (11 slots on 8 GPUs: 2 against 1.375; ONE 128 GiB slot on 8 GPUs), every slot is cut into
 *     S = 2^s units of nCells / S cells (cp2_slot_trees_build_*_units), the nSlots x S units are dealt out contiguously, the
 *     unit roots exchanged, and the log2 S upper layers of every slot tree plus the dataset tree built once; a proof input then
 *     takes the bottom of each path from whichever device holds the sampled cell; many proof inputs at once
 *     (cp2_multi_dataset_export_proof_inputs) take ONE batched gather per device, the devices in parallel.  Every kind of build
 *     follows this plan.  A STREAMED build cut by units is two-phase -- sampling needs the slot root, which exists only after the
 *     exchange of unit roots, so nothing of a proof input can be made while later units hash: the balanced unit build, the
 *     exchange, then every slot's input.json from the devices that hold its units, kept as text for _export_streamed /
 *     _streamed_json.  cp2_multi_set_split / CODEX_P2_SPLIT override (1 = whole slots only: the overlapped per-slot pipeline). */
typedef struct cp2_multi cp2_multi;
typedef struct cp2_multi_dataset cp2_multi_dataset;
enum { CP2_GATHER_AUTO = 0, CP2_GATHER_RCCL = 1, CP2_GATHER_HOST = 2, CP2_GATHER_COPY = 3 };
/* devices: n_dev HIP device indices (an index may repeat: several contexts on one device).  n_dev = 0: the environment
 * variable CODEX_P2_GPUS ("all" = every visible gfx950 device, "<count>" = the first <count> visible devices, or a
 * comma-separated index list), else ONE device: the first visible gfx950.  Several devices are OPT-IN -- by the variable or by
 * an explicit list -- until the exchange between two real devices has a committed record (this pipeline's GPU boxes hold one
 * device).  Every CODEX_P2_* variable is checked here (cp2_check_environment): a value that is not what
 * its variable takes is CP2_ERR_INVALID, not guessed at. */
int cp2_multi_init(const int* devices, int n_dev, cp2_multi** out);
/* frees the handle, its contexts and communicators: free every cp2_multi_dataset (and proof input) made through it first */
void cp2_multi_free(cp2_multi* m);
int cp2_multi_count(const cp2_multi* m);
int cp2_multi_device(const cp2_multi* m, int i);          /* HIP device index of entry i, -1 out of range */
cp2_ctx* cp2_multi_ctx(cp2_multi* m, int i);              /* NULL when the device is unusable            */
const char* cp2_multi_last_error(const cp2_multi* m);
const char* cp2_multi_gather_mode(const cp2_multi* m);
/* gather: CP2_GATHER_AUTO (RCCL when possible, else host), CP2_GATHER_RCCL (fail with CP2_ERR_INVALID when impossible),
 * CP2_GATHER_HOST, CP2_GATHER_COPY (the all-gather written out as device-to-device copies, hipMemcpyPeerAsync pair by pair,
 * through the RCCL path's own buffers, padded layout and compaction: no library, any mix of devices, repeated ones included);
 * cp2_multi_init reads the environment variable CODEX_P2_GATHER ("rccl" / "host" / "copy") as the initial value.  min_cells_per_device: 0 = the default above (or the environment variable CODEX_P2_MIN_CELLS, read by
 * cp2_multi_init); 1 = always spread over every device. */
int cp2_multi_set_policy(cp2_multi* m, int gather, uint64_t min_cells_per_device);
/* units per slot for cp2_multi_dataset_build: 0 = choose (above; or the environment variable CODEX_P2_SPLIT), 1 = whole slots
 * only, a power of two = exactly that many (ignored where the geometry does not allow it: units are >= 2 whole blocks). */
int cp2_multi_set_split(cp2_multi* m, int64_t units_per_slot);
/* the split rule: contiguous ranges, the first (n_items mod world) ranks hold one item more */
void cp2_shard_range(uint64_t n_items, int rank, int world, uint64_t* first, uint64_t* count);
/* The plan cp2_multi_dataset_build follows for `cfg` on `n_devices` devices (host-only arithmetic, no GPU needed): how many of
 * them get a shard and into how many units every slot is cut (1: whole slots).  min_cells_per_device / units_per_slot as for
 * cp2_multi_set_policy / cp2_multi_set_split (0 = the defaults). */
int cp2_multi_plan(const cp2_config* cfg, int n_devices, uint64_t min_cells_per_device, int64_t units_per_slot, int* n_shards,
                   uint64_t* units_per_slot_out);
/* cp2_dataset_build / _build_cached / _build_streamed for ALL cfg->n_slots slots over the devices of `m`, including the
 * exchange (verified and bounded, above) and the dataset tree on every device.  Cached: shard i of n uses "<cache_path>.shard<i>of<n>" (one shard: the path
 * itself; cut by S units per slot: "<cache_path>.units<S>.shard<i>of<n>", the unit trees of that shard).  Streamed: `threads` formatting threads in total, divided over the shards. */
int cp2_multi_dataset_build(cp2_multi* m, const cp2_config* cfg, cp2_multi_dataset** out);
int cp2_multi_dataset_build_cached(cp2_multi* m, const cp2_config* cfg, const char* cache_path, cp2_multi_dataset** out);
int cp2_multi_dataset_build_streamed(cp2_multi* m, const cp2_config* cfg, const uint8_t entropy[32], int threads, size_t group_slots,
                                     cp2_multi_dataset** out);
void cp2_multi_dataset_free(cp2_multi_dataset* mds);
int cp2_multi_dataset_shards(const cp2_multi_dataset* mds);
/* 1 when the dataset is cut by whole slots, else the number of units every slot was cut into */
uint64_t cp2_multi_dataset_units_per_slot(const cp2_multi_dataset* mds);
/* shard i: device index and its range of slots (or of units); by whole slots also its dataset (owned by mds; every
 * single-dataset call works on it), by units NULL */
cp2_dataset* cp2_multi_dataset_shard(cp2_multi_dataset* mds, int i, int* device, uint64_t* first, uint64_t* count);
int cp2_multi_dataset_root(cp2_multi_dataset* mds, uint8_t out[32]);
int cp2_multi_dataset_slot_roots(cp2_multi_dataset* mds, uint8_t* out /* n_slots x 32 */);
/* replaces `generateProofInputBN254`, gen_input/bn254.nim:35-79, on whichever device holds `slot_idx` */
int cp2_multi_proof_input_generate(cp2_multi_dataset* mds, uint64_t slot_idx, const uint8_t entropy[32], cp2_proof_input** out);
/* cp2_dataset_export_proof_inputs / _export_streamed / _streamed_json over all shards (every device works through its own
 * slots concurrently; `threads` host threads in total) */
int cp2_multi_dataset_export_proof_inputs(cp2_multi_dataset* mds, const uint64_t* slot_idx, size_t n, const uint8_t entropy[32],
                                          const char* dir, int threads, size_t batch, uint64_t* total_bytes);
int cp2_multi_dataset_export_streamed(cp2_multi_dataset* mds, const char* dir, int threads, uint64_t* total_bytes);
int cp2_multi_dataset_streamed_json(cp2_multi_dataset* mds, uint64_t slot_idx, char** text, size_t* len);

#ifdef __cplusplus
}
#endif
#endif /* CODEX_P2_H */

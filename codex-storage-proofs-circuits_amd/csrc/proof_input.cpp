// Datasets and proof inputs behind the C ABI (include/codex_p2.h).
//
// Mirrors reference/nim/proof_input/src/gen_input/bn254.nim:35-79 (generateProofInput), sample/bn254.nim:16-27
// (cellIndices), merkle.nim:21-42,86-100 (merkleProof, mergeMerkleProofs), types.nim:27-37 (padMerkleProof) and
// json/bn254.nim:19-78 + json/shared.nim:17-25 (exportProofInput).  Sampling, path lookup, gathers and cell
// regeneration run on the device; what stays on the host is byte packing and text formatting.
//
// Two ways through: (1) build the dataset, then generate proof inputs for any entropy (objects with accessors);
// (2) cp2_dataset_build_streamed: the entropy is known up front, so the sampling / gather / download / JSON body of
// every finished slot overlaps the hashing of the later slots, and only the lines that need ALL slot roots
// (dataSetRoot, slotProof: gen_input/bn254.nim:49-51,72) are added at the end by cp2_dataset_export_streamed.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "body_store.hpp"
#include "json_text.hpp"
#include "trees.hpp"

using namespace cp2i;
using namespace cp2text;

// ---------------------------------------------------------------------------------------------
// dataset
// ---------------------------------------------------------------------------------------------
struct cp2_dataset {
  cp2_ctx* ctx = nullptr;
  cp2_config cfg{};
  std::string file_base;
  bool from_file = false;
  uint64_t first_slot = 0, n_local = 0;
  cp2_slot_trees* trees = nullptr;            // every local slot tree (null for a roots-only dataset)
  // Roots-only dataset: the slot trees of a dataset whose nodes do not fit the device (3.1 % of the data: 256 MiB per 8 GiB
  // slot, 8 TiB for config 5's nominal 32 768 slots) are built batch by batch in pooled scratch and dropped again, only the
  // 32-byte roots stay; the tree of a slot that is proved is rebuilt on demand (0.2 s per 8 GiB), which is what the reference
  // does on EVERY run and once more per sample (gen_input/bn254.nim:42,57).
  DevBuf local_roots;                         // roots-only: n_local x 32 bytes
  // Compact dataset (between the two): of every local slot tree the part from the BLOCK ROOTS up stays (2 x nBlocks - 1 nodes:
  // 8 MiB per 8 GiB slot, 1/32 of the full tree), layer-major over the local slots: layer k of slot s starts at element
  // coff[k] + s * csizes[k].  The bottom of a path -- inside one network block -- is recomputed from the block's own cells
  // (<= nSamples blocks of 64 KiB per proof input: SURVEY.md section 7, "keep only block roots + upper layers and re-hash the
  // touched blocks"), checked against the stored block root.
  DevBuf compact;
  std::vector<size_t> csizes, coff;
  int tree_mode = 1;                          // 1 every node resident, 2 compact, 0 roots only
  bool have_roots = false;
  std::vector<size_t> dsizes;                 // dataset-tree layer sizes
  std::vector<uint8_t> dlayers;               // all dataset-tree layers, bottom first (host copy)
  // streamed build: one JSON body (", \"cellData\": ... }") per local slot, made while later slots were hashing
  bool prepared = false;
  uint8_t prep_entropy[32] = {};
  BodyStore bodies;
  ~cp2_dataset() { cp2_slot_trees_free(trees); }
};

// The field modulus r as four little-endian 64-bit words (README.md:76 of the reference), and a 32-byte value reduced into [0, r):
// `Entropy` is a field element in the reference (types/bn254.nim:21), so what is stored and printed is the canonical
// representative even when the caller hands in 32 arbitrary bytes (at most five subtractions: 2^256 / r < 5.3).
static const uint64_t FR_MODULUS_LE64[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static void canonical_felt(const uint8_t in[32], uint8_t out[32]) {
  uint64_t w[4];
  std::memcpy(w, in, 32);
  for (;;) {
    bool ge = true;
    for (int i = 3; i >= 0; --i)
      if (w[i] != FR_MODULUS_LE64[i]) { ge = w[i] > FR_MODULUS_LE64[i]; break; }
    if (!ge) break;
    unsigned __int128 borrow = 0;
    for (int i = 0; i < 4; ++i) {
      unsigned __int128 d = (unsigned __int128)w[i] - FR_MODULUS_LE64[i] - borrow;
      w[i] = (uint64_t)d;
      borrow = (d >> 64) & 1;
    }
  }
  std::memcpy(out, w, 32);
}

static int dataset_check(const cp2_config* cfg, uint64_t first_slot, uint64_t n_local) {
  if (n_local == 0 || first_slot > cfg->n_slots || n_local > cfg->n_slots - first_slot) return CP2_ERR_INVALID;   // (no wrap-around)
  if (cfg->max_depth < 0 || cfg->max_log2_nslots < 0) return CP2_ERR_INVALID;
  return CP2_OK;
}

static cp2_dataset* dataset_new(cp2_ctx* ctx, const cp2_config* cfg, uint64_t first_slot, uint64_t n_local) {
  cp2_dataset* ds = new (std::nothrow) cp2_dataset();
  if (!ds) return nullptr;
  ds->ctx = ctx;
  ds->cfg = *cfg;
  ds->from_file = cfg->file_base != nullptr;
  if (ds->from_file) ds->file_base = cfg->file_base;
  ds->cfg.file_base = nullptr;
  ds->first_slot = first_slot;
  ds->n_local = n_local;
  return ds;
}

static int dataset_build_trees(cp2_dataset* ds, size_t group, const SlotsDone& done) {
  const cp2_config& c = ds->cfg;
  if (ds->from_file)
    return trees_build_files(ds->ctx, ds->file_base, ds->first_slot, ds->n_local, c.cell_size, c.block_size, c.n_cells, group, done, &ds->trees);
  return trees_build_fake(ds->ctx, c.seed, ds->first_slot, ds->n_local, c.cell_size, c.block_size, c.n_cells, group, done, &ds->trees);
}

// the device buffer holding the roots of the local slots (n_local x 32 bytes)
static const void* dataset_roots_dev(const cp2_dataset* ds) {
  if (ds->trees) return cp2_slot_trees_roots_dev(ds->trees);
  if (ds->tree_mode == 2) return ds->compact.u8() + ds->coff.back() * 32;   // the last layer: one root per local slot
  return ds->local_roots.p;
}

// the trees of `n` local slots starting at local index `s0`, in pooled scratch (roots-only datasets: batches of the build, and
// the one slot a proof input is made for)
static int dataset_transient_trees(const cp2_dataset* ds, size_t s0, size_t n, cp2_slot_trees** out) {
  const cp2_config& c = ds->cfg;
  if (ds->from_file)
    return trees_build_files(ds->ctx, ds->file_base, ds->first_slot + s0, n, c.cell_size, c.block_size, c.n_cells, 0, nullptr, out, 1, true);
  return trees_build_fake(ds->ctx, c.seed, ds->first_slot + s0, n, c.cell_size, c.block_size, c.n_cells, 0, nullptr, out, 1, true);
}

// bytes of the compact part (block roots and up) of n_slots slot trees
static size_t compact_bytes(const cp2_config& c, size_t n_slots) {
  size_t per_slot = 0;
  for (size_t m : layer_sizes_of(c.n_cells / (c.block_size / c.cell_size))) per_slot += m;
  return n_slots * per_slot * 32;
}

// What of its trees does this dataset keep?  1 every node, 2 the compact part, 0 the roots.  The caller's word (cp2_set_keep_trees /
// CODEX_P2_KEEP_TREES), else the most that fits: a buffer must leave room for the builders' staging (two 2 GiB chunks), the batch in
// flight and some slack in what the device has free right now -- a SNAPSHOT (device_free_bytes: hipMemGetInfo, capped by
// CODEX_P2_MEM_LIMIT_MB), or the allowance a cp2_multi handed this context (its share of what the device had free before the shards started).  *automatic says whether the mode was
// chosen here (then dataset_build may step down when the allocation fails after all) or named by the caller (never changed).
#define CP2_TRY_MODE(call) do { if ((call) != CP2_OK) return 1; } while (0)   /* the device does not answer: plan as if everything fits */

// Slots per batch of a transient (compact / roots-only) build: half a staging chunk of nodes (1 GiB by default: 4 slots of 8 GiB), at
// least one slot.  The pipelined build holds two such node buffers, together one staging chunk's worth.
static size_t transient_batch_slots(const cp2_ctx* ctx, const cp2_config& c, uint64_t n_local) {
  const size_t per_slot = std::max<size_t>(1, trees_node_bytes(1, c.cell_size, c.block_size, c.n_cells));
  return std::max<size_t>(1, std::min<size_t>(n_local, (ctx->stage_bytes / 2) / per_slot));
}

static int dataset_tree_mode(cp2_ctx* ctx, const cp2_config& c, uint64_t n_local, bool* automatic = nullptr) {
  if (automatic) *automatic = false;
  int mode = ctx->keep_trees;
  if (mode < 0 && !env_keep_trees(&mode)) {
    ctx->err = "CODEX_P2_KEEP_TREES must be \"auto\", \"1\", \"2\" or \"0\"";
    return -1;
  }
  if (mode >= 0) return mode;
  if (automatic) *automatic = true;
  if (const char* t = std::getenv("CODEX_P2_TEST_OPTIMISTIC"))   // test-only: start at "every node" without looking, so that the step-down chain is what finds the mode that fits
    if (*t == '1') return 1;
  size_t free_b = ctx->mem_allowance;
  if (!free_b) CP2_TRY_MODE(device_free_bytes(&free_b));
  // What a build holds at its peak: what it keeps + the builders' staging (two chunks of at most `stage_bytes`, never more than the
  // data itself) + for the transient modes the node buffers of two batches in flight + headroom (sampling scratch, the dataset tree)
  typedef unsigned __int128 u128;
  const u128 data = (u128)n_local * c.n_cells * c.cell_size;
  const u128 staging = 2 * std::min<u128>(data, ctx->stage_bytes), headroom = std::min<u128>(data, (u128)1 << 30);
  const u128 nodes_all = (u128)trees_node_bytes(1, c.cell_size, c.block_size, c.n_cells) * n_local;
  const u128 in_flight = 2 * (u128)trees_node_bytes(transient_batch_slots(ctx, c, n_local), c.cell_size, c.block_size, c.n_cells);
  const u128 room = (u128)free_b * 9 / 10;
  if (nodes_all + staging + headroom <= room) return 1;
  if ((u128)compact_bytes(c, 1) * n_local + in_flight + staging + headroom <= room) return 2;
  return 0;
}

// compact / roots-only datasets: room for what stays of the local slots
static int dataset_alloc_kept(cp2_dataset* ds, int mode) {
  ds->tree_mode = mode;
  if (mode == 0) return ds->local_roots.alloc(ds->ctx, ds->n_local * 32);
  const cp2_config& c = ds->cfg;
  ds->csizes = layer_sizes_of(c.n_cells / (c.block_size / c.cell_size));
  ds->coff.clear();
  size_t off = 0;
  for (size_t m : ds->csizes) { ds->coff.push_back(off); off += ds->n_local * m; }
  return ds->compact.alloc(ds->ctx, off * 32);
}

// ... and what stays of a finished batch `t` (local slots [base, base + t->n_slots)): its roots, or its layers from the block roots up.
// Enqueued on `st` (default: the context's stream); the caller waits before the batch's nodes are used for anything else.
static int dataset_keep_from_batch(cp2_dataset* ds, const cp2_slot_trees* t, size_t base, hipStream_t st = nullptr) {
  cp2_ctx* ctx = ds->ctx;
  if (!st) st = ctx->stream;
  if (ds->tree_mode == 0) {
    CP2_HIP(ctx, hipMemcpyAsync(ds->local_roots.u8() + base * 32, cp2_slot_trees_roots_dev(t), t->n_slots * 32, hipMemcpyDeviceToDevice, st));
    return CP2_OK;
  }
  for (size_t k = 0; k < t->tsizes.size(); ++k)   // layer k of the batch's slots is contiguous, and so is its place in the dataset's layout
    CP2_HIP(ctx, hipMemcpyAsync(ds->compact.u8() + (ds->coff[k] + base * ds->csizes[k]) * 32, t->nodes.u8() + t->toff[k] * 32,
                                t->n_slots * t->tsizes[k] * 32, hipMemcpyDeviceToDevice, st));
  return CP2_OK;
}

// compact / roots-only build: batches of at most ~1 GiB of nodes (transient_batch_slots: 4 slots of 8 GiB; at least one slot), every
// batch a normal builder call; what the mode keeps is copied out, the rest is overwritten by the batch after next.
//
// The batches PIPELINE, from either source (BuildScratch: two node buffers used alternately, staging -- or, for slot files, the whole
// ingestion pipe -- that outlives a builder call, nothing synchronised per batch).  A batch ends with its tree-layer passes -- 22 launches for 2^22-cell slots, the top 16 of them one lone
// permutation latency each -- and the copy-out of what is kept, all on the context's THIRD stream; the next batch's generation and
// hashing go on alternating between the first two meanwhile, so the device does not drain between batches and the tail of a
// batch's last hash launch has the next batch's first one beside it (round 4 synchronised and freed here).  Node buffer b is handed
// to batch k + 2 once batch k's copy-out has completed (an event; long past by then).  (Until round 6 slot files went batch by batch, a new
// pipe set up and drained for each: "bound by the storage" is true of cold files, not of files in the page cache.)
static int dataset_build_transient(cp2_dataset* ds, int mode, bool allocated = false) {
  cp2_ctx* ctx = ds->ctx;
  const cp2_config& c = ds->cfg;
  const size_t batch = transient_batch_slots(ctx, c, ds->n_local);
  if (!allocated) CP2_TRY(dataset_alloc_kept(ds, mode));
  const char* what = mode == 2 ? "compact" : "roots-only";
  StageTimer trace;
  int st = CP2_OK;
  {
    BuildScratch scratch;                       // drains the context's streams before its buffers go, whatever path leaves this scope
    hipEvent_t kept[2] = {nullptr, nullptr};
    struct EvGuard { hipEvent_t* e; ~EvGuard() { for (int i = 0; i < 2; ++i) if (e[i]) (void)hipEventDestroy(e[i]); } } ev_guard{kept};
    for (int i = 0; i < 2; ++i) CP2_HIP(ctx, hipEventCreateWithFlags(&kept[i], hipEventDisableTiming));
    size_t k = 0;
    for (size_t s0 = 0; st == CP2_OK && s0 < ds->n_local; s0 += batch, ++k) {
      const size_t n = std::min(batch, (size_t)ds->n_local - s0);
      const int b = (int)(k & 1);
      if (k >= 2 && hipEventSynchronize(kept[b]) != hipSuccess) { ctx->err = std::string(what) + " build: a batch failed on the device"; st = CP2_ERR_HIP; break; }
      cp2_slot_trees* t = nullptr;
      st = ds->from_file ? trees_build_files(ctx, ds->file_base, ds->first_slot + s0, n, c.cell_size, c.block_size, c.n_cells, 0, nullptr, &t, 1, true, &scratch, b)
                         : trees_build_fake(ctx, c.seed, ds->first_slot + s0, n, c.cell_size, c.block_size, c.n_cells, 0, nullptr, &t, 1, true, &scratch, b);
      hipStream_t tail = scratch.tail_stream ? scratch.tail_stream : ctx->stream;
      if (st == CP2_OK) st = dataset_keep_from_batch(ds, t, s0, tail);    // follows the batch's layer passes on their stream (the context's third)
      if (st == CP2_OK && hipEventRecord(kept[b], tail) != hipSuccess) { ctx->err = "hipEventRecord failed"; st = CP2_ERR_HIP; }
      cp2_slot_trees_free(t);                                             // (the batch's nodes are the scratch's: nothing is waited for here)
      if (trace.on && ((k % 32) == 31 || s0 + n == ds->n_local))
        std::fprintf(stderr, "[cp2 trace] %s build: %zu of %llu slots enqueued\n", what, s0 + n, (unsigned long long)ds->n_local);
    }
    if (st == CP2_OK) {                         // everything landed (a failed launch or copy shows up here)
      if (hipStreamSynchronize(ctx->stream) != hipSuccess || (ctx->aux_stream && hipStreamSynchronize(ctx->aux_stream) != hipSuccess) ||
          (ctx->aux2_stream && hipStreamSynchronize(ctx->aux2_stream) != hipSuccess)) {
        (void)hipGetLastError();
        ctx->err = std::string(what) + " build: a batch failed on the device";
        st = CP2_ERR_HIP;
      }
    }
  }
  if (st != CP2_OK) return st;
  trace.lap(mode == 2 ? "compact build (block layers dropped)" : "roots-only build (trees dropped)");
  return CP2_OK;
}

// One mode down after an allocation failure of an automatically chosen mode: everything of the failed attempt is gone by now (the
// dataset object, its trees), the context's cached scratch goes too, and the trace / the error text say what happened.
static bool step_down(cp2_ctx* ctx, int* mode, const char* what) {
  if (*mode == 0) return false;
  const int next = *mode == 1 ? 2 : 0;
  (void)cp2_trim(ctx);
  const std::string why = ctx->err;
  if (std::getenv("CP2_TRACE"))
    std::fprintf(stderr, "[cp2 trace] %s: keeping %s did not fit after all (%s): retrying with %s\n", what, *mode == 1 ? "every node" : "the compact layers",
                 why.c_str(), next == 2 ? "the compact layers" : "the roots only");
  ctx->err.clear();
  *mode = next;
  return true;
}

static int dataset_build(cp2_ctx* ctx, const cp2_config* cfg, uint64_t first_slot, uint64_t n_local, bool always_keep_trees, cp2_dataset** out) {
  if (!ctx || !cfg || !out) return CP2_ERR_INVALID;
  *out = nullptr;
  CP2_REFUSE_STUCK(ctx);
  CP2_TRY(dataset_check(cfg, first_slot, n_local));
  CP2_TRY(trees_check_geometry(cfg->cell_size, cfg->block_size, cfg->n_cells, n_local));
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  bool automatic = false;
  int mode = always_keep_trees ? 1 : dataset_tree_mode(ctx, *cfg, n_local, &automatic);
  if (mode < 0) return CP2_ERR_INVALID;
  for (;;) {
    std::unique_ptr<cp2_dataset> ds(dataset_new(ctx, cfg, first_slot, n_local));
    if (!ds) return CP2_ERR_ALLOC;
    const int st = mode == 1 ? dataset_build_trees(ds.get(), 0, nullptr) : dataset_build_transient(ds.get(), mode);
    if (st == CP2_OK) {
      *out = ds.release();
      return CP2_OK;
    }
    ds.reset();                                   // (frees what the failed attempt holds before anything else is tried)
    if (st != CP2_ERR_ALLOC || !automatic || !step_down(ctx, &mode, "dataset build")) return st;
  }
}

extern "C" int cp2_dataset_build(cp2_ctx* ctx, const cp2_config* cfg, uint64_t first_slot, uint64_t n_local,
                                 cp2_dataset** out) try {
  return dataset_build(ctx, cfg, first_slot, n_local, false, out);
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

// Same as cp2_dataset_build, but what the dataset keeps of its slot trees is read from the cache when a file there is intact and
// matches the configuration AND (SlotFile source) the slot files' sizes and mtimes; built and written otherwise.
//
// Three representations can be cached -- every node ("CP2TREE3"), the compact layers or the roots ("CP2KEPT1", the mode in the
// header) -- and which one a run WANTS depends on what the device has free at that moment (dataset_tree_mode), which another
// tenant of the device can change from one run to the next.  So that such a change never costs a rebuild (hours at config 5's
// nominal size) or evicts a good cache:
//   * loading accepts ANY cached representation that is valid, describes this data and fits: the one wanted first, then the
//     smaller ones (a run that wanted every node but finds the compact layers keeps the dataset compact; CP2_TRACE says so);
//   * saving never overwrites a tree cache ("CP2TREE3") with the smaller kept form: that goes to "<cache_path>.kept" beside it,
//     and loading looks there too.
// To pin the representation, pin the mode: cp2_set_keep_trees / CODEX_P2_KEEP_TREES.
namespace {

bool file_has_magic(const char* path, const char* magic8) {
  char m[8] = {};
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return false;
  const bool ok = pread(fd, m, 8, 0) == 8 && std::memcmp(m, magic8, 8) == 0;
  close(fd);
  return ok;
}

// the kept form of `mode` (2 compact, 0 roots) from `path`: CP2_OK with *out set, CP2_ERR_IO when the file is not that, other errors as they come
int load_kept_dataset(cp2_ctx* ctx, const cp2_config* cfg, uint64_t first_slot, uint64_t n_local, int mode, const char* path, cp2_dataset** out) {
  if (!file_has_magic(path, "CP2KEPT1")) return CP2_ERR_IO;
  std::unique_ptr<cp2_dataset> ds(dataset_new(ctx, cfg, first_slot, n_local));
  if (!ds) return CP2_ERR_ALLOC;
  CP2_TRY(dataset_alloc_kept(ds.get(), mode));
  KeptMeta meta;
  meta.n_slots = n_local; meta.cell_size = cfg->cell_size; meta.block_size = cfg->block_size; meta.n_cells = cfg->n_cells;
  meta.src = (uint64_t)(ds->from_file ? CellSrc::File : CellSrc::Fake); meta.dataset_seed = cfg->seed; meta.first_slot = first_slot;
  meta.mode = (uint64_t)mode; meta.file_base = ds->file_base;
  void* buf = mode == 2 ? ds->compact.p : ds->local_roots.p;
  const size_t bytes = mode == 2 ? compact_bytes(*cfg, n_local) : (size_t)n_local * 32;
  CP2_TRY(kept_load(ctx, path, meta, buf, bytes));
  *out = ds.release();
  return CP2_OK;
}

int save_kept_dataset(cp2_dataset* ds, const char* cache_path) {
  const cp2_config& c = ds->cfg;
  KeptMeta meta;
  meta.n_slots = ds->n_local; meta.cell_size = c.cell_size; meta.block_size = c.block_size; meta.n_cells = c.n_cells;
  meta.src = (uint64_t)(ds->from_file ? CellSrc::File : CellSrc::Fake); meta.dataset_seed = c.seed; meta.first_slot = ds->first_slot;
  meta.mode = (uint64_t)ds->tree_mode; meta.file_base = ds->file_base;
  const void* buf = ds->tree_mode == 2 ? ds->compact.p : ds->local_roots.p;
  const size_t bytes = ds->tree_mode == 2 ? compact_bytes(c, ds->n_local) : (size_t)ds->n_local * 32;
  // a tree cache at the path is the richer representation: it stays, the kept form goes beside it
  const std::string path = file_has_magic(cache_path, "CP2TREE3") ? std::string(cache_path) + ".kept" : std::string(cache_path);
  return kept_save(ds->ctx, path.c_str(), meta, buf, bytes);
}

}  // namespace

extern "C" int cp2_dataset_build_cached(cp2_ctx* ctx, const cp2_config* cfg, uint64_t first_slot, uint64_t n_local,
                                        const char* cache_path, cp2_dataset** out) try {
  if (!ctx || !cfg || !out || !cache_path) return CP2_ERR_INVALID;
  *out = nullptr;
  CP2_REFUSE_STUCK(ctx);
  CP2_TRY(dataset_check(cfg, first_slot, n_local));
  CP2_TRY(trees_check_geometry(cfg->cell_size, cfg->block_size, cfg->n_cells, n_local));
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  bool automatic = false;
  int mode = dataset_tree_mode(ctx, *cfg, n_local, &automatic);
  if (mode < 0) return CP2_ERR_INVALID;
  const std::string beside = std::string(cache_path) + ".kept";
  StageTimer trace;
  // ---- load: the representation wanted, else (automatic choice only) any smaller one that is there
  if (mode == 1) {
    cp2_slot_trees* t = nullptr;
    const int lst = cp2_slot_trees_load(ctx, cache_path, &t);
    if (lst == CP2_OK) {
      const bool from_file = cfg->file_base != nullptr;
      bool match = t->n_slots == n_local && t->cell_size == cfg->cell_size && t->block_size == cfg->block_size &&
                   t->n_cells == cfg->n_cells && t->first_slot == first_slot && t->units_per_slot == 1 &&
                   (from_file ? (t->src == CellSrc::File && t->file_base == cfg->file_base)
                              : (t->src == CellSrc::Fake && t->dataset_seed == cfg->seed));
      if (match) {
        cp2_dataset* ds = dataset_new(ctx, cfg, first_slot, n_local);
        if (!ds) { cp2_slot_trees_free(t); return CP2_ERR_ALLOC; }
        ds->trees = t;
        *out = ds;
        return CP2_OK;
      }
      cp2_slot_trees_free(t);
    } else if (lst == CP2_ERR_ALLOC && automatic) {
      (void)step_down(ctx, &mode, "cached build (loading the tree cache)");   // the nodes do not fit after all: what is smaller may
    }
  }
  auto size_rank = [](int md) { return md == 1 ? 2 : (md == 2 ? 1 : 0); };   // every node > compact > roots only
  for (int m2 : {2, 0}) {
    if (size_rank(m2) > size_rank(mode) || (m2 != mode && !automatic)) continue;   // never a representation larger than what fits; a named mode is taken literally
    for (const char* path : {cache_path, beside.c_str()}) {
      cp2_dataset* ds = nullptr;
      const int lst = load_kept_dataset(ctx, cfg, first_slot, n_local, m2, path, &ds);
      if (lst == CP2_OK) {
        if (trace.on && m2 != mode) std::fprintf(stderr, "[cp2 trace] the cache holds %s (this run would have kept %s): taken as it is\n",
                                                 m2 == 2 ? "the compact layers" : "the slot roots", mode == 1 ? "every node" : "the compact layers");
        trace.lap(m2 == 2 ? "compact layers loaded from the cache" : "slot roots loaded from the cache");
        *out = ds;
        return CP2_OK;
      }
      if (lst != CP2_ERR_IO && lst != CP2_ERR_ALLOC) return lst;
    }
  }
  // ---- build (stepping down when an automatic choice does not fit after all) and write what the dataset keeps
  CP2_TRY(dataset_build(ctx, cfg, first_slot, n_local, mode == 1 && !automatic, out));
  cp2_dataset* ds = *out;
  int st = ds->trees ? cp2_slot_trees_save(ds->trees, cache_path) : save_kept_dataset(ds, cache_path);
  if (st != CP2_OK) { cp2_dataset_free(ds); *out = nullptr; return st; }
  trace.lap("built and written to the cache");
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" void cp2_dataset_free(cp2_dataset* ds) { delete ds; }

extern "C" int cp2_dataset_local_roots(cp2_dataset* ds, uint8_t* out) try {
  if (!ds || !out) return CP2_ERR_INVALID;
  if (ds->trees) return cp2_slot_trees_roots(ds->trees, out);
  cp2_ctx* ctx = ds->ctx;
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  CP2_HIP(ctx, hipMemcpyAsync(out, dataset_roots_dev(ds), ds->n_local * 32, hipMemcpyDeviceToHost, ctx->stream));
  CP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

// The dataset tree over the slot roots (gen_input/bn254.nim:49-50; odd layers use keys 2 / 3) from a DEVICE buffer holding
// all n_slots roots: the layers are built in device scratch and downloaded once (slotProof and the JSON heads are made on the
// host).  The buffer may live on another device or be the tree's own root layer; it is read on the context's stream.
static int dataset_tree_from_dev(cp2_dataset* ds, const void* d_all_roots) {
  cp2_ctx* ctx = ds->ctx;
  const size_t n = ds->cfg.n_slots;
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  ds->dsizes = layer_sizes_of(n);
  const size_t total = cp2_merkle_total(n);
  ds->dlayers.assign(total * 32, 0);
  StageTimer trace;
  DevBuf d;
  CP2_TRY(d.scratch(ctx, total * 32));
  CP2_TRY(merkle_trees_dev(ctx, d_all_roots, n, 1, d.p, false));
  CP2_HIP(ctx, hipMemcpyAsync(ds->dlayers.data(), d.p, total * 32, hipMemcpyDeviceToHost, ctx->stream));
  CP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
  trace.lap("dataset tree");
  ds->have_roots = true;
  return CP2_OK;
}

extern "C" int cp2_dataset_set_roots(cp2_dataset* ds, const uint8_t* all_roots) try {
  if (!ds) return CP2_ERR_INVALID;
  cp2_ctx* ctx = ds->ctx;
  const size_t n = ds->cfg.n_slots;
  if (!all_roots) {   // single GPU: the local roots are all of them and are already on the device
    if (ds->first_slot != 0 || ds->n_local != n) return CP2_ERR_INVALID;   // roots of other ranks' slots are missing
    return dataset_tree_from_dev(ds, dataset_roots_dev(ds));
  }
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  DevBuf d;
  CP2_TRY(d.scratch(ctx, n * 32));
  CP2_HIP(ctx, hipMemcpyAsync(d.p, all_roots, n * 32, hipMemcpyHostToDevice, ctx->stream));
  return dataset_tree_from_dev(ds, d.p);
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

// The same from a device buffer (what a device-to-device gather leaves behind: RCCL all-gather output, a torch tensor):
// no host copy of the roots is made on the way in.
extern "C" int cp2_dataset_set_roots_dev(cp2_dataset* ds, const void* d_all_roots) try {
  if (!ds || !d_all_roots) return CP2_ERR_INVALID;
  if (reinterpret_cast<uintptr_t>(d_all_roots) & 15) return CP2_ERR_ALIGN;
  return dataset_tree_from_dev(ds, d_all_roots);
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

int cp2i::dataset_own_roots_in_place(cp2_dataset* ds, bool* ok) {
  *ok = false;
  if (!ds->have_roots) return CP2_ERR_INVALID;
  std::vector<uint8_t> own(ds->n_local * 32);
  CP2_TRY(cp2_dataset_local_roots(ds, own.data()));
  *ok = std::memcmp(own.data(), &ds->dlayers[ds->first_slot * 32], own.size()) == 0;   // layer 0 of the dataset tree = all slot roots
  return CP2_OK;
}

extern "C" const void* cp2_dataset_local_roots_dev(const cp2_dataset* ds) { return ds ? dataset_roots_dev(ds) : nullptr; }
extern "C" int cp2_dataset_keeps_trees(const cp2_dataset* ds) { return !ds ? 0 : (ds->trees ? 1 : ds->tree_mode); }

extern "C" int cp2_dataset_copy_local_roots_dev(cp2_dataset* ds, void* d_out) try {
  if (!ds || !d_out) return CP2_ERR_INVALID;
  cp2_ctx* ctx = ds->ctx;
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  CP2_HIP(ctx, hipMemcpyAsync(d_out, dataset_roots_dev(ds), ds->n_local * 32, hipMemcpyDeviceToDevice, ctx->stream));
  return CP2_OK;
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" int cp2_dataset_range(const cp2_dataset* ds, uint64_t* first_slot, uint64_t* n_local) {
  if (!ds) return CP2_ERR_INVALID;
  if (first_slot) *first_slot = ds->first_slot;
  if (n_local) *n_local = ds->n_local;
  return CP2_OK;
}

extern "C" cp2_ctx* cp2_dataset_ctx(const cp2_dataset* ds) { return ds ? ds->ctx : nullptr; }

extern "C" int cp2_dataset_root(cp2_dataset* ds, uint8_t out[32]) try {
  if (!ds || !out) return CP2_ERR_INVALID;
  if (!ds->have_roots) CP2_TRY(cp2_dataset_set_roots(ds, nullptr));
  std::memcpy(out, &ds->dlayers[ds->dlayers.size() - 32], 32);
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

// ---------------------------------------------------------------------------------------------
// sampling + gathers on the device, results in pinned host memory
// ---------------------------------------------------------------------------------------------
namespace {

// device scratch of one sampling pass over up to `cap` (slot, counter) pairs
struct SampleDev {
  DevBuf entropy, slots, idx, gcell, rows, paths, leaves, cells;
  size_t cap = 0;
  int init(cp2_ctx* ctx, size_t cap_items, size_t ns, size_t md, size_t cs, bool need_cells) {
    cap = cap_items * ns;
    CP2_TRY(entropy.scratch(ctx, 32));
    CP2_TRY(slots.scratch(ctx, std::max<size_t>(cap_items, 1) * 8));
    CP2_TRY(idx.scratch(ctx, std::max<size_t>(cap, 1) * 8));
    CP2_TRY(gcell.scratch(ctx, std::max<size_t>(cap, 1) * 8));
    CP2_TRY(rows.scratch(ctx, std::max<size_t>(cap * md, 1) * 8));
    CP2_TRY(paths.scratch(ctx, std::max<size_t>(cap * md, 1) * 32));
    CP2_TRY(leaves.scratch(ctx, std::max<size_t>(cap, 1) * 32));
    if (need_cells) CP2_TRY(cells.scratch(ctx, std::max<size_t>(cap * cs, 1)));
    return CP2_OK;
  }
};

// pinned host landing zone of one pass
struct SampleHost {
  PinBuf idx, paths, leaves, cells;
  hipStream_t stream = nullptr;   // downloads into these buffers are enqueued here: drained before the blocks go back to the pool
  ~SampleHost() { if (stream) (void)hipStreamSynchronize(stream); }
  int init(cp2_ctx* ctx, size_t cap_items, size_t ns, size_t md, size_t cs, bool need_cells) {
    const size_t cap = cap_items * ns;
    CP2_TRY(idx.alloc(ctx, std::max<size_t>(cap, 1) * 8));
    CP2_TRY(paths.alloc(ctx, std::max<size_t>(cap * md, 1) * 32));
    CP2_TRY(leaves.alloc(ctx, std::max<size_t>(cap, 1) * 32));
    if (need_cells) CP2_TRY(cells.alloc(ctx, std::max<size_t>(cap * cs, 1)));
    return CP2_OK;
  }
};

// Enqueue on `st`: cellIndices for n_items slots (explicit batch-local list `h_slots`, or the range starting at slot0),
// path rows, the path gather, the regeneration / gather of the sampled cells (device-resident sources), and the
// downloads into `host`.  Nothing is synchronised.  d.entropy must already hold the entropy.
int enqueue_sampling(cp2_slot_trees* t, const cp2k::TreeGeom& g, SampleDev& d, SampleHost& host, const uint64_t* h_slots, uint64_t slot0,
                     size_t n_items, size_t ns, size_t md, bool fetch_cells, hipStream_t st) {
  cp2_ctx* ctx = t->ctx;
  const size_t total = n_items * ns, cs = t->cell_size;
  if (total == 0) return CP2_OK;
  host.stream = st;
  const uint64_t* d_slots = nullptr;
  if (h_slots) {
    CP2_HIP(ctx, hipMemcpyAsync(d.slots.p, h_slots, n_items * 8, hipMemcpyHostToDevice, st));
    d_slots = static_cast<const uint64_t*>(d.slots.p);
  }
  uint64_t* idx = static_cast<uint64_t*>(d.idx.p);
  uint64_t* gcell = static_cast<uint64_t*>(d.gcell.p);
  uint64_t* rows = static_cast<uint64_t*>(d.rows.p);
  CP2_HIP(ctx, cp2k::launch_sample_paths(g, t->nodes.p, d.entropy.p, d_slots, slot0, n_items, (uint32_t)ns, (uint32_t)md, idx, gcell, rows, st));
  CP2_HIP(ctx, cp2k::launch_gather_rows(t->nodes.p, rows, total * md, 32, d.paths.p, st));
  CP2_HIP(ctx, hipMemcpyAsync(host.idx.p, idx, total * 8, hipMemcpyDeviceToHost, st));
  CP2_HIP(ctx, hipMemcpyAsync(host.paths.p, d.paths.p, total * md * 32, hipMemcpyDeviceToHost, st));
  // the sampled cells' own hashes (leafValue of the merged proof, merkle.nim:86-100): layer 0 rows are slot * n_cells + cell
  CP2_HIP(ctx, cp2k::launch_gather_rows(t->nodes.p, gcell, total, 32, d.leaves.p, st));
  CP2_HIP(ctx, hipMemcpyAsync(host.leaves.p, d.leaves.p, total * 32, hipMemcpyDeviceToHost, st));
  if (!fetch_cells) return CP2_OK;   // host-side sources: the caller reads the sampled cells itself
  if (t->src == CellSrc::Fake) {
    CP2_HIP(ctx, cp2k::launch_gen_fake_cells(cp2_slot_seed(t->dataset_seed, t->units_per_slot > 1 ? 0 : t->first_slot), t->n_cells, 0, gcell, total, cs,
                                             d.cells.p, st, t->units_per_slot, t->first_slot));
  } else if (t->src == CellSrc::Dev && t->d_cells) {
    CP2_HIP(ctx, cp2k::launch_gather_rows(t->d_cells, gcell, total, cs, d.cells.p, st));
  } else {
    return CP2_ERR_INVALID;
  }
  CP2_HIP(ctx, hipMemcpyAsync(host.cells.p, d.cells.p, total * cs, hipMemcpyDeviceToHost, st));
  return CP2_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// proof input
// ---------------------------------------------------------------------------------------------
// cell bytes, Merkle paths and indices of a whole batch live in pinned blocks that every proof input of the batch
// shares (one download each, no per-slot copies); the blocks return to the context's pool with the last reference
struct BatchStore {
  PinBuf idx, paths, leaves, cells;
  std::vector<uint8_t> cells_heap;   // SlotFile / Host sources: sampled cells are read on the host
  std::vector<uint8_t> heap;         // cp2_proof_input_create: everything copied from the caller
};

struct cp2_proof_input {
  cp2_config cfg{};
  uint64_t slot_idx = 0;
  uint8_t entropy[32], dataset_root[32], slot_root[32];
  size_t n_samples = 0;
  std::vector<uint8_t> slot_proof;
  std::shared_ptr<BatchStore> store;
  const uint64_t* indices = nullptr;    // nSamples, inside store->idx
  const uint8_t* cell_data = nullptr;   // nSamples x cellSize, inside store->cells / cells_heap
  const uint8_t* paths = nullptr;       // nSamples x maxDepth x 32, inside store->paths
  const uint8_t* leaves = nullptr;      // nSamples x 32: hash of each sampled cell (may be null for caller-made inputs)
};

// slotProof = padMerkleProof(merkleProof(dsetTree, slotIdx), maxLog2NSlots), gen_input/bn254.nim:51,72
static void fill_slot_proof(const cp2_dataset* ds, uint64_t slot_idx, std::vector<uint8_t>& out) {
  out.assign((size_t)ds->cfg.max_log2_nslots * 32, 0);
  size_t k = slot_idx, m = ds->cfg.n_slots, off = 0;
  for (size_t i = 0; i + 1 < ds->dsizes.size(); ++i) {
    size_t j = k ^ 1;
    if (j < m) std::memcpy(&out[i * 32], &ds->dlayers[(off + j) * 32], 32);
    off += ds->dsizes[i];
    k >>= 1;
    m = (m + 1) >> 1;
  }
}

// sampled cells of the host-side sources: global index g = local_slot * n_cells + cell (slotLoadCellData, slot.nim:57-68)
static int host_cells_global(cp2_slot_trees* t, const uint64_t* g, size_t n, uint8_t* out) {
  cp2_ctx* ctx = t->ctx;
  const size_t cs = t->cell_size;
  if (t->src == CellSrc::Host) {
    if (!t->h_cells) return CP2_ERR_INVALID;   // loaded from a cache: attach the cells first
    for (size_t i = 0; i < n; ++i) std::memcpy(out + i * cs, t->h_cells + g[i] * cs, cs);
    return CP2_OK;
  }
  if (t->src == CellSrc::File) {
    // a batch of proof inputs samples hundreds of thousands of cells (4096 slots x 100): the reads are spread over the
    // context's fill threads, each taking a contiguous range of the (slot-ordered) list and opening a slot file once per run
    const int threads = (int)std::min<size_t>(ctx->ingest_threads > 0 ? (size_t)ctx->ingest_threads : 8, std::max<size_t>(1, n / 256));
    std::vector<std::string> failed(threads);
    auto work = [&](int w) {
      int fd = -1;
      size_t open_slot = ~(size_t)0;
      for (size_t i = n * w / threads; i < n * (w + 1) / threads; ++i) {
        size_t slot = g[i] / t->n_cells, cell = g[i] % t->n_cells;
        if (slot != open_slot) {
          if (fd >= 0) close(fd);
          std::string fname = slot_file_name(t->file_base, (t->first_slot + slot) / t->units_per_slot);
          fd = open(fname.c_str(), O_RDONLY);
          if (fd < 0) { failed[w] = fname; return; }
          open_slot = slot;
        }
        read_file_cell(fd, cs, ((t->first_slot + slot) % t->units_per_slot) * t->n_cells + cell, out + i * cs);
      }
      if (fd >= 0) close(fd);
    };
    if (threads <= 1) {
      work(0);
    } else {
      Workers pool(threads - 1);
      for (int w = 1; w < threads; ++w) pool.submit([&work, w] { work(w); });
      work(0);
      pool.wait_idle();
    }
    for (const auto& f : failed)
      if (!f.empty()) { ctx->err = "cannot open " + f; return CP2_ERR_IO; }
    return CP2_OK;
  }
  if (t->src == CellSrc::Dev) {   // cell sizes the row gather cannot take: plain copies
    if (!t->d_cells) return CP2_ERR_INVALID;
    for (size_t i = 0; i < n; ++i)
      CP2_HIP(ctx, hipMemcpyAsync(out + i * cs, t->d_cells + g[i] * cs, cs, hipMemcpyDeviceToHost, ctx->stream));
    CP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CP2_OK;
  }
  return CP2_ERR_INVALID;
}

static void fill_slot_proof(const cp2_dataset* ds, uint64_t slot_idx, std::vector<uint8_t>& out);

// generateProofInput (gen_input/bn254.nim:53-74) on a COMPACT dataset, for `n` slots in one pass: the top of every path -- block
// root to slot root -- is gathered from the stored layers; the bottom -- cell to block root (merkleProof on the block's tree,
// blocks/bn254.nim:60-67) -- comes from the trees of the touched blocks (at most nSamples per slot), rebuilt here from the blocks'
// own cells (regenerated, or read from the slot files) as ONE batch of one-block "slots" and checked against the stored block
// roots.  One sampling launch, one generator launch, one hash + layer pass and two gathers for all n slots (a node that proves
// every slot it holds each period: 4096 slots cost one pass over 26 GB of touched blocks, not 4096 latency-bound little ones).
// The caller chunks n so that the blocks fit the scratch; entropy canonical.
static int compact_proof_inputs(cp2_dataset* ds, const uint64_t* slots, size_t n, const uint8_t entropy[32], cp2_proof_input** out) {
  cp2_ctx* ctx = ds->ctx;
  const cp2_config& c = ds->cfg;
  const size_t ns = c.n_samples, md = (size_t)c.max_depth, cs = c.cell_size, cpb = c.block_size / c.cell_size, nblocks = c.n_cells / cpb;
  const size_t depth_b = layer_sizes_of(cpb).size() - 1, depth_t = ds->csizes.size() - 1;
  if (depth_b + depth_t > md) return CP2_ERR_INVALID;                                   // padMerkleProof assert, types.nim:29
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  const size_t total = n * ns;                                                          // (slot, counter) pairs = touched blocks
  std::vector<uint64_t> idx(total);
  if (total) {                                                                          // cellIndices, sample/bn254.nim:16-27, for all pairs at once
    std::vector<uint8_t> felts(total * 96, 0), dig(total * 32);
    for (size_t i = 0; i < n; ++i)
      for (size_t k = 0; k < ns; ++k) {
        uint8_t* f = &felts[(i * ns + k) * 96];
        std::memcpy(f, entropy, 32);
        std::memcpy(f + 32, &ds->dlayers[slots[i] * 32], 32);                           // the slot root: layer 0 of the dataset tree
        const uint64_t counter = k + 1;
        std::memcpy(f + 64, &counter, 8);
      }
    CP2_TRY(cp2_sponge2_felts_batch(ctx, felts.data(), 3, total, dig.data()));
    for (size_t p = 0; p < total; ++p) {
      uint64_t lo;
      std::memcpy(&lo, &dig[32 * p], 8);                                                // extractLowBits, types/bn254.nim:47-59
      idx[p] = lo & (c.n_cells - 1);
    }
  }
  std::vector<uint8_t> paths(total * md * 32, 0), leaves(total * 32), cells(total * cs);
  if (total) {
    // the cells of the touched blocks, block after block, in device scratch
    const size_t n_bc = total * cpb;
    DevBuf d_cells, d_list, d_rows, d_got;
    CP2_TRY(d_cells.scratch(ctx, n_bc * cs));
    std::vector<uint8_t> h_blocks;                                                       // SlotFile: read on the host first
    if (ds->from_file) {
      h_blocks.resize(n_bc * cs);
      for (size_t i = 0; i < n; ++i) {
        const std::string fname = slot_file_name(ds->file_base, slots[i]);
        const int fd = open(fname.c_str(), O_RDONLY);
        if (fd < 0) { ctx->err = "cannot open " + fname; return CP2_ERR_IO; }
        for (size_t k = 0; k < ns; ++k)
          for (size_t j = 0; j < cpb; ++j) read_file_cell(fd, cs, (idx[i * ns + k] / cpb) * cpb + j, &h_blocks[((i * ns + k) * cpb + j) * cs]);
        close(fd);
      }
      CP2_HIP(ctx, hipMemcpyAsync(d_cells.p, h_blocks.data(), h_blocks.size(), hipMemcpyHostToDevice, ctx->stream));
    } else {
      // the generator's list form over "global cells" of the local slots: g = local slot * nCells + cell, seed of local slot 0
      std::vector<uint64_t> list(n_bc);
      for (size_t i = 0; i < n; ++i)
        for (size_t k = 0; k < ns; ++k)
          for (size_t j = 0; j < cpb; ++j)
            list[(i * ns + k) * cpb + j] = (slots[i] - ds->first_slot) * c.n_cells + (idx[i * ns + k] / cpb) * cpb + j;
      CP2_TRY(d_list.scratch(ctx, n_bc * 8));
      CP2_HIP(ctx, hipMemcpyAsync(d_list.p, list.data(), n_bc * 8, hipMemcpyHostToDevice, ctx->stream));
      CP2_HIP(ctx, cp2k::launch_gen_fake_cells(cp2_slot_seed(c.seed, ds->first_slot), c.n_cells, 0, static_cast<const uint64_t*>(d_list.p), n_bc, cs,
                                               d_cells.p, ctx->stream));
      CP2_HIP(ctx, hipStreamSynchronize(ctx->stream));                                   // `list` leaves scope below
    }
    // one-block "slots": hash the cells, build the block trees (the singleton layer on top of each is not used)
    cp2_slot_trees* mini = nullptr;
    CP2_TRY(cp2_slot_trees_build_dev(ctx, d_cells.p, total, cs, c.block_size, cpb, &mini));
    struct Mini { cp2_slot_trees* t; ~Mini() { cp2_slot_trees_free(t); } } mini_guard{mini};
    // rows to gather: per pair the depth_b siblings inside its block, its leaf, the rebuilt block root (from `mini`), then the
    // depth_t siblings above the block and the stored block root (from the compact layers)
    const size_t per_m = depth_b + 2, per_c = depth_t + 1;
    std::vector<uint64_t> rows_m(total * per_m), rows_c(total * per_c), r(depth_b + 1);
    for (size_t i = 0; i < n; ++i) {
      const size_t ls = (size_t)(slots[i] - ds->first_slot);
      for (size_t k = 0; k < ns; ++k) {
        const size_t p = i * ns + k;
        const uint64_t in_block = idx[p] % cpb, b = idx[p] / cpb;
        path_rows(mini, p, in_block, depth_b + 1, r.data());                             // block layers, then the singleton's (unused) entry
        for (size_t d = 0; d < depth_b; ++d) rows_m[p * per_m + d] = r[d];
        rows_m[p * per_m + depth_b] = p * cpb + in_block;                                // the leaf: layer 0 of `mini`
        rows_m[p * per_m + depth_b + 1] = mini->toff[0] + p;                             // the block root: layer 0 of its singleton tree
        uint64_t q = b, m = nblocks;
        for (size_t d = 0; d < depth_t; ++d) {                                           // merkleProof(bigTree, blockIdx), merkle.nim:21-42
          const uint64_t sib = q ^ 1;
          rows_c[p * per_c + d] = sib < m ? ds->coff[d] + ls * ds->csizes[d] + sib : NO_ROW;
          q >>= 1;
          m = (m + 1) >> 1;
        }
        rows_c[p * per_c + depth_t] = ds->coff[0] + ls * ds->csizes[0] + b;              // the stored root of the block
      }
    }
    const size_t n_m = rows_m.size(), n_c = rows_c.size();
    CP2_TRY(d_rows.scratch(ctx, (n_m + n_c) * 8));
    CP2_TRY(d_got.scratch(ctx, (n_m + n_c) * 32));
    uint64_t* dr = static_cast<uint64_t*>(d_rows.p);
    CP2_HIP(ctx, hipMemcpyAsync(dr, rows_m.data(), n_m * 8, hipMemcpyHostToDevice, ctx->stream));
    CP2_HIP(ctx, hipMemcpyAsync(dr + n_m, rows_c.data(), n_c * 8, hipMemcpyHostToDevice, ctx->stream));
    CP2_HIP(ctx, cp2k::launch_gather_rows(mini->nodes.p, dr, n_m, 32, d_got.p, ctx->stream));
    CP2_HIP(ctx, cp2k::launch_gather_rows(ds->compact.p, dr + n_m, n_c, 32, d_got.u8() + n_m * 32, ctx->stream));
    std::vector<uint8_t> got((n_m + n_c) * 32);
    CP2_HIP(ctx, hipMemcpyAsync(got.data(), d_got.p, got.size(), hipMemcpyDeviceToHost, ctx->stream));
    // the sampled cells themselves: one row of cellSize bytes per pair out of the block scratch (device sources), or the host copy
    DevBuf d_sel, d_sel_rows;
    if (!ds->from_file) {
      if ((cs & 3) == 0) {
        std::vector<uint64_t> sel(total);
        for (size_t p = 0; p < total; ++p) sel[p] = p * cpb + idx[p] % cpb;
        CP2_TRY(d_sel_rows.scratch(ctx, total * 8));
        CP2_TRY(d_sel.scratch(ctx, total * cs));
        CP2_HIP(ctx, hipMemcpyAsync(d_sel_rows.p, sel.data(), total * 8, hipMemcpyHostToDevice, ctx->stream));
        CP2_HIP(ctx, cp2k::launch_gather_rows(d_cells.p, static_cast<const uint64_t*>(d_sel_rows.p), total, cs, d_sel.p, ctx->stream));
        CP2_HIP(ctx, hipMemcpyAsync(cells.data(), d_sel.p, total * cs, hipMemcpyDeviceToHost, ctx->stream));
        CP2_HIP(ctx, hipStreamSynchronize(ctx->stream));                                 // `sel` leaves scope
      } else {                                                                           // cell sizes the row gather cannot take: plain copies
        for (size_t p = 0; p < total; ++p)
          CP2_HIP(ctx, hipMemcpyAsync(&cells[p * cs], d_cells.u8() + (p * cpb + idx[p] % cpb) * cs, cs, hipMemcpyDeviceToHost, ctx->stream));
      }
    }
    CP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < n; ++i)
      for (size_t k = 0; k < ns; ++k) {
        const size_t p = i * ns + k;
        const uint8_t* gm = &got[p * per_m * 32];
        const uint8_t* gc = &got[(n_m + p * per_c) * 32];
        if (std::memcmp(gm + (depth_b + 1) * 32, gc + depth_t * 32, 32) != 0) {          // the data no longer hashes to the stored block root
          ctx->err = "block " + std::to_string(idx[p] / cpb) + " of slot " + std::to_string(slots[i]) +
                     " does not hash to its stored root (slot data changed since the build?)";
          return CP2_ERR_IO;
        }
        std::memcpy(&paths[p * md * 32], gm, depth_b * 32);
        std::memcpy(&paths[(p * md + depth_b) * 32], gc, depth_t * 32);
        std::memcpy(&leaves[p * 32], gm + depth_b * 32, 32);
        if (ds->from_file) std::memcpy(&cells[p * cs], &h_blocks[(p * cpb + idx[p] % cpb) * cs], cs);
      }
  }
  std::vector<uint8_t> proof;
  for (size_t i = 0; i < n; ++i) {
    fill_slot_proof(ds, slots[i], proof);
    int st = cp2_proof_input_create(&c, slots[i], &ds->dlayers[ds->dlayers.size() - 32], entropy, &ds->dlayers[slots[i] * 32], proof.data(), ns,
                                    &idx[i * ns], &cells[i * ns * cs], &paths[i * ns * md * 32], &leaves[i * ns * 32], out + i);
    if (st != CP2_OK) {
      for (size_t j = 0; j < i; ++j) { cp2_proof_input_free(out[j]); out[j] = nullptr; }
      return st;
    }
  }
  return CP2_OK;
}

// generateProofInput (gen_input/bn254.nim:35-79) for `n` slots of the dataset at once: one sampling launch,
// one path gather, one cell fetch for all of them.
extern "C" int cp2_proof_inputs_generate_batch(cp2_dataset* ds, const uint64_t* slot_idx, size_t n, const uint8_t entropy_in[32],
                                               cp2_proof_input** out) try {
  if (!ds || !entropy_in || (n && (!slot_idx || !out))) return CP2_ERR_INVALID;
  uint8_t entropy[32];
  canonical_felt(entropy_in, entropy);
  for (size_t i = 0; i < n; ++i) out[i] = nullptr;
  if (n == 0) return CP2_OK;
  const cp2_config& cfg = ds->cfg;
  cp2_ctx* ctx = ds->ctx;
  for (size_t i = 0; i < n; ++i)
    if (slot_idx[i] < ds->first_slot || slot_idx[i] >= ds->first_slot + ds->n_local) return CP2_ERR_INVALID;
  if (!is_pow2(cfg.n_cells)) return CP2_ERR_INVALID;                    // sample/bn254.nim:19-20
  if (cfg.n_samples && cfg.n_cells < 2) return CP2_ERR_INVALID;         // extractLowBits asserts k > 0, types/bn254.nim:48
  if (!ds->trees && ds->tree_mode == 2) {                               // compact dataset: stored upper layers + the touched blocks, slot by slot
    if (!ds->have_roots) CP2_TRY(cp2_dataset_set_roots(ds, nullptr));
    if (ds->dsizes.size() - 1 > (size_t)cfg.max_log2_nslots) return CP2_ERR_INVALID;   // padMerkleProof assert
    // as many slots per pass as keep the touched blocks within about 2 GiB of scratch (327 slots at 100 samples of 64 KiB blocks)
    const size_t per_slot = std::max<size_t>(1, (size_t)cfg.n_samples * cfg.block_size);
    const size_t chunk = std::max<size_t>(1, ((size_t)2 << 30) / per_slot);
    for (size_t i0 = 0; i0 < n; i0 += chunk) {
      const size_t m = std::min(chunk, n - i0);
      int st = compact_proof_inputs(ds, slot_idx + i0, m, entropy, out + i0);
      if (st != CP2_OK) {
        for (size_t j = 0; j < i0; ++j) { cp2_proof_input_free(out[j]); out[j] = nullptr; }
        return st;
      }
    }
    return CP2_OK;
  }
  if (!ds->trees && n > 1) {                                            // roots-only dataset: one slot, one rebuilt tree, at a time
    for (size_t i = 0; i < n; ++i) {
      int st = cp2_proof_inputs_generate_batch(ds, slot_idx + i, 1, entropy_in, out + i);
      if (st != CP2_OK) {
        for (size_t j = 0; j < i; ++j) { delete out[j]; out[j] = nullptr; }
        return st;
      }
    }
    return CP2_OK;
  }
  // the trees the paths come from: the dataset's own, or (roots-only) the tree of this one slot, rebuilt in pooled scratch
  cp2_slot_trees* t = ds->trees;
  struct Transient { cp2_slot_trees* t = nullptr; ~Transient() { cp2_slot_trees_free(t); } } transient;
  if (!t) {
    CP2_HIP(ctx, hipSetDevice(ctx->device));
    CP2_TRY(dataset_transient_trees(ds, (size_t)(slot_idx[0] - ds->first_slot), 1, &transient.t));
    t = transient.t;
    // slotRoot, slotProof and dataSetRoot come from the STORED roots; indices, paths and cells from the tree just rebuilt: the two must
    // be the same slot.  A slot file that changed since the build (or since the cache was written) is an I/O error, like the compact
    // mode's block check, never an input.json over mixed data.
    uint8_t rebuilt[32], stored[32];
    CP2_TRY(cp2_slot_trees_roots(t, rebuilt));
    CP2_HIP(ctx, hipMemcpyAsync(stored, static_cast<const uint8_t*>(dataset_roots_dev(ds)) + (slot_idx[0] - ds->first_slot) * 32, 32, hipMemcpyDeviceToHost, ctx->stream));
    CP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (std::memcmp(rebuilt, stored, 32) != 0) {
      ctx->err = "slot " + std::to_string(slot_idx[0]) + " does not hash to its stored root (slot data changed since the build?)";
      return CP2_ERR_IO;
    }
  }
  const uint64_t local_base = ds->trees ? ds->first_slot : slot_idx[0];  // index inside `t` = slot index - local_base
  if (!ds->have_roots) CP2_TRY(cp2_dataset_set_roots(ds, nullptr));
  if (ds->dsizes.size() - 1 > (size_t)cfg.max_log2_nslots) return CP2_ERR_INVALID;   // padMerkleProof assert
  if (cp2_slot_trees_depth(t) > (size_t)cfg.max_depth) return CP2_ERR_INVALID;        // padMerkleProof assert
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  const size_t ns = cfg.n_samples, md = (size_t)cfg.max_depth, cs = cfg.cell_size, total = n * ns;

  StageTimer trace;
  auto store = std::make_shared<BatchStore>();
  const bool dev_cells = t->src == CellSrc::Fake ||
                         (t->src == CellSrc::Dev && t->d_cells && (cs & 3) == 0 && (reinterpret_cast<uintptr_t>(t->d_cells) & 3) == 0);
  if (total) {
    cp2k::TreeGeom g;
    trees_geom(t, &g);
    std::vector<uint64_t> local(n);
    for (size_t i = 0; i < n; ++i) local[i] = slot_idx[i] - local_base;
    SampleDev dev;
    SampleHost host;
    CP2_TRY(dev.init(ctx, n, ns, md, cs, dev_cells));
    CP2_TRY(host.init(ctx, n, ns, md, cs, dev_cells));
    CP2_HIP(ctx, hipMemcpyAsync(dev.entropy.p, entropy, 32, hipMemcpyHostToDevice, ctx->stream));
    CP2_TRY(enqueue_sampling(t, g, dev, host, local.data(), 0, n, ns, md, dev_cells, ctx->stream));
    CP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
    trace.lap("sampling + gathers + downloads");
    store->idx.swap(host.idx);
    store->paths.swap(host.paths);
    store->leaves.swap(host.leaves);
    if (dev_cells) {
      store->cells.swap(host.cells);
    } else {
      const uint64_t* idx = static_cast<const uint64_t*>(store->idx.p);
      std::vector<uint64_t> gcell(total);
      for (size_t i = 0; i < n; ++i)
        for (size_t c = 0; c < ns; ++c) gcell[i * ns + c] = local[i] * t->n_cells + idx[i * ns + c];
      store->cells_heap.resize(total * cs);
      CP2_TRY(host_cells_global(t, gcell.data(), total, store->cells_heap.data()));
      trace.lap("cells read on the host");
    }
  }
  const uint8_t* cells = dev_cells ? store->cells.u8() : store->cells_heap.data();
  // ---- split
  for (size_t i = 0; i < n; ++i) {
    cp2_proof_input* p = new (std::nothrow) cp2_proof_input();
    if (!p) {
      for (size_t j = 0; j < i; ++j) { delete out[j]; out[j] = nullptr; }
      return CP2_ERR_ALLOC;
    }
    p->cfg = cfg;
    p->slot_idx = slot_idx[i];
    std::memcpy(p->entropy, entropy, 32);
    std::memcpy(p->dataset_root, &ds->dlayers[ds->dlayers.size() - 32], 32);
    std::memcpy(p->slot_root, &ds->dlayers[slot_idx[i] * 32], 32);      // layer 0 of the dataset tree = slot roots
    fill_slot_proof(ds, slot_idx[i], p->slot_proof);
    p->n_samples = ns;
    p->store = store;
    if (total) {
      p->indices = static_cast<const uint64_t*>(store->idx.p) + i * ns;
      p->cell_data = cells + i * ns * cs;
      p->paths = store->paths.u8() + i * ns * md * 32;
      p->leaves = store->leaves.u8() + i * ns * 32;
    }
    out[i] = p;
  }
  trace.lap("split into proof inputs");
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" int cp2_proof_input_generate(cp2_dataset* ds, uint64_t slot_idx, const uint8_t entropy[32], cp2_proof_input** out) try {
  if (!out) return CP2_ERR_INVALID;
  return cp2_proof_inputs_generate_batch(ds, &slot_idx, 1, entropy, out);
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" void cp2_proof_input_free(cp2_proof_input* p) { delete p; }

extern "C" int cp2_proof_input_roots(const cp2_proof_input* p, uint8_t dataset_root[32], uint8_t slot_root[32], uint8_t entropy[32]) try {
  if (!p) return CP2_ERR_INVALID;
  if (dataset_root) std::memcpy(dataset_root, p->dataset_root, 32);
  if (slot_root) std::memcpy(slot_root, p->slot_root, 32);
  if (entropy) std::memcpy(entropy, p->entropy, 32);
  return CP2_OK;
} catch (...) {
  return CP2_ERR_INVALID;
}
extern "C" size_t cp2_proof_input_nsamples(const cp2_proof_input* p) { return p ? p->n_samples : 0; }
extern "C" const uint64_t* cp2_proof_input_cell_indices(const cp2_proof_input* p) { return p ? p->indices : nullptr; }
extern "C" const uint8_t* cp2_proof_input_cell_data(const cp2_proof_input* p) { return p ? p->cell_data : nullptr; }
extern "C" const uint8_t* cp2_proof_input_merkle_paths(const cp2_proof_input* p) { return p ? p->paths : nullptr; }
extern "C" const uint8_t* cp2_proof_input_slot_proof(const cp2_proof_input* p) { return p ? p->slot_proof.data() : nullptr; }
extern "C" const uint8_t* cp2_proof_input_leaf_hashes(const cp2_proof_input* p) { return p ? p->leaves : nullptr; }

// A proof input assembled from caller arrays (what the Nim shim's exportProofInputBN254 holds: a SlotProofInput[Hash]
// value, types.nim:52-60), so that the byte-exact writer below can be used on it.  Everything is copied.
extern "C" int cp2_proof_input_create(const cp2_config* cfg, uint64_t slot_idx, const uint8_t dataset_root[32], const uint8_t entropy[32],
                                      const uint8_t slot_root[32], const uint8_t* slot_proof, size_t n_samples, const uint64_t* cell_indices,
                                      const uint8_t* cell_data, const uint8_t* merkle_paths, const uint8_t* leaf_hashes,
                                      cp2_proof_input** out) try {
  if (!cfg || !dataset_root || !entropy || !slot_root || !out) return CP2_ERR_INVALID;
  if (cfg->max_depth < 0 || cfg->max_log2_nslots < 0 || (cfg->max_log2_nslots && !slot_proof)) return CP2_ERR_INVALID;
  if (n_samples && (!cell_data || !merkle_paths)) return CP2_ERR_INVALID;
  *out = nullptr;
  std::unique_ptr<cp2_proof_input> p(new cp2_proof_input());
  p->cfg = *cfg;
  p->cfg.file_base = nullptr;
  p->cfg.n_samples = n_samples;
  p->slot_idx = slot_idx;
  std::memcpy(p->dataset_root, dataset_root, 32);
  canonical_felt(entropy, p->entropy);   // a field element in the reference (types/bn254.nim:21): stored and printed as its residue, like the generate paths
  std::memcpy(p->slot_root, slot_root, 32);
  p->slot_proof.assign(slot_proof, slot_proof + (size_t)cfg->max_log2_nslots * 32);
  p->n_samples = n_samples;
  auto store = std::make_shared<BatchStore>();
  const size_t md = (size_t)cfg->max_depth, cs = cfg->cell_size;
  const size_t o_idx = 0, o_cells = o_idx + n_samples * 8, o_paths = o_cells + n_samples * cs, o_leaves = o_paths + n_samples * md * 32;
  store->heap.assign(o_leaves + n_samples * 32 + 8, 0);
  uint8_t* h = store->heap.data();
  if (n_samples) {
    if (cell_indices) std::memcpy(h + o_idx, cell_indices, n_samples * 8);
    std::memcpy(h + o_cells, cell_data, n_samples * cs);
    std::memcpy(h + o_paths, merkle_paths, n_samples * md * 32);
    if (leaf_hashes) std::memcpy(h + o_leaves, leaf_hashes, n_samples * 32);
  }
  p->store = store;
  p->indices = reinterpret_cast<const uint64_t*>(h + o_idx);   // the vector's storage is 16-byte aligned
  p->cell_data = h + o_cells;
  p->paths = h + o_paths;
  p->leaves = leaf_hashes ? h + o_leaves : nullptr;
  *out = p.release();
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

// ---- JSON: the byte-exact formatter lives in json_text.hpp (cp2text::text_head / text_body) -----------------------------
namespace {

int write_parts(const char* path, const std::string& a, const std::string& b) {
  FILE* f = std::fopen(path, "wb");
  if (!f) return CP2_ERR_IO;
  size_t w = std::fwrite(a.data(), 1, a.size(), f) + (b.empty() ? 0 : std::fwrite(b.data(), 1, b.size(), f));
  int rc = std::fclose(f);
  return (w == a.size() + b.size() && rc == 0) ? CP2_OK : CP2_ERR_IO;
}

// head, then the stored body of local slot s (resident or spilled)
int write_head_and_body(const char* path, const std::string& head, const BodyStore& bodies, size_t s) {
  FILE* f = std::fopen(path, "wb");
  if (!f) return CP2_ERR_IO;
  int st = std::fwrite(head.data(), 1, head.size(), f) == head.size() ? CP2_OK : CP2_ERR_IO;
  if (st == CP2_OK) st = bodies.write_to(s, f);
  if (std::fclose(f) != 0 && st == CP2_OK) st = CP2_ERR_IO;
  return st;
}

}  // namespace

static void proof_input_text(const cp2_proof_input* p, std::string& s) {
  s.clear();
  s.reserve(body_bound(p->cfg, p->n_samples) + head_bound(p->cfg));
  text_head(s, p->cfg, p->slot_idx, p->dataset_root, p->entropy, p->slot_root, p->slot_proof.data());
  text_body(s, p->cfg, p->n_samples, p->cell_data, p->paths);
}

extern "C" int cp2_proof_input_json(const cp2_proof_input* p, char** text, size_t* len) try {
  if (!p || !text) return CP2_ERR_INVALID;
  std::string s;
  proof_input_text(p, s);
  char* buf = (char*)std::malloc(s.size() + 1);
  if (!buf) return CP2_ERR_ALLOC;
  std::memcpy(buf, s.data(), s.size());
  buf[s.size()] = 0;
  *text = buf;
  if (len) *len = s.size();
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

// Serialise (and optionally write) many proof inputs on `threads` host threads.  paths == NULL or
// paths[i] == NULL: serialise only.  total_bytes (may be NULL) receives the summed text length.
extern "C" int cp2_proof_inputs_write_json_batch(const cp2_proof_input* const* ps, size_t n, const char* const* paths,
                                                 int threads, uint64_t* total_bytes) try {
  if (n && !ps) return CP2_ERR_INVALID;
  for (size_t i = 0; i < n; ++i)
    if (!ps[i]) return CP2_ERR_INVALID;
  if (threads < 1) threads = 1;
  if ((size_t)threads > n) threads = n ? (int)n : 1;
  std::vector<int> status(threads, CP2_OK);
  std::vector<uint64_t> bytes(threads, 0);
  auto work = [&](int t) {
    try {
      std::string s;
      for (size_t i = t; i < n; i += threads) {
        proof_input_text(ps[i], s);
        bytes[t] += s.size();
        if (paths && paths[i]) {
          int st = write_parts(paths[i], s, std::string());
          if (st != CP2_OK) status[t] = st;
        }
      }
    } catch (...) {
      status[t] = CP2_ERR_ALLOC;
    }
  };
  {
    Workers pool(threads - 1 > 0 ? threads - 1 : 1);   // joined by its destructor on every path out of this scope
    for (int t = 1; t < threads; ++t) pool.submit([&, t] { work(t); });
    work(0);
    pool.wait_idle();
  }
  uint64_t tot = 0;
  for (int t = 0; t < threads; ++t) {
    tot += bytes[t];
    if (status[t] != CP2_OK) return status[t];
  }
  if (total_bytes) *total_bytes = tot;
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

// Proof inputs for many slots of a built dataset, generated and serialised as a two-stage pipeline: while the host
// threads turn batch k into JSON text (and write it when dir != NULL: "<dir>/input_<slot>.json"), the GPU already
// samples and gathers batch k+1.
extern "C" int cp2_dataset_export_proof_inputs(cp2_dataset* ds, const uint64_t* slot_idx, size_t n, const uint8_t entropy[32],
                                               const char* dir, int threads, size_t batch, uint64_t* total_bytes) try {
  if (!ds || !entropy || (n && !slot_idx)) return CP2_ERR_INVALID;
  if (batch == 0) batch = 512;
  if (threads < 1) threads = 1;
  uint64_t bytes = 0;
  int status = CP2_OK;
  std::vector<cp2_proof_input*> cur, next;
  auto generate = [&](size_t b0, std::vector<cp2_proof_input*>& out) -> int {
    size_t m = std::min(batch, n - b0);
    out.assign(m, nullptr);
    return cp2_proof_inputs_generate_batch(ds, slot_idx + b0, m, entropy, out.data());
  };
  struct Release {   // whatever leaves this scope, the objects are freed
    std::vector<cp2_proof_input*>& v;
    ~Release() { for (auto* p : v) delete p; v.clear(); }
  } rel_cur{cur}, rel_next{next};
  if (n) status = generate(0, cur);
  for (size_t b0 = 0; status == CP2_OK && b0 < n; b0 += batch) {
    const size_t b1 = b0 + batch;
    // names first: nothing below may throw while the producer task is in flight except into the pool's joining destructor
    std::vector<std::string> names;
    std::vector<const char*> paths;
    if (dir) {
      for (size_t i = 0; i < cur.size(); ++i) names.push_back(std::string(dir) + "/input_" + std::to_string(slot_idx[b0 + i]) + ".json");
      for (auto& s2 : names) paths.push_back(s2.c_str());
    }
    int gen_status = CP2_OK, st = CP2_OK;
    uint64_t got = 0;
    {
      Workers producer(1);                                   // GPU stage of the NEXT batch; joined when this scope ends
      if (b1 < n) producer.submit([&] { gen_status = generate(b1, next); });
      st = cp2_proof_inputs_write_json_batch(cur.data(), cur.size(), dir ? paths.data() : nullptr, threads, &got);
    }
    bytes += got;
    for (auto* p : cur) delete p;
    cur.clear();
    if (st != CP2_OK) status = st;
    else if (gen_status != CP2_OK) status = gen_status;
    cur.swap(next);
  }
  if (total_bytes) *total_bytes = bytes;
  return status;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

// ---------------------------------------------------------------------------------------------
// streamed build: proof-input bodies of finished slots while later slots are still hashing
// ---------------------------------------------------------------------------------------------
namespace {

struct StreamRing {
  // Fake data (depth 4): pass k being enqueued, k - 1 waiting for its group's hashing (two chunks are in flight), k - 2 landing, k - 3
  // being formatted.  Slot files (depth 8): the building thread is also what FILLS the ingestion ring, so it must never sleep on a
  // pass that has not landed yet; it hands out the passes it finds landed (a query per turn) and blocks only when this ring is full --
  // on a pass enqueued eight turns earlier, while the device runs at most four turns behind the host.  (Without sampled cells -- the
  // formatting workers read those from the slot files -- a landing buffer is a third of the fake path's.)
  static constexpr int MAX_DEPTH = 8;
  int depth = 4;
  SampleHost host[MAX_DEPTH];
  hipEvent_t landed[MAX_DEPTH] = {};
  size_t s0[MAX_DEPTH] = {}, s1[MAX_DEPTH] = {};   // slot range parked in host[r]
  ~StreamRing() { for (auto e : landed) if (e) (void)hipEventDestroy(e); }   // host[] drain their stream themselves
  // body tasks still reading host[r]: the build thread sleeps on the condition variable until a ring slot is free
  void begin(int r, size_t n) { std::lock_guard<std::mutex> lk(mu); pending[r] = n; }
  void task_done(int r) {
    std::lock_guard<std::mutex> lk(mu);
    if (--pending[r] == 0) cv.notify_all();
  }
  void wait_free(int r) {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return pending[r] == 0; });
  }

 private:
  std::mutex mu;
  std::condition_variable cv;
  size_t pending[MAX_DEPTH] = {};
};

}  // namespace

// the streamed build keeping its trees as `tree_mode` says (1 every node, 2 compact, 0 roots only)
static int build_streamed_in_mode(cp2_ctx* ctx, const cp2_config* cfg, uint64_t first_slot, uint64_t n_local,
                                  const uint8_t entropy_in[32], int threads, size_t group_slots, int tree_mode, cp2_dataset** out) {
  if (!ctx || !cfg || !out || !entropy_in) return CP2_ERR_INVALID;
  uint8_t entropy[32];
  canonical_felt(entropy_in, entropy);
  *out = nullptr;
  CP2_TRY(dataset_check(cfg, first_slot, n_local));
  if (!is_pow2(cfg->n_cells)) return CP2_ERR_INVALID;                    // sample/bn254.nim:19-20
  if (cfg->n_samples && cfg->n_cells < 2) return CP2_ERR_INVALID;        // extractLowBits asserts k > 0, types/bn254.nim:48
  CP2_TRY(trees_check_geometry(cfg->cell_size, cfg->block_size, cfg->n_cells, n_local));
  {
    const size_t cpb = cfg->block_size / cfg->cell_size, depth = (layer_sizes_of(cpb).size() - 1) + (layer_sizes_of(cfg->n_cells / cpb).size() - 1);
    if (depth > (size_t)cfg->max_depth) return CP2_ERR_INVALID;          // padMerkleProof assert, types.nim:29
  }
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  if (threads < 1) threads = 1;
  const size_t ns = cfg->n_samples, md = (size_t)cfg->max_depth, cs = cfg->cell_size;
  // group: slots per layer pass / sampling pass.  Default: what fills the fake builder's staging chunk (2 GiB), at least 1.
  if (group_slots == 0) group_slots = std::max<size_t>(1, ctx->stage_bytes / std::max<size_t>(1, cfg->n_cells * cs));
  group_slots = std::min<size_t>(group_slots, n_local);

  std::unique_ptr<cp2_dataset> ds(dataset_new(ctx, cfg, first_slot, n_local));
  if (!ds) return CP2_ERR_ALLOC;
  ds->bodies.init(ctx, n_local);
  std::memcpy(ds->prep_entropy, entropy, 32);
  const bool from_file = ds->from_file;
  const cp2_config cfgv = ds->cfg;
  const std::string file_base = ds->file_base;

  hipStream_t aux = nullptr;                                   // the sampling / download stream (the context's third)
  CP2_TRY(aux_stream(ctx, &aux, 2));
  SampleDev dev;
  StreamRing ring;
  CP2_TRY(dev.init(ctx, group_slots, ns, md, cs, !from_file));
  ring.depth = from_file ? StreamRing::MAX_DEPTH : 4;
  for (int r = 0; r < ring.depth; ++r) {
    CP2_TRY(ring.host[r].init(ctx, group_slots, ns, md, cs, !from_file));
    CP2_HIP(ctx, hipEventCreateWithFlags(&ring.landed[r], hipEventDisableTiming));
  }
  CP2_HIP(ctx, hipMemcpyAsync(dev.entropy.p, entropy, 32, hipMemcpyHostToDevice, aux));
  hipEvent_t trees_ready = nullptr;
  CP2_HIP(ctx, hipEventCreateWithFlags(&trees_ready, hipEventDisableTiming));
  struct EvGuard { hipEvent_t e; ~EvGuard() { (void)hipEventDestroy(e); } } ev_guard{trees_ready};

  std::atomic<int> task_status{CP2_OK};
  std::mutex io_mu;
  std::string io_error;         // first slot file a formatting worker could not open
  cp2_dataset* dsp = ds.get();
  size_t n_groups = 0;          // sampling passes enqueued so far
  size_t consumed = 0;          // passes whose body tasks have been handed to the workers
  cp2k::TreeGeom geom;
  bool have_geom = false;
  size_t slot_base = 0;         // roots-only build: dataset-local index of the first slot of the batch being built
  StageTimer trace;
  {
    Workers pool(threads, 10);  // declared last: joins before anything above is destroyed.  Niced: the bodies are needed at the end, the device is fed now

    // hand the body tasks of pass `k` to the workers once its downloads have landed
    auto consume = [&](size_t k) -> int {
      const int r = (int)(k % (size_t)ring.depth);
      CP2_HIP(ctx, hipEventSynchronize(ring.landed[r]));
      const size_t a = ring.s0[r], b = ring.s1[r];
      ring.begin(r, b - a);
      for (size_t s = a; s < b; ++s) {
        pool.submit([&, s, a, r] {
          try {
            const uint8_t* paths = ring.host[r].paths.u8() + (s - a) * ns * md * 32;
            const uint64_t* idx = static_cast<const uint64_t*>(ring.host[r].idx.p) + (s - a) * ns;
            static thread_local std::string body;   // worst-case sized once per worker; the store keeps an exact-size copy
            body.clear();
            body.reserve(body_reserve(cfgv, ns));
            bool have = true;
            if (from_file) {      // sampled cells straight from the slot file (slot.nim:57-68)
              std::vector<uint8_t> cells(ns * cs);
              const std::string fname = slot_file_name(file_base, first_slot + s);
              int fd = open(fname.c_str(), O_RDONLY);
              if (fd < 0) {       // reported like the classic path ("cannot open <file>"); no body for this slot
                task_status.store(CP2_ERR_IO);
                std::lock_guard<std::mutex> lk(io_mu);
                if (io_error.empty()) io_error = "cannot open " + fname;
                have = false;
              } else {
                for (size_t c = 0; c < ns; ++c) read_file_cell(fd, cs, idx[c], &cells[c * cs]);
                close(fd);
                text_body(body, cfgv, ns, cells.data(), paths);
              }
            } else {
              text_body(body, cfgv, ns, ring.host[r].cells.u8() + (s - a) * ns * cs, paths);
            }
            if (have && dsp->bodies.put(s, body) != CP2_OK) task_status.store(CP2_ERR_IO);   // the store names the path (bodies.error)
          } catch (...) {
            task_status.store(CP2_ERR_ALLOC);
          }
          ring.task_done(r);
        });
      }
      return CP2_OK;
    };

    // the builders call this each time the trees of slots [a, b) are complete on the context's stream
    double hook_wait_ms = 0, hook_enqueue_ms = 0, hook_handout_ms = 0;   // CP2_TRACE: where the hook's time goes (it runs on the building thread)
    auto clock_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    SlotsDone on_done = [&](cp2_slot_trees* t, size_t a, size_t b, hipStream_t tree_stream) -> int {
      if (!have_geom) { trees_geom(t, &geom); have_geom = true; }
      for (size_t g0 = a; g0 < b; g0 += group_slots) {
        const size_t g1 = std::min(b, g0 + group_slots);
        const size_t k = n_groups;
        const int r = (int)(k % (size_t)ring.depth);
        const double h0 = trace.on ? clock_ms() : 0;
        // ring slot r was last used by pass k - DEPTH: its tasks must have been handed out and finished
        while (consumed + (size_t)ring.depth <= k) { CP2_TRY(consume(consumed)); ++consumed; }
        ring.wait_free(r);
        const double h1 = trace.on ? clock_ms() : 0;
        CP2_HIP(ctx, hipEventRecord(trees_ready, tree_stream));   // the group's layer passes end on one of the two hashing streams
        CP2_HIP(ctx, hipStreamWaitEvent(aux, trees_ready, 0));
        CP2_TRY(enqueue_sampling(t, geom, dev, ring.host[r], nullptr, g0, g1 - g0, ns, md, !from_file, aux));
        CP2_HIP(ctx, hipEventRecord(ring.landed[r], aux));
        ring.s0[r] = slot_base + g0;                           // dataset-local slot indices (bodies, file names); g0 counts inside `t`
        ring.s1[r] = slot_base + g1;
        ++n_groups;
        const double h2 = trace.on ? clock_ms() : 0;
        // Two chunks are hashed at a time (one per hashing stream), so the pass before this one still waits for its group's
        // hashing: waiting for it here would keep the builder from enqueueing the next chunk until then.  The pass before THAT
        // has landed or is about to: hand it out.
        if (!from_file) {
          while (consumed + (stream_serial() ? 1 : 2) < n_groups) { CP2_TRY(consume(consumed)); ++consumed; }
        } else {
          // slot files: this thread fills the ingestion ring between the builder's turns and must not sleep on the device here
          // (StreamRing): only what has landed already is handed out
          while (consumed < n_groups) {
            const hipError_t q = hipEventQuery(ring.landed[consumed % (size_t)ring.depth]);
            if (q == hipErrorNotReady) { (void)hipGetLastError(); break; }
            CP2_HIP(ctx, q);
            CP2_TRY(consume(consumed));
            ++consumed;
          }
        }
        if (trace.on) { const double h3 = clock_ms(); hook_wait_ms += h1 - h0; hook_enqueue_ms += h2 - h1; hook_handout_ms += h3 - h2; }
      }
      return CP2_OK;
    };

    int st = CP2_OK;
    if (tree_mode == 1) {
      st = dataset_build_trees(dsp, group_slots, on_done);
      trace.lap("trees (sampling overlapped)");
      if (trace.on) std::fprintf(stderr, "[cp2 trace] the sampling hook over %zu passes: %.1f ms waiting for a free landing buffer, %.1f ms enqueueing, %.1f ms handing landed passes to the formatting threads\n", n_groups, hook_wait_ms, hook_enqueue_ms, hook_handout_ms);
      while (st == CP2_OK && consumed < n_groups) { st = consume(consumed); ++consumed; }
      pool.wait_idle();
      (void)hipStreamSynchronize(aux);
      trace.lap("last bodies");
    } else {
      // Compact / roots only (the nodes of all local slots do not fit the device, or the caller said so): the same pipeline over
      // batches of slots -- about 1 GiB of nodes: 4 slots of 8 GiB, a whole number of groups -- whose nodes are dropped once the
      // batch's bodies are made and what the mode keeps is copied out; roots (or compact layers) and bodies stay.
      size_t batch = transient_batch_slots(ctx, cfgv, n_local);
      batch = std::max(group_slots, batch / group_slots * group_slots);
      st = dataset_alloc_kept(dsp, tree_mode);
      {
        // The batches PIPELINE, fake data and slot files alike (BuildScratch: two node buffers used alternately, nothing synchronised per batch).  A
        // batch's tail -- the layer passes of its last group, that group's sampling, gathers and downloads, the copy-out of what
        // is kept: all on the third stream -- runs while the next batch's generation and hashing already occupy the two hashing
        // streams (round 4 drained the device here: 816 s against 806 s for the roots alone over 32 TiB).  Node buffer b goes to
        // batch k + 2 once batch k's copy-out and last sampling have completed.
        BuildScratch scratch;                                    // drains the context's streams before its buffers go
        hipStream_t layer_stream = nullptr;
        if (st == CP2_OK) st = aux_stream(ctx, &layer_stream, 1);
        hipEvent_t done_ev[2] = {nullptr, nullptr}, sampled[2] = {nullptr, nullptr};
        struct EvGuard { hipEvent_t* a; hipEvent_t* b; ~EvGuard() { for (int i = 0; i < 2; ++i) { if (a[i]) (void)hipEventDestroy(a[i]); if (b[i]) (void)hipEventDestroy(b[i]); } } } ev_guard2{done_ev, sampled};
        for (int i = 0; i < 2 && st == CP2_OK; ++i)
          if (hipEventCreateWithFlags(&done_ev[i], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&sampled[i], hipEventDisableTiming) != hipSuccess) {
            ctx->err = "hipEventCreate failed";
            st = CP2_ERR_HIP;
          }
        size_t k = 0;
        for (size_t base = 0; st == CP2_OK && base < n_local; base += batch, ++k) {
          const size_t nb = std::min(batch, (size_t)n_local - base);
          const int b = (int)(k & 1);
          if (k >= 2 && hipEventSynchronize(done_ev[b]) != hipSuccess) { ctx->err = "streamed build: a batch failed on the device"; st = CP2_ERR_HIP; break; }
          slot_base = base;
          have_geom = false;                                     // node offsets are those of THIS batch's layout
          cp2_slot_trees* t = nullptr;
          st = from_file ? trees_build_files(ctx, file_base, first_slot + base, nb, cfgv.cell_size, cfgv.block_size, cfgv.n_cells, group_slots, on_done, &t, 1, true, &scratch, b)
                         : trees_build_fake(ctx, cfgv.seed, first_slot + base, nb, cfgv.cell_size, cfgv.block_size, cfgv.n_cells, group_slots, on_done, &t, 1, true, &scratch, b);
          layer_stream = scratch.tail_stream ? scratch.tail_stream : layer_stream;         // where the builder put the batch's layer passes (the third stream)
          if (st == CP2_OK) st = dataset_keep_from_batch(dsp, t, base, layer_stream);     // follows the batch's last layer pass on that stream
          if (st == CP2_OK && (hipEventRecord(sampled[b], aux) != hipSuccess || hipStreamWaitEvent(layer_stream, sampled[b], 0) != hipSuccess ||
                               hipEventRecord(done_ev[b], layer_stream) != hipSuccess)) {
            ctx->err = "streamed build: event bookkeeping failed";
            st = CP2_ERR_HIP;
          }
          cp2_slot_trees_free(t);                                // (the batch's nodes are the scratch's: nothing is waited for here)
          if (trace.on && ((k % 32) == 31 || base + nb == n_local))
            std::fprintf(stderr, "[cp2 trace] streamed %s build: %zu of %llu slots enqueued\n", tree_mode == 2 ? "compact" : "roots-only", base + nb, (unsigned long long)n_local);
        }
        while (st == CP2_OK && consumed < n_groups) { st = consume(consumed); ++consumed; }
        pool.wait_idle();
        if (st == CP2_OK && (hipStreamSynchronize(ctx->stream) != hipSuccess || hipStreamSynchronize(layer_stream) != hipSuccess || hipStreamSynchronize(aux) != hipSuccess)) {
          (void)hipGetLastError();
          ctx->err = "streamed build: a batch failed on the device";
          st = CP2_ERR_HIP;
        }
      }
      trace.lap("trees + bodies, batch by batch (trees dropped)");
    }
    if (st != CP2_OK) return st;
  }
  if (task_status.load() != CP2_OK) {
    if (!io_error.empty()) ctx->err = io_error;
    else if (!ds->bodies.error.empty()) ctx->err = ds->bodies.error;
    return task_status.load();
  }
  ds->prepared = true;
  *out = ds.release();
  return CP2_OK;
}

extern "C" int cp2_dataset_build_streamed(cp2_ctx* ctx, const cp2_config* cfg, uint64_t first_slot, uint64_t n_local,
                                          const uint8_t entropy_in[32], int threads, size_t group_slots, cp2_dataset** out) try {
  if (!ctx || !cfg || !out || !entropy_in) return CP2_ERR_INVALID;
  *out = nullptr;
  CP2_REFUSE_STUCK(ctx);
  CP2_TRY(dataset_check(cfg, first_slot, n_local));
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  bool automatic = false;
  int mode = dataset_tree_mode(ctx, *cfg, n_local, &automatic);
  if (mode < 0) return CP2_ERR_INVALID;
  for (;;) {   // the same step-down chain as cp2_dataset_build: an automatic choice that does not fit after all is retried one mode down
    const int st = build_streamed_in_mode(ctx, cfg, first_slot, n_local, entropy_in, threads, group_slots, mode, out);
    if (st != CP2_ERR_ALLOC || !automatic || !step_down(ctx, &mode, "streamed build")) return st;
  }
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

// Finish the proof inputs a streamed build prepared: the dataset tree must be set (cp2_dataset_set_roots; implied when
// all slots are local).  Heads (dataSetRoot .. slotProof) are formatted and joined with the stored bodies on `threads`
// host threads; dir != NULL writes "<dir>/input_<slot>.json" for every local slot.  Text identical to cp2_proof_input_json.
extern "C" int cp2_dataset_export_streamed(cp2_dataset* ds, const char* dir, int threads, uint64_t* total_bytes) try {
  if (!ds || !ds->prepared) return CP2_ERR_INVALID;
  if (!ds->have_roots) CP2_TRY(cp2_dataset_set_roots(ds, nullptr));
  if (ds->dsizes.size() - 1 > (size_t)ds->cfg.max_log2_nslots) return CP2_ERR_INVALID;   // padMerkleProof assert
  if (threads < 1) threads = 1;
  const size_t n = ds->n_local;
  if ((size_t)threads > n) threads = (int)n;
  std::vector<int> status(threads, CP2_OK);
  std::vector<uint64_t> bytes(threads, 0);
  auto work = [&](int t) {
    try {
      std::string head;
      std::vector<uint8_t> proof;
      for (size_t s = t; s < n; s += threads) {
        const uint64_t slot = ds->first_slot + s;
        head.clear();
        fill_slot_proof(ds, slot, proof);
        text_head(head, ds->cfg, slot, &ds->dlayers[ds->dlayers.size() - 32], ds->prep_entropy, &ds->dlayers[slot * 32], proof.data());
        bytes[t] += head.size() + ds->bodies.size[s];
        if (dir) {
          std::string name = std::string(dir) + "/input_" + std::to_string(slot) + ".json";
          int st = write_head_and_body(name.c_str(), head, ds->bodies, s);
          if (st != CP2_OK) status[t] = st;
        }
      }
    } catch (...) {
      status[t] = CP2_ERR_ALLOC;
    }
  };
  {
    Workers pool(threads > 1 ? threads - 1 : 1);
    for (int t = 1; t < threads; ++t) pool.submit([&, t] { work(t); });
    work(0);
    pool.wait_idle();
  }
  uint64_t tot = 0;
  for (int t = 0; t < threads; ++t) {
    tot += bytes[t];
    if (status[t] != CP2_OK) return status[t];
  }
  if (total_bytes) *total_bytes = tot;
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

// the full text of one prepared slot (head + body) into a malloc'ed buffer (cp2_free_buffer)
extern "C" int cp2_dataset_streamed_json(cp2_dataset* ds, uint64_t slot_idx, char** text, size_t* len) try {
  if (!ds || !ds->prepared || !text) return CP2_ERR_INVALID;
  if (slot_idx < ds->first_slot || slot_idx >= ds->first_slot + ds->n_local) return CP2_ERR_INVALID;
  if (!ds->have_roots) CP2_TRY(cp2_dataset_set_roots(ds, nullptr));
  if (ds->dsizes.size() - 1 > (size_t)ds->cfg.max_log2_nslots) return CP2_ERR_INVALID;
  std::string head;
  std::vector<uint8_t> proof;
  fill_slot_proof(ds, slot_idx, proof);
  text_head(head, ds->cfg, slot_idx, &ds->dlayers[ds->dlayers.size() - 32], ds->prep_entropy, &ds->dlayers[slot_idx * 32], proof.data());
  CP2_TRY(ds->bodies.append(slot_idx - ds->first_slot, head));
  char* buf = (char*)std::malloc(head.size() + 1);
  if (!buf) return CP2_ERR_ALLOC;
  std::memcpy(buf, head.data(), head.size());
  buf[head.size()] = 0;
  *text = buf;
  if (len) *len = head.size();
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" void cp2_free_buffer(void* p) { std::free(p); }

extern "C" int cp2_proof_input_write_json(const cp2_proof_input* p, const char* path) try {
  if (!p || !path) return CP2_ERR_INVALID;
  StageTimer trace;
  std::string s;
  proof_input_text(p, s);
  const int st = write_parts(path, s, std::string());
  trace.lap("input.json formatted + written");
  return st;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" int cp2_write_circom_main(const cp2_config* cfg, const char* path) try {
  if (!cfg || !path || cfg->cell_size == 0) return CP2_ERR_INVALID;
  if (cfg->block_size % cfg->cell_size) return CP2_ERR_INVALID;
  uint64_t cpb = cfg->block_size / cfg->cell_size;
  if (!is_pow2(cpb)) return CP2_ERR_INVALID;                     // exactLog2 assert, misc.nim:25-28
  int depth = 0;
  while ((1ULL << depth) < cpb) ++depth;
  FILE* f = std::fopen(path, "wb");
  if (!f) return CP2_ERR_IO;
  std::fprintf(f, "pragma circom 2.0.0;\n");
  std::fprintf(f, "include \"sample_cells.circom\";\n");
  std::fprintf(f, "// SampleAndProven( maxDepth, maxLog2NSlots, blockTreeDepth, nFieldElemsPerCell, nSamples )\n");
  std::fprintf(f, "component main {public [entropy,dataSetRoot,slotIndex]} = SampleAndProve(%d, %d, %d, %llu, %llu);\n",
               cfg->max_depth, cfg->max_log2_nslots, depth, (unsigned long long)((cfg->cell_size + 30) / 31),
               (unsigned long long)cfg->n_samples);
  return std::fclose(f) == 0 ? CP2_OK : CP2_ERR_IO;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

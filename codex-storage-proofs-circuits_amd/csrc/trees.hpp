// Slot-tree batch shared by slot_trees.cpp and proof_input.cpp (not installed).
#pragma once
#include <functional>
#include <string>
#include <vector>

#include "internal.hpp"
#include "kernels.hpp"

enum class CellSrc { Fake, Dev, Host, File };

struct cp2_slot_trees {
  cp2_ctx* ctx = nullptr;
  size_t n_slots = 0, cell_size = 0, block_size = 0, n_cells = 0, cpb = 0, nblocks = 0;
  std::vector<size_t> bsizes, tsizes;   // per-tree layer sizes: block tree (cpb leaves), big tree (nblocks leaves)
  std::vector<size_t> boff, toff;       // element offsets of each layer in `nodes` (layer-major)
  cp2i::DevBuf nodes;
  // where sampled cells come from
  CellSrc src = CellSrc::Fake;
  uint64_t dataset_seed = 0, first_slot = 0;
  // Slots cut into units (several devices sharing ONE slot, multi_gpu.cpp): this batch then holds `n_slots` UNITS of `n_cells`
  // cells each, unit i being unit first_slot + i of the dataset, unit u = cells [(u % units_per_slot) * n_cells, +n_cells) of
  // slot u / units_per_slot.  1: a unit is a whole slot (everything outside multi_gpu.cpp).
  uint64_t units_per_slot = 1;
  bool pooled_nodes = false;            // node buffer from the context's scratch pool (transient batches: roots-only datasets)
  const uint8_t* d_cells = nullptr;     // not owned
  const uint8_t* h_cells = nullptr;     // not owned
  std::string file_base;
};

namespace cp2i {

constexpr uint64_t NO_ROW = ~0ULL;

// Called by the builders each time the trees of slots [s0, s1) (indices inside the batch) are complete ON STREAM `st`
// (everything enqueued, nothing synchronised): the streamed proof-input path hangs its sampling / gather / download
// of those slots on that stream (the context's third) while later slots are still hashing on the other two.
using SlotsDone = std::function<int(cp2_slot_trees* t, size_t s0, size_t s1, hipStream_t st)>;

// Scratch that outlives one builder call, so that consecutive batches of a transient (compact / roots-only) build PIPELINE instead of
// draining the device between them: two staging buffers for generated cells and two node buffers used alternately.  With it
// trees_build_fake returns with its work ENQUEUED (nothing synchronised, the batch's nodes a borrowed view of nodes[node_slot]):
// the next batch's generation and hashing start on the context's two hashing streams while this batch's layer passes and
// copy-outs still run on the third.  The owner waits for whatever reads nodes[b] before it hands slot b to another batch, and drains the
// context's streams before the scratch goes (its destructor does).
struct BuildScratch {
  cp2_ctx* ctx = nullptr;
  DevBuf stage[2], nodes[2];
  hipStream_t tail_stream = nullptr;   // set by the builder: the stream the last batch's layer passes were enqueued on
  std::shared_ptr<void> file_pipe;     // slot files: the ingestion pipe (pinned ring, device ring, copy stream) the batches share; declared
                                       // last, so it drains and goes before the buffers above
  ~BuildScratch();
};

int trees_check_geometry(size_t cell_size, size_t block_size, size_t n_cells, size_t n_slots);
// fake-data or slot-file trees; `group` = how many finished slots to batch per layer pass / callback (0: all at the end)
// units_per_slot > 1: first_slot / n_slots / n_cells count UNITS and the cells of one unit (cp2_slot_trees above)
// pooled_nodes: the node buffer comes from (and goes back to) the context's scratch pool instead of hipMalloc / hipFree
// scratch != nullptr: pipelined (BuildScratch above); the returned batch borrows scratch->nodes[node_slot] (both builders)
int trees_build_fake(cp2_ctx* ctx, uint64_t dataset_seed, uint64_t first_slot, size_t n_slots, size_t cell_size, size_t block_size,
                     size_t n_cells, size_t group, const SlotsDone& done, cp2_slot_trees** out, uint64_t units_per_slot = 1,
                     bool pooled_nodes = false, BuildScratch* scratch = nullptr, int node_slot = 0);
int trees_build_files(cp2_ctx* ctx, const std::string& base, uint64_t first_slot, size_t n_slots, size_t cell_size,
                      size_t block_size, size_t n_cells, size_t group, const SlotsDone& done, cp2_slot_trees** out,
                      uint64_t units_per_slot = 1, bool pooled_nodes = false, BuildScratch* scratch = nullptr, int node_slot = 0);
// bytes of the node buffer of a batch of n_slots slots of this geometry (all layers, 32 bytes per node)
size_t trees_node_bytes(size_t n_slots, size_t cell_size, size_t block_size, size_t n_cells);
void trees_geom(const cp2_slot_trees* t, cp2k::TreeGeom* g);
// node-row indices of the merged path of `cell` in slot `slot` (host twin of k_sample_paths' row arithmetic)
void path_rows(const cp2_slot_trees* t, size_t slot, uint64_t cell, size_t max_depth, uint64_t* rows);
// merged paths (padded to max_depth, + leaf hashes) of n (slot-in-batch, cell) pairs of a batch in one gather
int trees_paths_multi(cp2_slot_trees* t, const uint64_t* slot_idx, const uint64_t* cell_idx, size_t n, size_t max_depth, uint8_t* out, uint8_t* leaf_hashes);
// one cell of a slot file, zero-filled past EOF (slot.nim:57-68); fd < 0: all zeros
void read_file_cell(int fd, size_t cell_size, uint64_t cell, uint8_t* out);
std::string slot_file_name(const std::string& base, uint64_t slot);
bool is_pow2(uint64_t x);

// Persisted form of what a compact / roots-only dataset keeps (proof_input.cpp): the kept layers of `n_slots` local slots plus what
// they were built from.  kept_load fills the device buffer and returns CP2_OK only when the file is intact and describes exactly
// `want` (and, for the SlotFile source, slot files of unchanged size and mtime); anything else is CP2_ERR_IO and means "rebuild".
struct KeptMeta {
  uint64_t n_slots = 0, cell_size = 0, block_size = 0, n_cells = 0, src = 0, dataset_seed = 0, first_slot = 0, mode = 0;
  std::string file_base;
};
int kept_save(cp2_ctx* ctx, const char* path, const KeptMeta& meta, const void* d_buf, size_t bytes);
int kept_load(cp2_ctx* ctx, const char* path, const KeptMeta& want, void* d_buf, size_t bytes);

// After cp2_dataset_set_roots*: do rows [first_slot, first_slot + n_local) of the dataset tree's bottom layer hold THIS dataset's own
// slot roots?  One small download.  The multi-device exchange is verified with it (multi_gpu.cpp).
int dataset_own_roots_in_place(cp2_dataset* ds, bool* ok);

// stage timings on stderr when CP2_TRACE is set (the reference's only tracing is shell `time`, workflow/prove.sh:30-37)
bool stream_serial();   // CP2_STREAM_SERIAL=1 (A/B tooling): the streamed build hashes its groups one launch after the other, as rounds 2-4 did

struct StageTimer {
  bool on;
  double t0;
  StageTimer();
  void lap(const char* what);
};

}  // namespace cp2i

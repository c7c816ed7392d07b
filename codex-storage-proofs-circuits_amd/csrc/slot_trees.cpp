// Slot trees behind the C ABI (include/codex_p2.h): builders for the four cell sources, the streaming
// ingestion pipe, the persisted-tree cache and path lookup.
//
// Mirrors reference/nim/proof_input/src/gen_input/bn254.nim:21-33 (buildSlotTreeFull), blocks/bn254.nim:60-67
// (networkBlockTree), slot.nim:57-73 (slot data), merkle.nim:21-42,86-100 + types.nim:27-37 (paths).  All hashing
// runs in the HIP kernels; what stays on the host is index arithmetic and file / memory staging.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "trees.hpp"
#include "fake_turns.hpp"
#include "ingest_turns.hpp"
#include "fill_pipeline.hpp"

using namespace cp2i;

namespace cp2i {

bool is_pow2(uint64_t x) { return x && !(x & (x - 1)); }

static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
StageTimer::StageTimer() : on(std::getenv("CP2_TRACE") != nullptr), t0(now_ms()) {}
void StageTimer::lap(const char* what) {
  if (!on) return;
  double t1 = now_ms();
  std::fprintf(stderr, "[cp2 trace] %-34s %9.3f ms\n", what, t1 - t0);
  t0 = t1;
}

int aux_stream(cp2_ctx* ctx, hipStream_t* out, int which) {
  hipStream_t& st = which == 2 ? ctx->aux2_stream : ctx->aux_stream;
  if (!st) CP2_HIP(ctx, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  *out = st;
  return CP2_OK;
}

std::string slot_file_name(const std::string& base, uint64_t slot) { return fill_slot_file_name(base, slot); }   // dataset.nim:34

void read_file_cell(int fd, size_t cell_size, uint64_t cell, uint8_t* out) {
  size_t done = 0;
  while (fd >= 0 && done < cell_size) {
    ssize_t r = pread(fd, out + done, cell_size - done, (off_t)(cell * cell_size + done));
    if (r <= 0) break;
    done += (size_t)r;
  }
  if (done < cell_size) std::memset(out + done, 0, cell_size - done);
}

int trees_check_geometry(size_t cell_size, size_t block_size, size_t n_cells, size_t n_slots) {
  if (cell_size == 0 || block_size == 0 || n_cells == 0 || n_slots == 0) return CP2_ERR_INVALID;
  if (block_size % cell_size != 0) return CP2_ERR_INVALID;        // types.nim:104-107 cellsPerBlock assert
  size_t cpb = block_size / cell_size;
  if (n_cells % cpb != 0) return CP2_ERR_INVALID;                 // gen_input/bn254.nim:25 assert
  // sizes a kernel-argument geometry and 64-bit byte offsets can hold (also bounds what a cache header may claim)
  if (cell_size > ((size_t)1 << 30) || n_cells > ((size_t)1 << 40) || n_slots > ((size_t)1 << 32)) return CP2_ERR_INVALID;
  if ((unsigned __int128)n_cells * n_slots > ((unsigned __int128)1 << 44)) return CP2_ERR_INVALID;
  return CP2_OK;
}

}  // namespace cp2i

cp2i::BuildScratch::~BuildScratch() {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->aux_stream) (void)hipStreamSynchronize(ctx->aux_stream);
  if (ctx->aux2_stream) (void)hipStreamSynchronize(ctx->aux2_stream);
}

static int trees_layout(cp2_slot_trees* t, cp2i::DevBuf* borrow_from = nullptr) {
  t->bsizes = layer_sizes_of(t->cpb);
  t->tsizes = layer_sizes_of(t->nblocks);
  if (t->bsizes.size() > (size_t)cp2k::TreeGeom::MAX_LAYERS || t->tsizes.size() > (size_t)cp2k::TreeGeom::MAX_LAYERS) return CP2_ERR_INVALID;
  size_t off = 0;
  t->boff.clear();
  t->toff.clear();
  const size_t nb = t->n_slots * t->nblocks;
  for (size_t k = 0; k < t->bsizes.size(); ++k) {
    t->boff.push_back(off);
    if (k + 1 < t->bsizes.size()) off += nb * t->bsizes[k];
  }
  // the last block-tree layer (one root per block) is layer 0 of the big trees
  for (size_t k = 0; k < t->tsizes.size(); ++k) {
    t->toff.push_back(off);
    off += t->n_slots * t->tsizes[k];
  }
  if (borrow_from) {            // a pipeline's node buffer: (re)allocated only when too small (the first batch is the largest)
    if (borrow_from->bytes < off * 32) CP2_TRY(borrow_from->scratch(t->ctx, off * 32));
    t->nodes.borrow(borrow_from->p, off * 32);
    return CP2_OK;
  }
  return t->pooled_nodes ? t->nodes.scratch(t->ctx, off * 32) : t->nodes.alloc(t->ctx, off * 32);
}

size_t cp2i::trees_node_bytes(size_t n_slots, size_t cell_size, size_t block_size, size_t n_cells) {
  const size_t cpb = block_size / cell_size, nblocks = n_cells / cpb;
  size_t per_block = 0, per_slot = 0;
  const std::vector<size_t> b = layer_sizes_of(cpb), t = layer_sizes_of(nblocks);
  for (size_t k = 0; k + 1 < b.size(); ++k) per_block += b[k];     // the block roots are layer 0 of the slot tree
  for (size_t m : t) per_slot += m;
  return n_slots * (nblocks * per_block + per_slot) * 32;
}

void cp2i::trees_geom(const cp2_slot_trees* t, cp2k::TreeGeom* g) {
  std::memset(g, 0, sizeof *g);
  g->nb = (uint32_t)t->bsizes.size();
  g->nt = (uint32_t)t->tsizes.size();
  g->cpb = t->cpb;
  g->nblocks = t->nblocks;
  g->n_cells = t->n_cells;
  for (size_t k = 0; k < t->bsizes.size(); ++k) { g->boff[k] = t->boff[k]; g->bsz[k] = t->bsizes[k]; }
  for (size_t k = 0; k < t->tsizes.size(); ++k) { g->toff[k] = t->toff[k]; g->tsz[k] = t->tsizes[k]; }
}

static cp2_slot_trees* trees_new(cp2_ctx* ctx, size_t n_slots, size_t cell_size, size_t block_size, size_t n_cells) {
  cp2_slot_trees* t = new (std::nothrow) cp2_slot_trees();
  if (!t) return nullptr;
  t->ctx = ctx;
  t->n_slots = n_slots;
  t->cell_size = cell_size;
  t->block_size = block_size;
  t->n_cells = n_cells;
  t->cpb = block_size / cell_size;
  t->nblocks = n_cells / t->cpb;
  return t;
}

// all layers above the cell hashes of slots [s0, s1) (their cell hashes are already in layer 0)
static int trees_build_layers(cp2_slot_trees* t, size_t s0, size_t s1, hipStream_t st) {
  cp2_ctx* ctx = t->ctx;
  uint8_t* base = t->nodes.u8();
  const size_t ns = s1 - s0, nb = ns * t->nblocks;
  for (size_t k = 0; k + 1 < t->bsizes.size(); ++k)   // networkBlockTree, blocks/bn254.nim:60-67
    CP2_HIP(ctx, cp2k::launch_compress_layer(base + (t->boff[k] + s0 * t->nblocks * t->bsizes[k]) * 32,
                                             base + (t->boff[k + 1] + s0 * t->nblocks * t->bsizes[k + 1]) * 32, t->bsizes[k], nb,
                                             k == 0, t->bsizes[k], t->bsizes[k + 1], st));
  for (size_t k = 0; k + 1 < t->tsizes.size(); ++k)   // bigTree, gen_input/bn254.nim:28-29
    CP2_HIP(ctx, cp2k::launch_compress_layer(base + (t->toff[k] + s0 * t->tsizes[k]) * 32, base + (t->toff[k + 1] + s0 * t->tsizes[k + 1]) * 32,
                                             t->tsizes[k], ns, k == 0, t->tsizes[k], t->tsizes[k + 1], st));
  return CP2_OK;
}

// Tracks which slots have all their cells hashed and runs the layer passes (and the caller's hook) group by group.
// Cell hashing alternates between the context's two streams, chunk by chunk (hs[0] = ctx->stream, hs[1] = the second
// stream): consecutive kernels of one stream leave the GPU half empty while the last workgroups of one retire and the
// next has not started, and with two streams the next chunk's workgroups take the freed CUs at once (ingestion: +7...15 %,
// profiles/r02_ingest_scaling.txt; fake-data build of 4096 slots: -0.9 %).  With GROUPS (the streamed proof-input path) the
// group's layer passes -- small launches, each as long as one permutation chain (~0.1 ms) -- and whatever the caller's hook
// enqueues behind them go to the context's THIRD stream (on a hashing stream they would hold back the chunks queued behind them
// until the group's last chunk, hashed on the other stream, has finished), and the fake builder's hash launches leave a third
// of every CU free (launch_hash_cells' leave_room): next to a launch that holds every workgroup slot, a chain of small dependent
// kernels finishes only when that launch drains.  Rounds 2-4 hashed groups on the first stream alone, one launch after the
// other, every launch's tail exposed (43 GB/s per launch against 47 with two in flight); round 5's A/B on one box, configs[3]
// streamed: serial 0.858 s, two streams 0.837 s, two streams with room left 0.813 s (the plain tree build: 0.791 s).
// CP2_STREAM_SERIAL=1 (A/B tooling) restores the serial order at full occupancy.
bool cp2i::stream_serial() {
  static const bool v = [] { const char* e = std::getenv("CP2_STREAM_SERIAL"); return e && e[0] == '1'; }();
  return v;
}

namespace {
struct LayerScheduler {
  cp2_slot_trees* t;
  size_t group;
  const SlotsDone& done;
  size_t built = 0;
  bool take_all = false;                     // groups of varying size (the fake builder's ramp-down): every complete slot goes at once
  size_t ramp_min = 0;                       // groups of `group` slots whose last passes shrink down to this many slots (the slot-file builder; layer_take)
  hipStream_t hs[2] = {nullptr, nullptr};
  hipEvent_t ev[2] = {nullptr, nullptr};     // the latest hashing enqueued on each stream
  bool detached = false;                     // pipelined batches (BuildScratch): nothing is waited for here, the pipeline's owner does
  hipStream_t tail = nullptr;                // detached or groups: the layer passes go to the context's THIRD stream -- a hashing stream must not
                                             // wait for a chunk hashed on the other one, or nothing could overlap that chunk's tail: the next
                                             // chunk (or the next batch's first) is queued on it and runs beside that tail instead
  ~LayerScheduler() {
    for (int i = 0; i < 2; ++i) {
      if (hs[i] && !detached) (void)hipStreamSynchronize(hs[i]);
      if (ev[i]) (void)hipEventDestroy(ev[i]);   // (an event that is still pending is released once it has completed)
    }
    if (tail && !detached) (void)hipStreamSynchronize(tail);
  }
  int init() {
    cp2_ctx* ctx = t->ctx;
    hs[0] = ctx->stream;
    CP2_TRY(aux_stream(ctx, &hs[1]));
    for (int i = 0; i < 2; ++i) {
      CP2_HIP(ctx, hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
      CP2_HIP(ctx, hipEventRecord(ev[i], hs[i]));
    }
    if (detached || group) CP2_TRY(aux_stream(ctx, &tail, 2));
    return CP2_OK;
  }
  // the stream this batch's layer passes (and whatever follows them: the hook, the caller's copy-out) are enqueued on
  hipStream_t layer_stream() const { return tail ? tail : hs[0]; }
  // a chunk's hashing has just been enqueued on hs[s]
  int hashed_on(int s) {
    CP2_HIP(t->ctx, hipEventRecord(ev[s], hs[s]));
    return CP2_OK;
  }
  // cells [0, cells_hashed) of the batch are enqueued for hashing, the last chunk on hs[s]
  int advance(size_t cells_hashed, bool final, int s) {
    cp2_ctx* ctx = t->ctx;
    const size_t complete = cells_hashed / t->n_cells;
    for (;;) {
      const size_t take = layer_take(complete, built, group, take_all, final, t->n_slots, ramp_min);   // csrc/ingest_turns.hpp (walked by the CPU suite)
      if (!take) return CP2_OK;
      (void)s;
      hipStream_t ls = layer_stream();          // the third stream (groups, pipelined batches), else everything ends on the context's
      if (tail) CP2_HIP(ctx, hipStreamWaitEvent(ls, ev[0], 0));
      CP2_HIP(ctx, hipStreamWaitEvent(ls, ev[1], 0));            // cells of these slots were (also) hashed on the second stream
      CP2_TRY(trees_build_layers(t, built, built + take, ls));
      if (done) CP2_TRY(done(t, built, built + take, ls));
      built += take;
    }
  }
  int finish() {   // everything of both streams done
    cp2_ctx* ctx = t->ctx;
    if (detached) return CP2_OK;
    CP2_HIP(ctx, hipStreamSynchronize(hs[0]));
    CP2_HIP(ctx, hipStreamSynchronize(hs[1]));
    if (tail) CP2_HIP(ctx, hipStreamSynchronize(tail));
    return CP2_OK;
  }
};
}  // namespace

int cp2i::trees_build_fake(cp2_ctx* ctx, uint64_t dataset_seed, uint64_t first_slot, size_t n_slots, size_t cell_size,
                           size_t block_size, size_t n_cells, size_t group, const SlotsDone& done, cp2_slot_trees** out,
                           uint64_t units_per_slot, bool pooled_nodes, BuildScratch* scratch, int node_slot) {
  *out = nullptr;
  if (units_per_slot == 0) return CP2_ERR_INVALID;
  CP2_REFUSE_STUCK(ctx);
  CP2_TRY(trees_check_geometry(cell_size, block_size, n_cells, n_slots));
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  std::unique_ptr<cp2_slot_trees> t(trees_new(ctx, n_slots, cell_size, block_size, n_cells));
  if (!t) return CP2_ERR_ALLOC;
  t->src = CellSrc::Fake;
  t->dataset_seed = dataset_seed;
  t->first_slot = first_slot;
  t->units_per_slot = units_per_slot;
  t->pooled_nodes = pooled_nodes;
  StageTimer trace;
  if (scratch) scratch->ctx = ctx;
  CP2_TRY(trees_layout(t.get(), scratch ? &scratch->nodes[node_slot & 1] : nullptr));
  // how the batch is cut into turns -- staging chunks of up to 2 GiB, the ramp-down of the last groups, which of the two staging
  // buffers a turn uses: csrc/fake_turns.hpp (plain arithmetic; the CPU suite walks it over thousands of shapes)
  const char* ramp_env = std::getenv("CP2_STREAM_RAMP");                                                  // "0": A/B tooling
  const bool serial = group != 0 && stream_serial();            // A/B tooling: groups hashed on the first stream only, as in rounds 2-4
  const bool leave_room = group != 0 && !serial && ctx->hash_room;   // group builds: two workgroups per CU, the rest for the layer passes and sampling
  const FakeTurnPlan plan = fake_turn_plan(n_slots, n_cells, cell_size, ctx->stage_bytes, group, !(ramp_env && ramp_env[0] == '0'),
                                           leave_room ? FAKE_RESIDENCY_CELLS_WITH_ROOM : FAKE_RESIDENCY_CELLS);
  const size_t total_cells = plan.total_cells;
  // whole slots: the seed of the batch's first slot, the generator counts slots from there; units: the seed of slot 0 of the
  // dataset, the generator places unit first_slot + i inside slot (first_slot + i) / units_per_slot
  const uint64_t seed0 = cp2_slot_seed(dataset_seed, units_per_slot > 1 ? 0 : first_slot);
  LayerScheduler sched{t.get(), group, done};
  sched.detached = scratch != nullptr;        // (before init(): the choice of the layer stream depends on it)
  DevBuf own_stage[2];
  DevBuf* stage = scratch ? scratch->stage : own_stage;        // a pipeline's staging outlives this call (its last chunks may still be hashing)
  if (stage[0].bytes < plan.chunk * cell_size) CP2_TRY(stage[0].scratch(ctx, plan.chunk * cell_size));
  if (plan.two && stage[1].bytes < plan.chunk * cell_size) CP2_TRY(stage[1].scratch(ctx, plan.chunk * cell_size));
  trace.lap("fake slots: node + staging buffers");
  sched.take_all = plan.ramp;
  int st = sched.init();
  size_t turn = 0;
  for (size_t c0 = 0, n = 0; st == CP2_OK && c0 < total_cells; c0 += n, ++turn) {
    n = fake_turn_cells(plan, n_cells, c0);
    const int s = fake_turn_side(plan, turn, serial);           // generation + hashing of this turn on stream s, in its own staging buffer
    if (n == 0 || !stage[s].p || stage[s].bytes < n * cell_size) {   // (never: checked because a null or short buffer is a GPU fault)
      ctx->err = "fake builder: no staging buffer for a turn";
      st = CP2_ERR_INVALID;
      break;
    }
    hipError_t e = cp2k::launch_gen_fake_cells(seed0, n_cells, c0, nullptr, n, cell_size, stage[s].p, sched.hs[s], units_per_slot, first_slot);
    if (e == hipSuccess) e = cp2k::launch_hash_cells(stage[s].p, cell_size, n, t->nodes.u8() + c0 * 32, sched.hs[s], leave_room);
    if (e != hipSuccess) { ctx->err = hipGetErrorString(e); st = CP2_ERR_HIP; break; }
    st = sched.hashed_on(s);
    if (st == CP2_OK) st = sched.advance(c0 + n, c0 + n == total_cells, s);
  }
  if (scratch) scratch->tail_stream = sched.layer_stream();   // where the batch ends: the caller's copy-out follows the layer passes there
  int fin = sched.finish();
  trace.lap("fake slots: generate + hash + layers");
  if (st == CP2_OK) st = fin;
  if (st != CP2_OK) return st;
  *out = t.release();
  return CP2_OK;
}

extern "C" int cp2_slot_trees_build_fake(cp2_ctx* ctx, uint64_t dataset_seed, uint64_t first_slot, size_t n_slots,
                                         size_t cell_size, size_t block_size, size_t n_cells, cp2_slot_trees** out) try {
  if (!ctx || !out) return CP2_ERR_INVALID;
  return trees_build_fake(ctx, dataset_seed, first_slot, n_slots, cell_size, block_size, n_cells, 0, nullptr, out);
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

// Units: `n_units` consecutive pieces of `cells_per_unit` cells, unit u = cells [(u % units_per_slot) * cells_per_unit, ...) of
// slot u / units_per_slot.  The root of a unit is the node of that slot's tree (gen_input/bn254.nim:21-30) above its cells
// (cells_per_unit / cellsPerBlock >= 2 blocks, a power of two, so that the unit's layers ARE layers of the slot tree).
namespace {
bool unit_geometry_ok(uint64_t units_per_slot, size_t cell_size, size_t block_size, size_t cells_per_unit) {
  if (!is_pow2(units_per_slot) || cell_size == 0 || block_size < cell_size) return false;
  const size_t nb = cells_per_unit / (block_size / cell_size);
  return nb >= 2 && is_pow2(nb);
}
}  // namespace
extern "C" int cp2_slot_trees_build_fake_units(cp2_ctx* ctx, uint64_t dataset_seed, uint64_t units_per_slot, uint64_t first_unit,
                                               size_t n_units, size_t cell_size, size_t block_size, size_t cells_per_unit,
                                               cp2_slot_trees** out) try {
  if (!ctx || !out || units_per_slot == 0) return CP2_ERR_INVALID;
  CP2_TRY(trees_check_geometry(cell_size, block_size, cells_per_unit, n_units));
  if (units_per_slot > 1 && !unit_geometry_ok(units_per_slot, cell_size, block_size, cells_per_unit)) return CP2_ERR_INVALID;
  return trees_build_fake(ctx, dataset_seed, first_unit, n_units, cell_size, block_size, cells_per_unit, 0, nullptr, out, units_per_slot);
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" int cp2_slot_trees_build_file_units(cp2_ctx* ctx, const char* file_base, uint64_t units_per_slot, uint64_t first_unit,
                                               size_t n_units, size_t cell_size, size_t block_size, size_t cells_per_unit,
                                               cp2_slot_trees** out) try {
  if (!ctx || !out || !file_base || units_per_slot == 0) return CP2_ERR_INVALID;
  CP2_TRY(trees_check_geometry(cell_size, block_size, cells_per_unit, n_units));
  if (units_per_slot > 1 && !unit_geometry_ok(units_per_slot, cell_size, block_size, cells_per_unit)) return CP2_ERR_INVALID;
  return trees_build_files(ctx, file_base, first_unit, n_units, cell_size, block_size, cells_per_unit, 0, nullptr, out, units_per_slot);
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" int cp2_slot_trees_build_dev(cp2_ctx* ctx, const void* d_cells, size_t n_slots, size_t cell_size,
                                        size_t block_size, size_t n_cells, cp2_slot_trees** out) try {
  if (!ctx || !out || !d_cells) return CP2_ERR_INVALID;
  *out = nullptr;
  CP2_REFUSE_STUCK(ctx);
  CP2_TRY(trees_check_geometry(cell_size, block_size, n_cells, n_slots));
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  std::unique_ptr<cp2_slot_trees> t(trees_new(ctx, n_slots, cell_size, block_size, n_cells));
  if (!t) return CP2_ERR_ALLOC;
  t->src = CellSrc::Dev;
  t->d_cells = static_cast<const uint8_t*>(d_cells);
  CP2_TRY(trees_layout(t.get()));
  CP2_HIP(ctx, cp2k::launch_hash_cells(d_cells, cell_size, n_slots * n_cells, t->nodes.p, ctx->stream));
  CP2_TRY(trees_build_layers(t.get(), 0, n_slots, ctx->stream));
  // A `_dev` entry point: everything is enqueued on the context's stream, nothing is synchronised (header contract).
  // cp2_sync / cp2_slot_trees_roots / _paths synchronise and report a failed launch through cp2_last_error.
  *out = t.release();
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

// ---- streaming ingestion ----------------------------------------------------
// Overlapped stages: host threads fill a PINNED buffer (pread or memcpy), the copy engine moves it into a DEVICE buffer, one of
// the context's two hashing streams hashes it.  While chunk i is uploaded and hashed the host is already filling chunks i + 1 and
// i + 2, so disk / host memory, PCIe and the GPU work concurrently (the reference re-opens the slot file and reads one cell per
// call, slot.nim:57-68).  Fill threads, ring depth and chunk size are run-time knobs: cp2_set_ingest, or CP2_INGEST_THREADS /
// CP2_INGEST_RING / CP2_INGEST_CHUNK_MB.
//
// Chunk size is what matters (measured, tools/ingest_probe.cpp, profiles/r02_ingest_probe.txt): the hash kernel runs
// one CELL per lane, so a 64 MiB chunk of 2 KiB cells is 128 workgroups on a GPU that holds 768 of them -- the kernel
// then takes the lifetime of one workgroup (3.75 ms) whatever its size, and the pipe ran at 17 GB/s although host
// memcpy (130 GB/s on 8 threads), pinned H2D (56 GB/s) and the kernel from HBM (43 GB/s) are each far faster.  The
// default chunk is therefore a whole number of residencies of the kernel (wanted_chunk_bytes).
//
// Round 6 (profiles/r06_streamed_files_ab.txt, r06_streamed_files_trace.txt): (1) a chunk is a range of the BATCH's cells and may
// span many slot files (csrc/ingest_turns.hpp: a dataset of 8 MiB slots used to be hashed 16 workgroups at a time); (2) the pinned
// ring (3 buffers, free again once uploaded) is apart from the device ring (4); (3) fills are posted two turns deep in 4 MiB grains
// (fill_begin / fill_join); (4) a turn's upload rides on its own hashing stream, ahead of its kernel (upload_stream: a separate copy
// stream shared a hardware queue with the second hashing stream); (5) the pipe can outlive a builder call
// (BuildScratch::file_pipe): the batches of a transient build pipeline like the fake-data ones.
namespace {

size_t env_size(const char* name, size_t dflt) {
  const char* v = std::getenv(name);
  if (!v || !*v) return dflt;
  char* end = nullptr;
  unsigned long long x = std::strtoull(v, &end, 10);
  return (end && *end == 0 && x > 0) ? (size_t)x : dflt;
}

struct IngestPipe {
  static constexpr int MAX_DEPTH = 8;
  static constexpr size_t DIRECT_ALIGN = 4096;   // offset, length and address granule of O_DIRECT reads
  size_t cell_multiple = 1;                      // inside a large slot chunk sizes are multiples of this many cells (direct reads: whole 4 KiB blocks)
  cp2_ctx* ctx = nullptr;
  hipStream_t copy = nullptr;
  int pin_depth = 0, dev_depth = 0;
  PinBuf pinned[MAX_DEPTH];
  DevBuf dev[MAX_DEPTH];
  hipEvent_t copied[MAX_DEPTH] = {};     // per PINNED buffer: its upload is done, the host may fill it again
  hipEvent_t uploaded[MAX_DEPTH] = {};   // per DEVICE buffer: the chunk has landed in it (what the hash launch waits for)
  hipEvent_t hashed[MAX_DEPTH] = {};     // per DEVICE buffer: the kernel that read it is done (what the next upload into it waits for)
  size_t chunk = 0;                      // cells per full turn (what a ring buffer holds)
  size_t cap_bytes = 0;                  // bytes of every ring buffer
  size_t turn = 0;                       // turns since the pipe was set up (device buffer = turn % dev_depth)
  size_t pin_turn = 0;                   // ... of which through the pinned ring (pinned buffer = pin_turn % pin_depth)
  int threads = 1;
  bool serial = false;                   // CP2_STREAM_SERIAL=1 (A/B tooling): every chunk hashed on the first stream
  hipStream_t hash_stream[2] = {nullptr, nullptr};   // chunks alternate between the context's two streams: the next chunk's
                                                     // workgroups fill the CUs as the previous kernel's last ones retire
  int last_aux_dev = -1;                 // device buffer of the latest chunk hashed on the second stream

  ~IngestPipe() {
    if (!ctx) return;
    fill_join_all();                      // (an error path may leave with fills in flight: the workers write into the pinned ring)
    const bool trace = std::getenv("CP2_TRACE") != nullptr && (mapped_chunks + ring_chunks) > 0;
    const double t0 = now_ms();
    (void)hipSetDevice(ctx->device);
    (void)finish();
    (void)hipStreamSynchronize(ctx->stream);
    if (hash_stream[1]) (void)hipStreamSynchronize(hash_stream[1]);
    if (copy) (void)hipStreamSynchronize(copy);
    const double t1 = now_ms();
    for (auto& mp : mappings) mp.open = false;
    (void)release_mapped();
    const double t2 = now_ms();
    for (int b = 0; b < MAX_DEPTH; ++b) {
      if (copied[b]) (void)hipEventDestroy(copied[b]);
      if (uploaded[b]) (void)hipEventDestroy(uploaded[b]);
      if (hashed[b]) (void)hipEventDestroy(hashed[b]);
    }
    if (copy) (void)hipStreamDestroy(copy);
    if (trace)
      std::fprintf(stderr, "[cp2 trace] ingestion pipe torn down after %zu chunk(s) by mapping + %zu through the pinned ring: drain %.1f ms, release %.1f ms, events + copy stream %.1f ms\n",
                   mapped_chunks, ring_chunks, t1 - t0, t2 - t1, now_ms() - t2);
  }
  // what a ring buffer of this context holds, in bytes (before it is clipped to the batch)
  // Default: a whole number of residencies of the hash kernel at the occupancy it will be launched with -- 256 CUs x 3 workgroups x
  // 256 cells (one residency: 384 MiB at 2 KiB cells), and, when the launches leave room, 256 CUs x 2 workgroups x 256 cells THREE
  // times (768 MiB: also two residencies at full occupancy).  Every workgroup of a launch runs the same instruction stream for the same time, so a launch costs a whole number of
  // waves of workgroups: round 6's first trace of the streamed build from files showed 768-workgroup launches on the 512 slots a
  // launch with room has -- two waves for the price of 1.5, 11.6 ms per chunk where 8.9 would do (profiles/r06_streamed_files_trace.txt).
  // With passes of a group the size matters far less than while every ring turn carried a layer pass of its own (layer_take,
  // csrc/ingest_turns.hpp), but it still shows: 0.949 / 0.946 / 0.972 of the fake source's rate at 256 / 384 / 512 MiB, and in three
  // alternating rounds on one box 0.973-0.980 at 512, 0.997-0.998 at 768 (16 slots of 8 GiB: 1.004 and 1.013).
  static size_t wanted_chunk_bytes(const cp2_ctx* c, size_t cell_size, bool leave_room) {
    size_t chunk_bytes = c->ingest_chunk ? c->ingest_chunk : env_size("CP2_INGEST_CHUNK_MB", 0) << 20;
    if (chunk_bytes == 0) chunk_bytes = std::max<size_t>((size_t)64 << 20, std::min<size_t>((size_t)(leave_room ? 1536 : 768) * 256 * cell_size, (size_t)1 << 30));
    return chunk_bytes;
  }
  int init(cp2_ctx* c, size_t cell_size, size_t max_cells, bool leave_room = false) {
    ctx = c;
    int ring = c->ingest_ring ? c->ingest_ring : (int)env_size("CP2_INGEST_RING", 3);
    ring = std::max(2, std::min(ring, (int)MAX_DEPTH));
    const int want_dev = std::min(ring + 1, (int)MAX_DEPTH);
    threads = c->ingest_threads ? c->ingest_threads : (int)env_size("CP2_INGEST_THREADS", 8);
    threads = std::max(1, std::min(threads, 64));
    hash_stream[0] = ctx->stream;
    CP2_TRY(aux_stream(ctx, &hash_stream[1]));
    separate_copy_stream = env_size("CP2_INGEST_COPY_STREAM", 0) != 0;     // A/B tooling: uploads on a stream of their own, as until round 6
    // The rings: `ring` pinned + `ring + 1` device buffers of one turn each (2.25 + 3 GiB at the streamed builds' 768 MiB turns).  When
    // the host or the device cannot give that much (a small box, a shared device, CODEX_P2_MEM_LIMIT_MB), the turn is halved -- down to
    // 32 MiB -- rather than the build failed: smaller launches are slower, not wrong.
    size_t want_bytes = wanted_chunk_bytes(c, cell_size, leave_room);
    for (;;) {
      chunk = ingest_chunk_cells(want_bytes, cell_size, max_cells);
      cap_bytes = chunk * cell_size;
      int st = CP2_OK;
      for (int b = 0; b < ring && st == CP2_OK; ++b) st = pinned[b].alloc(ctx, cap_bytes);
      for (int b = 0; b < want_dev && st == CP2_OK; ++b) st = dev[b].scratch(ctx, cap_bytes);
      if (st == CP2_OK) break;
      for (int b = 0; b < MAX_DEPTH; ++b) { pinned[b].release(); dev[b].release(); }
      if (st != CP2_ERR_ALLOC || cap_bytes <= ((size_t)32 << 20) || chunk <= 1) return st;
      if (std::getenv("CP2_TRACE")) std::fprintf(stderr, "[cp2 trace] ingestion pipe: rings of %d + %d buffers of %zu MiB do not fit (%s): half the turn\n", ring, want_dev, cap_bytes >> 20, ctx->err.c_str());
      ctx->err.clear();
      want_bytes = cap_bytes / 2;
    }
    for (int b = 0; b < ring; ++b) {
      CP2_HIP(ctx, hipEventCreateWithFlags(&copied[b], hipEventDisableTiming));
      CP2_HIP(ctx, hipEventRecord(copied[b], ctx->stream));
      pin_depth = b + 1;
    }
    for (int b = 0; b < want_dev; ++b) {
      CP2_HIP(ctx, hipEventCreateWithFlags(&uploaded[b], hipEventDisableTiming));
      CP2_HIP(ctx, hipEventCreateWithFlags(&hashed[b], hipEventDisableTiming));
      CP2_HIP(ctx, hipEventRecord(hashed[b], ctx->stream));
      dev_depth = b + 1;
    }
    fill.reset(new FillPipeline(threads));
    return CP2_OK;
  }
  // a pipe that outlives builder calls: the next batch's chunk size (its buffers were sized by the first batch)
  bool fits(size_t cell_size) const { return cap_bytes >= cell_size; }
  void rebatch(size_t cell_size, size_t max_cells) { chunk = std::max<size_t>(1, std::min(max_cells, cap_bytes / cell_size)); }
  // the pinned buffer the host may fill next (blocks until the upload that last read it is done)
  int acquire(uint8_t** buf, size_t ahead = 0) {   // ahead = 1: the buffer of the ring turn AFTER the one about to be shipped
    const int b = (int)((pin_turn + ahead) % pin_depth);
    CP2_HIP(ctx, hipEventSynchronize(copied[b]));
    *buf = pinned[b].u8();
    return CP2_OK;
  }
  // Which stream carries a turn's upload?  The turn's OWN hashing stream (round 6): upload k, hash k, upload k + 2, hash k + 2 in order on
  // one stream, the odd turns on the other -- while one stream hashes, the other uploads, and nothing needs an event between an upload and
  // its kernel.  Until round 6 the uploads had a stream of their own; the runtime maps streams onto FOUR hardware queues, and with the
  // null stream, the context's three and that one it was the fifth: it shared a queue with the second hashing stream, and every odd
  // upload waited behind the hash launch queued there two turns earlier (first trace of the streamed build from files: uploads in pairs,
  // the first of each pair starting exactly when hash k - 2 ended; profiles/r06_streamed_files_trace.txt).  The serial A/B order (one
  // hashing stream) and CP2_INGEST_COPY_STREAM=1 keep the separate stream.
  bool separate_copy_stream = false;
  int upload_stream(int side, hipStream_t* us) {
    if (!serial && !separate_copy_stream) { *us = hash_stream[side]; return CP2_OK; }
    if (!copy) CP2_HIP(ctx, hipStreamCreateWithFlags(&copy, hipStreamNonBlocking));
    *us = copy;
    return CP2_OK;
  }
  int turn_side() const { return serial ? 0 : (int)(turn & 1); }
  // the hash launch of this turn, behind the upload(s) just enqueued on `us`; *s = the hashing stream it went to
  int hash_turn(int d, size_t m, size_t cell_size, uint8_t* leaves_out, bool leave_room, int* s, hipStream_t us) {
    const int side = turn_side();
    hipStream_t hs = hash_stream[side];
    if (us != hs) {
      CP2_HIP(ctx, hipEventRecord(uploaded[d], us));
      CP2_HIP(ctx, hipStreamWaitEvent(hs, uploaded[d], 0));
    }
    CP2_HIP(ctx, cp2k::launch_hash_cells(dev[d].p, cell_size, m, leaves_out, hs, leave_room));
    CP2_HIP(ctx, hipEventRecord(hashed[d], hs));
    if (side) last_aux_dev = d;
    if (s) *s = side;
    ++turn;
    return CP2_OK;
  }
  // ship the pinned buffer acquire() handed out: m cells -> leaf hashes at `leaves_out`
  int submit(size_t m, size_t cell_size, uint8_t* leaves_out, bool leave_room = false, int* s = nullptr) {
    const int b = (int)(pin_turn % pin_depth), d = (int)(turn % dev_depth);
    if (m == 0 || m * cell_size > cap_bytes || !pinned[b].p || !dev[d].p) {   // (never: checked because a short buffer is a GPU fault)
      ctx->err = "ingestion pipe: a turn that does not fit its ring buffer";
      return CP2_ERR_INVALID;
    }
    hipStream_t us = nullptr;
    CP2_TRY(upload_stream(turn_side(), &us));
    CP2_HIP(ctx, hipStreamWaitEvent(us, hashed[d], 0));         // the kernel that last read this device buffer (the same stream's own, with an even device ring)
    CP2_HIP(ctx, hipMemcpyAsync(dev[d].p, pinned[b].p, m * cell_size, hipMemcpyHostToDevice, us));
    CP2_HIP(ctx, hipEventRecord(copied[b], us));
    ++pin_turn;
    ++ring_chunks;
    return hash_turn(d, m, cell_size, leaves_out, leave_room, s, us);
  }

  // The fills of the turns (csrc/fill_pipeline.hpp: host threads, grains from a shared counter, two turns deep; no HIP in it -- the CPU
  // suite runs it against real files under ASan/UBSan and under TSan)
  std::unique_ptr<FillPipeline> fill;
  void fill_begin(const IngestGeom& g, const std::string& base, size_t c0, size_t m, uint8_t* buf, bool want_direct, const uint8_t* mem = nullptr) {
    fill->begin(g, base, c0, m, buf, want_direct, mem);
  }
  int fill_join() {
    std::string bad;
    if (fill && !fill->join(&bad)) {
      ctx->err = "cannot open " + bad;
      return CP2_ERR_IO;
    }
    return CP2_OK;
  }
  void fill_join_all() {
    if (fill) fill->join_all();
  }

  // ---- mapped mode (round 5): chunks of a slot file that sit in the PAGE CACHE go to the device without a CPU copy.  The file is
  // mmap'ed read-only; a chunk whose pages are resident (mincore, sampled) is REGISTERED with the runtime (hipHostRegister pins the
  // page-cache pages themselves: 0.7-2.5 ms per 384 MiB, and it does not wait for the device) and uploaded straight from the
  // mapping by the copy engine -- no host thread copies, one pass over host memory instead of three (tools/mmap_register_probe.cpp,
  // profiles/r05_mmap_register_probe.txt).  On one device the build is no faster than through the ring (the hash kernel bounds both:
  // profiles/r05_slot_files_nominal.txt), hence opt-in.  Two things the probe found shape this:
  //   * hipHostUnregister waits for the whole DEVICE (409 ms beside a 418 ms kernel): unregistering per chunk serialised the pipe to
  //     18 GB/s.  So windows stay registered -- and their mappings mapped -- until the pipe ends (it synchronises there anyway), or
  //     until `mapped_budget` bytes are registered, when everything uploaded so far is released in one go (one bubble per budget).
  //   * uploading from the UNregistered mapping (the runtime pins in place by itself) reaches 55 GB/s beside an idle device but only
  //     30 GB/s beside the hash kernel, and holds the host for the duration: slower than the ring.
  // A turn that is not (all) in the cache, reaches past the end of a file or cannot be registered goes through the ring as before; the
  // two mix freely, turn by turn: either way the turn's device buffer holds the chunk when its `uploaded` event fires.
  // What is mapped of a file is the range of ONE unit (round 6; a whole slot when slots are not cut): S devices sharing a 128 GiB slot no
  // longer map it S times over.
  bool mapped_allowed = false, mapped_broken = false;
  size_t mapped_chunks = 0, ring_chunks = 0, registered_bytes = 0, mapped_budget = (size_t)32 << 30, releases = 0;
  struct Window { void* p; size_t n; };
  struct Mapping { uint8_t* base; size_t len; uint64_t unit; size_t file_off; bool open; };   // bytes [file_off, file_off + len) of the file of dataset unit `unit`
  std::vector<Window> windows;                       // registered, uploads possibly in flight
  std::vector<Mapping> mappings;                     // every range mapped so far that still has (or may get) windows
  // all uploads from mappings are complete: unregister every window (this waits for the device) and unmap the ranges that are done
  int release_mapped() {
    const bool trace = std::getenv("CP2_TRACE") != nullptr;
    const double t0 = now_ms();
    const size_t n_windows = windows.size(), bytes = registered_bytes;
    if (!windows.empty()) {
      for (hipStream_t st : {copy, hash_stream[0], hash_stream[1]})      // whichever carried uploads from these windows
        if (st) CP2_HIP(ctx, hipStreamSynchronize(st));
      for (auto& w : windows) (void)hipHostUnregister(w.p);
      windows.clear();
      ++releases;
    }
    registered_bytes = 0;
    const double t1 = now_ms();
    // tearing down the page tables of a mapping whose every page was touched costs 1.1 ms per GiB (measured: 37 ms per 32 GiB) and needs
    // nothing of this pipe, the context or the runtime: a detached thread does it while the build goes on
    std::vector<Mapping> done;
    for (size_t i = 0; i < mappings.size();) {
      if (!mappings[i].open) { done.push_back(mappings[i]); mappings[i] = mappings.back(); mappings.pop_back(); }
      else ++i;
    }
    if (!done.empty()) {
      // ... in pieces of 64 MiB: an unmap holds the process's address-space lock, which the runtime's own allocations, frees and
      // registrations on the building thread need too; 0.07 ms at a time lets them in between
      auto unmap_in_pieces = [](const std::vector<Mapping>& v) {
        const size_t piece = (size_t)64 << 20;
        for (auto& mp : v)
          for (size_t at = 0; at < mp.len; at += piece) munmap(mp.base + at, std::min(piece, mp.len - at));
      };
      try {
        std::thread([done, unmap_in_pieces] { unmap_in_pieces(done); }).detach();
      } catch (...) {
        unmap_in_pieces(done);
      }
    }
    if (trace && n_windows)
      std::fprintf(stderr, "[cp2 trace] slot files: released %zu registered window(s), %.1f GiB: unregister %.1f ms, %zu mapping(s) handed to be unmapped %.1f ms\n", n_windows, bytes / 1073741824.0, t1 - t0,
                   done.size(), now_ms() - t1);
    return CP2_OK;
  }
  // the builder has moved past every unit before `unit`: their mappings go with the next release
  void mappings_done_before(uint64_t unit) {
    for (auto& mp : mappings)
      if (mp.unit < unit) mp.open = false;
  }
  // f(i) for i in [0, n) dealt out over the fill threads (the calling thread takes its share)
  template <typename F> void parallel_items(size_t n, F f) {
    const int nt = (int)std::min<size_t>((size_t)threads, n);
    Workers* pool = fill ? fill->workers() : nullptr;
    if (nt <= 1 || !pool) { for (size_t i = 0; i < n; ++i) f(i); return; }
    for (int t = 1; t < nt; ++t) pool->submit([=] { for (size_t i = (size_t)t; i < n; i += (size_t)nt) f(i); });
    for (size_t i = 0; i < n; i += (size_t)nt) f(i);
    pool->wait_idle();
  }
  // The turn [c0, c0 + m) straight from mappings of its files, when every piece of it is in the page cache and registered: uploads on
  // the copy stream, then the hash launch, like submit().  false (nothing enqueued): the ring takes the turn.
  //
  // One piece per file the turn touches.  Each piece needs its file mapped (the range of its unit, made on first use: a large slot's
  // mapping serves all its turns), its pages found resident -- mincore, SAMPLED: over every page of a 384 MiB chunk it costs as much as
  // the upload itself (98 304 page-cache lookups: 7 ms measured), so the first and last page and a page in every 256 KiB (at least every
  // 16th of the piece) are looked at; caches fill and evict in far larger runs than that, and a piece that passes with a hole in it is
  // still read correctly (the missing pages are faulted in while they are pinned), only more slowly -- and its window registered: whole
  // pages of its own, two registrations never share a page.  Round 6: a turn of MANY small files goes this way too, the open / mmap /
  // mincore / hipHostRegister of its pieces dealt out over the fill threads (48 files of 8 MiB per turn: registering them one after the
  // other on the building thread would cost what the copy saves).
  bool try_mapped_turn(const IngestGeom& g, const std::string& base, size_t c0, size_t m, uint8_t* leaves_out, bool leave_room, int* s, int* status) {
    *status = CP2_OK;
    if (!mapped_allowed || mapped_broken) return false;
    const size_t nbytes = m * g.cell_size, page = 4096, ub = g.unit_bytes();
    if (nbytes > cap_bytes) return false;
    size_t u0 = 0, u1 = 0;
    ingest_turn_units(g, c0, m, &u0, &u1);
    mappings_done_before(g.first_unit + u0);
    if (registered_bytes + nbytes + (u1 - u0 + 1) * page > mapped_budget && release_mapped() != CP2_OK) return false;
    struct Pc {
      IngestPiece q;
      size_t at = 0;                                   // byte position in the turn's buffer
      uint8_t* mbase = nullptr; size_t mlen = 0, moff = 0;   // the mapping (existing or made here) and where it starts in the file
      bool made = false, registered = false;
      size_t win_off = 0, win_n = 0;
    };
    std::vector<Pc> pcs;
    for (size_t p = 0; p < nbytes;) {
      Pc pc;
      pc.q = ingest_piece(g, c0, p, nbytes);
      pc.at = p;
      const uint64_t unit = g.first_unit + pc.q.unit;
      for (size_t i = mappings.size(); i-- > 0;)       // (the newest first: a large slot's mapping was made a few turns ago)
        if (mappings[i].open && mappings[i].unit == unit) { pc.mbase = mappings[i].base; pc.mlen = mappings[i].len; pc.moff = mappings[i].file_off; break; }
      pcs.push_back(pc);
      p += pc.q.len;
    }
    std::atomic<bool> refused{false};
    std::atomic<int> refused_code{0};
    const int device = ctx->device;
    parallel_items(pcs.size(), [&](size_t i) {
      Pc& pc = pcs[i];
      if (!pc.mbase) {                                 // map the range of this piece's unit, as far as the file reaches
        const uint64_t unit = g.first_unit + pc.q.unit;
        const size_t unit_off = (size_t)(unit % g.units_per_slot) * ub, lo = unit_off / page * page;
        const int fd = open(slot_file_name(base, pc.q.slot).c_str(), O_RDONLY);
        if (fd < 0) return;                            // (the ring reads the turn, and reports a file that is not there)
        struct stat sb;
        if (fstat(fd, &sb) == 0 && (size_t)sb.st_size > lo) {
          const size_t hi = std::min<size_t>((size_t)sb.st_size, unit_off + ub);
          void* mp = mmap(nullptr, hi - lo, PROT_READ, MAP_SHARED, fd, (off_t)lo);   // nothing is read by this: pages that are not in the cache stay where they are
          if (mp != MAP_FAILED) { pc.mbase = static_cast<uint8_t*>(mp); pc.mlen = hi - lo; pc.moff = lo; pc.made = true; }
        }
        close(fd);
        if (!pc.mbase) return;
      }
      if (pc.q.file_off < pc.moff) return;
      const size_t off = pc.q.file_off - pc.moff, len = pc.q.len;
      if (off + len > pc.mlen) return;                                                   // reaches past the end of the file: the ring zero-fills
      if (off % page || ((off + len) % page && off + len != pc.mlen)) return;            // windows are whole pages of their own
      const size_t stride = std::max<size_t>((size_t)256 << 10, len / 16 / page * page), last = (off + len - 1) / page * page;
      unsigned char r = 0;
      for (size_t at = off;; at += stride) {
        if (at > last) at = last;
        if (mincore(pc.mbase + at, page, &r) != 0 || !(r & 1)) return;                   // not in the page cache: the ring reads the turn (buffered or O_DIRECT)
        if (at == last) break;
      }
      const size_t n = std::min(pc.mlen, (off + len + page - 1) / page * page) - off;
      (void)hipSetDevice(device);
      const hipError_t e = hipHostRegister(pc.mbase + off, n, hipHostRegisterDefault);
      if (e != hipSuccess) { refused = true; refused_code = (int)e; return; }
      pc.registered = true;
      pc.win_off = off;
      pc.win_n = n;
    });
    bool all = true;
    for (auto& pc : pcs) {
      if (pc.made) mappings.push_back({pc.mbase, pc.mlen, g.first_unit + pc.q.unit, pc.moff, true});
      if (pc.registered) { windows.push_back({pc.mbase + pc.win_off, pc.win_n}); registered_bytes += pc.win_n; }
      else all = false;
    }
    if (refused) {
      (void)hipGetLastError();
      mapped_broken = true;                          // this stack does not register file-backed pages: the ring from here on
      if (std::getenv("CP2_TRACE")) std::fprintf(stderr, "[cp2 trace] slot files: hipHostRegister of mapped file pages refused (%s): the pinned ring from here on\n", hipGetErrorString((hipError_t)refused_code.load()));
    }
    if (!all) return false;
    const int d = (int)(turn % dev_depth);
    auto fail = [&](hipError_t e, const char* what) { ctx->err = std::string(what) + ": " + hipGetErrorString(e); *status = CP2_ERR_HIP; return true; };
    hipStream_t us = nullptr;
    if (upload_stream(turn_side(), &us) != CP2_OK) { *status = CP2_ERR_HIP; return true; }
    hipError_t e = hipStreamWaitEvent(us, hashed[d], 0);
    if (e != hipSuccess) return fail(e, "hipStreamWaitEvent");
    for (auto& pc : pcs) {
      e = hipMemcpyAsync(dev[d].u8() + pc.at, pc.mbase + pc.win_off, pc.q.len, hipMemcpyHostToDevice, us);
      if (e != hipSuccess) return fail(e, "hipMemcpyAsync from a mapped slot file");
    }
    ++mapped_chunks;
    *status = hash_turn(d, m, g.cell_size, leaves_out, leave_room, s, us);
    return true;
  }
  // everything hashed on the second stream is ordered before whatever the caller enqueues next on the context's stream
  int finish() {
    if (last_aux_dev >= 0) {
      CP2_HIP(ctx, hipStreamWaitEvent(ctx->stream, hashed[last_aux_dev], 0));
      last_aux_dev = -1;
    }
    return CP2_OK;
  }
};
}  // namespace

int cp2i::hash_host_cells_pipelined(cp2_ctx* ctx, const uint8_t* cells, size_t cell_size, size_t n, uint8_t* d_leaves) {
  IngestPipe pipe;
  CP2_TRY(pipe.init(ctx, cell_size, n));
  IngestGeom g;
  g.n_units = 1; g.n_cells = n; g.cell_size = cell_size;
  // turns like the slot-file builder's: the copy of turn k + 1 into its pinned buffer is posted behind turn k's before that is joined
  // (grains of 4 MiB from a shared counter: no fill thread waits at a turn's end), the pinned ring free again once uploaded
  const std::string none;
  size_t m = 0, m_next = 0;
  uint8_t* buf = nullptr;
  int st = CP2_OK;
  if (n) {
    m = ingest_turn_cells(g, pipe.chunk, 1, pipe.turn, 0);
    st = pipe.acquire(&buf);
    if (st == CP2_OK) pipe.fill_begin(g, none, 0, m, buf, false, cells);
  }
  for (size_t c0 = 0; st == CP2_OK && c0 < n; c0 += m, m = m_next) {
    const size_t c1 = c0 + m;
    if (c1 < n && pipe.pin_depth >= 2) {
      m_next = ingest_turn_cells(g, pipe.chunk, 1, pipe.turn + 1, c1);
      st = pipe.acquire(&buf, 1);
      if (st == CP2_OK) pipe.fill_begin(g, none, c1, m_next, buf, false, cells + c1 * cell_size);
    }
    if (st == CP2_OK) st = pipe.fill_join();
    if (st == CP2_OK) st = pipe.submit(m, cell_size, d_leaves + c0 * 32);
  }
  pipe.fill_join_all();
  if (st != CP2_OK) return st;
  return pipe.finish();   // the pipe's destructor waits for the streams
}

extern "C" int cp2_slot_trees_build_host(cp2_ctx* ctx, const uint8_t* cells, size_t n_slots, size_t cell_size,
                                         size_t block_size, size_t n_cells, cp2_slot_trees** out) try {
  if (!ctx || !out || !cells) return CP2_ERR_INVALID;
  *out = nullptr;
  CP2_REFUSE_STUCK(ctx);
  CP2_TRY(trees_check_geometry(cell_size, block_size, n_cells, n_slots));
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  std::unique_ptr<cp2_slot_trees> t(trees_new(ctx, n_slots, cell_size, block_size, n_cells));
  if (!t) return CP2_ERR_ALLOC;
  t->src = CellSrc::Host;
  t->h_cells = cells;
  CP2_TRY(trees_layout(t.get()));
  int st = hash_host_cells_pipelined(ctx, cells, cell_size, n_slots * n_cells, t->nodes.u8());
  if (st == CP2_OK) st = trees_build_layers(t.get(), 0, n_slots, ctx->stream);
  if (hipStreamSynchronize(ctx->stream) != hipSuccess && st == CP2_OK) st = CP2_ERR_HIP;
  if (st != CP2_OK) return st;
  *out = t.release();
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

// slot files "<base><k>.dat" (dataset.nim:34) streamed through the ingestion pipe; short files read as zeros.
// With GROUPS (the streamed proof-input path) the file builder does what the fake-data builder does (round 6): the layer passes of
// the slots a turn completed -- and the sampling, gathers and downloads the caller's hook hangs behind them -- go to the context's
// THIRD stream while the two hashing streams carry on with the next turns; the hash launches leave a third of every CU free for
// them (launch_hash_cells' leave_room); nothing joins the two hashing streams per slot.  Rounds 2-5 launched at full occupancy and
// made the first stream wait for the second after every slot.
int cp2i::trees_build_files(cp2_ctx* ctx, const std::string& base, uint64_t first_slot, size_t n_slots, size_t cell_size,
                            size_t block_size, size_t n_cells, size_t group, const SlotsDone& done, cp2_slot_trees** out,
                            uint64_t units_per_slot, bool pooled_nodes, BuildScratch* scratch, int node_slot) {
  *out = nullptr;
  if (units_per_slot == 0) return CP2_ERR_INVALID;
  CP2_REFUSE_STUCK(ctx);
  CP2_TRY(trees_check_geometry(cell_size, block_size, n_cells, n_slots));
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  std::unique_ptr<cp2_slot_trees> t(trees_new(ctx, n_slots, cell_size, block_size, n_cells));
  if (!t) return CP2_ERR_ALLOC;
  t->src = CellSrc::File;
  t->file_base = base;
  t->first_slot = first_slot;
  t->units_per_slot = units_per_slot;
  t->pooled_nodes = pooled_nodes;
  if (scratch) scratch->ctx = ctx;
  CP2_TRY(trees_layout(t.get(), scratch ? &scratch->nodes[node_slot & 1] : nullptr));
  IngestGeom g;
  g.n_units = n_slots; g.n_cells = n_cells; g.cell_size = cell_size; g.first_unit = first_slot; g.units_per_slot = units_per_slot;
  const size_t total_cells = g.total_cells();
  const bool want_direct = ctx->ingest_direct > 0 || (ctx->ingest_direct < 0 && env_size("CP2_INGEST_DIRECT", 0) != 0);
  int st = CP2_OK;
  {
    // the pipe: this call's own (drained and torn down when the call ends), or the pipeline's (BuildScratch: it outlives the call
    // with this batch's last chunks still in flight, and serves the next batch)
    IngestPipe own_pipe;
    IngestPipe* pipe = &own_pipe;
    if (scratch) {
      if (scratch->file_pipe && !static_cast<IngestPipe*>(scratch->file_pipe.get())->fits(cell_size)) scratch->file_pipe.reset();
      if (!scratch->file_pipe) scratch->file_pipe = std::make_shared<IngestPipe>();
      pipe = static_cast<IngestPipe*>(scratch->file_pipe.get());
    }
    const bool serial = group != 0 && stream_serial();             // A/B tooling: groups hashed on the first stream only, full occupancy
    const bool leave_room = group != 0 && !serial && ctx->hash_room;
    StageTimer init_trace;
    if (!pipe->ctx) {
      st = pipe->init(ctx, cell_size, total_cells, leave_room);
      if (init_trace.on) init_trace.lap("slot files: pipe set up (rings, events, copy stream)");
    } else {
      pipe->rebatch(cell_size, total_cells);
    }
    LayerScheduler sched{t.get(), group, done};
    sched.detached = scratch != nullptr;          // (before init(): the choice of the layer stream depends on it)
    // groups: passes of `group` slots (the caller's landing buffers hold that many), the last ones halving down to one ring turn's
    // worth -- NOT a pass per turn: a pass costs the third stream ~10 ms whatever its size (layer_take, csrc/ingest_turns.hpp)
    const char* ramp_env = std::getenv("CP2_STREAM_RAMP");                                                  // "0": A/B tooling
    if (group != 0 && !(ramp_env && ramp_env[0] == '0')) sched.ramp_min = std::max<size_t>(1, pipe->chunk / n_cells);
    pipe->serial = serial;
    if (st == CP2_OK) st = sched.init();
    pipe->cell_multiple = 1;
    if (want_direct) {
      size_t a = cell_size, h = IngestPipe::DIRECT_ALIGN;
      while (h) { size_t r = a % h; a = h; h = r; }               // gcd(cell_size, 4096)
      pipe->cell_multiple = IngestPipe::DIRECT_ALIGN / a;
    }
    // mapped mode (cp2_set_ingest_mapped / CP2_INGEST_MAPPED, default off): chunks that sit in the page cache are uploaded straight
    // from a mapping of the file, no CPU copy; not with O_DIRECT, whose point is to leave the page cache alone
    pipe->mapped_allowed = !want_direct && (ctx->ingest_mapped > 0 || (ctx->ingest_mapped < 0 && env_size("CP2_INGEST_MAPPED", 0) != 0));
    const size_t mapped0 = pipe->mapped_chunks, ring0 = pipe->ring_chunks;
    StageTimer ingest_trace;
    // One turn = cells [c0, c0 + m) of the batch.  Ring turns are double-buffered on the HOST side too: turn k + 1's fill is posted
    // behind turn k's before that is joined (the workers never idle between turns), and turn k's scheduling work -- the layer passes of
    // the slots it completed and the caller's hook behind them (sampling, gathers, downloads, the hand-out of landed passes to the
    // formatting threads: 30-odd runtime calls per turn) -- is done on this thread while they read.  With mapped ingestion allowed the
    // turns go one after the other (whether a turn can be mapped is only known by trying, and a mapped turn has no fill to overlap).
    struct Turn { size_t m = 0; uint8_t* buf = nullptr; bool filling = false; };
    auto turn_cells = [&](size_t c0, size_t ahead, size_t* m) -> int {
      *m = ingest_turn_cells(g, pipe->chunk, pipe->cell_multiple, pipe->turn + ahead, c0);
      if (*m == 0 || *m > pipe->chunk || c0 + *m > total_cells) {   // (never: csrc/ingest_turns.hpp is walked by the CPU suite; a wrong turn is a GPU fault)
        ctx->err = "slot-file builder: a turn outside its ring buffer or its batch";
        return CP2_ERR_INVALID;
      }
      return CP2_OK;
    };
    auto begin_turn = [&](size_t c0, size_t ahead, Turn* tn) -> int {      // size the turn, take a pinned buffer, post its fill
      CP2_TRY(turn_cells(c0, ahead, &tn->m));
      CP2_TRY(pipe->acquire(&tn->buf, ahead));
      pipe->fill_begin(g, base, c0, tn->m, tn->buf, want_direct);
      tn->filling = true;
      return CP2_OK;
    };
    const bool ahead_ok = !pipe->mapped_allowed && pipe->pin_depth >= 2;
    Turn cur;
    double t_post = 0, t_join = 0, t_submit = 0, t_sched = 0;   // CP2_TRACE: where the building thread's time goes
    size_t n_turns = 0;
    for (size_t c0 = 0; st == CP2_OK && c0 < total_cells; ++n_turns) {
      int side = 0;
      bool mapped = false;
      double w0 = ingest_trace.on ? now_ms() : 0;
      if (!cur.filling) {
        st = turn_cells(c0, 0, &cur.m);
        if (st != CP2_OK) break;
        if (pipe->mapped_allowed) mapped = pipe->try_mapped_turn(g, base, c0, cur.m, t->nodes.u8() + c0 * 32, leave_room, &side, &st);
        if (!mapped && st == CP2_OK) st = begin_turn(c0, 0, &cur);
      }
      const size_t c1 = c0 + cur.m;
      Turn nxt;
      if (!mapped && st == CP2_OK && c1 < total_cells && ahead_ok) st = begin_turn(c1, 1, &nxt);   // posted BEHIND this turn's fill, before it is joined
      double w1 = ingest_trace.on ? now_ms() : 0;
      double w2 = w1;
      if (!mapped && st == CP2_OK) {
        st = pipe->fill_join();
        cur.filling = false;
        w2 = ingest_trace.on ? now_ms() : 0;
        if (st == CP2_OK) st = pipe->submit(cur.m, cell_size, t->nodes.u8() + c0 * 32, leave_room, &side);
      }
      double w3 = ingest_trace.on ? now_ms() : 0;
      if (st == CP2_OK) st = sched.hashed_on(side);
      if (st == CP2_OK) st = sched.advance(c1, c1 == total_cells, side);
      if (ingest_trace.on) { const double w4 = now_ms(); t_post += w1 - w0; t_join += w2 - w1; t_submit += w3 - w2; t_sched += w4 - w3; }
      c0 = c1;
      cur = nxt;
    }
    pipe->fill_join_all();                        // (an error above may leave fills in flight)
    if (ingest_trace.on)
      std::fprintf(stderr, "[cp2 trace] slot files: %zu turn(s); the building thread spent %.1f ms posting fills (incl. waiting for a pinned buffer), %.1f ms in fills (its own grains + waiting for the workers), %.1f ms submitting, %.1f ms on layer passes + the caller's hook\n",
                   n_turns, t_post, t_join, t_submit, t_sched);
    pipe->mappings_done_before(~0ULL);            // (they stay mapped until their windows are released: the pipe's end, or the next release)
    if (scratch) scratch->tail_stream = sched.layer_stream();   // where the batch ends: the caller's copy-out follows the layer passes there
    int fin = sched.finish();
    if (st == CP2_OK) st = fin;
    if (ingest_trace.on) {
      char what[160];
      std::snprintf(what, sizeof what, "slot files: %zu chunk(s) from the page cache by mapping, %zu through the pinned ring%s", pipe->mapped_chunks - mapped0,
                    pipe->ring_chunks - ring0, pipe->mapped_broken ? " (registration refused)" : "");
      ingest_trace.lap(what);
    }
  }
  if (st != CP2_OK) return st;
  *out = t.release();
  return CP2_OK;
}

extern "C" int cp2_set_ingest_mapped(cp2_ctx* ctx, int on) try {
  if (!ctx) return CP2_ERR_INVALID;
  ctx->ingest_mapped = on < 0 ? -1 : (on ? 1 : 0);
  return CP2_OK;
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" int cp2_set_ingest_direct(cp2_ctx* ctx, int on) try {
  if (!ctx) return CP2_ERR_INVALID;
  ctx->ingest_direct = on < 0 ? -1 : (on ? 1 : 0);
  return CP2_OK;
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" int cp2_set_ingest(cp2_ctx* ctx, int fill_threads, int ring_depth, size_t chunk_bytes) try {
  if (!ctx || fill_threads < 0 || ring_depth < 0) return CP2_ERR_INVALID;
  ctx->ingest_threads = fill_threads;
  ctx->ingest_ring = ring_depth;
  ctx->ingest_chunk = chunk_bytes;
  return CP2_OK;
} catch (...) {
  return CP2_ERR_INVALID;
}

// ---- persisted slot trees ---------------------------------------------------
// File = header + data-source description + every node of the layer-major buffer (canonical 32-byte elements).
// A later run with new entropy then needs only 2 permutations per sample plus gathers instead of re-hashing every
// slot (the reference re-hashes all slots per run AND the proving slot once per sample, gen_input/bn254.nim:42,57).
// The file is written to "<path>.tmp.<pid>" and renamed into place; it carries a checksum of the nodes and, for the
// SlotFile source, size + mtime of every slot file so that a cache is never reused over changed data.
namespace {
struct TreeFileHeader {
  char magic[8];            // "CP2TREE3"
  uint64_t n_slots, cell_size, block_size, n_cells;   // a batch of units: units and the cells of one unit (trees.hpp)
  uint64_t units_per_slot;  // 1: whole slots
  uint64_t src;             // CellSrc
  uint64_t dataset_seed, first_slot;
  uint64_t file_base_len;   // bytes following the header
  uint64_t n_stamps;        // (size, mtime_ns) pairs following the base name: one per slot for CellSrc::File
  uint64_t n_nodes;
  uint64_t node_checksum;
};

// 64-bit multiply-mix over 8-byte words (not cryptographic: detects truncation and bit rot, not an adversary), fed chunk by
// chunk while the nodes stream between the device and the file.  Every chunk but the last must be a multiple of 32 bytes.
struct Checksum64 {
  uint64_t h[4] = {0x9e3779b97f4a7c15ULL, 0xc2b2ae3d27d4eb4fULL, 0x165667b19e3779f9ULL, 0x27d4eb2f165667c5ULL};
  uint64_t total = 0;
  uint64_t tail = 0;
  bool tailed = false;
  void update(const uint8_t* p, size_t n) {
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
      uint64_t w[4];
      std::memcpy(w, p + i, 32);
      for (int k = 0; k < 4; ++k) {
        h[k] = (h[k] ^ w[k]) * 0x100000001b3ULL;
        h[k] = (h[k] << 29) | (h[k] >> 35);
      }
    }
    total += n;
    if (i < n) {                 // only ever the last chunk
      tailed = true;
      tail = h[0] ^ (h[1] * 3) ^ (h[2] * 5) ^ (h[3] * 7) ^ total;
      for (; i < n; ++i) tail = (tail ^ p[i]) * 0x100000001b3ULL;
    }
  }
  uint64_t finish() const {
    uint64_t r = tailed ? tail : (h[0] ^ (h[1] * 3) ^ (h[2] * 5) ^ (h[3] * 7) ^ total);
    r ^= r >> 33; r *= 0xff51afd7ed558ccdULL; r ^= r >> 33;
    return r;
  }
};

// ring-slot flags shared between the streaming thread and its I/O worker
struct SlotFlags {
  static constexpr int K = 3;
  std::mutex mu;
  std::condition_variable cv;
  bool flag[K] = {false, false, false};
  bool failed = false;
  void set(int r, bool v) { { std::lock_guard<std::mutex> lk(mu); flag[r] = v; } cv.notify_all(); }
  void fail() { { std::lock_guard<std::mutex> lk(mu); failed = true; } cv.notify_all(); }
  // waits until flag[r] == v; false if the worker reported a failure
  bool wait(int r, bool v) {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return failed || flag[r] == v; });
    return !failed;
  }
};
constexpr size_t CACHE_CHUNK = (size_t)64 << 20;

bool pwrite_all(int fd, const uint8_t* p, size_t n, off_t off) {
  while (n) {
    ssize_t w = pwrite(fd, p, n, off);
    if (w <= 0) return false;
    p += w; n -= (size_t)w; off += w;
  }
  return true;
}
bool pread_all(int fd, uint8_t* p, size_t n, off_t off) {
  while (n) {
    ssize_t r = pread(fd, p, n, off);
    if (r <= 0) return false;
    p += r; n -= (size_t)r; off += r;
  }
  return true;
}

// (size, mtime in ns) of the slot file behind each slot (each unit: several units then stamp the same file); a missing file
// stamps as (~0, ~0)
std::vector<uint64_t> file_stamps(const std::string& base, uint64_t first_slot, size_t n_slots, uint64_t units_per_slot = 1) {
  std::vector<uint64_t> v(2 * n_slots);
  for (size_t s = 0; s < n_slots; ++s) {
    struct stat sb;
    if (stat(slot_file_name(base, (first_slot + s) / units_per_slot).c_str(), &sb) == 0) {
      v[2 * s] = (uint64_t)sb.st_size;
      v[2 * s + 1] = (uint64_t)sb.st_mtim.tv_sec * 1000000000ULL + (uint64_t)sb.st_mtim.tv_nsec;
    } else {
      v[2 * s] = v[2 * s + 1] = ~0ULL;
    }
  }
  return v;
}
}  // namespace

// The nodes stream device -> pinned ring -> file in 64 MiB chunks: the download of chunk i+1, the checksum of chunk i and the
// write of chunk i-1 overlap, and no host copy of the whole node buffer exists (8 GiB for 32 768 slots of 2^12 cells).
extern "C" int cp2_slot_trees_save(cp2_slot_trees* t, const char* path) try {
  if (!t || !path) return CP2_ERR_INVALID;
  cp2_ctx* ctx = t->ctx;
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  TreeFileHeader h{};
  std::memcpy(h.magic, "CP2TREE3", 8);
  h.units_per_slot = t->units_per_slot;
  h.n_slots = t->n_slots; h.cell_size = t->cell_size; h.block_size = t->block_size; h.n_cells = t->n_cells;
  h.src = (uint64_t)t->src; h.dataset_seed = t->dataset_seed; h.first_slot = t->first_slot;
  h.file_base_len = t->file_base.size();
  std::vector<uint64_t> stamps;
  if (t->src == CellSrc::File) stamps = file_stamps(t->file_base, t->first_slot, t->n_slots, t->units_per_slot);
  h.n_stamps = stamps.size() / 2;
  h.n_nodes = t->nodes.bytes / 32;
  const size_t total = t->nodes.bytes;
  const std::string tmp = std::string(path) + ".tmp." + std::to_string((long)getpid());
  const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_NOFOLLOW | O_CLOEXEC, 0644);
  if (fd < 0) { ctx->err = "cannot create " + tmp; return CP2_ERR_IO; }
  struct FdGuard { int fd; std::string tmp; bool keep = false; ~FdGuard() { if (fd >= 0) close(fd); if (!keep) std::remove(tmp.c_str()); } } guard{fd, tmp};
  const off_t data_off = (off_t)(sizeof h + h.file_base_len + stamps.size() * 8);
  bool ok = (h.file_base_len == 0 || pwrite_all(fd, reinterpret_cast<const uint8_t*>(t->file_base.data()), h.file_base_len, sizeof h)) &&
            (stamps.empty() || pwrite_all(fd, reinterpret_cast<const uint8_t*>(stamps.data()), stamps.size() * 8, (off_t)(sizeof h + h.file_base_len)));
  if (!ok) return CP2_ERR_IO;
  CP2_HIP(ctx, hipStreamSynchronize(ctx->stream));                  // the trees are complete
  if (ctx->aux_stream) CP2_HIP(ctx, hipStreamSynchronize(ctx->aux_stream));
  Checksum64 sum;
  {
    constexpr int K = SlotFlags::K;
    PinBuf pin[K];
    hipEvent_t ev[K] = {};
    struct EvGuard { hipEvent_t* e; ~EvGuard() { for (int i = 0; i < K; ++i) if (e[i]) (void)hipEventDestroy(e[i]); } } ev_guard{ev};
    const size_t chunk = std::min(CACHE_CHUNK, std::max<size_t>(total, 32));
    for (int r = 0; r < K; ++r) {
      CP2_TRY(pin[r].alloc(ctx, chunk));
      CP2_HIP(ctx, hipEventCreateWithFlags(&ev[r], hipEventDisableTiming));
    }
    SlotFlags busy;                                                 // flag[r]: the writer still reads pin[r]
    const size_t n_chunks = (total + chunk - 1) / chunk;
    int st = CP2_OK;
    {
      Workers writer(1);                                            // joins before the pinned blocks go back to the pool
      auto finish_chunk = [&](size_t c) -> int {                    // landed -> checksum -> hand to the writer
        const int r = (int)(c % K);
        const size_t m = std::min(chunk, total - c * chunk);
        CP2_HIP(ctx, hipEventSynchronize(ev[r]));
        sum.update(pin[r].u8(), m);
        busy.set(r, true);
        const uint8_t* src = pin[r].u8();
        const off_t off = data_off + (off_t)(c * chunk);
        writer.submit([&busy, r, fd, src, m, off] {
          if (pwrite_all(fd, src, m, off)) busy.set(r, false);
          else busy.fail();
        });
        return CP2_OK;
      };
      for (size_t c = 0; c < n_chunks && st == CP2_OK; ++c) {
        const int r = (int)(c % K);
        const size_t m = std::min(chunk, total - c * chunk);
        if (!busy.wait(r, false)) { st = CP2_ERR_IO; break; }       // chunk c - K has left pin[r]
        hipError_t e = hipMemcpyAsync(pin[r].p, t->nodes.u8() + c * chunk, m, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipEventRecord(ev[r], ctx->stream);
        if (e != hipSuccess) { ctx->err = std::string("cache download: ") + hipGetErrorString(e); st = CP2_ERR_HIP; break; }
        if (c > 0) st = finish_chunk(c - 1);                        // overlaps the download just enqueued
      }
      if (st == CP2_OK && n_chunks) st = finish_chunk(n_chunks - 1);
      writer.wait_idle();
      (void)hipStreamSynchronize(ctx->stream);
      if (st == CP2_OK && busy.failed) st = CP2_ERR_IO;
    }
    if (st != CP2_OK) return st;
  }
  h.node_checksum = sum.finish();
  if (!pwrite_all(fd, reinterpret_cast<const uint8_t*>(&h), sizeof h, 0)) return CP2_ERR_IO;
  if (close(fd) != 0) { guard.fd = -1; return CP2_ERR_IO; }
  guard.fd = -1;
  if (std::rename(tmp.c_str(), path) != 0) return CP2_ERR_IO;
  guard.keep = true;
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

// file -> pinned ring -> device in 64 MiB chunks: the read of chunk i+1, the checksum of chunk i and the upload of chunk i-1
// overlap.  The nodes reach the device before the checksum is known; a mismatch discards them.
extern "C" int cp2_slot_trees_load(cp2_ctx* ctx, const char* path, cp2_slot_trees** out) try {
  if (!ctx || !path || !out) return CP2_ERR_INVALID;
  *out = nullptr;
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return CP2_ERR_IO;
  struct Closer { int fd; ~Closer() { close(fd); } } closer{fd};
  TreeFileHeader h{};
  // every header field is bounded before anything is sized from it
  if (!pread_all(fd, reinterpret_cast<uint8_t*>(&h), sizeof h, 0) || std::memcmp(h.magic, "CP2TREE3", 8) != 0 || h.file_base_len > 4096 ||
      trees_check_geometry(h.cell_size, h.block_size, h.n_cells, h.n_slots) != CP2_OK || h.src > (uint64_t)CellSrc::File ||
      h.units_per_slot == 0 || (h.units_per_slot > 1 && !unit_geometry_ok(h.units_per_slot, h.cell_size, h.block_size, h.n_cells)) ||
      (h.n_stamps != 0 && h.n_stamps != h.n_slots) || (h.src == (uint64_t)CellSrc::File) != (h.n_stamps != 0)) {
    ctx->err = std::string("not a slot-tree cache of this version: ") + path;
    return CP2_ERR_IO;
  }
  {   // the stamps the header announces are in the file before anything is sized from their count
    struct stat sb;
    if (fstat(fd, &sb) != 0 || h.n_stamps > ((uint64_t)1 << 40) || (uint64_t)sb.st_size < sizeof h + h.file_base_len + 16 * h.n_stamps) {
      ctx->err = std::string("slot-tree cache is truncated: ") + path;
      return CP2_ERR_IO;
    }
  }
  std::string base(h.file_base_len, '\0');
  if (h.file_base_len && !pread_all(fd, reinterpret_cast<uint8_t*>(&base[0]), h.file_base_len, sizeof h)) return CP2_ERR_IO;
  std::vector<uint64_t> stamps(2 * h.n_stamps);
  if (!stamps.empty() && !pread_all(fd, reinterpret_cast<uint8_t*>(stamps.data()), stamps.size() * 8, (off_t)(sizeof h + h.file_base_len))) return CP2_ERR_IO;
  if (h.src == (uint64_t)CellSrc::File && stamps != file_stamps(base, h.first_slot, h.n_slots, h.units_per_slot)) {
    ctx->err = std::string("slot files changed since the cache was written: ") + path;
    return CP2_ERR_IO;   // size or mtime of a slot file differs: the trees no longer describe the data
  }
  const off_t data_off = (off_t)(sizeof h + h.file_base_len + stamps.size() * 8);
  if (hipSetDevice(ctx->device) != hipSuccess) return CP2_ERR_HIP;
  std::unique_ptr<cp2_slot_trees> t(trees_new(ctx, h.n_slots, h.cell_size, h.block_size, h.n_cells));
  if (!t) return CP2_ERR_ALLOC;
  t->src = (CellSrc)h.src;
  t->dataset_seed = h.dataset_seed;
  t->first_slot = h.first_slot;
  t->units_per_slot = h.units_per_slot;
  t->file_base = base;
  CP2_TRY(trees_layout(t.get()));
  if (t->nodes.bytes / 32 != h.n_nodes) return CP2_ERR_IO;
  {
    struct stat sb;
    if (fstat(fd, &sb) != 0 || (uint64_t)sb.st_size != (uint64_t)data_off + t->nodes.bytes) {
      ctx->err = std::string("slot-tree cache is truncated: ") + path;
      return CP2_ERR_IO;
    }
  }
  const size_t total = t->nodes.bytes;
  Checksum64 sum;
  {
    constexpr int K = SlotFlags::K;
    PinBuf pin[K];
    hipEvent_t ev[K] = {};
    struct EvGuard { hipEvent_t* e; ~EvGuard() { for (int i = 0; i < K; ++i) if (e[i]) (void)hipEventDestroy(e[i]); } } ev_guard{ev};
    const size_t chunk = std::min(CACHE_CHUNK, std::max<size_t>(total, 32));
    for (int r = 0; r < K; ++r) {
      CP2_TRY(pin[r].alloc(ctx, chunk));
      CP2_HIP(ctx, hipEventCreateWithFlags(&ev[r], hipEventDisableTiming));
      CP2_HIP(ctx, hipEventRecord(ev[r], ctx->stream));
    }
    SlotFlags ready;                                                // flag[r]: pin[r] holds the chunk the main thread waits for
    const size_t n_chunks = (total + chunk - 1) / chunk;
    int st = CP2_OK;
    {
      Workers reader(1);
      size_t submitted = 0;
      auto submit_read = [&](size_t c) -> int {                     // pin[c % K] is free once the upload of chunk c - K has finished
        const int r = (int)(c % K);
        CP2_HIP(ctx, hipEventSynchronize(ev[r]));
        uint8_t* dst = pin[r].u8();
        const size_t m = std::min(chunk, total - c * chunk);
        const off_t off = data_off + (off_t)(c * chunk);
        reader.submit([&ready, r, fd, dst, m, off] {
          if (pread_all(fd, dst, m, off)) ready.set(r, true);
          else ready.fail();
        });
        return CP2_OK;
      };
      for (size_t c = 0; c < n_chunks && st == CP2_OK; ++c) {
        while (st == CP2_OK && submitted < n_chunks && submitted < c + K) st = submit_read(submitted++);
        if (st != CP2_OK) break;
        const int r = (int)(c % K);
        const size_t m = std::min(chunk, total - c * chunk);
        if (!ready.wait(r, true)) { st = CP2_ERR_IO; break; }
        ready.set(r, false);
        sum.update(pin[r].u8(), m);
        hipError_t e = hipMemcpyAsync(t->nodes.u8() + c * chunk, pin[r].p, m, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipEventRecord(ev[r], ctx->stream);
        if (e != hipSuccess) { ctx->err = std::string("cache upload: ") + hipGetErrorString(e); st = CP2_ERR_HIP; }
      }
      reader.wait_idle();
      if (hipStreamSynchronize(ctx->stream) != hipSuccess && st == CP2_OK) st = CP2_ERR_HIP;
    }
    if (st != CP2_OK) return st;
  }
  if (sum.finish() != h.node_checksum) {
    ctx->err = std::string("slot-tree cache is corrupt (checksum): ") + path;
    return CP2_ERR_IO;
  }
  *out = t.release();
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

// ---- persisted compact layers / roots (what a dataset that does not keep every node keeps) ------------------------------------
// Same discipline as the tree cache above: written to "<path>.tmp.<pid>" and renamed, a checksum over the payload, size + mtime of
// every slot file for the SlotFile source.  The payload moves through one pinned 64 MiB buffer, chunk by chunk.
namespace {
struct KeptFileHeader {
  char magic[8];            // "CP2KEPT1"
  uint64_t n_slots, cell_size, block_size, n_cells, src, dataset_seed, first_slot, mode;
  uint64_t file_base_len, n_stamps, payload_bytes, checksum;
};
}  // namespace

int cp2i::kept_save(cp2_ctx* ctx, const char* path, const KeptMeta& m, const void* d_buf, size_t bytes) {
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  KeptFileHeader h{};
  std::memcpy(h.magic, "CP2KEPT1", 8);
  h.n_slots = m.n_slots; h.cell_size = m.cell_size; h.block_size = m.block_size; h.n_cells = m.n_cells;
  h.src = m.src; h.dataset_seed = m.dataset_seed; h.first_slot = m.first_slot; h.mode = m.mode;
  h.file_base_len = m.file_base.size();
  std::vector<uint64_t> stamps;
  if (m.src == (uint64_t)CellSrc::File) stamps = file_stamps(m.file_base, m.first_slot, m.n_slots);
  h.n_stamps = stamps.size() / 2;
  h.payload_bytes = bytes;
  const std::string tmp = std::string(path) + ".tmp." + std::to_string((long)getpid());
  const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_NOFOLLOW | O_CLOEXEC, 0644);
  if (fd < 0) { ctx->err = "cannot create " + tmp; return CP2_ERR_IO; }
  struct FdGuard { int fd; std::string tmp; bool keep = false; ~FdGuard() { if (fd >= 0) close(fd); if (!keep) std::remove(tmp.c_str()); } } guard{fd, tmp};
  off_t off = (off_t)sizeof h;
  if (h.file_base_len && !pwrite_all(fd, reinterpret_cast<const uint8_t*>(m.file_base.data()), h.file_base_len, off)) return CP2_ERR_IO;
  off += (off_t)h.file_base_len;
  if (!stamps.empty() && !pwrite_all(fd, reinterpret_cast<const uint8_t*>(stamps.data()), stamps.size() * 8, off)) return CP2_ERR_IO;
  off += (off_t)(stamps.size() * 8);
  PinBuf pin;
  CP2_TRY(pin.alloc(ctx, std::min(CACHE_CHUNK, std::max<size_t>(bytes, 32))));
  Checksum64 sum;
  for (size_t at = 0; at < bytes; at += CACHE_CHUNK) {
    const size_t n = std::min(CACHE_CHUNK, bytes - at);
    CP2_HIP(ctx, hipMemcpyAsync(pin.p, static_cast<const uint8_t*>(d_buf) + at, n, hipMemcpyDeviceToHost, ctx->stream));
    CP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
    sum.update(pin.u8(), n);
    if (!pwrite_all(fd, pin.u8(), n, off + (off_t)at)) return CP2_ERR_IO;
  }
  h.checksum = sum.finish();
  if (!pwrite_all(fd, reinterpret_cast<const uint8_t*>(&h), sizeof h, 0)) return CP2_ERR_IO;
  if (close(fd) != 0) { guard.fd = -1; return CP2_ERR_IO; }
  guard.fd = -1;
  if (std::rename(tmp.c_str(), path) != 0) return CP2_ERR_IO;
  guard.keep = true;
  return CP2_OK;
}

int cp2i::kept_load(cp2_ctx* ctx, const char* path, const KeptMeta& w, void* d_buf, size_t bytes) {
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return CP2_ERR_IO;
  struct Closer { int fd; ~Closer() { close(fd); } } closer{fd};
  KeptFileHeader h{};
  if (!pread_all(fd, reinterpret_cast<uint8_t*>(&h), sizeof h, 0) || std::memcmp(h.magic, "CP2KEPT1", 8) != 0 || h.file_base_len > 4096) return CP2_ERR_IO;
  if (h.n_slots != w.n_slots || h.cell_size != w.cell_size || h.block_size != w.block_size || h.n_cells != w.n_cells || h.src != w.src ||
      h.first_slot != w.first_slot || h.mode != w.mode || h.payload_bytes != bytes || h.file_base_len != w.file_base.size() ||
      (w.src == (uint64_t)CellSrc::Fake && h.dataset_seed != w.dataset_seed) || (h.n_stamps != 0 && h.n_stamps != h.n_slots) ||
      (h.src == (uint64_t)CellSrc::File) != (h.n_stamps != 0))
    return CP2_ERR_IO;
  off_t off = (off_t)sizeof h;
  std::string base(h.file_base_len, '\0');
  if (h.file_base_len && !pread_all(fd, reinterpret_cast<uint8_t*>(&base[0]), h.file_base_len, off)) return CP2_ERR_IO;
  if (base != w.file_base) return CP2_ERR_IO;
  off += (off_t)h.file_base_len;
  std::vector<uint64_t> stamps(2 * h.n_stamps);
  if (!stamps.empty() && !pread_all(fd, reinterpret_cast<uint8_t*>(stamps.data()), stamps.size() * 8, off)) return CP2_ERR_IO;
  off += (off_t)(stamps.size() * 8);
  if (h.src == (uint64_t)CellSrc::File && stamps != file_stamps(base, h.first_slot, h.n_slots)) return CP2_ERR_IO;   // the slot files changed
  struct stat sb;
  if (fstat(fd, &sb) != 0 || (uint64_t)sb.st_size != (uint64_t)off + bytes) return CP2_ERR_IO;                          // truncated / appended
  if (hipSetDevice(ctx->device) != hipSuccess) return CP2_ERR_HIP;
  PinBuf pin;
  CP2_TRY(pin.alloc(ctx, std::min(CACHE_CHUNK, std::max<size_t>(bytes, 32))));
  Checksum64 sum;
  for (size_t at = 0; at < bytes; at += CACHE_CHUNK) {
    const size_t n = std::min(CACHE_CHUNK, bytes - at);
    if (!pread_all(fd, pin.u8(), n, off + (off_t)at)) return CP2_ERR_IO;
    sum.update(pin.u8(), n);
    CP2_HIP(ctx, hipMemcpyAsync(static_cast<uint8_t*>(d_buf) + at, pin.p, n, hipMemcpyHostToDevice, ctx->stream));
    CP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return sum.finish() == h.checksum ? CP2_OK : CP2_ERR_IO;
}

// trees loaded from a cache that were built from caller memory have no cell source until one is attached
extern "C" int cp2_slot_trees_attach_cells(cp2_slot_trees* t, const uint8_t* host_cells, const void* dev_cells) try {
  if (!t || (host_cells && dev_cells)) return CP2_ERR_INVALID;
  if (host_cells) { t->src = CellSrc::Host; t->h_cells = host_cells; t->d_cells = nullptr; }
  else if (dev_cells) { t->src = CellSrc::Dev; t->d_cells = static_cast<const uint8_t*>(dev_cells); t->h_cells = nullptr; }
  else return CP2_ERR_INVALID;
  return CP2_OK;
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" void cp2_slot_trees_free(cp2_slot_trees* t) {
  if (!t) return;
  if (t->nodes.borrowed) { delete t; return; }   // a pipelined batch: its nodes belong to the pipeline's scratch, whose owner does the waiting
  if (t->ctx->stuck) { delete t; return; }        // its streams will not drain: nothing is waited for, the node buffer is dropped from the books (DevBuf::release)
  (void)hipSetDevice(t->ctx->device);
  (void)hipStreamSynchronize(t->ctx->stream);
  if (t->ctx->aux_stream) (void)hipStreamSynchronize(t->ctx->aux_stream);
  if (t->ctx->aux2_stream) (void)hipStreamSynchronize(t->ctx->aux2_stream);
  delete t;
}

extern "C" size_t cp2_slot_trees_count(const cp2_slot_trees* t) { return t ? t->n_slots : 0; }
extern "C" size_t cp2_slot_trees_depth(const cp2_slot_trees* t) {
  return t ? (t->bsizes.size() - 1) + (t->tsizes.size() - 1) : 0;
}

extern "C" const void* cp2_slot_trees_roots_dev(const cp2_slot_trees* t) {
  return t ? t->nodes.u8() + t->toff.back() * 32 : nullptr;
}

extern "C" int cp2_slot_trees_roots(cp2_slot_trees* t, uint8_t* out) try {
  if (!t || !out) return CP2_ERR_INVALID;
  cp2_ctx* ctx = t->ctx;
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  CP2_HIP(ctx, hipMemcpyAsync(out, cp2_slot_trees_roots_dev(t), t->n_slots * 32, hipMemcpyDeviceToHost, ctx->stream));
  CP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

// node-row indices of the merged path of `cell` in slot `slot` (merkle.nim:21-42 twice, then :86-100)
void cp2i::path_rows(const cp2_slot_trees* t, size_t slot, uint64_t cell, size_t max_depth, uint64_t* rows) {
  size_t b = cell / t->cpb, j = cell % t->cpb, d = 0;
  size_t m = t->cpb;
  for (size_t k = 0; k + 1 < t->bsizes.size(); ++k, ++d) {
    size_t sib = j ^ 1;
    rows[d] = (sib < m) ? t->boff[k] + (slot * t->nblocks + b) * t->bsizes[k] + sib : NO_ROW;   // zero if out of range
    j >>= 1;
    m = (m + 1) >> 1;
  }
  size_t i = b;
  m = t->nblocks;
  for (size_t k = 0; k + 1 < t->tsizes.size(); ++k, ++d) {
    size_t sib = i ^ 1;
    rows[d] = (sib < m) ? t->toff[k] + slot * t->tsizes[k] + sib : NO_ROW;
    i >>= 1;
    m = (m + 1) >> 1;
  }
  for (; d < max_depth; ++d) rows[d] = NO_ROW;                                                  // padMerkleProof
}

// merged paths of n (slot-in-batch, cell) pairs in ONE gather: what a device contributes to the proof inputs of many slots of a
// dataset cut by units (multi_gpu.cpp), and, with a constant slot, cp2_slot_trees_paths
int cp2i::trees_paths_multi(cp2_slot_trees* t, const uint64_t* slot_idx, const uint64_t* cell_idx, size_t n, size_t max_depth, uint8_t* out,
                            uint8_t* leaf_hashes) {
  if (!t || (n && (!slot_idx || !cell_idx || !out))) return CP2_ERR_INVALID;
  if (cp2_slot_trees_depth(t) > max_depth) return CP2_ERR_INVALID;     // types.nim:29 assert(pad >= 0)
  if (n == 0) return CP2_OK;
  cp2_ctx* ctx = t->ctx;
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  const size_t per = max_depth + 1;                                     // + the leaf itself
  std::vector<uint64_t> rows(n * per);
  for (size_t i = 0; i < n; ++i) {
    if (slot_idx[i] >= t->n_slots || cell_idx[i] >= t->n_cells) return CP2_ERR_INVALID;   // merkle.nim:27 assert
    path_rows(t, slot_idx[i], cell_idx[i], max_depth, &rows[i * per]);
    rows[i * per + max_depth] = slot_idx[i] * t->n_cells + cell_idx[i];
  }
  DevBuf d_rows, d_out;
  CP2_TRY(d_rows.scratch(ctx, rows.size() * 8));
  CP2_TRY(d_out.scratch(ctx, rows.size() * 32));
  CP2_HIP(ctx, hipMemcpyAsync(d_rows.p, rows.data(), rows.size() * 8, hipMemcpyHostToDevice, ctx->stream));
  CP2_HIP(ctx, cp2k::launch_gather_rows(t->nodes.p, static_cast<const uint64_t*>(d_rows.p), rows.size(), 32, d_out.p, ctx->stream));
  std::vector<uint8_t> tmp(rows.size() * 32);
  CP2_HIP(ctx, hipMemcpyAsync(tmp.data(), d_out.p, tmp.size(), hipMemcpyDeviceToHost, ctx->stream));
  CP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (size_t i = 0; i < n; ++i) {
    std::memcpy(out + i * max_depth * 32, &tmp[i * per * 32], max_depth * 32);
    if (leaf_hashes) std::memcpy(leaf_hashes + i * 32, &tmp[(i * per + max_depth) * 32], 32);
  }
  return CP2_OK;
}

extern "C" int cp2_slot_trees_paths(cp2_slot_trees* t, size_t slot, const uint64_t* cell_idx, size_t n, size_t max_depth,
                                    uint8_t* out, uint8_t* leaf_hashes) try {
  if (!t || (n && (!cell_idx || !out)) || slot >= t->n_slots) return CP2_ERR_INVALID;
  const std::vector<uint64_t> slots(n, (uint64_t)slot);
  return trees_paths_multi(t, slots.data(), cell_idx, n, max_depth, out, leaf_hashes);
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

// Slot trees, datasets and proof inputs behind the C ABI (include/codex_p2.h).
//
// Mirrors reference/nim/proof_input/src/gen_input/bn254.nim:21-79 (buildSlotTreeFull, generateProofInput),
// merkle.nim:21-42,86-100 (merkleProof, mergeMerkleProofs), types.nim:27-37 (padMerkleProof) and
// json/bn254.nim:19-78 + json/shared.nim:17-25 (exportProofInput).  All hashing runs in the HIP kernels;
// what stays on the host is index arithmetic, byte packing and text formatting.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "internal.hpp"
#include "kernels.hpp"

using namespace cp2i;

namespace {
// CP2_TRACE=1: stage timings of the batched proof-input path on stderr (the reference's only tracing is shell
// `time` around whole steps, workflow/prove.sh:30-37)
struct StageTimer {
  bool on;
  std::chrono::steady_clock::time_point t0;
  StageTimer() : on(std::getenv("CP2_TRACE") != nullptr), t0(std::chrono::steady_clock::now()) {}
  void lap(const char* what) {
    if (!on) return;
    auto t1 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[cp2 trace] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};
constexpr uint64_t NO_ROW = ~0ULL;
constexpr size_t STAGE_BYTES = (size_t)1 << 31;   // device staging buffer for generated / uploaded cells

bool is_pow2(uint64_t x) { return x && !(x & (x - 1)); }
}  // namespace

// ---------------------------------------------------------------------------------------------
// slot trees
// ---------------------------------------------------------------------------------------------
enum class CellSrc { Fake, Dev, Host, File };

struct cp2_slot_trees {
  cp2_ctx* ctx = nullptr;
  size_t n_slots = 0, cell_size = 0, block_size = 0, n_cells = 0, cpb = 0, nblocks = 0;
  std::vector<size_t> bsizes, tsizes;   // per-tree layer sizes: block tree (cpb leaves), big tree (nblocks leaves)
  std::vector<size_t> boff, toff;       // element offsets of each layer in `nodes` (layer-major)
  DevBuf nodes;
  // where sampled cells come from
  CellSrc src = CellSrc::Fake;
  uint64_t dataset_seed = 0, first_slot = 0;
  const uint8_t* d_cells = nullptr;     // not owned
  const uint8_t* h_cells = nullptr;     // not owned
  std::string file_base;
};

static int trees_layout(cp2_slot_trees* t) {
  t->bsizes = layer_sizes_of(t->cpb);
  t->tsizes = layer_sizes_of(t->nblocks);
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
  return t->nodes.alloc(t->ctx, off * 32);
}

static int trees_check_geometry(size_t cell_size, size_t block_size, size_t n_cells, size_t n_slots) {
  if (cell_size == 0 || block_size == 0 || n_cells == 0 || n_slots == 0) return CP2_ERR_INVALID;
  if (block_size % cell_size != 0) return CP2_ERR_INVALID;        // types.nim:104-107 cellsPerBlock assert
  size_t cpb = block_size / cell_size;
  if (n_cells % cpb != 0) return CP2_ERR_INVALID;                 // gen_input/bn254.nim:25 assert
  return CP2_OK;
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

// all layers above the cell hashes (which are already in nodes[0 .. n_slots*n_cells))
static int trees_build_layers(cp2_slot_trees* t) {
  cp2_ctx* ctx = t->ctx;
  uint8_t* base = t->nodes.u8();
  const size_t nb = t->n_slots * t->nblocks;
  for (size_t k = 0; k + 1 < t->bsizes.size(); ++k)   // networkBlockTree, blocks/bn254.nim:60-67
    CP2_HIP(ctx, cp2k::launch_compress_layer(base + t->boff[k] * 32, base + t->boff[k + 1] * 32, t->bsizes[k], nb, k == 0,
                                             t->bsizes[k], t->bsizes[k + 1], ctx->stream));
  for (size_t k = 0; k + 1 < t->tsizes.size(); ++k)   // bigTree, gen_input/bn254.nim:28-29
    CP2_HIP(ctx, cp2k::launch_compress_layer(base + t->toff[k] * 32, base + t->toff[k + 1] * 32, t->tsizes[k], t->n_slots,
                                             k == 0, t->tsizes[k], t->tsizes[k + 1], ctx->stream));
  return CP2_OK;
}

extern "C" int cp2_slot_trees_build_fake(cp2_ctx* ctx, uint64_t dataset_seed, uint64_t first_slot, size_t n_slots,
                                         size_t cell_size, size_t block_size, size_t n_cells, cp2_slot_trees** out) try {
  if (!ctx || !out) return CP2_ERR_INVALID;
  *out = nullptr;
  CP2_TRY(trees_check_geometry(cell_size, block_size, n_cells, n_slots));
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  cp2_slot_trees* t = trees_new(ctx, n_slots, cell_size, block_size, n_cells);
  if (!t) return CP2_ERR_ALLOC;
  t->src = CellSrc::Fake;
  t->dataset_seed = dataset_seed;
  t->first_slot = first_slot;
  int st = trees_layout(t);
  if (st != CP2_OK) { delete t; return st; }
  const size_t total_cells = n_slots * n_cells;
  const size_t chunk = std::max<size_t>(1, std::min(total_cells, STAGE_BYTES / cell_size));
  DevBuf stage;
  st = stage.alloc(ctx, chunk * cell_size);
  if (st != CP2_OK) { delete t; return st; }
  const uint64_t seed0 = cp2_slot_seed(dataset_seed, first_slot);
  for (size_t c0 = 0; c0 < total_cells; c0 += chunk) {
    size_t n = std::min(chunk, total_cells - c0);
    hipError_t e = cp2k::launch_gen_fake_cells(seed0, n_cells, c0, nullptr, n, cell_size, stage.p, ctx->stream);
    if (e == hipSuccess) e = cp2k::launch_hash_cells(stage.p, cell_size, n, t->nodes.u8() + c0 * 32, ctx->stream);
    if (e != hipSuccess) { ctx->err = hipGetErrorString(e); delete t; return CP2_ERR_HIP; }
  }
  st = trees_build_layers(t);
  if (st == CP2_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = CP2_ERR_HIP;
  if (st != CP2_OK) { delete t; return st; }
  *out = t;
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" int cp2_slot_trees_build_dev(cp2_ctx* ctx, const void* d_cells, size_t n_slots, size_t cell_size,
                                        size_t block_size, size_t n_cells, cp2_slot_trees** out) try {
  if (!ctx || !out || !d_cells) return CP2_ERR_INVALID;
  *out = nullptr;
  CP2_TRY(trees_check_geometry(cell_size, block_size, n_cells, n_slots));
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  cp2_slot_trees* t = trees_new(ctx, n_slots, cell_size, block_size, n_cells);
  if (!t) return CP2_ERR_ALLOC;
  t->src = CellSrc::Dev;
  t->d_cells = static_cast<const uint8_t*>(d_cells);
  int st = trees_layout(t);
  if (st == CP2_OK) {
    hipError_t e = cp2k::launch_hash_cells(d_cells, cell_size, n_slots * n_cells, t->nodes.p, ctx->stream);
    if (e != hipSuccess) { ctx->err = hipGetErrorString(e); st = CP2_ERR_HIP; }
  }
  if (st == CP2_OK) st = trees_build_layers(t);
  if (st != CP2_OK) { delete t; return st; }
  *out = t;   // asynchronous: the caller syncs (cp2_sync) or reads roots (which syncs)
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

// host-side fill of a pinned chunk on a few threads (one memcpy / pread stream is ~14 GB/s here, below PCIe Gen5)
namespace {
constexpr int FILL_THREADS = 4;
template <typename F> void parallel_ranges(size_t n, size_t grain, F f) {
  int nt = (int)std::min<size_t>(FILL_THREADS, std::max<size_t>(1, n / grain));
  if (nt <= 1) { f(0, n); return; }
  std::vector<std::thread> pool;
  for (int t = 1; t < nt; ++t) pool.emplace_back(f, n * t / nt, n * (t + 1) / nt);
  f(0, n / nt);
  for (auto& th : pool) th.join();
}
void parallel_memcpy(uint8_t* dst, const uint8_t* src, size_t n) {
  parallel_ranges(n, (size_t)4 << 20, [&](size_t a, size_t b) { std::memcpy(dst + a, src + a, b - a); });
}
// bytes [off, off+n) of file `fd` into dst, zero-filled past EOF (short files read as zeros, slot.nim:61-66)
void parallel_pread(int fd, uint8_t* dst, size_t off, size_t n) {
  parallel_ranges(n, (size_t)4 << 20, [&](size_t a, size_t b) {
    size_t done = a;
    while (done < b) {
      ssize_t r = pread(fd, dst + done, b - done, (off_t)(off + done));
      if (r <= 0) break;
      done += (size_t)r;
    }
    if (done < b) std::memset(dst + done, 0, b - done);
  });
}
}  // namespace

// ---- streaming ingestion ----------------------------------------------------
// Three overlapped stages over two slots of a ring: the host fills a PINNED buffer (fread or memcpy),
// a dedicated copy stream moves it to the device, the context's stream hashes it.  While chunk i is copied and
// hashed the host is already filling chunk i+1, so disk, PCIe and the GPU work concurrently
// (the reference re-opens the slot file and reads one cell per call, slot.nim:57-68).
struct IngestPipe {
  static constexpr size_t CHUNK_BYTES = (size_t)64 << 20;
  cp2_ctx* ctx = nullptr;
  hipStream_t copy = nullptr;
  void* pinned[2] = {nullptr, nullptr};
  DevBuf dev[2];
  hipEvent_t copied[2] = {nullptr, nullptr}, hashed[2] = {nullptr, nullptr};
  size_t chunk = 0, turn = 0;

  ~IngestPipe() {
    if (!ctx) return;
    (void)hipStreamSynchronize(ctx->stream);
    for (int b = 0; b < 2; ++b) {
      if (copied[b]) (void)hipEventDestroy(copied[b]);
      if (hashed[b]) (void)hipEventDestroy(hashed[b]);
      if (pinned[b]) (void)hipHostFree(pinned[b]);
    }
    if (copy) (void)hipStreamDestroy(copy);
  }
  int init(cp2_ctx* c, size_t cell_size, size_t max_cells) {
    ctx = c;
    chunk = std::max<size_t>(1, std::min(max_cells, CHUNK_BYTES / cell_size));
    CP2_HIP(ctx, hipStreamCreateWithFlags(&copy, hipStreamNonBlocking));
    for (int b = 0; b < 2; ++b) {
      CP2_HIP(ctx, hipHostMalloc(&pinned[b], chunk * cell_size, hipHostMallocDefault));
      CP2_TRY(dev[b].alloc(ctx, chunk * cell_size));
      CP2_HIP(ctx, hipEventCreateWithFlags(&copied[b], hipEventDisableTiming));
      CP2_HIP(ctx, hipEventCreateWithFlags(&hashed[b], hipEventDisableTiming));
      CP2_HIP(ctx, hipEventRecord(hashed[b], ctx->stream));
    }
    return CP2_OK;
  }
  // the pinned buffer the host may fill next (blocks until the kernel that last used this ring slot is done)
  int acquire(uint8_t** buf) {
    int b = (int)(turn & 1);
    CP2_HIP(ctx, hipEventSynchronize(hashed[b]));
    *buf = static_cast<uint8_t*>(pinned[b]);
    return CP2_OK;
  }
  // ship the filled buffer: m cells -> leaf hashes at `leaves_out`
  int submit(size_t m, size_t cell_size, uint8_t* leaves_out) {
    int b = (int)(turn & 1);
    CP2_HIP(ctx, hipMemcpyAsync(dev[b].p, pinned[b], m * cell_size, hipMemcpyHostToDevice, copy));
    CP2_HIP(ctx, hipEventRecord(copied[b], copy));
    CP2_HIP(ctx, hipStreamWaitEvent(ctx->stream, copied[b], 0));
    CP2_HIP(ctx, cp2k::launch_hash_cells(dev[b].p, cell_size, m, leaves_out, ctx->stream));
    CP2_HIP(ctx, hipEventRecord(hashed[b], ctx->stream));
    ++turn;
    return CP2_OK;
  }
};

int cp2i::hash_host_cells_pipelined(cp2_ctx* ctx, const uint8_t* cells, size_t cell_size, size_t n, uint8_t* d_leaves) {
  IngestPipe pipe;
  CP2_TRY(pipe.init(ctx, cell_size, n));
  for (size_t c0 = 0; c0 < n; c0 += pipe.chunk) {
    size_t m = std::min(pipe.chunk, n - c0);
    uint8_t* buf = nullptr;
    CP2_TRY(pipe.acquire(&buf));
    parallel_memcpy(buf, cells + c0 * cell_size, m * cell_size);
    CP2_TRY(pipe.submit(m, cell_size, d_leaves + c0 * 32));
  }
  return CP2_OK;   // the pipe's destructor waits for the stream
}

extern "C" int cp2_slot_trees_build_host(cp2_ctx* ctx, const uint8_t* cells, size_t n_slots, size_t cell_size,
                                         size_t block_size, size_t n_cells, cp2_slot_trees** out) try {
  if (!ctx || !out || !cells) return CP2_ERR_INVALID;
  *out = nullptr;
  CP2_TRY(trees_check_geometry(cell_size, block_size, n_cells, n_slots));
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  cp2_slot_trees* t = trees_new(ctx, n_slots, cell_size, block_size, n_cells);
  if (!t) return CP2_ERR_ALLOC;
  t->src = CellSrc::Host;
  t->h_cells = cells;
  int st = trees_layout(t);
  if (st == CP2_OK) st = hash_host_cells_pipelined(ctx, cells, cell_size, n_slots * n_cells, t->nodes.u8());
  if (st == CP2_OK) st = trees_build_layers(t);
  if (st == CP2_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = CP2_ERR_HIP;
  if (st != CP2_OK) { delete t; return st; }
  *out = t;
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

// slot files "<base><k>.dat" (dataset.nim:34) streamed through the ingestion pipe; short files read as zeros
static int trees_build_files(cp2_ctx* ctx, const std::string& base, uint64_t first_slot, size_t n_slots, size_t cell_size,
                             size_t block_size, size_t n_cells, cp2_slot_trees** out) {
  *out = nullptr;
  CP2_TRY(trees_check_geometry(cell_size, block_size, n_cells, n_slots));
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  cp2_slot_trees* t = trees_new(ctx, n_slots, cell_size, block_size, n_cells);
  if (!t) return CP2_ERR_ALLOC;
  t->src = CellSrc::File;
  t->file_base = base;
  t->first_slot = first_slot;
  int st = trees_layout(t);
  {
    IngestPipe pipe;
    if (st == CP2_OK) st = pipe.init(ctx, cell_size, n_cells);
    for (size_t s = 0; st == CP2_OK && s < n_slots; ++s) {
      std::string fname = base + std::to_string(first_slot + s) + ".dat";
      int fd = open(fname.c_str(), O_RDONLY);
      if (fd < 0) { ctx->err = "cannot open " + fname; st = CP2_ERR_IO; break; }
      for (size_t c0 = 0; st == CP2_OK && c0 < n_cells; c0 += pipe.chunk) {
        size_t m = std::min(pipe.chunk, n_cells - c0);
        uint8_t* buf = nullptr;
        st = pipe.acquire(&buf);
        if (st != CP2_OK) break;
        parallel_pread(fd, buf, c0 * cell_size, m * cell_size);
        st = pipe.submit(m, cell_size, t->nodes.u8() + (s * n_cells + c0) * 32);
      }
      close(fd);
    }
    if (st == CP2_OK) st = trees_build_layers(t);
    if (st == CP2_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = CP2_ERR_HIP;
  }
  if (st != CP2_OK) { delete t; return st; }
  *out = t;
  return CP2_OK;
}

// ---- persisted slot trees ---------------------------------------------------
// File = header + every node of the layer-major buffer (canonical 32-byte elements).  A later run with new
// entropy then needs only 2 permutations per sample plus gathers instead of re-hashing every slot
// (the reference re-hashes all slots per run AND the proving slot once per sample, gen_input/bn254.nim:42,57).
namespace {
struct TreeFileHeader {
  char magic[8];            // "CP2TREE1"
  uint64_t n_slots, cell_size, block_size, n_cells;
  uint64_t src;             // CellSrc
  uint64_t dataset_seed, first_slot;
  uint64_t file_base_len;   // bytes following the header
  uint64_t n_nodes;
};
}  // namespace

extern "C" int cp2_slot_trees_save(cp2_slot_trees* t, const char* path) try {
  if (!t || !path) return CP2_ERR_INVALID;
  cp2_ctx* ctx = t->ctx;
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  TreeFileHeader h{};
  std::memcpy(h.magic, "CP2TREE1", 8);
  h.n_slots = t->n_slots; h.cell_size = t->cell_size; h.block_size = t->block_size; h.n_cells = t->n_cells;
  h.src = (uint64_t)t->src; h.dataset_seed = t->dataset_seed; h.first_slot = t->first_slot;
  h.file_base_len = t->file_base.size();
  h.n_nodes = t->nodes.bytes / 32;
  std::vector<uint8_t> host(t->nodes.bytes);
  CP2_HIP(ctx, hipMemcpyAsync(host.data(), t->nodes.p, host.size(), hipMemcpyDeviceToHost, ctx->stream));
  CP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
  FILE* f = std::fopen(path, "wb");
  if (!f) return CP2_ERR_IO;
  bool ok = std::fwrite(&h, sizeof h, 1, f) == 1 &&
            (h.file_base_len == 0 || std::fwrite(t->file_base.data(), 1, h.file_base_len, f) == h.file_base_len) &&
            std::fwrite(host.data(), 1, host.size(), f) == host.size();
  ok = (std::fclose(f) == 0) && ok;
  return ok ? CP2_OK : CP2_ERR_IO;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" int cp2_slot_trees_load(cp2_ctx* ctx, const char* path, cp2_slot_trees** out) try {
  if (!ctx || !path || !out) return CP2_ERR_INVALID;
  *out = nullptr;
  FILE* f = std::fopen(path, "rb");
  if (!f) return CP2_ERR_IO;
  TreeFileHeader h{};
  if (std::fread(&h, sizeof h, 1, f) != 1 || std::memcmp(h.magic, "CP2TREE1", 8) != 0 || h.file_base_len > 4096 ||
      trees_check_geometry(h.cell_size, h.block_size, h.n_cells, h.n_slots) != CP2_OK || h.src > (uint64_t)CellSrc::File) {
    std::fclose(f);
    return CP2_ERR_IO;
  }
  std::string base(h.file_base_len, '\0');
  if (h.file_base_len && std::fread(&base[0], 1, h.file_base_len, f) != h.file_base_len) { std::fclose(f); return CP2_ERR_IO; }
  if (hipSetDevice(ctx->device) != hipSuccess) { std::fclose(f); return CP2_ERR_HIP; }
  cp2_slot_trees* t = trees_new(ctx, h.n_slots, h.cell_size, h.block_size, h.n_cells);
  if (!t) { std::fclose(f); return CP2_ERR_ALLOC; }
  t->src = (CellSrc)h.src;
  t->dataset_seed = h.dataset_seed;
  t->first_slot = h.first_slot;
  t->file_base = base;
  int st = trees_layout(t);
  if (st == CP2_OK && t->nodes.bytes / 32 != h.n_nodes) st = CP2_ERR_IO;
  std::vector<uint8_t> host;
  if (st == CP2_OK) {
    host.resize(t->nodes.bytes);
    if (std::fread(host.data(), 1, host.size(), f) != host.size()) st = CP2_ERR_IO;
  }
  std::fclose(f);
  if (st == CP2_OK) {
    hipError_t e = hipMemcpyAsync(t->nodes.p, host.data(), host.size(), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { ctx->err = hipGetErrorString(e); st = CP2_ERR_HIP; }
  }
  if (st != CP2_OK) { delete t; return st; }
  *out = t;
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

// trees loaded from a cache that were built from caller memory have no cell source until one is attached
extern "C" int cp2_slot_trees_attach_cells(cp2_slot_trees* t, const uint8_t* host_cells, const void* dev_cells) try {
  if (!t || (host_cells && dev_cells)) return CP2_ERR_INVALID;
  if (host_cells) { t->src = CellSrc::Host; t->h_cells = host_cells; t->d_cells = nullptr; }
  else if (dev_cells) { t->src = CellSrc::Dev; t->d_cells = static_cast<const uint8_t*>(dev_cells); t->h_cells = nullptr; }
  else return CP2_ERR_INVALID;
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" void cp2_slot_trees_free(cp2_slot_trees* t) {
  if (!t) return;
  (void)hipSetDevice(t->ctx->device);
  (void)hipStreamSynchronize(t->ctx->stream);
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
static void path_rows(const cp2_slot_trees* t, size_t slot, uint64_t cell, size_t max_depth, uint64_t* rows) {
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

extern "C" int cp2_slot_trees_paths(cp2_slot_trees* t, size_t slot, const uint64_t* cell_idx, size_t n, size_t max_depth,
                                    uint8_t* out, uint8_t* leaf_hashes) try {
  if (!t || (n && (!cell_idx || !out)) || slot >= t->n_slots) return CP2_ERR_INVALID;
  if (cp2_slot_trees_depth(t) > max_depth) return CP2_ERR_INVALID;     // types.nim:29 assert(pad >= 0)
  if (n == 0) return CP2_OK;
  cp2_ctx* ctx = t->ctx;
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  const size_t per = max_depth + 1;                                     // + the leaf itself
  std::vector<uint64_t> rows(n * per);
  for (size_t i = 0; i < n; ++i) {
    if (cell_idx[i] >= t->n_cells) return CP2_ERR_INVALID;              // merkle.nim:27 assert
    path_rows(t, slot, cell_idx[i], max_depth, &rows[i * per]);
    rows[i * per + max_depth] = slot * t->n_cells + cell_idx[i];
  }
  DevBuf d_rows, d_out;
  CP2_TRY(d_rows.alloc(ctx, rows.size() * 8));
  CP2_TRY(d_out.alloc(ctx, rows.size() * 32));
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
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

// the bytes of n cells given by global index g = local_slot * n_cells + cell (slotLoadCellData, slot.nim:57-68)
static int trees_cells_global(cp2_slot_trees* t, const uint64_t* g, size_t n, uint8_t* out) {
  cp2_ctx* ctx = t->ctx;
  const size_t cs = t->cell_size;
  if (n == 0) return CP2_OK;
  switch (t->src) {
    case CellSrc::Host:
      if (!t->h_cells) return CP2_ERR_INVALID;   // loaded from a cache: attach the cells first
      for (size_t i = 0; i < n; ++i) std::memcpy(out + i * cs, t->h_cells + g[i] * cs, cs);
      return CP2_OK;
    case CellSrc::File: {
      FILE* f = nullptr;
      size_t open_slot = ~(size_t)0;
      for (size_t i = 0; i < n; ++i) {
        size_t slot = g[i] / t->n_cells, cell = g[i] % t->n_cells;
        if (slot != open_slot) {
          if (f) std::fclose(f);
          std::string fname = t->file_base + std::to_string(t->first_slot + slot) + ".dat";
          f = std::fopen(fname.c_str(), "rb");
          if (!f) { ctx->err = "cannot open " + fname; return CP2_ERR_IO; }
          open_slot = slot;
        }
        std::memset(out + i * cs, 0, cs);
        if (std::fseek(f, (long)(cell * cs), SEEK_SET) == 0) (void)!std::fread(out + i * cs, 1, cs, f);
      }
      if (f) std::fclose(f);
      return CP2_OK;
    }
    case CellSrc::Fake:
    case CellSrc::Dev: {
      DevBuf d_g, d_out;
      CP2_TRY(d_g.alloc(ctx, n * 8));
      CP2_TRY(d_out.alloc(ctx, n * cs));
      CP2_HIP(ctx, hipMemcpyAsync(d_g.p, g, n * 8, hipMemcpyHostToDevice, ctx->stream));
      if (t->src == CellSrc::Dev && !t->d_cells) return CP2_ERR_INVALID;   // loaded from a cache: attach the cells first
      if (t->src == CellSrc::Fake) {
        CP2_HIP(ctx, cp2k::launch_gen_fake_cells(cp2_slot_seed(t->dataset_seed, t->first_slot), t->n_cells, 0,
                                                 static_cast<const uint64_t*>(d_g.p), n, cs, d_out.p, ctx->stream));
      } else if ((cs & 3) == 0) {
        CP2_HIP(ctx, cp2k::launch_gather_rows(t->d_cells, static_cast<const uint64_t*>(d_g.p), n, cs, d_out.p, ctx->stream));
      } else {
        for (size_t i = 0; i < n; ++i)
          CP2_HIP(ctx, hipMemcpyAsync(d_out.u8() + i * cs, t->d_cells + g[i] * cs, cs, hipMemcpyDeviceToDevice, ctx->stream));
      }
      CP2_HIP(ctx, hipMemcpyAsync(out, d_out.p, n * cs, hipMemcpyDeviceToHost, ctx->stream));
      CP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
      return CP2_OK;
    }
  }
  return CP2_ERR_INVALID;
}

// ---------------------------------------------------------------------------------------------
// dataset
// ---------------------------------------------------------------------------------------------
struct cp2_dataset {
  cp2_ctx* ctx = nullptr;
  cp2_config cfg{};
  std::string file_base;
  bool from_file = false;
  uint64_t first_slot = 0, n_local = 0;
  cp2_slot_trees* trees = nullptr;
  bool have_roots = false;
  std::vector<size_t> dsizes;                 // dataset-tree layer sizes
  std::vector<uint8_t> dlayers;               // all dataset-tree layers, bottom first (host copy)
};

extern "C" int cp2_dataset_build(cp2_ctx* ctx, const cp2_config* cfg, uint64_t first_slot, uint64_t n_local,
                                 cp2_dataset** out) try {
  if (!ctx || !cfg || !out) return CP2_ERR_INVALID;
  *out = nullptr;
  if (n_local == 0 || first_slot + n_local > cfg->n_slots) return CP2_ERR_INVALID;
  if (cfg->max_depth < 0 || cfg->max_log2_nslots < 0) return CP2_ERR_INVALID;
  cp2_dataset* ds = new (std::nothrow) cp2_dataset();
  if (!ds) return CP2_ERR_ALLOC;
  ds->ctx = ctx;
  ds->cfg = *cfg;
  ds->from_file = cfg->file_base != nullptr;
  if (ds->from_file) ds->file_base = cfg->file_base;
  ds->cfg.file_base = nullptr;
  ds->first_slot = first_slot;
  ds->n_local = n_local;
  int st;
  if (ds->from_file)
    st = trees_build_files(ctx, ds->file_base, first_slot, n_local, cfg->cell_size, cfg->block_size, cfg->n_cells, &ds->trees);
  else
    st = cp2_slot_trees_build_fake(ctx, cfg->seed, first_slot, n_local, cfg->cell_size, cfg->block_size, cfg->n_cells, &ds->trees);
  if (st != CP2_OK) { delete ds; return st; }
  *out = ds;
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

// Same as cp2_dataset_build, but the slot trees are read from `cache_path` when that file exists and matches
// the configuration, and written there after a build otherwise.
extern "C" int cp2_dataset_build_cached(cp2_ctx* ctx, const cp2_config* cfg, uint64_t first_slot, uint64_t n_local,
                                        const char* cache_path, cp2_dataset** out) try {
  if (!ctx || !cfg || !out || !cache_path) return CP2_ERR_INVALID;
  *out = nullptr;
  if (n_local == 0 || first_slot + n_local > cfg->n_slots) return CP2_ERR_INVALID;
  cp2_slot_trees* t = nullptr;
  if (cp2_slot_trees_load(ctx, cache_path, &t) == CP2_OK) {
    const bool from_file = cfg->file_base != nullptr;
    bool match = t->n_slots == n_local && t->cell_size == cfg->cell_size && t->block_size == cfg->block_size &&
                 t->n_cells == cfg->n_cells && t->first_slot == first_slot &&
                 (from_file ? (t->src == CellSrc::File && t->file_base == cfg->file_base)
                            : (t->src == CellSrc::Fake && t->dataset_seed == cfg->seed));
    if (match) {
      cp2_dataset* ds = new (std::nothrow) cp2_dataset();
      if (!ds) { cp2_slot_trees_free(t); return CP2_ERR_ALLOC; }
      ds->ctx = ctx;
      ds->cfg = *cfg;
      ds->from_file = from_file;
      if (from_file) ds->file_base = cfg->file_base;
      ds->cfg.file_base = nullptr;
      ds->first_slot = first_slot;
      ds->n_local = n_local;
      ds->trees = t;
      *out = ds;
      return CP2_OK;
    }
    cp2_slot_trees_free(t);
  }
  CP2_TRY(cp2_dataset_build(ctx, cfg, first_slot, n_local, out));
  int st = cp2_slot_trees_save((*out)->trees, cache_path);
  if (st != CP2_OK) { cp2_dataset_free(*out); *out = nullptr; }
  return st;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" void cp2_dataset_free(cp2_dataset* ds) {
  if (!ds) return;
  cp2_slot_trees_free(ds->trees);
  delete ds;
}

extern "C" int cp2_dataset_local_roots(cp2_dataset* ds, uint8_t* out) try {
  if (!ds || !out) return CP2_ERR_INVALID;
  return cp2_slot_trees_roots(ds->trees, out);
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" int cp2_dataset_set_roots(cp2_dataset* ds, const uint8_t* all_roots) try {
  if (!ds) return CP2_ERR_INVALID;
  cp2_ctx* ctx = ds->ctx;
  const size_t n = ds->cfg.n_slots;
  std::vector<uint8_t> roots(n * 32);
  if (all_roots) {
    std::memcpy(roots.data(), all_roots, n * 32);
  } else {
    if (ds->first_slot != 0 || ds->n_local != n) return CP2_ERR_INVALID;   // roots of other ranks' slots are missing
    CP2_TRY(cp2_slot_trees_roots(ds->trees, roots.data()));
  }
  // dataset tree over the slot roots (gen_input/bn254.nim:49-50); odd layers use keys 2/3
  ds->dsizes = layer_sizes_of(n);
  size_t total = cp2_merkle_total(n);
  ds->dlayers.assign(total * 32, 0);
  CP2_TRY(cp2_merkle_tree(ctx, roots.data(), n, ds->dlayers.data(), nullptr, nullptr));
  ds->have_roots = true;
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

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
// proof input
// ---------------------------------------------------------------------------------------------
// cell bytes and Merkle paths of a whole batch live in one uninitialised allocation that every proof input of
// the batch shares (no zero-fill, one device-to-host copy, no per-slot copies)
struct BatchStore {
  std::unique_ptr<uint8_t[]> cells, paths;
};

struct cp2_proof_input {
  cp2_config cfg{};
  uint64_t slot_idx = 0;
  uint8_t entropy[32], dataset_root[32], slot_root[32];
  std::vector<uint64_t> indices;
  std::vector<uint8_t> slot_proof;
  std::shared_ptr<BatchStore> store;
  const uint8_t* cell_data = nullptr;   // nSamples x cellSize, inside store->cells
  const uint8_t* paths = nullptr;       // nSamples x maxDepth x 32, inside store->paths
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

// generateProofInput (gen_input/bn254.nim:35-79) for `n` slots of the dataset at once: one sampling launch,
// one path gather, one cell fetch for all of them.
extern "C" int cp2_proof_inputs_generate_batch(cp2_dataset* ds, const uint64_t* slot_idx, size_t n, const uint8_t entropy[32],
                                               cp2_proof_input** out) try {
  if (!ds || !entropy || (n && (!slot_idx || !out))) return CP2_ERR_INVALID;
  for (size_t i = 0; i < n; ++i) out[i] = nullptr;
  if (n == 0) return CP2_OK;
  const cp2_config& cfg = ds->cfg;
  cp2_ctx* ctx = ds->ctx;
  cp2_slot_trees* t = ds->trees;
  for (size_t i = 0; i < n; ++i)
    if (slot_idx[i] < ds->first_slot || slot_idx[i] >= ds->first_slot + ds->n_local) return CP2_ERR_INVALID;
  if (!is_pow2(cfg.n_cells)) return CP2_ERR_INVALID;                    // sample/bn254.nim:19-20
  if (!ds->have_roots) CP2_TRY(cp2_dataset_set_roots(ds, nullptr));
  if (ds->dsizes.size() - 1 > (size_t)cfg.max_log2_nslots) return CP2_ERR_INVALID;   // padMerkleProof assert
  if (cp2_slot_trees_depth(t) > (size_t)cfg.max_depth) return CP2_ERR_INVALID;        // padMerkleProof assert
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  const size_t ns = cfg.n_samples, md = (size_t)cfg.max_depth, cs = cfg.cell_size, total = n * ns;

  StageTimer trace;
  // ---- sampling: cellIndices for every (slot, counter), sample/bn254.nim:16-27
  std::vector<uint64_t> indices(total);
  if (total) {
    std::vector<uint8_t> felts(total * 96, 0), dig(total * 32);
    for (size_t i = 0; i < n; ++i) {
      const uint8_t* root = &ds->dlayers[slot_idx[i] * 32];
      for (size_t c = 0; c < ns; ++c) {
        uint8_t* f = &felts[(i * ns + c) * 96];
        std::memcpy(f, entropy, 32);
        std::memcpy(f + 32, root, 32);
        uint64_t counter = c + 1;
        std::memcpy(f + 64, &counter, 8);
      }
    }
    CP2_TRY(cp2_sponge2_felts_batch(ctx, felts.data(), 3, total, dig.data()));
    for (size_t k = 0; k < total; ++k) {
      uint64_t lo;
      std::memcpy(&lo, &dig[32 * k], 8);
      indices[k] = lo & (cfg.n_cells - 1);
    }
  }
  trace.lap("sampling");
  // ---- paths (one gather) and cells (one fetch)
  auto store = std::make_shared<BatchStore>();
  store->paths.reset(new (std::nothrow) uint8_t[std::max<size_t>(total * md * 32, 1)]);
  store->cells.reset(new (std::nothrow) uint8_t[std::max<size_t>(total * cs, 1)]);
  if (!store->paths || !store->cells) return CP2_ERR_ALLOC;
  uint8_t* paths = store->paths.get();
  uint8_t* cells = store->cells.get();
  trace.lap("host buffers");
  if (total) {
    std::vector<uint64_t> rows(total * md);
    std::vector<uint64_t> gcell(total);
    for (size_t i = 0; i < n; ++i) {
      size_t local = slot_idx[i] - ds->first_slot;
      for (size_t c = 0; c < ns; ++c) {
        path_rows(t, local, indices[i * ns + c], md, &rows[(i * ns + c) * md]);
        gcell[i * ns + c] = local * t->n_cells + indices[i * ns + c];
      }
    }
    DevBuf d_rows, d_out;
    CP2_TRY(d_rows.alloc(ctx, rows.size() * 8));
    CP2_TRY(d_out.alloc(ctx, rows.size() * 32));
    CP2_HIP(ctx, hipMemcpyAsync(d_rows.p, rows.data(), rows.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    CP2_HIP(ctx, cp2k::launch_gather_rows(t->nodes.p, static_cast<const uint64_t*>(d_rows.p), rows.size(), 32, d_out.p, ctx->stream));
    CP2_HIP(ctx, hipMemcpyAsync(paths, d_out.p, total * md * 32, hipMemcpyDeviceToHost, ctx->stream));
    CP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
    trace.lap("path rows + gather + D2H");
    CP2_TRY(trees_cells_global(t, gcell.data(), total, cells));
    trace.lap("cells fetch + D2H");
  }
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
    p->indices.assign(indices.begin() + i * ns, indices.begin() + (i + 1) * ns);
    p->store = store;
    p->cell_data = cells + i * ns * cs;
    p->paths = paths + i * ns * md * 32;
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
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}
extern "C" size_t cp2_proof_input_nsamples(const cp2_proof_input* p) { return p ? p->indices.size() : 0; }
extern "C" const uint64_t* cp2_proof_input_cell_indices(const cp2_proof_input* p) { return p ? p->indices.data() : nullptr; }
extern "C" const uint8_t* cp2_proof_input_cell_data(const cp2_proof_input* p) { return p ? p->cell_data : nullptr; }
extern "C" const uint8_t* cp2_proof_input_merkle_paths(const cp2_proof_input* p) { return p ? p->paths : nullptr; }
extern "C" const uint8_t* cp2_proof_input_slot_proof(const cp2_proof_input* p) { return p ? p->slot_proof.data() : nullptr; }

// ---- JSON (json/bn254.nim:57-74, json/shared.nim:17-25, types/bn254.nim:29-43) -----------------
namespace {

// canonical decimal of a 256-bit little-endian integer: no leading zeros, "0" for zero (toDecimalF)
void append_quoted_decimal(std::string& s, const uint8_t* le32) {
  uint64_t w[4];
  std::memcpy(w, le32, 32);
  char buf[80];
  int pos = 80;
  const uint64_t CH = 10000000000000000000ULL;   // 10^19
  bool nonzero = (w[0] | w[1] | w[2] | w[3]) != 0;
  if (!nonzero) buf[--pos] = '0';
  while (nonzero) {
    unsigned __int128 rem = 0;
    for (int i = 3; i >= 0; --i) {
      unsigned __int128 cur = (rem << 64) | w[i];
      w[i] = (uint64_t)(cur / CH);
      rem = cur % CH;
    }
    nonzero = (w[0] | w[1] | w[2] | w[3]) != 0;
    uint64_t r = (uint64_t)rem;
    if (nonzero) {
      for (int d = 0; d < 19; ++d) { buf[--pos] = (char)('0' + r % 10); r /= 10; }
    } else {
      while (r) { buf[--pos] = (char)('0' + r % 10); r /= 10; }
    }
  }
  s.push_back('"');
  s.append(buf + pos, 80 - pos);
  s.push_back('"');
}

// writeList specialised to field elements: first "<prefix>[ x", then "<indent>, x", close "<indent>]"
void write_felt_list(std::string& s, const std::string& prefix, const uint8_t* felts, size_t n) {
  std::string indent(prefix.size(), ' ');
  for (size_t i = 0; i < n; ++i) {
    s += (i == 0) ? prefix + "[ " : indent + ", ";
    append_quoted_decimal(s, felts + 32 * i);
    s.push_back('\n');
  }
  s += indent + "]\n";
}

}  // namespace

static void proof_input_text(const cp2_proof_input* p, std::string& s) {
  const cp2_config& cfg = p->cfg;
  s.clear();
  s.reserve((p->indices.size() * (cp2_felts_per_bytes(cfg.cell_size) + (size_t)cfg.max_depth) + 64) * 90);
  s += "{\n";
  s += "  \"dataSetRoot\":      "; append_quoted_decimal(s, p->dataset_root); s += "\n";
  s += ", \"entropy\":          "; append_quoted_decimal(s, p->entropy); s += "\n";
  s += ", \"nCellsPerSlot\":    " + std::to_string(cfg.n_cells) + "\n";
  s += ", \"nSlotsPerDataSet\": " + std::to_string(cfg.n_slots) + "\n";
  s += ", \"slotIndex\":        " + std::to_string(p->slot_idx) + "\n";
  s += ", \"slotRoot\":         "; append_quoted_decimal(s, p->slot_root); s += "\n";
  s += ", \"slotProof\":\n";
  write_felt_list(s, "    ", p->slot_proof.data(), (size_t)cfg.max_log2_nslots);
  const size_t ns = p->indices.size();
  const std::string outer = "    ", outer_indent(outer.size(), ' ');
  s += ", \"cellData\":\n";
  {
    size_t nf = cp2_felts_per_bytes(cfg.cell_size);
    std::vector<uint8_t> felts(nf * 32);
    for (size_t i = 0; i < ns; ++i) {
      cp2_bytes_to_felts(p->cell_data + i * cfg.cell_size, cfg.cell_size, felts.data());   // json/bn254.nim:25
      write_felt_list(s, (i == 0) ? outer + "[ " : outer_indent + ", ", felts.data(), nf);
    }
    s += outer_indent + "]\n";
  }
  s += ", \"merklePaths\":\n";
  for (size_t i = 0; i < ns; ++i)
    write_felt_list(s, (i == 0) ? outer + "[ " : outer_indent + ", ", p->paths + i * (size_t)cfg.max_depth * 32, (size_t)cfg.max_depth);
  s += outer_indent + "]\n";
  s += "}\n";
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

static int write_text_file(const std::string& s, const char* path) {
  FILE* f = std::fopen(path, "wb");
  if (!f) return CP2_ERR_IO;
  size_t w = std::fwrite(s.data(), 1, s.size(), f);
  int rc = std::fclose(f);
  return (w == s.size() && rc == 0) ? CP2_OK : CP2_ERR_IO;
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
    std::string s;
    for (size_t i = t; i < n; i += threads) {
      proof_input_text(ps[i], s);
      bytes[t] += s.size();
      if (paths && paths[i]) {
        int st = write_text_file(s, paths[i]);
        if (st != CP2_OK) status[t] = st;
      }
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < threads; ++t) pool.emplace_back(work, t);
  work(0);
  for (auto& th : pool) th.join();
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

// Proof inputs for many slots, generated and serialised as a two-stage pipeline: while the host threads turn
// batch k into JSON text (and write it when dir != NULL: "<dir>/input_<slot>.json"), the GPU already samples and
// gathers batch k+1.  Config 4's metric (witnesses/s) is this call after cp2_dataset_build.
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
  auto release = [](std::vector<cp2_proof_input*>& v) {
    for (auto* p : v) delete p;
    v.clear();
  };
  if (n) status = generate(0, cur);
  for (size_t b0 = 0; status == CP2_OK && b0 < n; b0 += batch) {
    const size_t b1 = b0 + batch;
    int gen_status = CP2_OK;
    std::thread producer;                                   // GPU stage of the NEXT batch
    if (b1 < n) producer = std::thread([&] { gen_status = generate(b1, next); });
    std::vector<std::string> names;                          // host stage of THIS batch
    std::vector<const char*> paths;
    if (dir) {
      for (size_t i = 0; i < cur.size(); ++i) names.push_back(std::string(dir) + "/input_" + std::to_string(slot_idx[b0 + i]) + ".json");
      for (auto& s2 : names) paths.push_back(s2.c_str());
    }
    uint64_t got = 0;
    int st = cp2_proof_inputs_write_json_batch(cur.data(), cur.size(), dir ? paths.data() : nullptr, threads, &got);
    bytes += got;
    if (producer.joinable()) producer.join();
    release(cur);
    if (st != CP2_OK) status = st;
    else if (gen_status != CP2_OK) status = gen_status;
    cur.swap(next);
  }
  release(cur);
  release(next);
  if (total_bytes) *total_bytes = bytes;
  return status;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" void cp2_free_buffer(void* p) { std::free(p); }

extern "C" int cp2_proof_input_write_json(const cp2_proof_input* p, const char* path) try {
  if (!p || !path) return CP2_ERR_INVALID;
  std::string s;
  proof_input_text(p, s);
  return write_text_file(s, path);
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

// Several GPUs of one node behind ONE handle of the C ABI (include/codex_p2.h, section "e"): SURVEY.md 8(e).
//
// Slots are independent until the dataset tree (reference/nim/proof_input/src/gen_input/bn254.nim:41-49), so a dataset is cut
// into contiguous slot ranges, one per device; every device builds its slot trees on its own host thread and context with no
// communication; then ONE exchange -- an all-gather of the 32-byte slot roots, device to device (RCCL over xGMI, in-place
// ncclAllGather on each context's stream) -- and every device builds the identical dataset tree (gen_input/bn254.nim:49-51)
// from the gathered device buffer and serves the proof inputs of its own slots (gen_input/bn254.nim:53-74).  Nothing else
// crosses between devices.  One process: a Nim or C caller gets every GPU of the node through the same calls it makes for one.
//
// librccl is opened at run time (dlopen), and only when at least two DISTINCT devices hold a shard: a one-GPU run of the cli
// twin never pays for loading it, and a box without RCCL still works -- the gather then goes through host memory (one
// download per shard, one upload per device; 1 MiB at 32 768 slots) and cp2_multi_gather_mode says so.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and prototypes only: the entry points are resolved with dlsym

#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "trees.hpp"

using namespace cp2i;

namespace {

// The RCCL entry points this file needs, resolved once per process.
struct Rccl {
  void* lib = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string why;   // why it could not be loaded

  static Rccl& get() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] { r.load(); });
    return r;
  }
  bool ok() const { return lib != nullptr; }

 private:
  void load() {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) { why = std::string("librccl not found: ") + (dlerror() ? dlerror() : "dlopen failed"); return; }
    bool all = true;
    auto sym = [&](const char* name) { void* p = dlsym(lib, name); if (!p) all = false; return p; };
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
    AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
    if (!all) { why = "librccl lacks an expected entry point"; dlclose(lib); lib = nullptr; }
  }
};

// one hash-kernel residency: 768 workgroups x 256 cells.  A device with less than this finishes no sooner than one with
// exactly this (a launch lasts at least the lifetime of one wave), so smaller datasets use fewer devices.
constexpr uint64_t RESIDENCY_CELLS = (uint64_t)768 * 256;

}  // namespace

extern "C" void cp2_shard_range(uint64_t n_items, int rank, int world, uint64_t* first, uint64_t* count) {
  if (world < 1 || rank < 0 || rank >= world) { if (first) *first = 0; if (count) *count = 0; return; }
  const uint64_t base = n_items / (uint64_t)world, rem = n_items % (uint64_t)world;
  if (count) *count = base + ((uint64_t)rank < rem ? 1 : 0);
  if (first) *first = (uint64_t)rank * base + std::min<uint64_t>((uint64_t)rank, rem);
}

struct cp2_multi {
  std::vector<int> devices;
  std::vector<cp2_ctx*> ctxs;                 // made on first use
  std::vector<ncclComm_t> comms;              // communicators over devices[0 .. comm_world), made on first use
  int comm_world = 0;
  int gather = CP2_GATHER_AUTO;
  uint64_t min_cells = 0;                     // 0: RESIDENCY_CELLS
  std::string err, gather_note = "none yet";
  std::mutex mu;

  cp2_ctx* ctx_of(int i, int* status) {
    std::lock_guard<std::mutex> lk(mu);
    if (!ctxs[i]) {
      int st = cp2_init(devices[i], &ctxs[i]);
      if (status) *status = st;
      if (st != CP2_OK) return nullptr;
    } else if (status) {
      *status = CP2_OK;
    }
    return ctxs[i];
  }
  void drop_comms() {
    if (comm_world) {
      Rccl& r = Rccl::get();
      for (ncclComm_t c : comms) if (c) (void)r.CommDestroy(c);
    }
    comms.clear();
    comm_world = 0;
  }
};

// device indices from the environment: CODEX_P2_GPUS = "<count>" (the first <count> visible devices) or a comma-separated
// list of indices ("0,2,3"; "0,0" = two contexts on device 0; "2," = device 2 only)
static bool devices_from_env(int visible, std::vector<int>& out) {
  const char* e = std::getenv("CODEX_P2_GPUS");
  if (!e || !*e) return false;
  const std::string s(e);
  if (s.find(',') == std::string::npos) {
    const long n = std::strtol(e, nullptr, 10);
    if (n < 1) return false;
    for (int d = 0; d < std::min<long>(n, visible); ++d) out.push_back(d);
    return !out.empty();
  }
  size_t at = 0;
  while (at < s.size()) {
    size_t c = s.find(',', at);
    if (c == std::string::npos) c = s.size();
    if (c > at) out.push_back((int)std::strtol(s.substr(at, c - at).c_str(), nullptr, 10));
    at = c + 1;
  }
  return !out.empty();
}

extern "C" int cp2_multi_init(const int* devices, int n_dev, cp2_multi** out) try {
  if (!out || n_dev < 0 || (n_dev > 0 && !devices)) return CP2_ERR_INVALID;
  *out = nullptr;
  StageTimer trace;
  int visible = 0;
  if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) { (void)hipGetLastError(); return CP2_ERR_NO_DEVICE; }
  trace.lap("HIP runtime init (device count)");
  std::unique_ptr<cp2_multi> m(new cp2_multi());
  if (n_dev > 0) {
    m->devices.assign(devices, devices + n_dev);
  } else if (!devices_from_env(visible, m->devices)) {
    for (int d = 0; d < visible; ++d) {          // every visible gfx950 device
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, d) == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) m->devices.push_back(d);
    }
  }
  if (m->devices.empty()) return CP2_ERR_NO_DEVICE;
  for (int d : m->devices)
    if (d < 0 || d >= visible) return CP2_ERR_NO_DEVICE;
  m->ctxs.assign(m->devices.size(), nullptr);
  if (const char* e = std::getenv("CODEX_P2_MIN_CELLS")) m->min_cells = std::strtoull(e, nullptr, 10);   // see cp2_multi_set_policy
  *out = m.release();
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" void cp2_multi_free(cp2_multi* m) {
  if (!m) return;
  m->drop_comms();
  for (cp2_ctx* c : m->ctxs) cp2_free(c);
  delete m;
}

extern "C" int cp2_multi_count(const cp2_multi* m) { return m ? (int)m->devices.size() : 0; }
extern "C" int cp2_multi_device(const cp2_multi* m, int i) { return (m && i >= 0 && i < (int)m->devices.size()) ? m->devices[i] : -1; }
extern "C" cp2_ctx* cp2_multi_ctx(cp2_multi* m, int i) {
  if (!m || i < 0 || i >= (int)m->devices.size()) return nullptr;
  return m->ctx_of(i, nullptr);
}
extern "C" const char* cp2_multi_last_error(const cp2_multi* m) { return m ? m->err.c_str() : "no handle"; }
extern "C" const char* cp2_multi_gather_mode(const cp2_multi* m) { return m ? m->gather_note.c_str() : "no handle"; }

extern "C" int cp2_multi_set_policy(cp2_multi* m, int gather, uint64_t min_cells_per_device) {
  if (!m || gather < CP2_GATHER_AUTO || gather > CP2_GATHER_HOST) return CP2_ERR_INVALID;
  m->gather = gather;
  m->min_cells = min_cells_per_device;
  return CP2_OK;
}

// ---------------------------------------------------------------------------------------------
// sharded dataset
// ---------------------------------------------------------------------------------------------
struct cp2_multi_dataset {
  cp2_multi* m = nullptr;
  cp2_config cfg{};
  std::string file_base;
  struct Shard { int dev = 0; uint64_t first = 0, count = 0; cp2_dataset* ds = nullptr; };
  std::vector<Shard> shards;
  ~cp2_multi_dataset() { for (auto& s : shards) cp2_dataset_free(s.ds); }
  Shard* owner(uint64_t slot) {
    for (auto& s : shards)
      if (slot >= s.first && slot < s.first + s.count) return &s;
    return nullptr;
  }
};

namespace {

// f(i) for every shard on its own host thread (the calling thread takes shard 0); returns the first non-OK status
template <typename F> int for_each_shard(size_t n, F f) {
  std::vector<int> st(n, CP2_OK);
  std::vector<std::thread> th;
  for (size_t i = 1; i < n; ++i) th.emplace_back([&, i] { try { st[i] = f(i); } catch (...) { st[i] = CP2_ERR_ALLOC; } });
  try { st[0] = f(0); } catch (...) { st[0] = CP2_ERR_ALLOC; }
  for (auto& t : th) t.join();
  for (int s : st)
    if (s != CP2_OK) return s;
  return CP2_OK;
}

struct DeviceRestore {   // the caller's current device is left as it was found
  int dev = -1;
  DeviceRestore() { if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = -1; } }
  ~DeviceRestore() { if (dev >= 0) (void)hipSetDevice(dev); }
};

// The exchange step: every shard's slot roots to every device, then the dataset tree everywhere.
int gather_roots_and_build_trees(cp2_multi_dataset* mds) {
  cp2_multi* m = mds->m;
  const size_t world = mds->shards.size();
  const uint64_t n = mds->cfg.n_slots;
  if (world == 1 && m->gather != CP2_GATHER_RCCL) {   // (RCCL asked for by name: a communicator of one rank, as a self-test of the path)
    m->gather_note = "none (one shard: nothing to exchange)";
    return cp2_dataset_set_roots(mds->shards[0].ds, nullptr);
  }
  DeviceRestore restore;
  // RCCL needs one rank per DISTINCT device
  bool distinct = true;
  for (size_t i = 0; i < world; ++i)
    for (size_t j = 0; j < i; ++j)
      if (m->devices[mds->shards[i].dev] == m->devices[mds->shards[j].dev]) distinct = false;
  std::string why;
  bool use_rccl = m->gather != CP2_GATHER_HOST;
  if (use_rccl && !distinct) { use_rccl = false; why = "a device holds more than one shard"; }
  if (use_rccl && !Rccl::get().ok()) { use_rccl = false; why = Rccl::get().why; }
  if (use_rccl) {
    Rccl& r = Rccl::get();
    if (m->comm_world != (int)world) {          // shards are always devices[0 .. world)
      m->drop_comms();
      m->comms.assign(world, nullptr);
      ncclResult_t e = r.CommInitAll(m->comms.data(), (int)world, m->devices.data());
      if (e != ncclSuccess) {
        why = std::string("ncclCommInitAll: ") + r.GetErrorString(e);
        m->comms.clear();
        use_rccl = false;
      } else {
        m->comm_world = (int)world;
      }
    }
  }
  if (!use_rccl && m->gather == CP2_GATHER_RCCL) {
    m->err = "RCCL gather requested but unavailable: " + why;
    return CP2_ERR_INVALID;
  }
  if (use_rccl) {
    Rccl& r = Rccl::get();
    const uint64_t max_rows = (n + world - 1) / world;
    const bool even = n % world == 0;
    std::vector<DevBuf> gath(world), all(world);
    for (size_t i = 0; i < world; ++i) {        // this shard's roots into its own row block of its gather buffer
      cp2_ctx* ctx = cp2_dataset_ctx(mds->shards[i].ds);
      CP2_HIP(ctx, hipSetDevice(ctx->device));
      CP2_TRY(gath[i].scratch(ctx, world * max_rows * 32));
      CP2_TRY(cp2_dataset_copy_local_roots_dev(mds->shards[i].ds, gath[i].u8() + i * max_rows * 32));
    }
    ncclResult_t e = r.GroupStart();
    for (size_t i = 0; i < world && e == ncclSuccess; ++i) {
      cp2_ctx* ctx = cp2_dataset_ctx(mds->shards[i].ds);
      e = r.AllGather(gath[i].u8() + i * max_rows * 32, gath[i].p, max_rows * 32, ncclUint8, m->comms[i], ctx->stream);   // in place
    }
    ncclResult_t e2 = r.GroupEnd();
    if (e == ncclSuccess) e = e2;
    if (e != ncclSuccess) {
      m->err = std::string("ncclAllGather: ") + r.GetErrorString(e);
      return CP2_ERR_HIP;
    }
    m->gather_note = "rccl (in-place ncclAllGather of " + std::to_string(max_rows * 32) + " bytes per rank over " + std::to_string(world) + " devices)";
    return for_each_shard(world, [&](size_t i) -> int {
      cp2_ctx* ctx = cp2_dataset_ctx(mds->shards[i].ds);
      CP2_HIP(ctx, hipSetDevice(ctx->device));
      const void* roots = gath[i].p;
      if (!even) {                               // shards differ by one row: close the gaps of the padded layout
        CP2_TRY(all[i].scratch(ctx, n * 32));
        for (size_t r2 = 0; r2 < world; ++r2)
          CP2_HIP(ctx, hipMemcpyAsync(all[i].u8() + mds->shards[r2].first * 32, gath[i].u8() + r2 * max_rows * 32, mds->shards[r2].count * 32,
                                      hipMemcpyDeviceToDevice, ctx->stream));
        roots = all[i].p;
      }
      return cp2_dataset_set_roots_dev(mds->shards[i].ds, roots);
    });
  }
  // host gather: one download per shard, one upload per device
  m->gather_note = "host (" + (why.empty() ? std::string("requested") : why) + ")";
  std::vector<uint8_t> roots(n * 32);
  for (auto& s : mds->shards) CP2_TRY(cp2_dataset_local_roots(s.ds, roots.data() + s.first * 32));
  return for_each_shard(world, [&](size_t i) { return cp2_dataset_set_roots(mds->shards[i].ds, roots.data()); });
}

enum class BuildKind { Plain, Streamed, Cached };

int multi_build(cp2_multi* m, const cp2_config* cfg, BuildKind kind, const uint8_t* entropy, int threads, size_t group_slots,
                const char* cache_path, cp2_multi_dataset** out) {
  if (!m || !cfg || !out) return CP2_ERR_INVALID;
  *out = nullptr;
  m->err.clear();
  if (cfg->n_slots == 0) return CP2_ERR_INVALID;
  // how many devices get a shard: every device when there is a residency of hashing for each, fewer for small datasets
  const uint64_t min_cells = m->min_cells ? m->min_cells : RESIDENCY_CELLS;
  const unsigned __int128 total_cells = (unsigned __int128)cfg->n_slots * cfg->n_cells;
  uint64_t world = (uint64_t)std::min<unsigned __int128>((total_cells + min_cells - 1) / min_cells, m->devices.size());
  world = std::max<uint64_t>(1, std::min<uint64_t>(world, cfg->n_slots));
  std::unique_ptr<cp2_multi_dataset> mds(new cp2_multi_dataset());
  mds->m = m;
  mds->cfg = *cfg;
  if (cfg->file_base) { mds->file_base = cfg->file_base; mds->cfg.file_base = mds->file_base.c_str(); }
  mds->shards.resize(world);
  for (uint64_t r = 0; r < world; ++r) {
    mds->shards[r].dev = (int)r;
    cp2_shard_range(cfg->n_slots, (int)r, (int)world, &mds->shards[r].first, &mds->shards[r].count);
  }
  const int per = std::max(1, threads / (int)world);
  std::vector<std::string> errs(world);
  StageTimer trace;
  int st = for_each_shard(world, [&](size_t i) -> int {
    auto& s = mds->shards[i];
    int cst = CP2_OK;
    cp2_ctx* ctx = m->ctx_of(s.dev, &cst);
    if (!ctx) { errs[i] = "device " + std::to_string(m->devices[s.dev]) + ": " + cp2_strerror(cst); return cst; }
    int r = CP2_OK;
    if (kind == BuildKind::Streamed) r = cp2_dataset_build_streamed(ctx, &mds->cfg, s.first, s.count, entropy, per, group_slots, &s.ds);
    else if (kind == BuildKind::Cached) {
      const std::string path = world == 1 ? std::string(cache_path) : std::string(cache_path) + ".shard" + std::to_string(i) + "of" + std::to_string(world);
      r = cp2_dataset_build_cached(ctx, &mds->cfg, s.first, s.count, path.c_str(), &s.ds);
    } else r = cp2_dataset_build(ctx, &mds->cfg, s.first, s.count, &s.ds);
    if (r != CP2_OK) errs[i] = "device " + std::to_string(m->devices[s.dev]) + ", slots " + std::to_string(s.first) + ".." + std::to_string(s.first + s.count) +
                               ": " + (*cp2_last_error(ctx) ? cp2_last_error(ctx) : cp2_strerror(r));
    return r;
  });
  if (st != CP2_OK) {
    for (auto& e : errs) if (!e.empty()) { m->err = e; break; }
    return st;
  }
  trace.lap(("slot trees on " + std::to_string(world) + " device context(s)").c_str());
  st = gather_roots_and_build_trees(mds.get());
  trace.lap(("slot roots exchanged: " + m->gather_note).c_str());
  if (st != CP2_OK) {
    if (m->err.empty())
      for (auto& s : mds->shards) {
        const char* e = cp2_last_error(cp2_dataset_ctx(s.ds));
        if (e && *e) { m->err = e; break; }
      }
    return st;
  }
  *out = mds.release();
  return CP2_OK;
}

}  // namespace

extern "C" int cp2_multi_dataset_build(cp2_multi* m, const cp2_config* cfg, cp2_multi_dataset** out) try {
  return multi_build(m, cfg, BuildKind::Plain, nullptr, 1, 0, nullptr, out);
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" int cp2_multi_dataset_build_cached(cp2_multi* m, const cp2_config* cfg, const char* cache_path, cp2_multi_dataset** out) try {
  if (!cache_path) return CP2_ERR_INVALID;
  return multi_build(m, cfg, BuildKind::Cached, nullptr, 1, 0, cache_path, out);
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" int cp2_multi_dataset_build_streamed(cp2_multi* m, const cp2_config* cfg, const uint8_t entropy[32], int threads, size_t group_slots,
                                                cp2_multi_dataset** out) try {
  if (!entropy) return CP2_ERR_INVALID;
  return multi_build(m, cfg, BuildKind::Streamed, entropy, std::max(1, threads), group_slots, nullptr, out);
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" void cp2_multi_dataset_free(cp2_multi_dataset* mds) { delete mds; }

extern "C" int cp2_multi_dataset_shards(const cp2_multi_dataset* mds) { return mds ? (int)mds->shards.size() : 0; }

extern "C" cp2_dataset* cp2_multi_dataset_shard(cp2_multi_dataset* mds, int i, int* device, uint64_t* first_slot, uint64_t* n_local) {
  if (!mds || i < 0 || i >= (int)mds->shards.size()) return nullptr;
  const auto& s = mds->shards[i];
  if (device) *device = mds->m->devices[s.dev];
  if (first_slot) *first_slot = s.first;
  if (n_local) *n_local = s.count;
  return s.ds;
}

extern "C" int cp2_multi_dataset_root(cp2_multi_dataset* mds, uint8_t out[32]) try {
  if (!mds || !out || mds->shards.empty()) return CP2_ERR_INVALID;
  return cp2_dataset_root(mds->shards[0].ds, out);
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" int cp2_multi_dataset_slot_roots(cp2_multi_dataset* mds, uint8_t* out) try {
  if (!mds || !out) return CP2_ERR_INVALID;
  for (auto& s : mds->shards) CP2_TRY(cp2_dataset_local_roots(s.ds, out + s.first * 32));
  return CP2_OK;
} catch (...) {
  return CP2_ERR_INVALID;
}

// generateProofInputBN254 (gen_input/bn254.nim:35-79) on the device that holds the slot
extern "C" int cp2_multi_proof_input_generate(cp2_multi_dataset* mds, uint64_t slot_idx, const uint8_t entropy[32], cp2_proof_input** out) try {
  if (!mds || !out) return CP2_ERR_INVALID;
  *out = nullptr;
  auto* s = mds->owner(slot_idx);
  if (!s) return CP2_ERR_INVALID;                                  // slot index out of range
  int st = cp2_proof_input_generate(s->ds, slot_idx, entropy, out);
  if (st != CP2_OK) mds->m->err = cp2_last_error(cp2_dataset_ctx(s->ds));
  return st;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

// cp2_dataset_export_proof_inputs over all shards: the list is cut by owner, every device works through its own part
extern "C" int cp2_multi_dataset_export_proof_inputs(cp2_multi_dataset* mds, const uint64_t* slot_idx, size_t n, const uint8_t entropy[32],
                                                     const char* dir, int threads, size_t batch, uint64_t* total_bytes) try {
  if (!mds || !entropy || (n && !slot_idx)) return CP2_ERR_INVALID;
  const size_t world = mds->shards.size();
  std::vector<std::vector<uint64_t>> part(world);
  for (size_t i = 0; i < n; ++i) {
    auto* s = mds->owner(slot_idx[i]);
    if (!s) return CP2_ERR_INVALID;
    part[(size_t)(s - mds->shards.data())].push_back(slot_idx[i]);
  }
  const int per = std::max(1, threads / (int)world);
  std::vector<uint64_t> bytes(world, 0);
  int st = for_each_shard(world, [&](size_t i) -> int {
    if (part[i].empty()) return CP2_OK;
    return cp2_dataset_export_proof_inputs(mds->shards[i].ds, part[i].data(), part[i].size(), entropy, dir, per, batch, &bytes[i]);
  });
  uint64_t tot = 0;
  for (uint64_t b : bytes) tot += b;
  if (total_bytes) *total_bytes = tot;
  return st;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" int cp2_multi_dataset_export_streamed(cp2_multi_dataset* mds, const char* dir, int threads, uint64_t* total_bytes) try {
  if (!mds) return CP2_ERR_INVALID;
  const size_t world = mds->shards.size();
  const int per = std::max(1, threads / (int)world);
  std::vector<uint64_t> bytes(world, 0);
  int st = for_each_shard(world, [&](size_t i) { return cp2_dataset_export_streamed(mds->shards[i].ds, dir, per, &bytes[i]); });
  uint64_t tot = 0;
  for (uint64_t b : bytes) tot += b;
  if (total_bytes) *total_bytes = tot;
  return st;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" int cp2_multi_dataset_streamed_json(cp2_multi_dataset* mds, uint64_t slot_idx, char** text, size_t* len) try {
  if (!mds || !text) return CP2_ERR_INVALID;
  auto* s = mds->owner(slot_idx);
  if (!s) return CP2_ERR_INVALID;
  return cp2_dataset_streamed_json(s->ds, slot_idx, text, len);
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

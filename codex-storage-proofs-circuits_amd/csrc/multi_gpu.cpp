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
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "trees.hpp"
#include "exchange_policy.hpp"

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
  std::string why;   // why it could not be loaded -- or why it is no longer to be used (an initialisation that never returned)

  static Rccl& get() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] { r.load(); });
    return r;
  }
  bool ok() const { return lib != nullptr && !abandoned; }
  bool abandoned = false;   // ncclCommInitAll did not return in time: its thread still runs somewhere inside the library, hands off

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
  int64_t split = 0;                          // units per slot: 0 = choose, 1 = whole slots only, 2^k = exactly that (cp2_multi_set_split)
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

// device indices from the environment: CODEX_P2_GPUS = "all", "<count>" (the first <count> visible devices) or a comma-separated
// list of indices ("0,2,3"; "0,0" = two contexts on device 0; "2," = device 2 only)
// returns 1 when the variable named devices, 2 for "all", 0 when it is unset, -1 when it is set to anything else
static int devices_from_env(int visible, std::vector<int>& out) {
  const char* e = std::getenv("CODEX_P2_GPUS");
  if (!e || !*e) return 0;
  const std::string s(e);
  if (s == "all") return 2;                      // every visible gfx950 device
  auto number = [](const std::string& t, long* v) {
    if (t.empty() || t.size() > 6 || t.find_first_not_of("0123456789") != std::string::npos) return false;
    *v = std::strtol(t.c_str(), nullptr, 10);
    return true;
  };
  long n = 0;
  if (s.find(',') == std::string::npos) {
    if (!number(s, &n) || n < 1) return -1;
    for (int d = 0; d < std::min<long>(n, visible); ++d) out.push_back(d);
    return 1;
  }
  size_t at = 0;
  while (at < s.size()) {
    size_t c = s.find(',', at);
    if (c == std::string::npos) c = s.size();
    if (c > at) {
      if (!number(s.substr(at, c - at), &n)) return -1;
      out.push_back((int)n);
    }
    at = c + 1;
  }
  return out.empty() ? -1 : 1;
}

extern "C" int cp2_multi_init(const int* devices, int n_dev, cp2_multi** out) try {
  if (!out || n_dev < 0 || (n_dev > 0 && !devices)) return CP2_ERR_INVALID;
  *out = nullptr;
  // the environment first, before anything touches HIP: a mistyped knob is refused (and named by cp2_check_environment) whatever the box
  if (cp2_check_environment(nullptr, 0) != CP2_OK) return CP2_ERR_INVALID;
  StageTimer trace;
  int visible = 0;
  if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) { (void)hipGetLastError(); return CP2_ERR_NO_DEVICE; }
  trace.lap("HIP runtime init (device count)");
  std::unique_ptr<cp2_multi> m(new cp2_multi());
  const int from_env = n_dev > 0 ? 0 : devices_from_env(visible, m->devices);
  if (from_env < 0) return CP2_ERR_INVALID;      // CODEX_P2_GPUS is set but is neither "all", "<count>" nor an index list: not guessed at
  if (n_dev > 0) {
    m->devices.assign(devices, devices + n_dev);
  } else if (from_env == 0 || from_env == 2) {
    // No device named: ONE device, the first visible gfx950 -- several devices are opt-in (CODEX_P2_GPUS=all | <count> | <list>, or an
    // explicit device list from the caller) until the exchange between two REAL devices has a committed record (DESIGN.md section 7:
    // this pipeline's GPU boxes hold one device; tests/test_gpu_multi.py -k real_device is the run that produces it).
    for (int d = 0; d < visible; ++d) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, d) == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) {
        m->devices.push_back(d);
        if (from_env == 0) break;
      }
    }
  }
  if (m->devices.empty()) return CP2_ERR_NO_DEVICE;
  for (int d : m->devices)
    if (d < 0 || d >= visible) return CP2_ERR_NO_DEVICE;
  m->ctxs.assign(m->devices.size(), nullptr);
  // CODEX_P2_MIN_CELLS (cp2_multi_set_policy), CODEX_P2_SPLIT (cp2_multi_set_split), CODEX_P2_GATHER: checked above, read here
  uint64_t split = 0;
  (void)env_decimal("CODEX_P2_MIN_CELLS", &m->min_cells, nullptr);
  (void)env_decimal("CODEX_P2_SPLIT", &split, nullptr);
  m->split = (int64_t)split;
  if (const char* e = std::getenv("CODEX_P2_GATHER")) {
    if (std::strcmp(e, "rccl") == 0) m->gather = CP2_GATHER_RCCL;
    else if (std::strcmp(e, "host") == 0) m->gather = CP2_GATHER_HOST;
    else if (std::strcmp(e, "copy") == 0) m->gather = CP2_GATHER_COPY;
  }
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
  if (!m || gather < CP2_GATHER_AUTO || gather > CP2_GATHER_COPY) return CP2_ERR_INVALID;
  m->gather = gather;
  m->min_cells = min_cells_per_device;
  return CP2_OK;
}

extern "C" int cp2_multi_set_split(cp2_multi* m, int64_t units_per_slot) {
  if (!m || units_per_slot < 0 || (units_per_slot > 1 && !is_pow2((uint64_t)units_per_slot))) return CP2_ERR_INVALID;
  m->split = units_per_slot;
  return CP2_OK;
}

// ---------------------------------------------------------------------------------------------
// sharded dataset
// ---------------------------------------------------------------------------------------------
// Two ways to cut a dataset over the devices (same contiguous-range rule, cp2_shard_range):
//   by SLOTS  every device holds whole slots as a cp2_dataset of its own: everything the single-context API offers works per
//             shard (streamed builds, caches, batched exports), and each device serves its own slots' proof inputs.
//   by UNITS  every slot is cut into S = 2^s units of n_cells / S cells (whole blocks, at least two), the n_slots * S units are
//             dealt out contiguously: what SURVEY.md 8(e) calls "the same scheme one level down" -- for datasets of FEW, LARGE
//             slots (one 128 GiB slot over eight GPUs; 11 slots over 8 GPUs, where whole slots would leave the busiest device
//             with 2 and the others with 1).  A unit's root is a node of its slot's tree; after the exchange of unit roots the
//             log2 S upper layers of every slot tree and the dataset tree are built once, and a proof input is put together
//             from the unit paths of whichever devices hold the sampled cells plus the upper path.
struct cp2_multi_dataset {
  cp2_multi* m = nullptr;
  cp2_config cfg{};
  std::string file_base;
  struct Shard { int dev = 0; uint64_t first = 0, count = 0; cp2_dataset* ds = nullptr; cp2_slot_trees* units = nullptr; };
  std::vector<Shard> shards;                 // ranges count slots (by slots) or units (by units)
  uint64_t units_per_slot = 1;               // S; 1 = by slots
  // by units: host copies of the upper layers (layer k of ALL slots contiguous: n_slots * (S >> k) elements) and the dataset tree
  std::vector<uint8_t> upper;
  std::vector<size_t> upper_off;
  std::vector<uint8_t> dlayers;
  std::vector<size_t> dsizes;
  // by units, streamed kind: the finished input.json of every slot for the entropy of the build (few, large slots: a few MB)
  bool prepared = false;
  std::vector<std::string> texts;
  ~cp2_multi_dataset() {
    for (auto& s : shards) { cp2_dataset_free(s.ds); cp2_slot_trees_free(s.units); }
  }
  bool by_units() const { return units_per_slot > 1; }
  Shard* owner(uint64_t item) {              // the shard holding slot `item` (by slots) or unit `item` (by units)
    for (auto& s : shards)
      if (item >= s.first && item < s.first + s.count) return &s;
    return nullptr;
  }
  const uint8_t* slot_root(uint64_t slot) const { return &upper[(upper_off.back() + slot) * 32]; }
};

namespace {

// f(i) for every shard on its own host thread (the calling thread takes shard 0); returns the first non-OK status
template <typename F> int for_each_shard(size_t n, F f) {
  std::vector<int> st(n, CP2_OK);
  std::vector<std::thread> th;
  th.reserve(n);
  size_t started = 1;
  try {
    for (; started < n; ++started) th.emplace_back([&, i = started] { try { st[i] = f(i); } catch (...) { st[i] = CP2_ERR_ALLOC; } });
  } catch (...) {   // the host refused another thread: the shards without one run here, after shard 0
  }
  try { st[0] = f(0); } catch (...) { st[0] = CP2_ERR_ALLOC; }
  for (size_t i = started; i < n; ++i) { try { st[i] = f(i); } catch (...) { st[i] = CP2_ERR_ALLOC; } }
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

// One shard's contribution to the exchange: `count` roots in device memory of `ctx`, rows [first, first + count) of the whole.
struct RootsPart { cp2_ctx* ctx; const void* d_roots; uint64_t first, count; };
// What the exchange leaves behind: all n roots on EVERY device (dev[i], in memory owned by this object), or in host memory.
struct Exchanged {
  std::vector<DevBuf> gath, all;
  std::vector<const void*> dev;
  std::vector<uint8_t> host;
  bool on_device = false;
  bool timed_out = false;   // the exchange was given up with work still queued on the participating streams: NOT a failed or mis-verified
                            // exchange -- nothing more may be enqueued on those streams (a retry through host memory would wait on them for ever)
  // work that may still be reading or writing these buffers could not be waited for (it timed out): they must never go back to
  // a pool or to the device's allocator -- they are dropped from the books instead (a leak, said so in the error message)
  void abandon() {
    for (auto* v : {&gath, &all})
      for (auto& b : *v) { b.p = nullptr; b.bytes = 0; b.owner = nullptr; }
  }
};

// seconds an exchange may take before it is given up (CODEX_P2_EXCHANGE_TIMEOUT_S; 0 = wait for ever).  The exchange moves 1 MiB;
// what this bounds is a collective that never completes on first contact with a machine (a peer that cannot be reached, a
// communicator whose creation hangs) -- the caller gets an error that says so instead of a process that never returns.
double exchange_timeout_s() {
  uint64_t v = 120;
  bool set = false;
  if (env_decimal("CODEX_P2_EXCHANGE_TIMEOUT_S", &v, &set) && set) return (double)v;
  return 120.0;
}

// Waits until every participating context's stream has drained.  Bounded: false when `timeout_s` (> 0) passed first.
bool drain_streams(const std::vector<RootsPart>& parts, double timeout_s) {
  const auto t0 = std::chrono::steady_clock::now();
  for (auto& p : parts) {
    (void)hipSetDevice(p.ctx->device);
    if (timeout_s <= 0) { (void)hipStreamSynchronize(p.ctx->stream); continue; }
    for (;;) {
      hipError_t e = hipStreamQuery(p.ctx->stream);
      if (e != hipErrorNotReady) { (void)hipGetLastError(); break; }
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
      std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
  }
  return true;
}

// Whatever path leaves exchange_roots, the participating streams are drained before its buffers can be destroyed by the caller:
// a context's gather buffer is read by its PEERS (RCCL kernels, peer copies), so its own stream alone says nothing.  When even the
// bounded drain does not finish, the buffers are abandoned rather than handed back while something may still touch them.
struct ExchangeGuard {
  const std::vector<RootsPart>& parts;
  Exchanged& ex;
  double timeout_s;
  bool done = false;                   // set on the success path, which drains (and checks) by itself
  ~ExchangeGuard() {
    if (done) return;
    if (!drain_streams(parts, timeout_s)) {
      ex.abandon();
      ex.timed_out = true;
      for (auto& p : parts) p.ctx->stuck = true;
    }
  }
};

// ncclCommInitAll on a helper thread, so that a creation that never returns (first contact with a machine's fabric) costs a
// timeout and not the process: on timeout the thread is left behind with everything it references kept alive, RCCL is marked
// abandoned for the rest of the process and the caller falls back (automatic mode) or reports it (RCCL by name).
struct CommInit {
  std::mutex mu;
  std::condition_variable cv;
  bool finished = false;
  ncclResult_t result = ncclSuccess;
  std::vector<ncclComm_t> comms;
  std::vector<int> devices;
};
bool comm_init_all(Rccl& r, const std::vector<int>& devices, size_t world, double timeout_s, std::vector<ncclComm_t>& out, std::string& why) {
  auto st = std::make_shared<CommInit>();
  st->comms.assign(world, nullptr);
  st->devices.assign(devices.begin(), devices.begin() + (long)world);
  std::thread th([st, &r] {
    if (const char* fault = std::getenv("CODEX_P2_TEST_EXCHANGE_FAULT"))      // test-only: a communicator creation that never returns
      if (std::strcmp(fault, "hang_init") == 0)
        for (;;) std::this_thread::sleep_for(std::chrono::seconds(3600));
    ncclResult_t e = r.CommInitAll(st->comms.data(), (int)st->comms.size(), st->devices.data());
    std::lock_guard<std::mutex> lk(st->mu);
    st->result = e;
    st->finished = true;
    st->cv.notify_all();
  });
  std::unique_lock<std::mutex> lk(st->mu);
  const bool in_time = timeout_s <= 0 ? (st->cv.wait(lk, [&] { return st->finished; }), true)
                                      : st->cv.wait_for(lk, std::chrono::duration<double>(timeout_s), [&] { return st->finished; });
  if (!in_time) {
    lk.unlock();
    th.detach();                       // `st` stays alive through the thread's copy of the shared pointer
    r.abandoned = true;
    r.why = why = "ncclCommInitAll did not return within " + std::to_string((int)timeout_s) + " s (abandoned for the rest of this process)";
    return false;
  }
  lk.unlock();
  th.join();
  if (st->result != ncclSuccess) {
    why = std::string("ncclCommInitAll: ") + r.GetErrorString(st->result);
    return false;
  }
  out = st->comms;
  return true;
}

// THE exchange step: every shard's roots to every device.  RCCL (device to device) when every shard sits on its own device and
// librccl loads; host memory otherwise; by name also plain device-to-device copies (CP2_GATHER_COPY: the RCCL path's buffers,
// layout and compaction with the collective written out as peer copies -- no library, any mix of devices).  One shard and no
// RCCL by name: nothing moves.  `force_host`: the automatic mode's second attempt after a device path failed its verification.
int exchange_roots(cp2_multi* m, const std::vector<RootsPart>& parts, uint64_t n, Exchanged& ex, bool force_host = false) {
  const size_t world = parts.size();
  ex.dev.assign(world, nullptr);
  if (world == 1 && m->gather != CP2_GATHER_RCCL) {   // (RCCL asked for by name: a communicator of one rank, as a self-test of the path)
    m->gather_note = "none (one shard: nothing to exchange)";
    ex.dev[0] = parts[0].d_roots;
    ex.on_device = true;
    return CP2_OK;
  }
  DeviceRestore restore;
  const double timeout_s = exchange_timeout_s();
  ExchangeGuard guard{parts, ex, timeout_s};
  bool distinct = true;                        // RCCL needs one rank per DISTINCT device
  for (size_t i = 0; i < world; ++i)
    for (size_t j = 0; j < i; ++j)
      if (parts[i].ctx->device == parts[j].ctx->device) distinct = false;
  std::string why = force_host ? "the device-to-device exchange failed its verification" : "";
  const bool use_copy = m->gather == CP2_GATHER_COPY && !force_host;   // device-to-device copies, pair by pair: same layout and compaction as RCCL, no library
  bool use_rccl = m->gather != CP2_GATHER_HOST && !use_copy && !force_host;
  if (use_rccl && !distinct) { use_rccl = false; why = "a device holds more than one shard"; }
  if (use_rccl && !Rccl::get().ok()) { use_rccl = false; why = Rccl::get().why; }
  if (use_rccl) {
    Rccl& r = Rccl::get();
    if (m->comm_world != (int)world) {          // shards are always devices[0 .. world)
      m->drop_comms();
      std::vector<ncclComm_t> comms;
      if (comm_init_all(r, m->devices, world, timeout_s, comms, why)) {
        m->comms = comms;
        m->comm_world = (int)world;
      } else {
        use_rccl = false;
      }
    }
  }
  if (!use_rccl && m->gather == CP2_GATHER_RCCL && !force_host) {
    m->err = "RCCL gather requested but unavailable: " + why;
    return CP2_ERR_INVALID;
  }
  if (use_rccl || use_copy) {
    uint64_t max_rows = 0;
    bool even = true;
    for (auto& p : parts) { max_rows = std::max(max_rows, p.count); even = even && p.count == parts[0].count; }
    ex.gath = std::vector<DevBuf>(world);
    ex.all = std::vector<DevBuf>(world);
    for (size_t i = 0; i < world; ++i) {        // this shard's roots into its own row block of its gather buffer
      cp2_ctx* ctx = parts[i].ctx;
      CP2_HIP(ctx, hipSetDevice(ctx->device));
      CP2_TRY(ex.gath[i].scratch(ctx, world * max_rows * 32));
      CP2_HIP(ctx, hipMemcpyAsync(ex.gath[i].u8() + i * max_rows * 32, parts[i].d_roots, parts[i].count * 32, hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (use_rccl) {
      Rccl& r = Rccl::get();
      ncclResult_t e = r.GroupStart();
      for (size_t i = 0; i < world && e == ncclSuccess; ++i)
        e = r.AllGather(ex.gath[i].u8() + i * max_rows * 32, ex.gath[i].p, max_rows * 32, ncclUint8, m->comms[i], parts[i].ctx->stream);   // in place
      ncclResult_t e2 = r.GroupEnd();
      if (e == ncclSuccess) e = e2;
      if (e != ncclSuccess) {
        m->err = std::string("ncclAllGather: ") + r.GetErrorString(e);
        return CP2_ERR_HIP;                       // (the guard drains what was enqueued before the buffers go)
      }
      m->gather_note = "rccl (in-place ncclAllGather of " + std::to_string(max_rows * 32) + " bytes per rank over " + std::to_string(world) + " devices)";
    } else {
      // the same all-gather written out as world x (world - 1) peer copies: every shard's row block, once it is staged, into the
      // same place of every other context's buffer, on the RECEIVING context's stream (what consumes it there follows in order)
      for (auto& p : parts) {
        CP2_HIP(p.ctx, hipSetDevice(p.ctx->device));
        CP2_HIP(p.ctx, hipStreamSynchronize(p.ctx->stream));
      }
      for (size_t i = 0; i < world; ++i) {
        cp2_ctx* ctx = parts[i].ctx;
        CP2_HIP(ctx, hipSetDevice(ctx->device));
        for (size_t j = 0; j < world; ++j)
          if (j != i && parts[j].count)
            CP2_HIP(ctx, hipMemcpyPeerAsync(ex.gath[i].u8() + j * max_rows * 32, ctx->device, ex.gath[j].u8() + j * max_rows * 32, parts[j].ctx->device,
                                            parts[j].count * 32, ctx->stream));
      }
      m->gather_note = "copy (" + std::to_string(world * (world - 1)) + " device-to-device copies of at most " + std::to_string(max_rows * 32) + " bytes over " +
                       std::to_string(world) + " contexts)";
    }
    for (size_t i = 0; i < world; ++i) {
      cp2_ctx* ctx = parts[i].ctx;
      ex.dev[i] = ex.gath[i].p;
      if (even) continue;
      CP2_HIP(ctx, hipSetDevice(ctx->device));   // shards differ by one row: close the gaps of the padded layout
      CP2_TRY(ex.all[i].scratch(ctx, n * 32));
      for (size_t r2 = 0; r2 < world; ++r2)
        CP2_HIP(ctx, hipMemcpyAsync(ex.all[i].u8() + parts[r2].first * 32, ex.gath[i].u8() + r2 * max_rows * 32, parts[r2].count * 32,
                                    hipMemcpyDeviceToDevice, ctx->stream));
      ex.dev[i] = ex.all[i].p;
    }
    if (const char* fault = std::getenv("CODEX_P2_TEST_EXCHANGE_FAULT")) {
      // test-only: overwrite one byte of the first row in every context's gathered copy -- what a wrong rank-to-device mapping or a
      // misplaced block would look like to the verification that follows the exchange (tests/test_gpu_round5.py)
      if (std::strcmp(fault, "corrupt") == 0)
        for (size_t i = 0; i < world; ++i) {
          cp2_ctx* ctx = parts[i].ctx;
          CP2_HIP(ctx, hipSetDevice(ctx->device));
          CP2_HIP(ctx, hipMemsetAsync(const_cast<uint8_t*>(static_cast<const uint8_t*>(ex.dev[i])) + 5, 0x5a, 1, ctx->stream));
        }
    }
    if (const char* fault = std::getenv("CODEX_P2_TEST_EXCHANGE_FAULT")) {
      // test-only: a collective that does not complete in time -- a host function that sleeps for the time-out and five seconds more on
      // every participating stream (bounded: the streams do drain in the end, the device itself is never made to spin)
      if (std::strcmp(fault, "hang_collective") == 0)
        for (auto& p : parts) {
          CP2_HIP(p.ctx, hipSetDevice(p.ctx->device));
          CP2_HIP(p.ctx, hipLaunchHostFunc(p.ctx->stream, [](void* ms) { std::this_thread::sleep_for(std::chrono::milliseconds((long)(intptr_t)ms)); },
                                           reinterpret_cast<void*>((intptr_t)((timeout_s > 0 ? timeout_s : 1) * 1000 + 5000))));
        }
    }
    // the exchange is COMPLETE when this returns: a context's buffers are read by its peers (RCCL kernels, peer copies), so none
    // of them may go back to its pool on the strength of its own stream alone
    if (!drain_streams(parts, timeout_s)) {
      ex.abandon();
      ex.timed_out = true;
      for (auto& p : parts) p.ctx->stuck = true;   // whatever is enqueued on these streams from now on waits behind the collective
      guard.done = true;
      m->err = "the exchange of slot roots (" + m->gather_note + ") did not complete within " + std::to_string((int)timeout_s) +
               " s; its device buffers are abandoned and the participating contexts take no further work (CODEX_P2_EXCHANGE_TIMEOUT_S, CODEX_P2_GATHER=host)";
      if (use_rccl) { Rccl::get().abandoned = true; Rccl::get().why = "an all-gather did not complete in time"; m->comms.clear(); m->comm_world = 0; }
      return CP2_ERR_HIP;
    }
    for (auto& p : parts) {                       // a failed copy or kernel shows up here, not in the next unrelated call
      CP2_HIP(p.ctx, hipSetDevice(p.ctx->device));
      CP2_HIP(p.ctx, hipStreamSynchronize(p.ctx->stream));
    }
    guard.done = true;
    ex.on_device = true;
    return CP2_OK;
  }
  // host gather: one download per shard (and one upload per device by whoever consumes ex.host)
  m->gather_note = "host (" + (why.empty() ? std::string("requested") : why) + ")";
  ex.host.assign(n * 32, 0);
  for (auto& p : parts) {
    CP2_HIP(p.ctx, hipSetDevice(p.ctx->device));
    CP2_HIP(p.ctx, hipMemcpyAsync(ex.host.data() + p.first * 32, p.d_roots, p.count * 32, hipMemcpyDeviceToHost, p.ctx->stream));
    CP2_HIP(p.ctx, hipStreamSynchronize(p.ctx->stream));
  }
  guard.done = true;
  return CP2_OK;
}

// by slots: the exchange, then the dataset tree on every device -- and a check that costs one 32-byte-per-slot download per
// shard: (1) every device finds ITS OWN roots at its own rows of the list it was handed, (2) every device computed the same
// dataset root.  Together: every device holds the same list and every block of it is where its owner put it, i.e. the exchange
// did what it is for -- whatever carried it.  A wrong rank-to-device mapping, a misplaced block of the padded layout or a peer
// that delivered stale memory gives CP2_ERR_HIP here, never a wrong dataSetRoot in an input.json.  In the automatic mode a
// device path that fails the check is retried once through host memory.
int gather_roots_and_build_trees(cp2_multi_dataset* mds) {
  cp2_multi* m = mds->m;
  const size_t world = mds->shards.size();
  std::vector<RootsPart> parts;
  for (auto& s : mds->shards) parts.push_back({cp2_dataset_ctx(s.ds), cp2_dataset_local_roots_dev(s.ds), s.first, s.count});
  for (int attempt = 0; attempt < 2; ++attempt) {
    Exchanged ex;
    int st = exchange_roots(m, parts, mds->cfg.n_slots, ex, attempt == 1);
    if (st != CP2_OK) {
      if (exchange_may_retry_on_host(st, ex.timed_out, m->gather, world, attempt)) {   // the device path broke (NOT: timed out): host memory carries 1 MiB just as well
        const std::string first = m->err;
        m->err.clear();
        Exchanged ex2;
        st = exchange_roots(m, parts, mds->cfg.n_slots, ex2, true);
        if (st != CP2_OK) { m->err = first; return st; }
        m->gather_note += " [first attempt: " + first + "]";
        CP2_TRY(for_each_shard(world, [&](size_t i) -> int { return cp2_dataset_set_roots(mds->shards[i].ds, ex2.host.data()); }));
        attempt = 1;
      } else {
        return st;
      }
    } else {
      CP2_TRY(for_each_shard(world, [&](size_t i) -> int {
        return ex.on_device ? cp2_dataset_set_roots_dev(mds->shards[i].ds, ex.dev[i]) : cp2_dataset_set_roots(mds->shards[i].ds, ex.host.data());
      }));
    }
    if (m->gather_note.rfind("none", 0) == 0) return CP2_OK;   // nothing moved
    // ---- verification (also of a one-rank communicator asked for by name: the RCCL path's self-test)
    std::vector<char> in_place(world, 1);
    std::vector<std::array<uint8_t, 32>> roots(world);
    CP2_TRY(for_each_shard(world, [&](size_t i) -> int {
      bool ok = false;
      CP2_TRY(dataset_own_roots_in_place(mds->shards[i].ds, &ok));
      in_place[i] = ok ? 1 : 0;
      return cp2_dataset_root(mds->shards[i].ds, roots[i].data());
    }));
    std::string bad;
    for (size_t i = 0; i < world && bad.empty(); ++i) {
      if (!in_place[i]) bad = "the device of shard " + std::to_string(i) + " does not find its own slot roots at rows " + std::to_string(mds->shards[i].first) + ".." +
                              std::to_string(mds->shards[i].first + mds->shards[i].count) + " of the gathered list";
      else if (roots[i] != roots[0]) bad = "shards 0 and " + std::to_string(i) + " computed different dataset roots from the gathered list";
    }
    if (bad.empty()) return CP2_OK;
    if (attempt == 0 && m->gather == CP2_GATHER_AUTO && ex.on_device) {
      if (std::getenv("CP2_TRACE")) std::fprintf(stderr, "[cp2 trace] exchange verification FAILED (%s: %s): once more through host memory\n", m->gather_note.c_str(), bad.c_str());
      continue;
    }
    m->err = "exchange verification failed (" + m->gather_note + "): " + bad;
    return CP2_ERR_HIP;
  }
  return CP2_ERR_HIP;   // not reached
}

// by units: the exchange of unit roots, then -- once, on the first device -- the log2 S upper layers of every slot tree
// (inner layers of gen_input/bn254.nim:29's tree: keys 0, never the bottom rule) and the dataset tree over the slot roots.
// Verified like the exchange of slot roots: every shard finds its own unit roots at its rows of the list the first device built from.
int gather_unit_roots_and_build_upper(cp2_multi_dataset* mds) {
  cp2_multi* m = mds->m;
  const uint64_t S = mds->units_per_slot, n_slots = mds->cfg.n_slots, n_units = n_slots * S;
  std::vector<RootsPart> parts;
  for (auto& s : mds->shards) parts.push_back({s.units->ctx, cp2_slot_trees_roots_dev(s.units), s.first, s.count});
  for (auto& p : parts) {                          // the unit trees are complete before their roots travel
    CP2_HIP(p.ctx, hipSetDevice(p.ctx->device));
    CP2_HIP(p.ctx, hipStreamSynchronize(p.ctx->stream));
  }
  DeviceRestore restore;
  cp2_ctx* ctx = parts[0].ctx;
  size_t levels = 0;
  while (((uint64_t)1 << levels) < S) ++levels;
  size_t total = 0;
  mds->upper_off.clear();
  for (size_t k = 0; k <= levels; ++k) { mds->upper_off.push_back(total); total += n_slots * (S >> k); }
  for (int attempt = 0; attempt < 2; ++attempt) {
    Exchanged ex;
    int st = exchange_roots(m, parts, n_units, ex, attempt == 1);
    if (st != CP2_OK) {
      if (exchange_may_retry_on_host(st, ex.timed_out, m->gather, parts.size(), attempt)) { m->gather_note += " [failed: " + m->err + "]"; m->err.clear(); continue; }
      return st;
    }
    CP2_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf up, dtree;
    CP2_TRY(up.scratch(ctx, total * 32));
    if (ex.on_device) CP2_HIP(ctx, hipMemcpyAsync(up.p, ex.dev[0], n_units * 32, hipMemcpyDeviceToDevice, ctx->stream));
    else CP2_HIP(ctx, hipMemcpyAsync(up.p, ex.host.data(), n_units * 32, hipMemcpyHostToDevice, ctx->stream));
    for (size_t k = 0; k < levels; ++k)
      CP2_HIP(ctx, cp2k::launch_compress_layer(up.u8() + mds->upper_off[k] * 32, up.u8() + mds->upper_off[k + 1] * 32, S >> k, n_slots, false,
                                               S >> k, S >> (k + 1), ctx->stream));
    mds->dsizes = layer_sizes_of(n_slots);
    const size_t dtotal = cp2_merkle_total(n_slots);
    CP2_TRY(dtree.scratch(ctx, dtotal * 32));
    CP2_TRY(merkle_trees_dev(ctx, up.u8() + mds->upper_off[levels] * 32, n_slots, 1, dtree.p, false));   // gen_input/bn254.nim:49-50
    mds->upper.assign(total * 32, 0);
    mds->dlayers.assign(dtotal * 32, 0);
    CP2_HIP(ctx, hipMemcpyAsync(mds->upper.data(), up.p, total * 32, hipMemcpyDeviceToHost, ctx->stream));
    CP2_HIP(ctx, hipMemcpyAsync(mds->dlayers.data(), dtree.p, dtotal * 32, hipMemcpyDeviceToHost, ctx->stream));
    CP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (parts.size() == 1) return CP2_OK;
    // every shard's own unit roots against rows [first, first + count) of what the first device built from
    std::string bad;
    std::vector<uint8_t> own;
    for (size_t i = 0; i < parts.size() && bad.empty(); ++i) {
      own.resize(parts[i].count * 32);
      CP2_HIP(parts[i].ctx, hipSetDevice(parts[i].ctx->device));
      CP2_HIP(parts[i].ctx, hipMemcpyAsync(own.data(), parts[i].d_roots, own.size(), hipMemcpyDeviceToHost, parts[i].ctx->stream));
      CP2_HIP(parts[i].ctx, hipStreamSynchronize(parts[i].ctx->stream));
      if (std::memcmp(own.data(), &mds->upper[parts[i].first * 32], own.size()) != 0)
        bad = "the unit roots of shard " + std::to_string(i) + " are not at rows " + std::to_string(parts[i].first) + ".." + std::to_string(parts[i].first + parts[i].count) +
              " of the list the first device received";
    }
    if (bad.empty()) return CP2_OK;
    if (attempt == 0 && m->gather == CP2_GATHER_AUTO && ex.on_device) {
      if (std::getenv("CP2_TRACE")) std::fprintf(stderr, "[cp2 trace] exchange verification FAILED (%s: %s): once more through host memory\n", m->gather_note.c_str(), bad.c_str());
      continue;
    }
    m->err = "exchange verification failed (" + m->gather_note + "): " + bad;
    return CP2_ERR_HIP;
  }
  return CP2_ERR_HIP;   // not reached
}

double imbalance(uint64_t items, uint64_t world) { return (double)((items + world - 1) / world * world) / (double)items; }

// How many units to cut every slot into: 1 (whole slots) while the busiest device holds at most 6 % more than its share,
// else the smallest power of two that gets there (units are at least two whole blocks).
uint64_t choose_units_per_slot(const cp2_config& c, uint64_t world, int64_t forced) {
  if (c.cell_size == 0 || c.block_size % c.cell_size) return 1;
  const uint64_t cpb = c.block_size / c.cell_size;
  if (!is_pow2(c.n_cells) || !is_pow2(cpb) || c.n_cells / cpb < 4) return 1;
  const uint64_t max_s = c.n_cells / cpb / 2;
  if (forced >= 1) return (is_pow2((uint64_t)forced) && (uint64_t)forced <= max_s) ? (uint64_t)forced : 1;
  if (world <= 1 || imbalance(c.n_slots, world) <= 1.06) return 1;
  uint64_t best = 1;
  double best_imb = imbalance(c.n_slots, world);
  for (uint64_t S = 2; S <= max_s && S <= 4096; S <<= 1) {
    const double imb = imbalance(c.n_slots * S, world);
    if (imb < best_imb - 0.03) { best = S; best_imb = imb; }
    if (imb <= 1.06) break;
  }
  return best;
}

// How a dataset is cut over `n_dev` devices: how many of them get a shard -- every device when there is a residency of hashing
// for each, fewer for small datasets -- and into how many units every slot is cut (1: whole slots).  Pure arithmetic.
void plan_shards(const cp2_config& c, size_t n_dev, uint64_t min_cells_per_device, int64_t split, uint64_t* world_out, uint64_t* units_out) {
  const uint64_t min_cells = min_cells_per_device ? min_cells_per_device : RESIDENCY_CELLS;
  const unsigned __int128 total_cells = (unsigned __int128)c.n_slots * c.n_cells;
  uint64_t world = (uint64_t)std::min<unsigned __int128>((total_cells + min_cells - 1) / min_cells, n_dev);
  world = std::max<uint64_t>(1, world);
  const uint64_t S = split == 1 ? 1 : choose_units_per_slot(c, world, split);
  *world_out = std::min<uint64_t>(world, c.n_slots * S);
  *units_out = S;
}

enum class BuildKind { Plain, Streamed, Cached };
int units_export(cp2_multi_dataset* mds, const uint64_t* slot_idx, size_t n, const uint8_t entropy[32], const char* dir, int threads, size_t batch,
                 uint64_t* total_bytes, std::vector<std::string>* keep);

int multi_build(cp2_multi* m, const cp2_config* cfg, BuildKind kind, const uint8_t* entropy, int threads, size_t group_slots,
                const char* cache_path, cp2_multi_dataset** out) {
  if (!m || !cfg || !out) return CP2_ERR_INVALID;
  *out = nullptr;
  m->err.clear();
  if (cfg->n_slots == 0 || cfg->max_depth < 0 || cfg->max_log2_nslots < 0) return CP2_ERR_INVALID;
  // whole slots or units -- the same plan for every kind of build.  A STREAMED build cut by units is two-phase: nothing of a proof
  // input can be made while later units hash (sampling needs the slot root, which exists only after the exchange of unit roots), so
  // it is the balanced unit build, the exchange, then every slot's input.json from the devices that hold its units -- batched per
  // device, the devices in parallel (units_export) -- kept as text for cp2_multi_dataset_export_streamed / _streamed_json.  (Round 4
  // fell back to whole slots here: 11 slots of 8 GiB on 8 GPUs left the busiest device with 2 slots against a share of 1.375.)
  uint64_t world = 1, S = 1;
  plan_shards(*cfg, m->devices.size(), m->min_cells, m->split, &world, &S);
  // how many shards build on each PHYSICAL device at the same time (an index may repeat): each context's automatic residency choice
  // takes its share of what that device has free, not all of it (proof_input.cpp dataset_tree_mode)
  auto share_of = [&](uint64_t r, uint64_t w) { int k = 0; for (uint64_t q = 0; q < w; ++q) k += m->devices[q] == m->devices[r]; return k; };
  if (S > 1 && m->split == 0) {
    // Cut by units every node of every unit tree stays resident (the compact / roots-only modes exist for whole slots).  When that
    // does not fit the devices -- many LARGE slots: 9 x 1 TiB over 8 GPUs -- the plan falls back to whole slots, whose datasets
    // choose their residency themselves.  (A split the caller named is taken literally.)
    DeviceRestore restore;
    bool fits = true;
    for (uint64_t r = 0; r < world && fits; ++r) {
      uint64_t first = 0, count = 0;
      cp2_shard_range(cfg->n_slots * S, (int)r, (int)world, &first, &count);
      size_t free_b = 0;
      if (hipSetDevice(m->devices[r]) != hipSuccess || device_free_bytes(&free_b) != CP2_OK) { (void)hipGetLastError(); continue; }
      const unsigned __int128 need = (unsigned __int128)trees_node_bytes(1, cfg->cell_size, cfg->block_size, cfg->n_cells / S) * count +
                                     3 * std::min<unsigned __int128>((unsigned __int128)count * (cfg->n_cells / S) * cfg->cell_size, (unsigned __int128)2 << 30) + ((unsigned __int128)1 << 30);
      fits = need <= (unsigned __int128)(free_b / (size_t)share_of(r, world)) * 9 / 10;
    }
    if (!fits) {
      if (std::getenv("CP2_TRACE")) std::fprintf(stderr, "[cp2 trace] cut by %llu units the unit trees would not fit the devices: whole slots instead\n", (unsigned long long)S);
      plan_shards(*cfg, m->devices.size(), m->min_cells, 1, &world, &S);
    }
  }
  std::unique_ptr<cp2_multi_dataset> mds(new cp2_multi_dataset());
  mds->m = m;
  mds->cfg = *cfg;
  if (cfg->file_base) { mds->file_base = cfg->file_base; mds->cfg.file_base = mds->file_base.c_str(); }
  mds->units_per_slot = S;
  mds->shards.resize(world);
  for (uint64_t r = 0; r < world; ++r) {
    mds->shards[r].dev = (int)r;
    cp2_shard_range(cfg->n_slots * S, (int)r, (int)world, &mds->shards[r].first, &mds->shards[r].count);
  }
  const int per = std::max(1, threads / (int)world);
  std::vector<std::string> errs(world);
  // what every device has free NOW, before any shard allocates, shared out among the contexts placed on it: each context's
  // automatic residency choice plans with its share (0: the device did not answer, the context asks for itself)
  std::vector<size_t> allowance(world, 0);
  {
    DeviceRestore restore;
    for (uint64_t r = 0; r < world; ++r) {
      size_t free_b = 0;
      if (hipSetDevice(m->devices[r]) == hipSuccess && device_free_bytes(&free_b) == CP2_OK) allowance[r] = std::max<size_t>(1, free_b / (size_t)share_of(r, world));
      else (void)hipGetLastError();
    }
  }
  StageTimer trace;
  int st = for_each_shard(world, [&](size_t i) -> int {
    auto& s = mds->shards[i];
    int cst = CP2_OK;
    cp2_ctx* ctx = m->ctx_of(s.dev, &cst);
    if (!ctx) { errs[i] = "device " + std::to_string(m->devices[s.dev]) + ": " + cp2_strerror(cst); return cst; }
    ctx->mem_allowance = allowance[i];
    struct AllowanceReset { cp2_ctx* c; ~AllowanceReset() { c->mem_allowance = 0; } } allowance_reset{ctx};
    int r = CP2_OK;
    if (S > 1) {
      // cached: this shard's unit trees from "<cache>.units<S>.shard<i>of<world>" when that file is intact and describes exactly
      // these units of this data (cp2_slot_trees_load checks the checksum and, for slot files, their sizes and mtimes)
      const std::string path = kind == BuildKind::Cached ? std::string(cache_path) + ".units" + std::to_string(S) + ".shard" + std::to_string(i) + "of" + std::to_string(world) : std::string();
      if (kind == BuildKind::Cached && cp2_slot_trees_load(ctx, path.c_str(), &s.units) == CP2_OK) {
        const cp2_slot_trees* t = s.units;
        const bool match = t->n_slots == s.count && t->first_slot == s.first && t->units_per_slot == S && t->cell_size == cfg->cell_size &&
                           t->block_size == cfg->block_size && t->n_cells == cfg->n_cells / S &&
                           (cfg->file_base ? (t->src == CellSrc::File && t->file_base == mds->file_base) : (t->src == CellSrc::Fake && t->dataset_seed == cfg->seed));
        if (match) return CP2_OK;
        cp2_slot_trees_free(s.units);
        s.units = nullptr;
      }
      r = cfg->file_base ? cp2_slot_trees_build_file_units(ctx, mds->file_base.c_str(), S, s.first, s.count, cfg->cell_size, cfg->block_size, cfg->n_cells / S, &s.units)
                         : cp2_slot_trees_build_fake_units(ctx, cfg->seed, S, s.first, s.count, cfg->cell_size, cfg->block_size, cfg->n_cells / S, &s.units);
      if (r == CP2_OK && kind == BuildKind::Cached) r = cp2_slot_trees_save(s.units, path.c_str());
    } else if (kind == BuildKind::Streamed) {
      r = cp2_dataset_build_streamed(ctx, &mds->cfg, s.first, s.count, entropy, per, group_slots, &s.ds);
    } else if (kind == BuildKind::Cached) {
      const std::string path = world == 1 ? std::string(cache_path) : std::string(cache_path) + ".shard" + std::to_string(i) + "of" + std::to_string(world);
      r = cp2_dataset_build_cached(ctx, &mds->cfg, s.first, s.count, path.c_str(), &s.ds);
    } else {
      r = cp2_dataset_build(ctx, &mds->cfg, s.first, s.count, &s.ds);
    }
    if (r != CP2_OK) errs[i] = "device " + std::to_string(m->devices[s.dev]) + (S > 1 ? ", units " : ", slots ") + std::to_string(s.first) + ".." +
                               std::to_string(s.first + s.count) + ": " + (*cp2_last_error(ctx) ? cp2_last_error(ctx) : cp2_strerror(r));
    return r;
  });
  if (st != CP2_OK) {
    for (auto& e : errs) if (!e.empty()) { m->err = e; break; }
    return st;
  }
  trace.lap((std::string(S > 1 ? "unit trees on " : "slot trees on ") + std::to_string(world) + " device context(s)").c_str());
  st = S > 1 ? gather_unit_roots_and_build_upper(mds.get()) : gather_roots_and_build_trees(mds.get());
  trace.lap(("slot roots exchanged: " + m->gather_note).c_str());
  if (st != CP2_OK) {
    if (m->err.empty())
      for (auto& s : mds->shards) {
        const char* e = cp2_last_error(s.ds ? cp2_dataset_ctx(s.ds) : s.units->ctx);
        if (e && *e) { m->err = e; break; }
      }
    return st;
  }
  if (S > 1 && kind == BuildKind::Streamed) {
    mds->texts.assign(cfg->n_slots, std::string());
    std::vector<uint64_t> all(cfg->n_slots);
    for (uint64_t i = 0; i < cfg->n_slots; ++i) all[i] = i;
    uint64_t tot = 0;
    CP2_TRY(units_export(mds.get(), all.data(), all.size(), entropy, nullptr, threads, 0, &tot, &mds->texts));
    mds->prepared = true;
    trace.lap("every input.json from the devices holding the units");
  }
  *out = mds.release();
  return CP2_OK;
}

// generateProofInput (gen_input/bn254.nim:35-79) for `n` slots of a dataset cut by units, shaped like the by-slots path: ONE
// sampling launch for all (slot, counter) pairs (sample/bn254.nim:16-27), the touched units grouped by the device that holds
// them -- one batched gather per device, the devices in parallel (the bottom of every path: merkleProof inside the unit) -- the
// top of every path from the upper layers, the sampled cells regenerated in one launch (or read from the slot files on `threads`
// host threads), all of it handed to the byte-exact writer's objects (cp2_proof_input_create).  Round 4 did one
// cp2_slot_trees_paths + synchronisation per touched unit, one slot after the other.
int units_proof_inputs(cp2_multi_dataset* mds, const uint64_t* slots, size_t n, const uint8_t entropy_in[32], int threads, cp2_proof_input** out) {
  const cp2_config& c = mds->cfg;
  cp2_multi* m = mds->m;
  for (size_t i = 0; i < n; ++i) { out[i] = nullptr; if (slots[i] >= c.n_slots) return CP2_ERR_INVALID; }
  if (n == 0) return CP2_OK;
  if (c.n_samples && c.n_cells < 2) return CP2_ERR_INVALID;                        // extractLowBits asserts k > 0, types/bn254.nim:48
  if (!is_pow2(c.n_cells)) return CP2_ERR_INVALID;                                 // sample/bn254.nim:19-20
  const uint64_t S = mds->units_per_slot, P = c.n_cells / S, cpb = c.block_size / c.cell_size;
  const size_t ns = c.n_samples, md = (size_t)c.max_depth, cs = c.cell_size, total = n * ns;
  size_t levels = 0;
  while (((uint64_t)1 << levels) < S) ++levels;
  const size_t depth_unit = (layer_sizes_of(cpb).size() - 1) + (layer_sizes_of(P / cpb).size() - 1);
  if (depth_unit + levels > md) return CP2_ERR_INVALID;                            // padMerkleProof assert, types.nim:29
  if (mds->dsizes.size() - 1 > (size_t)c.max_log2_nslots) return CP2_ERR_INVALID;  // the same for slotProof
  cp2_ctx* ctx0 = mds->shards[0].units->ctx;
  // (the entropy is a field element, types/bn254.nim:21: the sponge takes any 32 bytes mod r, and cp2_proof_input_create stores and
  // prints the canonical residue -- so what is hashed and what is printed agree without reducing it here)
  const uint8_t* entropy = entropy_in;
  // ---- cellIndices for all pairs at once
  std::vector<uint64_t> idx(total);
  if (total) {
    DeviceRestore restore;
    std::vector<uint8_t> felts(total * 96, 0), dig(total * 32);
    for (size_t i = 0; i < n; ++i)
      for (size_t k = 0; k < ns; ++k) {
        uint8_t* f = &felts[(i * ns + k) * 96];
        std::memcpy(f, entropy, 32);
        std::memcpy(f + 32, mds->slot_root(slots[i]), 32);
        const uint64_t counter = k + 1;
        std::memcpy(f + 64, &counter, 8);
      }
    int st = cp2_sponge2_felts_batch(ctx0, felts.data(), 3, total, dig.data());
    if (st != CP2_OK) { m->err = cp2_last_error(ctx0); return st; }
    for (size_t p = 0; p < total; ++p) {
      uint64_t lo;
      std::memcpy(&lo, &dig[32 * p], 8);                                            // extractLowBits, types/bn254.nim:47-59
      idx[p] = lo & (c.n_cells - 1);
    }
  }
  std::vector<uint8_t> paths(total * md * 32, 0), leaves(total * 32), cells(total * cs);
  // ---- the bottom of every path from the device that holds the unit: one gather per device, devices in parallel
  const size_t world = mds->shards.size();
  std::vector<std::vector<size_t>> by_shard(world);                                 // pair indices per owning shard
  for (size_t p = 0; p < total; ++p) {
    const uint64_t unit = slots[p / ns] * S + idx[p] / P;
    auto* sh = mds->owner(unit);
    if (!sh) return CP2_ERR_INVALID;
    by_shard[(size_t)(sh - mds->shards.data())].push_back(p);
  }
  std::vector<std::string> errs(world);
  int st = for_each_shard(world, [&](size_t w) -> int {
    const auto& mine = by_shard[w];
    if (mine.empty()) return CP2_OK;
    auto& sh = mds->shards[w];
    const size_t k = mine.size();
    std::vector<uint64_t> unit_local(k), cell_local(k);
    for (size_t j = 0; j < k; ++j) {
      const size_t p = mine[j];
      unit_local[j] = slots[p / ns] * S + idx[p] / P - sh.first;
      cell_local[j] = idx[p] % P;
    }
    std::vector<uint8_t> up(k * depth_unit * 32), lf(k * 32);
    int r = trees_paths_multi(sh.units, unit_local.data(), cell_local.data(), k, depth_unit, up.data(), lf.data());
    if (r != CP2_OK) { errs[w] = cp2_last_error(sh.units->ctx); return r; }
    for (size_t j = 0; j < k; ++j) {
      const size_t p = mine[j];
      std::memcpy(&paths[p * md * 32], &up[j * depth_unit * 32], depth_unit * 32);                          // merkleProof inside the unit
      std::memcpy(&leaves[p * 32], &lf[j * 32], 32);
    }
    return CP2_OK;
  });
  if (st != CP2_OK) {
    for (auto& e : errs) if (!e.empty()) { m->err = e; break; }
    return st;
  }
  for (size_t p = 0; p < total; ++p) {                                              // ... and above the unit, from the upper layers
    const uint64_t slot = slots[p / ns], q = idx[p] / P;
    for (size_t lv = 0; lv < levels; ++lv) {
      const uint64_t sib = (q >> lv) ^ 1;
      std::memcpy(&paths[(p * md + depth_unit + lv) * 32], &mds->upper[(mds->upper_off[lv] + slot * (S >> lv) + sib) * 32], 32);
    }
  }
  // ---- the sampled cells (slot.nim:57-73)
  if (c.file_base && total) {
    const int nt = (int)std::min<size_t>((size_t)std::max(1, threads), n);
    std::vector<std::string> failed(nt);
    auto work = [&](int t) {
      for (size_t i = (size_t)t; i < n; i += (size_t)nt) {
        const std::string fname = slot_file_name(mds->file_base, slots[i]);
        const int fd = open(fname.c_str(), O_RDONLY);
        if (fd < 0) { failed[t] = fname; return; }
        for (size_t k = 0; k < ns; ++k) read_file_cell(fd, cs, idx[i * ns + k], &cells[(i * ns + k) * cs]);
        close(fd);
      }
    };
    {
      Workers pool(nt > 1 ? nt - 1 : 1);
      for (int t = 1; t < nt; ++t) pool.submit([&work, t] { work(t); });
      work(0);
      pool.wait_idle();
    }
    for (auto& f : failed) if (!f.empty()) { m->err = "cannot open " + f; return CP2_ERR_IO; }
  } else if (total) {   // genFakeCell for every sampled index in one launch: the list form over "global cells" slot * nCells + cell, seed of slot 0
    DeviceRestore restore;
    CP2_HIP(ctx0, hipSetDevice(ctx0->device));
    std::vector<uint64_t> list(total);
    for (size_t p = 0; p < total; ++p) list[p] = slots[p / ns] * c.n_cells + idx[p];
    DevBuf d_idx, d_cells;
    CP2_TRY(d_idx.scratch(ctx0, total * 8));
    CP2_TRY(d_cells.scratch(ctx0, total * cs));
    CP2_HIP(ctx0, hipMemcpyAsync(d_idx.p, list.data(), total * 8, hipMemcpyHostToDevice, ctx0->stream));
    CP2_HIP(ctx0, cp2k::launch_gen_fake_cells(cp2_slot_seed(c.seed, 0), c.n_cells, 0, static_cast<const uint64_t*>(d_idx.p), total, cs, d_cells.p, ctx0->stream));
    CP2_HIP(ctx0, hipMemcpyAsync(cells.data(), d_cells.p, total * cs, hipMemcpyDeviceToHost, ctx0->stream));
    CP2_HIP(ctx0, hipStreamSynchronize(ctx0->stream));
  }
  // ---- slotProof = padMerkleProof(merkleProof(dsetTree, slotIdx), maxLog2NSlots), gen_input/bn254.nim:51,72; the objects
  std::vector<uint8_t> proof((size_t)c.max_log2_nslots * 32);
  for (size_t i = 0; i < n; ++i) {
    std::fill(proof.begin(), proof.end(), 0);
    size_t k = slots[i], mm = c.n_slots, off = 0;
    for (size_t l = 0; l + 1 < mds->dsizes.size(); ++l) {
      const size_t j = k ^ 1;
      if (j < mm) std::memcpy(&proof[l * 32], &mds->dlayers[(off + j) * 32], 32);
      off += mds->dsizes[l];
      k >>= 1;
      mm = (mm + 1) >> 1;
    }
    int r = cp2_proof_input_create(&c, slots[i], &mds->dlayers[mds->dlayers.size() - 32], entropy, mds->slot_root(slots[i]), proof.data(), ns, &idx[i * ns],
                                   &cells[i * ns * cs], &paths[i * ns * md * 32], &leaves[i * ns * 32], out + i);
    if (r != CP2_OK) {
      for (size_t j = 0; j < i; ++j) { cp2_proof_input_free(out[j]); out[j] = nullptr; }
      return r;
    }
  }
  return CP2_OK;
}

// the texts (and, with a directory, the files) of many slots of a dataset cut by units: batches of `batch` slots through
// units_proof_inputs, formatted on `threads` host threads (cp2_proof_inputs_write_json_batch); keep != nullptr: the texts are kept
int units_export(cp2_multi_dataset* mds, const uint64_t* slot_idx, size_t n, const uint8_t entropy[32], const char* dir, int threads, size_t batch,
                 uint64_t* total_bytes, std::vector<std::string>* keep) {
  if (batch == 0) batch = 256;
  if (threads < 1) threads = 1;
  uint64_t tot = 0;
  for (size_t b0 = 0; b0 < n; b0 += batch) {
    const size_t k = std::min(batch, n - b0);
    std::vector<cp2_proof_input*> ps(k, nullptr);
    struct Release { std::vector<cp2_proof_input*>& v; ~Release() { for (auto* p : v) cp2_proof_input_free(p); } } rel{ps};
    CP2_TRY(units_proof_inputs(mds, slot_idx + b0, k, entropy, threads, ps.data()));
    if (keep) {
      for (size_t i = 0; i < k; ++i) {
        char* text = nullptr;
        size_t len = 0;
        CP2_TRY(cp2_proof_input_json(ps[i], &text, &len));
        (*keep)[slot_idx[b0 + i]].assign(text, len);
        cp2_free_buffer(text);
        tot += len;
      }
      continue;
    }
    std::vector<std::string> names;
    std::vector<const char*> paths;
    if (dir) {
      for (size_t i = 0; i < k; ++i) names.push_back(std::string(dir) + "/input_" + std::to_string(slot_idx[b0 + i]) + ".json");
      for (auto& s2 : names) paths.push_back(s2.c_str());
    }
    uint64_t got = 0;
    int st = cp2_proof_inputs_write_json_batch(ps.data(), k, dir ? paths.data() : nullptr, threads, &got);
    if (st != CP2_OK) { if (st == CP2_ERR_IO) mds->m->err = std::string("cannot write into ") + (dir ? dir : "?"); return st; }
    tot += got;
  }
  if (total_bytes) *total_bytes = tot;
  return CP2_OK;
}

}  // namespace

// the plan cp2_multi_dataset_build would follow, without a device (host-only arithmetic)
extern "C" int cp2_multi_plan(const cp2_config* cfg, int n_devices, uint64_t min_cells_per_device, int64_t units_per_slot, int* n_shards,
                              uint64_t* units_per_slot_out) try {
  if (!cfg || n_devices < 1 || cfg->n_slots == 0 || cfg->n_cells == 0 || units_per_slot < 0 ||
      (units_per_slot > 1 && !is_pow2((uint64_t)units_per_slot))) return CP2_ERR_INVALID;
  uint64_t world = 1, S = 1;
  plan_shards(*cfg, (size_t)n_devices, min_cells_per_device, units_per_slot, &world, &S);
  if (n_shards) *n_shards = (int)world;
  if (units_per_slot_out) *units_per_slot_out = S;
  return CP2_OK;
} catch (...) {
  return CP2_ERR_INVALID;
}

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
extern "C" uint64_t cp2_multi_dataset_units_per_slot(const cp2_multi_dataset* mds) { return mds ? mds->units_per_slot : 0; }

extern "C" cp2_dataset* cp2_multi_dataset_shard(cp2_multi_dataset* mds, int i, int* device, uint64_t* first, uint64_t* count) {
  if (!mds || i < 0 || i >= (int)mds->shards.size()) return nullptr;
  const auto& s = mds->shards[i];
  if (device) *device = mds->m->devices[s.dev];
  if (first) *first = s.first;
  if (count) *count = s.count;
  return s.ds;
}

extern "C" int cp2_multi_dataset_root(cp2_multi_dataset* mds, uint8_t out[32]) try {
  if (!mds || !out || mds->shards.empty()) return CP2_ERR_INVALID;
  if (mds->by_units()) { std::memcpy(out, &mds->dlayers[mds->dlayers.size() - 32], 32); return CP2_OK; }
  return cp2_dataset_root(mds->shards[0].ds, out);
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" int cp2_multi_dataset_slot_roots(cp2_multi_dataset* mds, uint8_t* out) try {
  if (!mds || !out) return CP2_ERR_INVALID;
  if (mds->by_units()) { std::memcpy(out, mds->slot_root(0), mds->cfg.n_slots * 32); return CP2_OK; }
  for (auto& s : mds->shards) CP2_TRY(cp2_dataset_local_roots(s.ds, out + s.first * 32));
  return CP2_OK;
} catch (...) {
  return CP2_ERR_INVALID;
}

// generateProofInputBN254 (gen_input/bn254.nim:35-79) on the device that holds the slot (or, by units, from every device that
// holds one of its sampled cells)
extern "C" int cp2_multi_proof_input_generate(cp2_multi_dataset* mds, uint64_t slot_idx, const uint8_t entropy[32], cp2_proof_input** out) try {
  if (!mds || !out || !entropy) return CP2_ERR_INVALID;
  *out = nullptr;
  if (mds->by_units()) return units_proof_inputs(mds, &slot_idx, 1, entropy, 1, out);
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
  if (mds->by_units()) return units_export(mds, slot_idx, n, entropy, dir, threads, batch, total_bytes, nullptr);   // batched per device, devices in parallel
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
  if (mds->by_units()) {                                             // two-phase streamed build: the texts are there
    if (!mds->prepared) return CP2_ERR_INVALID;
    uint64_t tot = 0;
    for (size_t s = 0; s < mds->texts.size(); ++s) {
      tot += mds->texts[s].size();
      if (!dir) continue;
      const std::string name = std::string(dir) + "/input_" + std::to_string(s) + ".json";
      FILE* f = std::fopen(name.c_str(), "wb");
      const bool ok = f && std::fwrite(mds->texts[s].data(), 1, mds->texts[s].size(), f) == mds->texts[s].size();
      if ((f && std::fclose(f) != 0) || !ok) { mds->m->err = "cannot write " + name; return CP2_ERR_IO; }
    }
    if (total_bytes) *total_bytes = tot;
    return CP2_OK;
  }
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
  if (mds->by_units()) {
    if (!mds->prepared || slot_idx >= mds->texts.size()) return CP2_ERR_INVALID;
    const std::string& t = mds->texts[slot_idx];
    char* buf = (char*)std::malloc(t.size() + 1);
    if (!buf) return CP2_ERR_ALLOC;
    std::memcpy(buf, t.data(), t.size());
    buf[t.size()] = 0;
    *text = buf;
    if (len) *len = t.size();
    return CP2_OK;
  }
  auto* s = mds->owner(slot_idx);
  if (!s) return CP2_ERR_INVALID;
  return cp2_dataset_streamed_json(s->ds, slot_idx, text, len);
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

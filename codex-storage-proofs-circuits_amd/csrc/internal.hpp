// Internals shared by the C-ABI translation units (not installed).
#pragma once
#include <hip/hip_runtime.h>

#include <sys/resource.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <condition_variable>
#include <cstdint>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/codex_p2.h"
#include "workers.hpp"

namespace cp2i {

// ---- the environment, parsed strictly (environment.cpp lives in codex_p2_abi.cpp) ----------------------------------------------
// Every CODEX_P2_* variable either holds exactly what it takes or the call that reads it fails with CP2_ERR_INVALID and a message
// that names the variable: a mistyped knob must never silently mean "automatic" (cp2_check_environment, include/codex_p2.h).
//   env_decimal  unset / empty: true, *set = false.  A plain decimal number of at most 18 digits: true, *set = true.  Else false.
bool env_decimal(const char* name, uint64_t* value, bool* set);
// CODEX_P2_KEEP_TREES: "0" / "1" / "2", or "auto" (= unset).  false when it holds anything else.  *mode = -1 when automatic.
bool env_keep_trees(int* mode);
// CODEX_P2_MEM_LIMIT_MB (tests, rehearsals): a cap on the device memory this PROCESS may hold per device through the library's own
// allocations, in MiB; 0 / unset = none.  The automatic residency choice sees min(what the device has free, what the cap leaves),
// and an allocation that would exceed the cap fails like a real out-of-memory -- so the three residency modes and the fallback
// between them can be reached on a device with 288 GB free.
size_t mem_limit_bytes();
// device allocations of this process through the library, per HIP device (for the cap and for the residency estimate)
hipError_t dev_malloc(void** p, size_t n);
void dev_free(void* p, size_t n);
size_t dev_bytes_held();            // on the current device
// what the CURRENT device has free for this process right now: hipMemGetInfo, and no more than the cap leaves
int device_free_bytes(size_t* out);

// Per-context cache of device and pinned-host scratch blocks.  The host-pointer entry points (one hipMalloc +
// hipFree pair per call before) and the staging buffers of the builders draw from it, so a context that is
// called repeatedly stops allocating after its first calls.  Blocks are handed back only when no work that
// touches them is in flight (DevBuf / PinBuf synchronise the context's stream first).  Thread-safe.
class BlockPool {
 public:
  static constexpr size_t MAX_CACHED_DEV = (size_t)6 << 30, MAX_CACHED_PIN = (size_t)4 << 30;
  ~BlockPool() { trim(); }
  // returns nullptr on failure; *got = usable size (>= n)
  void* get(bool pinned, size_t n, size_t* got) {
    if (n == 0) n = 16;
    {
      std::lock_guard<std::mutex> lk(mu_);
      auto& fl = pinned ? pin_ : dev_;
      size_t best = fl.size();
      for (size_t i = 0; i < fl.size(); ++i)
        if (fl[i].bytes >= n && fl[i].bytes / 4 <= n && (best == fl.size() || fl[i].bytes < fl[best].bytes)) best = i;
      if (best != fl.size()) {
        Blk b = fl[best];
        fl.erase(fl.begin() + (long)best);
        (pinned ? cached_pin_ : cached_dev_) -= b.bytes;
        *got = b.bytes;
        return b.p;
      }
    }
    size_t want = round_up(n);
    void* p = nullptr;
    hipError_t e = pinned ? hipHostMalloc(&p, want, hipHostMallocDefault) : dev_malloc(&p, want);
    if (e != hipSuccess) {   // make room and try the exact size once
      (void)hipGetLastError();
      trim();
      want = n;
      e = pinned ? hipHostMalloc(&p, want, hipHostMallocDefault) : dev_malloc(&p, want);
      if (e != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    }
    *got = want;
    return p;
  }
  void put(bool pinned, void* p, size_t bytes) {
    if (!p) return;
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (dead_) return;
      size_t& cached = pinned ? cached_pin_ : cached_dev_;
      if (cached + bytes <= (pinned ? MAX_CACHED_PIN : MAX_CACHED_DEV)) {
        (pinned ? pin_ : dev_).push_back({p, bytes});
        cached += bytes;
        return;
      }
    }
    if (pinned) (void)hipHostFree(p); else dev_free(p, bytes);
  }
  // the owning context's stream will not drain (cp2_ctx::stuck): cached blocks are forgotten, not freed (freeing device memory
  // waits for the device), and blocks that come back later are dropped too
  void abandon() {
    std::lock_guard<std::mutex> lk(mu_);
    dead_ = true;
    dev_.clear(); pin_.clear();
    cached_dev_ = cached_pin_ = 0;
  }
  void trim() {
    std::vector<Blk> d, h;
    {
      std::lock_guard<std::mutex> lk(mu_);
      d.swap(dev_); h.swap(pin_);
      cached_dev_ = cached_pin_ = 0;
    }
    for (auto& b : d) dev_free(b.p, b.bytes);
    for (auto& b : h) (void)hipHostFree(b.p);
  }

 private:
  struct Blk { void* p; size_t bytes; };
  static size_t round_up(size_t n) {   // 64 KiB granules below 1 MiB, then 1/8-octave steps: sizes that recur hit the cache
    if (n <= ((size_t)1 << 20)) return (n + 0xffff) & ~(size_t)0xffff;
    size_t step = (size_t)1 << 17;
    while (step * 16 < n) step <<= 1;
    return (n + step - 1) / step * step;
  }
  std::mutex mu_;
  std::vector<Blk> dev_, pin_;
  size_t cached_dev_ = 0, cached_pin_ = 0;
  bool dead_ = false;
};

}  // namespace cp2i

struct cp2_ctx {
  int device = 0;
  bool native = false;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipStream_t aux_stream = nullptr;   // second hashing stream: chunks alternate between `stream` and this one (created on first use)
  hipStream_t aux2_stream = nullptr;  // third stream: the layer passes of groups / pipelined batches, sampling, gathers and downloads of the streamed proof-input path (created on first use)
  std::shared_ptr<cp2i::BlockPool> pool = std::make_shared<cp2i::BlockPool>();
  size_t stage_bytes = (size_t)1 << 31;   // device staging chunk of the fake-data builder
  int ingest_threads = 0, ingest_ring = 0;   // 0: CP2_INGEST_* environment or the built-in default (cp2_set_ingest)
  size_t ingest_chunk = 0;
  int ingest_mapped = -1;                    // SlotFile chunks that sit in the page cache uploaded straight from a mapping of the file (no CPU copy): 1 on, 0 off, -1 = environment CP2_INGEST_MAPPED, default off (cp2_set_ingest_mapped)
  int ingest_direct = -1;                    // SlotFile reads with O_DIRECT: 1 on, 0 off, -1 = environment CP2_INGEST_DIRECT (cp2_set_ingest_direct)
  size_t body_budget = 0;                    // streamed proof-input bodies kept in host memory; 0: CP2_BODY_BUDGET_MB or 4 GiB (cp2_set_body_budget)
  size_t mem_allowance = 0;                  // device bytes this context may plan with in its automatic residency choice; 0: ask the device.  cp2_multi sets it for the duration of a build: what the device had free BEFORE its shards started, divided by the number of contexts placed on that device
  bool stuck = false;                        // work that will not complete is queued on this context's stream (a multi-device exchange timed out): nothing waits
                                             // for its streams any more -- pooled buffers are dropped from the books, builders refuse, cp2_free does not drain
  bool hash_room = false;                    // this device takes k_hash_cells launches that leave a third of every CU free (decided once, by cp2_init: kernels.hip)
  int keep_trees = -1;                       // what cp2_dataset_build keeps of the slot trees in device memory: 1 every node, 2 block roots and up, 0 roots only, -1 = CODEX_P2_KEEP_TREES or the most that fits (cp2_set_keep_trees)
  std::string spill_dir;                     // where bodies beyond the budget go; empty: $TMPDIR or /tmp
  std::string err;
};

namespace cp2i {

#define CP2_HIP(ctx, call)                                                                        \
  do {                                                                                            \
    hipError_t e__ = (call);                                                                      \
    if (e__ != hipSuccess) {                                                                      \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                            \
      return CP2_ERR_HIP;                                                                         \
    }                                                                                             \
  } while (0)

// builders and synchronising entry points refuse a context whose stream will not drain (cp2_ctx::stuck) instead of waiting on it
#define CP2_REFUSE_STUCK(ctx)                                                                                                  \
  do {                                                                                                                         \
    if ((ctx)->stuck) {                                                                                                        \
      (ctx)->err = "this context takes no further work: a multi-device exchange timed out with work still queued on its stream"; \
      return CP2_ERR_HIP;                                                                                                      \
    }                                                                                                                          \
  } while (0)

#define CP2_TRY(call)                 \
  do {                                \
    int s__ = (call);                 \
    if (s__ != CP2_OK) return s__;    \
  } while (0)

// RAII device buffer.  alloc() = plain hipMalloc (long-lived: tree nodes); scratch() = from the context's
// pool (staging / temporaries).  A pooled block goes back only after the context's stream has drained.
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  cp2_ctx* owner = nullptr;   // non-null: pooled
  cp2_ctx* home = nullptr;    // alloc(): the context the block was allocated for (a block of a stuck context is dropped, not freed: hipFree waits for the device)
  bool borrowed = false;      // a view of memory someone else owns (a batch's nodes inside a pipeline's scratch): never freed from here
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void borrow(void* q, size_t n) {
    release();
    p = q;
    bytes = n;
    borrowed = true;
  }
  void release() {
    if (p && !borrowed) {
      if (owner && owner->stuck) {
        // the context's stream will not drain: the block may still be touched by what is queued there -- it is dropped, not recycled
      } else if (owner) {
        (void)hipStreamSynchronize(owner->stream);
        if (owner->aux_stream) (void)hipStreamSynchronize(owner->aux_stream);
        if (owner->aux2_stream) (void)hipStreamSynchronize(owner->aux2_stream);
        owner->pool->put(false, p, bytes);
      } else if (!(home && home->stuck)) {
        dev_free(p, bytes);
      }
    }
    p = nullptr;
    bytes = 0;
    owner = nullptr;
    home = nullptr;
    borrowed = false;
  }
  int alloc(cp2_ctx* ctx, size_t n) {
    release();
    if (n == 0) n = 16;
    hipError_t e = dev_malloc(&p, n);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      ctx->pool->trim();        // cached scratch may be what is in the way
      e = dev_malloc(&p, n);
    }
    if (e != hipSuccess) {
      (void)hipGetLastError();   // reported here; must not resurface from a later hipGetLastError()
      p = nullptr;
      ctx->err = std::string("hipMalloc(") + std::to_string(n) + "): " + hipGetErrorString(e);
      return CP2_ERR_ALLOC;
    }
    bytes = n;
    home = ctx;
    return CP2_OK;
  }
  int scratch(cp2_ctx* ctx, size_t n) {
    release();
    size_t got = 0;
    p = ctx->pool->get(false, n, &got);
    if (!p) {
      ctx->err = std::string("device scratch of ") + std::to_string(n) + " bytes: allocation failed";
      return CP2_ERR_ALLOC;
    }
    bytes = got;
    owner = ctx;
    return CP2_OK;
  }
  uint8_t* u8() const { return static_cast<uint8_t*>(p); }
};

// RAII pinned host buffer from the context's pool (the pool outlives the context through the shared_ptr, so
// proof inputs that still reference their batch storage stay valid after cp2_free).
struct PinBuf {
  void* p = nullptr;
  size_t bytes = 0;
  std::shared_ptr<BlockPool> pool;
  PinBuf() = default;
  PinBuf(const PinBuf&) = delete;
  PinBuf& operator=(const PinBuf&) = delete;
  ~PinBuf() { release(); }
  void release() {
    if (p && pool) pool->put(true, p, bytes);
    p = nullptr;
    bytes = 0;
  }
  int alloc(cp2_ctx* ctx, size_t n) {
    release();
    pool = ctx->pool;
    size_t got = 0;
    p = pool->get(true, n, &got);
    if (!p) {
      ctx->err = std::string("pinned host buffer of ") + std::to_string(n) + " bytes: allocation failed";
      return CP2_ERR_ALLOC;
    }
    bytes = got;
    return CP2_OK;
  }
  void swap(PinBuf& o) {
    std::swap(p, o.p);
    std::swap(bytes, o.bytes);
    std::swap(pool, o.pool);
  }
  uint8_t* u8() const { return static_cast<uint8_t*>(p); }
};

// element counts of all layers of a tree over n leaves, bottom first (merkle/bn254.nim:29-58):
// the bottom layer always gets one round of compression, even for a singleton.
inline std::vector<size_t> layer_sizes_of(size_t n) {
  std::vector<size_t> s;
  if (n == 0) return s;
  size_t m = n;
  bool bottom = true;
  for (;;) {
    s.push_back(m);
    if (m == 1 && !bottom) break;
    m = (m + 1) / 2;
    bottom = false;
  }
  return s;
}

int merkle_trees_dev(cp2_ctx* ctx, const void* d_leaves, size_t n, size_t nseg, void* d_layers_out, bool leaves_in_place);
// hash n host-resident cells into d_leaves (device, n x 32 bytes) through the pinned ingestion pipe
int hash_host_cells_pipelined(cp2_ctx* ctx, const uint8_t* cells, size_t cell_size, size_t n, uint8_t* d_leaves);
// the context's second (which = 1) or third (which = 2) stream, created on first use
int aux_stream(cp2_ctx* ctx, hipStream_t* out, int which = 1);

}  // namespace cp2i

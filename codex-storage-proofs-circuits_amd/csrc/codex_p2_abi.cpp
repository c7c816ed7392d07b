// C ABI of libcodex_p2.so (include/codex_p2.h): contexts, device memory, kernel sequencing.
// No CPU fallback lives here: every hash goes through the HIP kernels of kernels.hip.
#include "../../include/codex_p2.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "trees.hpp"

using namespace cp2i;

// ---------------------------------------------------------------------------------------------
// the environment, parsed strictly; device-memory accounting (internal.hpp)
// ---------------------------------------------------------------------------------------------
namespace cp2i {

bool env_decimal(const char* name, uint64_t* value, bool* set) {
  if (set) *set = false;
  const char* e = std::getenv(name);
  if (!e || !*e) return true;
  const std::string t(e);
  if (t.size() > 18 || t.find_first_not_of("0123456789") != std::string::npos) return false;
  if (value) *value = std::strtoull(e, nullptr, 10);
  if (set) *set = true;
  return true;
}

bool env_keep_trees(int* mode) {
  if (mode) *mode = -1;
  const char* e = std::getenv("CODEX_P2_KEEP_TREES");
  if (!e || !*e || std::strcmp(e, "auto") == 0) return true;
  if ((e[0] == '0' || e[0] == '1' || e[0] == '2') && e[1] == 0) {
    if (mode) *mode = e[0] - '0';
    return true;
  }
  return false;
}

size_t mem_limit_bytes() {
  uint64_t mb = 0;
  bool set = false;
  if (!env_decimal("CODEX_P2_MEM_LIMIT_MB", &mb, &set) || !set) return 0;   // (a malformed value is refused by cp2_check_environment before any build)
  return (size_t)mb << 20;
}

namespace {
struct DevLedger {
  std::mutex mu;
  std::unordered_map<void*, std::pair<int, size_t>> blocks;   // pointer -> (device, bytes)
  size_t held[64] = {};
};
DevLedger& ledger() {
  static DevLedger* l = new DevLedger();   // never destroyed: frees may still arrive while the process shuts down
  return *l;
}
}  // namespace

hipError_t dev_malloc(void** p, size_t n) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
  dev &= 63;
  const size_t limit = mem_limit_bytes();
  DevLedger& l = ledger();
  if (limit) {
    std::lock_guard<std::mutex> lk(l.mu);
    if (l.held[dev] + n > limit) return hipErrorOutOfMemory;      // the cap behaves like the device running out
  }
  hipError_t e = hipMalloc(p, n);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lk(l.mu);
  l.blocks[*p] = {dev, n};
  l.held[dev] += n;
  return hipSuccess;
}

void dev_free(void* p, size_t) {
  if (!p) return;
  {
    DevLedger& l = ledger();
    std::lock_guard<std::mutex> lk(l.mu);
    auto it = l.blocks.find(p);
    if (it != l.blocks.end()) {
      l.held[it->second.first] -= it->second.second;
      l.blocks.erase(it);
    }
  }
  (void)hipFree(p);
}

int device_free_bytes(size_t* out) {
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return CP2_ERR_HIP; }
  if (const size_t limit = mem_limit_bytes()) {
    const size_t held = dev_bytes_held();
    free_b = std::min(free_b, limit > held ? limit - held : 0);
  }
  *out = free_b;
  return CP2_OK;
}

size_t dev_bytes_held() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
  DevLedger& l = ledger();
  std::lock_guard<std::mutex> lk(l.mu);
  return l.held[dev & 63];
}

}  // namespace cp2i

extern "C" int cp2_abi_version(void) { return CP2_ABI_VERSION; }   // what THIS library was built from (include/codex_p2.h)

// Host-only: every CODEX_P2_* variable holds what it takes, or the first one that does not is named in `msg`.
extern "C" int cp2_check_environment(char* msg, size_t msg_len) try {
  auto fail = [&](const char* var, const char* takes) {
    if (msg && msg_len) std::snprintf(msg, msg_len, "%s=\"%s\" is not what the variable takes: %s", var, std::getenv(var) ? std::getenv(var) : "", takes);
    return CP2_ERR_INVALID;
  };
  if (msg && msg_len) msg[0] = 0;
  if (const char* e = std::getenv("CODEX_P2_GPUS")) {
    const std::string s(e);
    bool ok = !s.empty();
    if (s == "all") {
      ok = true;
    } else if (s.find(',') == std::string::npos) {
      ok = ok && s.size() <= 6 && s.find_first_not_of("0123456789") == std::string::npos && std::strtol(s.c_str(), nullptr, 10) >= 1;
    } else {
      size_t at = 0, n = 0;
      while (ok && at < s.size()) {
        size_t c = s.find(',', at);
        if (c == std::string::npos) c = s.size();
        if (c > at) {
          const std::string t = s.substr(at, c - at);
          ok = t.size() <= 6 && t.find_first_not_of("0123456789") == std::string::npos;
          ++n;
        }
        at = c + 1;
      }
      ok = ok && n > 0;
    }
    if (*e && !ok) return fail("CODEX_P2_GPUS", "\"all\", a device count (\"4\") or a comma-separated list of device indices (\"0,2,3\"; \"2,\" = device 2 only)");
  }
  uint64_t v = 0;
  bool set = false;
  if (!env_decimal("CODEX_P2_MIN_CELLS", &v, &set)) return fail("CODEX_P2_MIN_CELLS", "a decimal number of cells");
  if (!env_decimal("CODEX_P2_SPLIT", &v, &set) || (set && v > 1 && (v & (v - 1)))) return fail("CODEX_P2_SPLIT", "0 (choose), 1 (whole slots) or a power of two (units per slot)");
  if (!env_decimal("CODEX_P2_MEM_LIMIT_MB", &v, &set)) return fail("CODEX_P2_MEM_LIMIT_MB", "a decimal number of MiB (0 = no cap)");
  if (!env_decimal("CODEX_P2_EXCHANGE_TIMEOUT_S", &v, &set)) return fail("CODEX_P2_EXCHANGE_TIMEOUT_S", "a decimal number of seconds (0 = wait for ever)");
  if (!env_decimal("CODEX_P2_TEST_LDS_LIMIT", &v, &set)) return fail("CODEX_P2_TEST_LDS_LIMIT", "a decimal number of bytes (test hook: the LDS per workgroup cp2_init's launch-shape decision sees)");
  if (!env_decimal("CODEX_P2_STAGE_MB", &v, &set) || (set && (v < 1 || v > 65536))) return fail("CODEX_P2_STAGE_MB", "a decimal number of MiB between 1 and 65536");
  if (const char* e = std::getenv("CODEX_P2_GATHER"))
    if (*e && std::strcmp(e, "auto") && std::strcmp(e, "rccl") && std::strcmp(e, "host") && std::strcmp(e, "copy"))
      return fail("CODEX_P2_GATHER", "\"auto\", \"rccl\", \"copy\" or \"host\"");
  int mode = -1;
  if (!env_keep_trees(&mode)) return fail("CODEX_P2_KEEP_TREES", "\"auto\", \"1\" (every node), \"2\" (compact) or \"0\" (roots only)");
  return CP2_OK;
} catch (...) {
  return CP2_ERR_INVALID;
}

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
extern "C" int cp2_init(int device, cp2_ctx** out) try {
  if (!out) return CP2_ERR_INVALID;
  *out = nullptr;
  if (cp2_check_environment(nullptr, 0) != CP2_OK) return CP2_ERR_INVALID;   // a mistyped CODEX_P2_* knob is refused, never read as "automatic"
  StageTimer trace;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return CP2_ERR_NO_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return CP2_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return CP2_ERR_NO_DEVICE;
  trace.lap("context: HIP runtime + device query");
  cp2_ctx* c = new (std::nothrow) cp2_ctx();
  if (!c) return CP2_ERR_ALLOC;
  c->device = device;
  c->native = std::strncmp(prop.gcnArchName, "gfx950", 6) == 0;
  if (!c->native) {   // the code object only holds gfx950 ISA: fail loudly instead of faulting at launch
    delete c;
    return CP2_ERR_NO_DEVICE;
  }
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return CP2_ERR_HIP;
  }
  c->stream = c->own_stream;
  {   // CODEX_P2_STAGE_MB: the device staging chunk of the fake-data builder (default 2048); a transient batch holds half of it in nodes
    uint64_t mb = 0;
    bool set = false;
    if (env_decimal("CODEX_P2_STAGE_MB", &mb, &set) && set) c->stage_bytes = (size_t)mb << 20;
  }
  trace.lap("context: stream");
  {   // Can the hash kernel be launched two workgroups to a CU (the streamed builds' co-residency, kernels.hip)?  Decided here, once;
      // CODEX_P2_TEST_LDS_LIMIT (test hook) caps the LDS per workgroup the decision sees, so that the "no" branch can be reached on gfx950.
    uint64_t cap = 0;
    bool set = false;
    (void)env_decimal("CODEX_P2_TEST_LDS_LIMIT", &cap, &set);
    std::string why;
    c->hash_room = cp2k::hash_cells_can_leave_room(set ? (size_t)cap : 0, &why);
    if (trace.on) {
      std::fprintf(stderr, "[cp2 trace] context: hash launches beside the streamed builds' small kernels %s (%s)\n", c->hash_room ? "leave room: two workgroups per CU" : "hold every workgroup slot: no room", why.c_str());
      trace.lap("context: code object load + launch shape");
    }
  }
  *out = c;
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" void cp2_free(cp2_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stuck) {        // its stream will not drain: nothing is waited for, nothing of it is handed back to the device (a stated leak)
    ctx->pool->abandon();
    delete ctx;
    return;
  }
  (void)hipStreamSynchronize(ctx->stream);
  for (hipStream_t st : {ctx->aux_stream, ctx->aux2_stream})
    if (st) {
      (void)hipStreamSynchronize(st);
      (void)hipStreamDestroy(st);
    }
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  ctx->pool->trim();   // cached scratch goes now; pinned blocks still held by live proof inputs return to the pool later
  delete ctx;
}

extern "C" int cp2_set_stream(cp2_ctx* ctx, void* hip_stream) try {
  if (!ctx) return CP2_ERR_INVALID;
  ctx->stream = (hipStream_t)hip_stream;
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" int cp2_reset_stream(cp2_ctx* ctx) try {
  if (!ctx) return CP2_ERR_INVALID;
  ctx->stream = ctx->own_stream;
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" int cp2_sync(cp2_ctx* ctx) try {
  if (!ctx) return CP2_ERR_INVALID;
  CP2_REFUSE_STUCK(ctx);
  CP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" int cp2_trim(cp2_ctx* ctx) try {
  if (!ctx) return CP2_ERR_INVALID;
  CP2_REFUSE_STUCK(ctx);
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  CP2_HIP(ctx, hipStreamSynchronize(ctx->stream));            // nothing in flight may still touch a cached block
  if (ctx->aux_stream) CP2_HIP(ctx, hipStreamSynchronize(ctx->aux_stream));
  if (ctx->aux2_stream) CP2_HIP(ctx, hipStreamSynchronize(ctx->aux2_stream));
  ctx->pool->trim();
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" int cp2_set_keep_trees(cp2_ctx* ctx, int mode) try {
  if (!ctx || mode < -1 || mode > 2) return CP2_ERR_INVALID;
  ctx->keep_trees = mode;
  return CP2_OK;
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" int cp2_set_body_budget(cp2_ctx* ctx, size_t max_resident_bytes, const char* spill_dir) try {
  if (!ctx) return CP2_ERR_INVALID;
  if (max_resident_bytes) ctx->body_budget = max_resident_bytes;
  ctx->spill_dir = spill_dir ? spill_dir : "";
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" const char* cp2_strerror(int status) {
  switch (status) {
    case CP2_OK: return "ok";
    case CP2_ERR_INVALID: return "invalid argument";
    case CP2_ERR_NO_DEVICE: return "no usable gfx950 HIP device";
    case CP2_ERR_HIP: return "HIP runtime error";
    case CP2_ERR_ALLOC: return "allocation failed";
    case CP2_ERR_IO: return "I/O error";
    case CP2_ERR_ALIGN: return "device pointer not 16-byte aligned";
    default: return "unknown status";
  }
}

extern "C" const char* cp2_last_error(const cp2_ctx* ctx) { return ctx ? ctx->err.c_str() : "no context"; }
extern "C" int cp2_device_is_native(const cp2_ctx* ctx) { return ctx && ctx->native ? 1 : 0; }

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---------------------------------------------------------------------------------------------
// a1 permutation
// ---------------------------------------------------------------------------------------------
extern "C" int cp2_permute_batch_dev(cp2_ctx* ctx, const void* d_in, void* d_out, size_t n) try {
  if (!ctx || (n && (!d_in || !d_out))) return CP2_ERR_INVALID;
  if (!aligned16(d_in) || !aligned16(d_out)) return CP2_ERR_ALIGN;
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  CP2_HIP(ctx, cp2k::launch_permute_batch(d_in, d_out, n, ctx->stream));
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

// Host arrays through the GPU in chunks: out[i] = kernel(in[i]) for n items.  Three stages on three streams over a
// 4-deep ring of pinned + device buffers -- upload of chunk i+1, kernel of chunk i and download of chunk i-1 overlap
// (PCIe is full duplex), and a few host threads copy between the caller's pageable arrays and the pinned ring.
// Arrays the caller has pinned (hipHostMalloc, hipHostRegister) skip the ring and the copies: the copy engines work on them
// in place.  Small inputs take one upload / launch / download on the context's stream with pooled scratch.
namespace {
// is p..p+bytes host memory the runtime can DMA from / to directly (hipHostMalloc'ed or hipHostRegister'ed by the caller)?
static bool host_pinned(const void* p) {
  hipPointerAttribute_t a{};
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeHost;
}
// ... the whole range: probed at the start of every `step` bytes (one chunk of the direct path) and at its last byte, so an array
// stitched together from several registrations with a pageable hole between them is not taken for pinned on the strength of its
// two ends (a hole would still be copied correctly -- the runtime stages pageable memory itself -- only synchronously and slowly)
static bool host_range_pinned(const uint8_t* p, size_t bytes, size_t step) {
  if (bytes == 0) return false;
  for (size_t at = 0; at < bytes; at += step)
    if (!host_pinned(p + at)) return false;
  return host_pinned(p + bytes - 1);
}
template <typename Launch>
int stream_map(cp2_ctx* ctx, const uint8_t* in, size_t in_item, uint8_t* out, size_t out_item, size_t n, Launch launch) {
  constexpr int MAX_DEPTH = 4;
  // items per chunk and ring depth: swept on 2^24 states (profiles/r05_host_array_sweep.txt): 2^18 x 4 is the fastest cell of the table at
  // 8 and 16 copy threads (38 ms against 44-48 ms for round 4's 2^20 x 3); deeper rings are slower (more pinned memory in flight
  // for the copy threads and both DMA directions to share)
  constexpr size_t CHUNK = (size_t)1 << 18;
  constexpr int DEPTH = 4;
  if (n <= ((size_t)1 << 20)) {
    DevBuf d_in, d_out;
    CP2_TRY(d_in.scratch(ctx, n * in_item));
    CP2_TRY(d_out.scratch(ctx, n * out_item));
    CP2_HIP(ctx, hipMemcpyAsync(d_in.p, in, n * in_item, hipMemcpyHostToDevice, ctx->stream));
    CP2_HIP(ctx, launch(d_in.p, d_out.p, n, ctx->stream));
    CP2_HIP(ctx, hipMemcpyAsync(out, d_out.p, n * out_item, hipMemcpyDeviceToHost, ctx->stream));
    CP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CP2_OK;
  }
  struct Ring {
    cp2_ctx* ctx;
    hipStream_t up = nullptr, down = nullptr;
    hipEvent_t e_up[MAX_DEPTH] = {}, e_k[MAX_DEPTH] = {}, e_down[MAX_DEPTH] = {};
    PinBuf pin_in[MAX_DEPTH], pin_out[MAX_DEPTH];
    DevBuf d_in[MAX_DEPTH], d_out[MAX_DEPTH];
    ~Ring() {   // drain everything before the buffers go back to the pools
      if (up) (void)hipStreamSynchronize(up);
      (void)hipStreamSynchronize(ctx->stream);
      if (down) (void)hipStreamSynchronize(down);
      for (int r = 0; r < MAX_DEPTH; ++r) {
        if (e_up[r]) (void)hipEventDestroy(e_up[r]);
        if (e_k[r]) (void)hipEventDestroy(e_k[r]);
        if (e_down[r]) (void)hipEventDestroy(e_down[r]);
      }
      if (up) (void)hipStreamDestroy(up);
      if (down) (void)hipStreamDestroy(down);
    }
  } ring{ctx};
  CP2_HIP(ctx, hipStreamCreateWithFlags(&ring.up, hipStreamNonBlocking));
  CP2_HIP(ctx, hipStreamCreateWithFlags(&ring.down, hipStreamNonBlocking));
  const bool trace = std::getenv("CP2_TRACE") != nullptr;
  // the caller's arrays are pinned (hipHostMalloc / hipHostRegister): the copy engines read and write them in place -- no pinned
  // ring, no host thread copies; chunks of 2^20 items, every dependency a device-side event wait, one host wait at the end
  const bool direct = host_range_pinned(in, n * in_item, ((size_t)1 << 20) * in_item) && host_range_pinned(out, n * out_item, ((size_t)1 << 20) * out_item);
  const size_t chunk = direct ? (size_t)1 << 20 : CHUNK;
  for (int r = 0; r < DEPTH; ++r) {
    CP2_HIP(ctx, hipEventCreateWithFlags(&ring.e_up[r], hipEventDisableTiming));
    CP2_HIP(ctx, hipEventCreateWithFlags(&ring.e_k[r], hipEventDisableTiming));
    CP2_HIP(ctx, hipEventCreateWithFlags(&ring.e_down[r], hipEventDisableTiming));
    if (!direct) {
      CP2_TRY(ring.pin_in[r].alloc(ctx, chunk * in_item));
      CP2_TRY(ring.pin_out[r].alloc(ctx, chunk * out_item));
    }
    CP2_TRY(ring.d_in[r].scratch(ctx, chunk * in_item));
    CP2_TRY(ring.d_out[r].scratch(ctx, chunk * out_item));
  }
  if (direct) {
    const size_t n_chunks = (n + chunk - 1) / chunk;
    for (size_t c = 0; c < n_chunks; ++c) {
      const int r = (int)(c % DEPTH);
      const size_t m = std::min(chunk, n - c * chunk);
      if (c >= (size_t)DEPTH) {
        CP2_HIP(ctx, hipStreamWaitEvent(ring.up, ring.e_k[r], 0));            // d_in[r] was read by the kernel of chunk c - DEPTH
        CP2_HIP(ctx, hipStreamWaitEvent(ctx->stream, ring.e_down[r], 0));     // d_out[r] was downloaded
      }
      CP2_HIP(ctx, hipMemcpyAsync(ring.d_in[r].p, in + c * chunk * in_item, m * in_item, hipMemcpyHostToDevice, ring.up));
      CP2_HIP(ctx, hipEventRecord(ring.e_up[r], ring.up));
      CP2_HIP(ctx, hipStreamWaitEvent(ctx->stream, ring.e_up[r], 0));
      CP2_HIP(ctx, launch(ring.d_in[r].p, ring.d_out[r].p, m, ctx->stream));
      CP2_HIP(ctx, hipEventRecord(ring.e_k[r], ctx->stream));
      CP2_HIP(ctx, hipStreamWaitEvent(ring.down, ring.e_k[r], 0));
      CP2_HIP(ctx, hipMemcpyAsync(out + c * chunk * out_item, ring.d_out[r].p, m * out_item, hipMemcpyDeviceToHost, ring.down));
      CP2_HIP(ctx, hipEventRecord(ring.e_down[r], ring.down));
    }
    CP2_HIP(ctx, hipStreamSynchronize(ring.down));
    if (trace) std::fprintf(stderr, "[cp2 trace] stream_map %zu chunks straight from / to the caller's pinned arrays\n", n_chunks);
    return CP2_OK;
  }
  const int threads = ctx->ingest_threads > 0 ? ctx->ingest_threads : 8;
  Workers pool(threads > 1 ? threads - 1 : 1);
  auto par_copy = [&](uint8_t* dst, const uint8_t* src, size_t bytes) {
    const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)threads, bytes >> 21));
    for (int t = 1; t < nt; ++t) pool.submit([=] { std::memcpy(dst + bytes * t / nt, src + bytes * t / nt, bytes * (t + 1) / nt - bytes * t / nt); });
    std::memcpy(dst, src, bytes / nt);
    pool.wait_idle();
  };
  const size_t n_chunks = (n + CHUNK - 1) / CHUNK;
  double t_wait = 0, t_drain = 0, t_fill = 0, t_enq = 0;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  auto drain = [&](size_t c) -> int {   // chunk c's results from the pinned ring into the caller's array
    const int r = (int)(c % DEPTH);
    double a = now();
    CP2_HIP(ctx, hipEventSynchronize(ring.e_down[r]));
    double b = now();
    const size_t m = std::min(CHUNK, n - c * CHUNK);
    par_copy(out + c * CHUNK * out_item, ring.pin_out[r].u8(), m * out_item);
    t_wait += b - a;
    t_drain += now() - b;
    return CP2_OK;
  };
  for (size_t c = 0; c < n_chunks; ++c) {
    const int r = (int)(c % DEPTH);
    if (c >= DEPTH) CP2_TRY(drain(c - DEPTH));   // frees ring slot r (its upload and kernel finished before its download)
    const size_t m = std::min(CHUNK, n - c * CHUNK);
    double a = now();
    par_copy(ring.pin_in[r].u8(), in + c * CHUNK * in_item, m * in_item);
    double b = now();
    t_fill += b - a;
    CP2_HIP(ctx, hipMemcpyAsync(ring.d_in[r].p, ring.pin_in[r].p, m * in_item, hipMemcpyHostToDevice, ring.up));
    CP2_HIP(ctx, hipEventRecord(ring.e_up[r], ring.up));
    CP2_HIP(ctx, hipStreamWaitEvent(ctx->stream, ring.e_up[r], 0));
    CP2_HIP(ctx, launch(ring.d_in[r].p, ring.d_out[r].p, m, ctx->stream));
    CP2_HIP(ctx, hipEventRecord(ring.e_k[r], ctx->stream));
    CP2_HIP(ctx, hipStreamWaitEvent(ring.down, ring.e_k[r], 0));
    CP2_HIP(ctx, hipMemcpyAsync(ring.pin_out[r].p, ring.d_out[r].p, m * out_item, hipMemcpyDeviceToHost, ring.down));
    CP2_HIP(ctx, hipEventRecord(ring.e_down[r], ring.down));
    t_enq += now() - b;
  }
  for (size_t c = n_chunks > DEPTH ? n_chunks - DEPTH : 0; c < n_chunks; ++c) CP2_TRY(drain(c));
  if (trace)
    std::fprintf(stderr, "[cp2 trace] stream_map %zu chunks: fill %.2f ms, enqueue %.2f ms, wait for downloads %.2f ms, drain %.2f ms\n",
                 n_chunks, t_fill, t_enq, t_wait, t_drain);
  return CP2_OK;
}
}  // namespace

extern "C" int cp2_permute_batch(cp2_ctx* ctx, const uint8_t* in, uint8_t* out, size_t n) try {
  if (!ctx || (n && (!in || !out))) return CP2_ERR_INVALID;
  if (n == 0) return CP2_OK;
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  return stream_map(ctx, in, 96, out, 96, n, [](const void* i, void* o, size_t m, hipStream_t st) { return cp2k::launch_permute_batch(i, o, m, st); });
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

// ---------------------------------------------------------------------------------------------
// a6 keyed compression
// ---------------------------------------------------------------------------------------------
extern "C" int cp2_compress_batch(cp2_ctx* ctx, const uint8_t* xy, uint32_t key, uint8_t* out, size_t n) try {
  if (!ctx || key > 3 || (n && (!xy || !out))) return CP2_ERR_INVALID;
  if (n == 0) return CP2_OK;
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  return stream_map(ctx, xy, 64, out, 32, n,
                    [key](const void* i, void* o, size_t m, hipStream_t st) { return cp2k::launch_compress_pairs(i, key, o, m, st); });
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

// ---------------------------------------------------------------------------------------------
// a3 sponge over field elements
// ---------------------------------------------------------------------------------------------
extern "C" int cp2_sponge2_felts_batch_dev(cp2_ctx* ctx, const void* d_felts, size_t nf, size_t nitems, void* d_out) try {
  if (!ctx || (nitems && (!d_out || (nf && !d_felts)))) return CP2_ERR_INVALID;
  if (!aligned16(d_felts) || !aligned16(d_out)) return CP2_ERR_ALIGN;
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  CP2_HIP(ctx, cp2k::launch_sponge2_felts(d_felts, nf, nitems, d_out, ctx->stream));
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" int cp2_sponge2_felts_batch(cp2_ctx* ctx, const uint8_t* felts, size_t nf, size_t nitems, uint8_t* out) try {
  if (!ctx || (nitems && (!out || (nf && !felts)))) return CP2_ERR_INVALID;
  if (nitems == 0) return CP2_OK;
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  DevBuf d_in, d_out;
  CP2_TRY(d_in.scratch(ctx, std::max<size_t>(nf * nitems * 32, 32)));
  CP2_TRY(d_out.scratch(ctx, nitems * 32));
  if (nf) CP2_HIP(ctx, hipMemcpyAsync(d_in.p, felts, nf * nitems * 32, hipMemcpyHostToDevice, ctx->stream));
  CP2_TRY(cp2_sponge2_felts_batch_dev(ctx, d_in.p, nf, nitems, d_out.p));
  CP2_HIP(ctx, hipMemcpyAsync(out, d_out.p, nitems * 32, hipMemcpyDeviceToHost, ctx->stream));
  CP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" int cp2_sponge2_felts(cp2_ctx* ctx, const uint8_t* felts, size_t n, uint8_t out[32]) try {
  return cp2_sponge2_felts_batch(ctx, felts, n, 1, out);
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

// ---------------------------------------------------------------------------------------------
// a4 bytes -> field elements (pure byte packing, host)
// ---------------------------------------------------------------------------------------------
extern "C" size_t cp2_felts_per_bytes(size_t len) { return (len + 1 + 30) / 31; }

extern "C" int cp2_bytes_to_felts(const uint8_t* data, size_t len, uint8_t* out) try {
  if ((len && !data) || !out) return CP2_ERR_INVALID;
  size_t n = cp2_felts_per_bytes(len);
  const size_t full = len / 31;                         // chunks made of data bytes only
  for (size_t k = 0; k < full; ++k) {
    std::memcpy(out + 32 * k, data + 31 * k, 31);
    out[32 * k + 31] = 0;
  }
  if (full < n) {                                       // the chunk holding the 0x01 marker (and, before it, the data tail)
    uint8_t* last = out + 32 * full;
    std::memset(last, 0, 32);
    const size_t tail = len - 31 * full;
    if (tail) std::memcpy(last, data + 31 * full, tail);
    last[tail] = 0x01;
  }
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

// ---------------------------------------------------------------------------------------------
// a5 hashCell
// ---------------------------------------------------------------------------------------------
extern "C" int cp2_hash_cells_dev(cp2_ctx* ctx, const void* d_cells, size_t cell_size, size_t n_cells, void* d_out) try {
  if (!ctx || (n_cells && (!d_out || (cell_size && !d_cells)))) return CP2_ERR_INVALID;
  if (!aligned16(d_out)) return CP2_ERR_ALIGN;
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  CP2_HIP(ctx, cp2k::launch_hash_cells(d_cells, cell_size, n_cells, d_out, ctx->stream));
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" int cp2_hash_cells(cp2_ctx* ctx, const uint8_t* cells, size_t cell_size, size_t n_cells, uint8_t* out) try {
  if (!ctx || (n_cells && (!out || (cell_size && !cells)))) return CP2_ERR_INVALID;
  if (n_cells == 0) return CP2_OK;
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  DevBuf d_in, d_out;
  CP2_TRY(d_out.scratch(ctx, n_cells * 32));
  if (cell_size * n_cells > ((size_t)32 << 20)) {
    // large inputs stream through the pinned ring (upload and hashing overlapped)
    CP2_TRY(hash_host_cells_pipelined(ctx, cells, cell_size, n_cells, d_out.u8()));
  } else {
    CP2_TRY(d_in.scratch(ctx, std::max<size_t>(cell_size * n_cells, 16)));
    if (cell_size) CP2_HIP(ctx, hipMemcpyAsync(d_in.p, cells, cell_size * n_cells, hipMemcpyHostToDevice, ctx->stream));
    CP2_TRY(cp2_hash_cells_dev(ctx, d_in.p, cell_size, n_cells, d_out.p));
  }
  CP2_HIP(ctx, hipMemcpyAsync(out, d_out.p, n_cells * 32, hipMemcpyDeviceToHost, ctx->stream));
  CP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" int cp2_hash_bytes(cp2_ctx* ctx, const uint8_t* data, size_t len, uint8_t out[32]) try {
  return cp2_hash_cells(ctx, data, len, 1, out);
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

// ---------------------------------------------------------------------------------------------
// a7 Merkle trees
// ---------------------------------------------------------------------------------------------
extern "C" size_t cp2_merkle_num_layers(size_t n) { return layer_sizes_of(n).size(); }

extern "C" size_t cp2_merkle_total(size_t n) {
  size_t t = 0;
  for (size_t m : layer_sizes_of(n)) t += m;
  return t;
}

// layer-major: layer k of all nseg trees is contiguous, tree s at s * size_k inside it
int cp2i::merkle_trees_dev(cp2_ctx* ctx, const void* d_leaves, size_t n, size_t nseg, void* d_layers_out,
                           bool leaves_in_place) {
  if (n == 0 || nseg == 0) return CP2_OK;
  std::vector<size_t> sizes = layer_sizes_of(n);
  uint8_t* base = static_cast<uint8_t*>(d_layers_out);
  if (!leaves_in_place)
    CP2_HIP(ctx, hipMemcpyAsync(base, d_leaves, n * nseg * 32, hipMemcpyDeviceToDevice, ctx->stream));
  size_t off = 0;
  for (size_t k = 0; k + 1 < sizes.size(); ++k) {
    const uint8_t* in = base + off * 32;
    uint8_t* out = base + (off + sizes[k] * nseg) * 32;
    CP2_HIP(ctx, cp2k::launch_compress_layer(in, out, sizes[k], nseg, k == 0, sizes[k], sizes[k + 1], ctx->stream));
    off += sizes[k] * nseg;
  }
  return CP2_OK;
}

extern "C" int cp2_merkle_trees_dev(cp2_ctx* ctx, const void* d_leaves, size_t n, size_t nseg, void* d_layers_out) try {
  if (!ctx || !d_leaves || !d_layers_out) return CP2_ERR_INVALID;
  if (!aligned16(d_leaves) || !aligned16(d_layers_out)) return CP2_ERR_ALIGN;
  if (n == 0) return CP2_ERR_INVALID;   // Merkle.hs:72 "input is empty"
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  return merkle_trees_dev(ctx, d_leaves, n, nseg, d_layers_out, d_leaves == d_layers_out);
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" int cp2_merkle_tree(cp2_ctx* ctx, const uint8_t* leaves, size_t n, uint8_t* layers_out, size_t* layer_sizes,
                               size_t* n_layers) try {
  if (!ctx || !leaves || !layers_out || n == 0) return CP2_ERR_INVALID;
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  std::vector<size_t> sizes = layer_sizes_of(n);
  size_t total = cp2_merkle_total(n);
  DevBuf d;
  CP2_TRY(d.scratch(ctx, total * 32));
  CP2_HIP(ctx, hipMemcpyAsync(d.p, leaves, n * 32, hipMemcpyHostToDevice, ctx->stream));
  CP2_TRY(merkle_trees_dev(ctx, d.p, n, 1, d.p, true));
  CP2_HIP(ctx, hipMemcpyAsync(layers_out, d.p, total * 32, hipMemcpyDeviceToHost, ctx->stream));
  CP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (layer_sizes) std::copy(sizes.begin(), sizes.end(), layer_sizes);
  if (n_layers) *n_layers = sizes.size();
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" int cp2_merkle_root(cp2_ctx* ctx, const uint8_t* leaves, size_t n, uint8_t out[32]) try {
  if (!ctx || !leaves || !out || n == 0) return CP2_ERR_INVALID;
  size_t total = cp2_merkle_total(n);
  std::vector<uint8_t> layers(total * 32);
  CP2_TRY(cp2_merkle_tree(ctx, leaves, n, layers.data(), nullptr, nullptr));
  std::memcpy(out, &layers[(total - 1) * 32], 32);
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

// ---------------------------------------------------------------------------------------------
// a10 fake data
// ---------------------------------------------------------------------------------------------
extern "C" uint64_t cp2_slot_seed(uint64_t dataset_seed, uint64_t slot_idx) { return dataset_seed + 72 + 1001 * slot_idx; }

extern "C" int cp2_gen_fake_cells_dev(cp2_ctx* ctx, uint64_t seed, uint64_t first, size_t n, size_t cell_size, void* d_out) try {
  if (!ctx || (n && cell_size && !d_out)) return CP2_ERR_INVALID;
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  CP2_HIP(ctx, cp2k::launch_gen_fake_cells(seed, 0, first, nullptr, n, cell_size, d_out, ctx->stream));
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

extern "C" int cp2_gen_fake_cells(cp2_ctx* ctx, uint64_t seed, uint64_t first, size_t n, size_t cell_size, uint8_t* out) try {
  if (!ctx || (n && cell_size && !out)) return CP2_ERR_INVALID;
  if (n == 0 || cell_size == 0) return CP2_OK;
  CP2_HIP(ctx, hipSetDevice(ctx->device));
  DevBuf d;
  CP2_TRY(d.scratch(ctx, n * cell_size));
  CP2_TRY(cp2_gen_fake_cells_dev(ctx, seed, first, n, cell_size, d.p));
  CP2_HIP(ctx, hipMemcpyAsync(out, d.p, n * cell_size, hipMemcpyDeviceToHost, ctx->stream));
  CP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

// ---------------------------------------------------------------------------------------------
// a12 sampling
// ---------------------------------------------------------------------------------------------
extern "C" int cp2_cell_indices(cp2_ctx* ctx, const uint8_t entropy[32], const uint8_t slot_root[32], uint64_t n_cells,
                                size_t n_samples, uint64_t* out) try {
  if (!ctx || !entropy || !slot_root || (n_samples && !out)) return CP2_ERR_INVALID;
  if (n_cells < 2 || (n_cells & (n_cells - 1)) != 0) return CP2_ERR_INVALID;    // sample/bn254.nim:19-20; one cell: extractLowBits asserts k > 0 (types/bn254.nim:48)
  if (n_samples == 0) return CP2_OK;
  std::vector<uint8_t> felts(n_samples * 96, 0), dig(n_samples * 32);
  for (size_t i = 0; i < n_samples; ++i) {
    std::memcpy(&felts[96 * i], entropy, 32);
    std::memcpy(&felts[96 * i + 32], slot_root, 32);
    uint64_t counter = i + 1;                                                    // sample/bn254.nim:27
    std::memcpy(&felts[96 * i + 64], &counter, 8);
  }
  CP2_TRY(cp2_sponge2_felts_batch(ctx, felts.data(), 3, n_samples, dig.data()));
  for (size_t i = 0; i < n_samples; ++i) {
    uint64_t lo;
    std::memcpy(&lo, &dig[32 * i], 8);                                           // extractLowBits, types/bn254.nim:47-59
    out[i] = lo & (n_cells - 1);
  }
  return CP2_OK;
} catch (const std::bad_alloc&) {
  return CP2_ERR_ALLOC;   // nothing may unwind across the C ABI
} catch (...) {
  return CP2_ERR_INVALID;
}

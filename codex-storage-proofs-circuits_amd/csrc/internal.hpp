// Internals shared by the C-ABI translation units (not installed).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/codex_p2.h"

struct cp2_ctx {
  int device = 0;
  bool native = false;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
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

#define CP2_TRY(call)                 \
  do {                                \
    int s__ = (call);                 \
    if (s__ != CP2_OK) return s__;    \
  } while (0)

// RAII device buffer
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  int alloc(cp2_ctx* ctx, size_t n) {
    release();
    if (n == 0) n = 16;
    hipError_t e = hipMalloc(&p, n);
    if (e != hipSuccess) {
      p = nullptr;
      ctx->err = std::string("hipMalloc(") + std::to_string(n) + "): " + hipGetErrorString(e);
      return CP2_ERR_ALLOC;
    }
    bytes = n;
    return CP2_OK;
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

}  // namespace cp2i

// Where does host -> HBM ingestion lose its bandwidth?  (run on the GPU box)
//   hipcc -O2 -std=c++17 -pthread tools/ingest_probe.cpp -Icodex-storage-proofs-circuits_amd -Lcodex-storage-proofs-circuits_amd \
//         -lcodex_p2 -Wl,-rpath,$PWD/codex-storage-proofs-circuits_amd -o /tmp/ingest_probe && /tmp/ingest_probe
// Measures, each alone: multi-threaded memcpy pageable->pageable and pageable->pinned (several hipHostMalloc flags),
// pinned hipMemcpyAsync H2D, the hash kernel from HBM; then H2D and the hash kernel together on two streams.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/codex_p2.h"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); std::exit(1); } } while (0)

static double par_memcpy(uint8_t* dst, const uint8_t* src, size_t n, int threads, int reps) {
  double best = 0;
  for (int r = 0; r < reps; ++r) {
    double t0 = now();
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t)
      th.emplace_back([=] { size_t a = n * t / threads, b = n * (t + 1) / threads; std::memcpy(dst + a, src + a, b - a); });
    for (auto& x : th) x.join();
    double gbps = n / (now() - t0) / 1e9;
    if (gbps > best) best = gbps;
  }
  return best;
}

int main() {
  const size_t N = (size_t)1 << 30;
  uint8_t* a = (uint8_t*)std::malloc(N);
  uint8_t* b = (uint8_t*)std::malloc(N);
  std::memset(a, 1, N);
  std::memset(b, 2, N);
  std::printf("hardware_concurrency %u\n", std::thread::hardware_concurrency());
  for (int t : {1, 2, 4, 8, 16}) std::printf("memcpy pageable->pageable  %2d threads: %6.1f GB/s\n", t, par_memcpy(b, a, N, t, 3));
  struct { const char* name; unsigned flags; } kinds[] = {{"hipHostMallocDefault", hipHostMallocDefault},
                                                          {"hipHostMallocNonCoherent", hipHostMallocNonCoherent},
                                                          {"hipHostMallocCoherent", hipHostMallocCoherent},
                                                          {"hipHostMallocNumaUser", hipHostMallocNumaUser},
                                                          {"hipHostMallocWriteCombined", hipHostMallocWriteCombined}};
  void* d = nullptr;
  CK(hipMalloc(&d, N));
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (auto& k : kinds) {
    void* p = nullptr;
    double t0 = now();
    if (hipHostMalloc(&p, N, k.flags) != hipSuccess) { (void)hipGetLastError(); std::printf("%s: allocation failed\n", k.name); continue; }
    double alloc_ms = (now() - t0) * 1e3;
    std::memset(p, 3, N);
    std::printf("%s: hipHostMalloc(1 GiB) %.1f ms\n", k.name, alloc_ms);
    for (int t : {1, 4, 8, 16}) std::printf("  memcpy pageable->pinned %2d threads: %6.1f GB/s\n", t, par_memcpy((uint8_t*)p, a, N, t, 3));
    for (int t : {1, 8}) std::printf("  memcpy pinned->pageable %2d threads: %6.1f GB/s\n", t, par_memcpy(b, (uint8_t*)p, N, t, 2));
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
      CK(hipEventRecord(e0, s1));
      CK(hipMemcpyAsync(d, p, N, hipMemcpyHostToDevice, s1));
      CK(hipEventRecord(e1, s1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    std::printf("  hipMemcpyAsync H2D 1 GiB: %.1f GB/s; in 64 MiB pieces:", N / (best * 1e-3) / 1e9);
    CK(hipEventRecord(e0, s1));
    for (size_t o = 0; o < N; o += (size_t)64 << 20) CK(hipMemcpyAsync((uint8_t*)d + o, (uint8_t*)p + o, (size_t)64 << 20, hipMemcpyHostToDevice, s1));
    CK(hipEventRecord(e1, s1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::printf(" %.1f GB/s\n", N / (ms * 1e-3) / 1e9);
    CK(hipHostFree(p));
  }
  // the hash kernel alone and with a concurrent H2D
  cp2_ctx* ctx = nullptr;
  if (cp2_init(0, &ctx) != CP2_OK) { std::printf("cp2_init failed\n"); return 1; }
  cp2_set_stream(ctx, s2);
  void *cells = nullptr, *leaves = nullptr, *p = nullptr;
  const size_t ncell = N / 2048;
  CK(hipMalloc(&cells, N));
  CK(hipMalloc(&leaves, ncell * 32));
  CK(hipHostMalloc(&p, N, hipHostMallocDefault));
  std::memset(p, 5, N);
  cp2_gen_fake_cells_dev(ctx, 1, 0, ncell, 2048, cells);
  cp2_hash_cells_dev(ctx, cells, 2048, ncell, leaves);
  CK(hipStreamSynchronize(s2));
  hipEvent_t k0, k1;
  CK(hipEventCreate(&k0));
  CK(hipEventCreate(&k1));
  float ms_k, ms_c;
  CK(hipEventRecord(k0, s2));
  cp2_hash_cells_dev(ctx, cells, 2048, ncell, leaves);
  CK(hipEventRecord(k1, s2));
  CK(hipEventSynchronize(k1));
  CK(hipEventElapsedTime(&ms_k, k0, k1));
  std::printf("hash kernel alone, 1 GiB of cells from HBM: %.2f ms = %.1f GB/s\n", ms_k, N / (ms_k * 1e-3) / 1e9);
  CK(hipEventRecord(k0, s2));
  cp2_hash_cells_dev(ctx, cells, 2048, ncell, leaves);
  CK(hipEventRecord(k1, s2));
  CK(hipEventRecord(e0, s1));
  CK(hipMemcpyAsync(d, p, N, hipMemcpyHostToDevice, s1));
  CK(hipEventRecord(e1, s1));
  CK(hipEventSynchronize(k1));
  CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms_k, k0, k1));
  CK(hipEventElapsedTime(&ms_c, e0, e1));
  std::printf("together: hash kernel %.2f ms (%.1f GB/s), H2D 1 GiB %.2f ms (%.1f GB/s)\n", ms_k, N / (ms_k * 1e-3) / 1e9, ms_c, N / (ms_c * 1e-3) / 1e9);
  // chunked pipelines, pinned source, no host fill at all: which structure overlaps copy(i+1) with hash(i)?
  auto ring = [&](const char* name, size_t CH, int depth, bool per_slot_streams, bool timeline) {
    const size_t NCH = N / CH;
    std::vector<hipEvent_t> cp(depth), hs(depth);
    std::vector<void*> dv(depth);
    std::vector<hipStream_t> ss(depth);
    std::vector<hipEvent_t> tl;
    for (int i = 0; i < depth; ++i) {
      CK(hipEventCreateWithFlags(&cp[i], hipEventDisableTiming));
      CK(hipEventCreateWithFlags(&hs[i], hipEventDisableTiming));
      CK(hipMalloc(&dv[i], CH));
      CK(hipStreamCreateWithFlags(&ss[i], hipStreamNonBlocking));
      CK(hipEventRecord(hs[i], s2));
    }
    if (timeline) { tl.resize(NCH * 4); for (size_t q = 0; q < tl.size(); ++q) CK(hipEventCreate(&tl[q])); }
    CK(hipDeviceSynchronize());
    double t0 = now();
    for (size_t i = 0; i < NCH; ++i) {
      int r = (int)(i % depth);
      CK(hipEventSynchronize(hs[r]));
      hipStream_t cs = per_slot_streams ? ss[r] : s1, ks = per_slot_streams ? ss[r] : s2;
      if (timeline) CK(hipEventRecord(tl[4 * i], cs));
      CK(hipMemcpyAsync(dv[r], (uint8_t*)p + i * CH, CH, hipMemcpyHostToDevice, cs));
      if (timeline) CK(hipEventRecord(tl[4 * i + 1], cs));
      if (!per_slot_streams) { CK(hipEventRecord(cp[r], cs)); CK(hipStreamWaitEvent(ks, cp[r], 0)); }
      cp2_set_stream(ctx, ks);
      if (timeline) CK(hipEventRecord(tl[4 * i + 2], ks));
      cp2_hash_cells_dev(ctx, dv[r], 2048, CH / 2048, (uint8_t*)leaves + i * (CH / 2048) * 32);
      if (timeline) CK(hipEventRecord(tl[4 * i + 3], ks));
      CK(hipEventRecord(hs[r], ks));
    }
    CK(hipDeviceSynchronize());
    std::printf("%-58s %6.1f GB/s\n", name, N / (now() - t0) / 1e9);
    if (timeline) {
      for (size_t i = 0; i < NCH && i < 8; ++i) {
        float a0, a1, b0, b1;
        CK(hipEventElapsedTime(&a0, tl[0], tl[4 * i]));
        CK(hipEventElapsedTime(&a1, tl[0], tl[4 * i + 1]));
        CK(hipEventElapsedTime(&b0, tl[0], tl[4 * i + 2]));
        CK(hipEventElapsedTime(&b1, tl[0], tl[4 * i + 3]));
        std::printf("    chunk %zu: copy %.2f..%.2f ms   hash %.2f..%.2f ms\n", i, a0, a1, b0, b1);
      }
      for (auto& e : tl) (void)hipEventDestroy(e);
    }
    for (int i = 0; i < depth; ++i) { (void)hipEventDestroy(cp[i]); (void)hipEventDestroy(hs[i]); (void)hipFree(dv[i]); (void)hipStreamDestroy(ss[i]); }
  };
  ring("copy stream + hash stream + events, 3 x 64 MiB", (size_t)64 << 20, 3, false, false);
  ring("copy stream + hash stream + events, 3 x 64 MiB (timeline)", (size_t)64 << 20, 3, false, true);
  ring("copy stream + hash stream + events, 3 x 256 MiB", (size_t)256 << 20, 3, false, false);
  ring("one stream per ring slot, 3 x 64 MiB", (size_t)64 << 20, 3, true, false);
  ring("one stream per ring slot, 3 x 64 MiB (timeline)", (size_t)64 << 20, 3, true, true);
  ring("one stream per ring slot, 4 x 32 MiB", (size_t)32 << 20, 4, true, false);
  ring("one stream per ring slot, 2 x 128 MiB", (size_t)128 << 20, 2, true, false);
  return 0;
}

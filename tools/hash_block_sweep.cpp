// k_hash_cells throughput against batch size for its two workgroup sizes.
//   hipcc -O2 -std=c++17 tools/hash_block_sweep.cpp -Icodex-storage-proofs-circuits_amd/csrc -Lcodex-storage-proofs-circuits_amd \
//         -lcodex_p2 -Wl,-rpath,$PWD/codex-storage-proofs-circuits_amd -o tools/hash_block_sweep && tools/hash_block_sweep
// For batches of 2 KiB cells from 2 MiB to 8 GiB, resident in HBM: the launch time (HIP events, best and median of several
// launches) of the 256-lane and of the 64-lane instantiation, GB/s of cell data, and whether both wrote identical digests.
// A second table runs two launches of half the batch on two streams (how the ingestion pipe overlaps consecutive chunks).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "kernels.hpp"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); std::exit(1); } } while (0)

static uint64_t digest(const std::vector<uint8_t>& v) {
  uint64_t h = 0xcbf29ce484222325ULL;
  for (size_t i = 0; i + 8 <= v.size(); i += 8) {
    uint64_t w;
    std::memcpy(&w, &v[i], 8);
    h = (h ^ w) * 0x100000001b3ULL;
    h = (h << 27) | (h >> 37);
  }
  return h;
}

int main(int argc, char** argv) {
  const size_t cs = 2048;
  const size_t max_bytes = argc > 1 ? (size_t)std::atoll(argv[1]) << 20 : (size_t)8 << 30;
  uint8_t *cells = nullptr, *out = nullptr;
  CK(hipMalloc((void**)&cells, max_bytes));
  CK(hipMalloc((void**)&out, max_bytes / cs * 32));
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(cp2k::launch_gen_fake_cells(12345 + 72, 0, 0, nullptr, max_bytes / cs, cs, cells, s0));
  CK(hipStreamSynchronize(s0));
  hipEvent_t e0, e1, e2;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventCreate(&e2));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  std::printf("device %s, %d CUs; cell size %zu; one launch per row, HIP events on the launch stream\n", prop.gcnArchName, prop.multiProcessorCount, cs);
  std::printf("%10s %9s | %10s %10s %8s | %10s %10s %8s | %s\n", "batch", "waves", "256: best", "median ms", "GB/s", "64: best", "median ms", "GB/s", "same digests");
  for (size_t bytes = (size_t)2 << 20; bytes <= max_bytes; bytes *= 2) {
    const size_t n = bytes / cs;
    const int reps = bytes >= ((size_t)1 << 32) ? 3 : (bytes >= ((size_t)1 << 28) ? 5 : 9);
    double best[2], med[2];
    uint64_t dig[2];
    for (int v = 0; v < 2; ++v) {
      const int block = v == 0 ? 256 : 64;
      std::vector<float> ms;
      CK(hipMemsetAsync(out, 0, n * 32, s0));
      CK(cp2k::launch_hash_cells_block(block, cells, cs, n, out, s0));   // warm-up
      CK(hipStreamSynchronize(s0));
      for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, s0));
        CK(cp2k::launch_hash_cells_block(block, cells, cs, n, out, s0));
        CK(hipEventRecord(e1, s0));
        CK(hipEventSynchronize(e1));
        float t = 0;
        CK(hipEventElapsedTime(&t, e0, e1));
        ms.push_back(t);
      }
      std::sort(ms.begin(), ms.end());
      best[v] = ms.front();
      med[v] = ms[ms.size() / 2];
      std::vector<uint8_t> h(std::min<size_t>(n * 32, (size_t)64 << 20));
      CK(hipMemcpy(h.data(), out, h.size(), hipMemcpyDeviceToHost));
      dig[v] = digest(h);
    }
    std::printf("%7zu MiB %9zu | %10.3f %10.3f %8.2f | %10.3f %10.3f %8.2f | %s\n", bytes >> 20, n / 64, best[0], med[0], bytes / (med[0] * 1e-3) / 1e9,
                best[1], med[1], bytes / (med[1] * 1e-3) / 1e9, dig[0] == dig[1] ? "yes" : "NO");
    std::fflush(stdout);
  }
  std::printf("\ntwo launches of half the batch each, on two streams (events around both)\n");
  std::printf("%10s | %10s %8s | %10s %8s\n", "batch", "256: ms", "GB/s", "64: ms", "GB/s");
  for (size_t bytes = (size_t)8 << 20; bytes <= std::min<size_t>(max_bytes, (size_t)1 << 30); bytes *= 2) {
    const size_t n = bytes / cs, half = n / 2;
    double med[2];
    for (int v = 0; v < 2; ++v) {
      const int block = v == 0 ? 256 : 64;
      std::vector<float> ms;
      for (int r = 0; r < 6; ++r) {
        CK(hipEventRecord(e0, s0));
        CK(hipStreamWaitEvent(s1, e0, 0));
        CK(cp2k::launch_hash_cells_block(block, cells, cs, half, out, s0));
        CK(cp2k::launch_hash_cells_block(block, cells + half * cs, cs, n - half, out + half * 32, s1));
        CK(hipEventRecord(e2, s1));
        CK(hipStreamWaitEvent(s0, e2, 0));
        CK(hipEventRecord(e1, s0));
        CK(hipEventSynchronize(e1));
        float t = 0;
        CK(hipEventElapsedTime(&t, e0, e1));
        if (r) ms.push_back(t);
      }
      std::sort(ms.begin(), ms.end());
      med[v] = ms[ms.size() / 2];
    }
    std::printf("%7zu MiB | %10.3f %8.2f | %10.3f %8.2f\n", bytes >> 20, med[0], bytes / (med[0] * 1e-3) / 1e9, med[1], bytes / (med[1] * 1e-3) / 1e9);
    std::fflush(stdout);
  }
  return 0;
}

// Which kernels START while a k_hash_cells launch fills the device?  (Round 5: the streamed proof-input build.)
//   hipcc -O2 -std=c++17 tools/coresidency_probe.cpp -Icodex-storage-proofs-circuits_amd/csrc -Lcodex-storage-proofs-circuits_amd \
//         -lcodex_p2 -Wl,-rpath,$PWD/codex-storage-proofs-circuits_amd -o tools/coresidency_probe && tools/coresidency_probe
// Stream A hashes 2^20 cells of 2 KiB (one staging chunk, ~50 ms).  10 ms later the host enqueues a second kernel on stream B and
// waits for it: if it finishes long before the hash launch does, it ran beside it; if it finishes when the hash launch ends, it
// sat waiting for the launch's last workgroups.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "kernels.hpp"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); std::exit(1); } } while (0)

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const size_t cs = 2048, n = (size_t)1 << 20;
  uint8_t *cells = nullptr, *cells2 = nullptr, *out = nullptr, *out2 = nullptr, *nodes = nullptr;
  CK(hipMalloc((void**)&cells, n * cs));
  CK(hipMalloc((void**)&cells2, n * cs));
  CK(hipMalloc((void**)&out, n * 32));
  CK(hipMalloc((void**)&out2, n * 32));
  CK(hipMalloc((void**)&nodes, n * 32));
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  CK(cp2k::launch_gen_fake_cells(12345 + 72, 0, 0, nullptr, n, cs, cells, a));
  CK(cp2k::launch_gen_fake_cells(999 + 72, 0, 0, nullptr, n, cs, cells2, a));
  CK(cp2k::launch_hash_cells(cells, cs, n, out, a));
  CK(hipStreamSynchronize(a));
  struct Case { const char* name; int kind; size_t m; };
  const Case cases[] = {{"generator (36 KB LDS stage), 2^20 cells", 0, n}, {"generator (36 KB LDS stage), 25 600 cells", 0, 25600},
                        {"k_compress_layer, 2^19 pairs", 2, n / 2}, {"k_compress_layer, 256 pairs", 2, 256}, {"k_hash_cells, 2^20 cells (a second chunk)", 3, n},
                        {"k_permute_batch, 2^16 states", 4, 1 << 16}};
  for (const Case& c : cases) {
    for (int beside = 0; beside < 2; ++beside) {
      CK(hipDeviceSynchronize());
      const double t0 = now_ms();
      if (beside) CK(cp2k::launch_hash_cells(cells, cs, n, out, a));
      std::this_thread::sleep_for(std::chrono::milliseconds(10));
      const double t1 = now_ms();
      switch (c.kind) {
        case 0: CK(cp2k::launch_gen_fake_cells(777 + 72, 0, 0, nullptr, c.m, cs, cells2, b)); break;
        case 2: CK(cp2k::launch_compress_layer(out2, nodes, 2 * c.m, 1, true, 2 * c.m, c.m, b)); break;
        case 3: CK(cp2k::launch_hash_cells(cells2, cs, c.m, out2, b)); break;
        case 4: CK(cp2k::launch_permute_batch(cells2, cells2, c.m, b)); break;
      }
      CK(hipStreamSynchronize(b));
      const double t2 = now_ms();
      CK(hipStreamSynchronize(a));
      const double t3 = now_ms();
      std::printf("%-46s %s: done %6.2f ms after it was enqueued; the hash launch %s %6.2f ms after ITS enqueue\n", c.name, beside ? "beside a hash launch" : "alone               ",
                  t2 - t1, beside ? "ended" : "(none)", beside ? t3 - t0 : 0.0);
    }
  }
  // a CHAIN of small dependent kernels (one group's twelve layer passes of the streamed build) beside a hash launch, on a stream of
  // normal and of high priority: does each link wait for the launch's next residency boundary?
  int least = 0, greatest = 0;
  CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
  hipStream_t hi;
  CK(hipStreamCreateWithPriority(&hi, hipStreamNonBlocking, greatest));
  std::printf("stream priorities: least %d, greatest %d\n", least, greatest);
  for (int links : {1, 12, 16}) {
    for (int which = 0; which < 3; ++which) {      // 0 alone on b, 1 beside hash on b, 2 beside hash on the high-priority stream
      hipStream_t st = which == 2 ? hi : b;
      CK(hipDeviceSynchronize());
      const double t0 = now_ms();
      if (which) CK(cp2k::launch_hash_cells(cells, cs, n, out, a));
      std::this_thread::sleep_for(std::chrono::milliseconds(10));
      const double t1 = now_ms();
      size_t m = (size_t)256 << 11;                // 256 slots x 2^11 pairs at the bottom, halving upwards like a tree
      for (int l = 0; l < links; ++l, m = std::max<size_t>(256, m / 2)) CK(cp2k::launch_compress_layer(out2, nodes, 2 * m, 1, l == 0, 2 * m, m, st));
      CK(hipStreamSynchronize(st));
      const double t2 = now_ms();
      CK(hipStreamSynchronize(a));
      const double t3 = now_ms();
      std::printf("chain of %2d layer kernels %-42s: done %6.2f ms after it was enqueued%s\n", links,
                  which == 0 ? "alone" : which == 1 ? "beside a hash launch, normal priority" : "beside a hash launch, HIGH priority", t2 - t1,
                  which ? "" : "");
      if (which) std::printf("        (the hash launch ended %6.2f ms after its enqueue)\n", t3 - t0);
    }
  }
  return 0;
}

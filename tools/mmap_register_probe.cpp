// Can slot files that sit in the page cache reach the device WITHOUT a CPU copy?  (the ingestion pipe copies every
// byte from the page cache into its pinned ring -- pread -- before the upload; from a page-cache-warm file it runs at 37-38 GB/s where
// the hash kernel sustains 44-47.)  Probe: mmap the file read-only, hipHostRegister the mapping, hipMemcpyAsync straight from it.
//   hipcc -O2 -o tools/mmap_register_probe tools/mmap_register_probe.cpp && tools/mmap_register_probe <file> [GiB to write if missing]
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
__global__ void spin(unsigned long long cycles, unsigned* out) {   // keeps the device busy for `cycles` shader clocks
  const unsigned long long t0 = clock64();
  unsigned x = threadIdx.x;
  while (clock64() - t0 < cycles) x = x * 1664525u + 1013904223u;
  if (x == 42) *out = x;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
  const char* path = argc > 1 ? argv[1] : "/tmp/mmap_probe.dat";
  const size_t gib = argc > 2 ? (size_t)atoi(argv[2]) : 4, bytes = gib << 30, chunk = (size_t)384 << 20;
  struct stat sb;
  if (stat(path, &sb) != 0 || (size_t)sb.st_size != bytes) {
    std::vector<char> buf(64 << 20, 7);
    FILE* f = fopen(path, "wb");
    if (!f) { perror(path); return 4; }
    for (size_t at = 0; at < bytes; at += buf.size()) { for (size_t i = 0; i < buf.size(); i += 4096) buf[i] = (char)(at >> 20); if (fwrite(buf.data(), 1, buf.size(), f) != buf.size()) { perror("fwrite"); return 4; } }
    if (fclose(f) != 0) { perror("fclose"); return 4; }
    printf("wrote %zu GiB to %s\n", gib, path); fflush(stdout);
  }
  int fd = open(path, O_RDONLY);
  if (fd < 0) { perror("open"); return 5; }
  setvbuf(stdout, nullptr, _IOLBF, 0);
  void* dev = nullptr; CK(hipMalloc(&dev, chunk));
  hipStream_t st; CK(hipStreamCreate(&st));
  // (a) the pipe's way: pread into a pinned buffer (one thread here), then upload
  void* pin = nullptr; CK(hipHostMalloc(&pin, chunk, hipHostMallocDefault));
  for (int rep = 0; rep < 2; ++rep) {
    double t0 = now();
    for (size_t at = 0; at < bytes; at += chunk) { const size_t m = bytes - at < chunk ? bytes - at : chunk; for (size_t got = 0; got < m;) { ssize_t r = pread(fd, (char*)pin + got, m - got, (off_t)(at + got)); if (r <= 0) { perror("pread"); return 2; } got += (size_t)r; } CK(hipMemcpyAsync(dev, pin, m, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); }
    printf("pread (1 thread) + upload, serial: %.2f GB/s\n", bytes / (now() - t0) / 1e9);
  }
  // (b) mmap + register + upload straight from the page cache
  void* map = mmap(nullptr, bytes, PROT_READ, MAP_SHARED | MAP_POPULATE, fd, 0);
  if (map == MAP_FAILED) { perror("mmap"); return 3; }
  double t0 = now();
  hipError_t e = hipHostRegister(map, bytes, hipHostRegisterDefault);
  printf("hipHostRegister of the whole mapping (%zu GiB): %s, %.1f ms\n", gib, hipGetErrorString(e), (now() - t0) * 1e3);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    t0 = now();
    e = hipHostRegister(map, chunk, hipHostRegisterDefault);
    printf("hipHostRegister of one 384 MiB chunk: %s, %.1f ms\n", hipGetErrorString(e), (now() - t0) * 1e3);
    if (e != hipSuccess) { printf("file-backed mappings cannot be registered on this stack: the pinned ring stays\n"); return 0; }
    for (int rep = 0; rep < 3; ++rep) { t0 = now(); CK(hipMemcpyAsync(dev, map, chunk, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); printf("upload from the registered mapping: %.2f GB/s\n", chunk / (now() - t0) / 1e9); }
    return 0;
  }
  for (int rep = 0; rep < 3; ++rep) {
    t0 = now();
    for (size_t at = 0; at < bytes; at += chunk) CK(hipMemcpyAsync(dev, (char*)map + at, bytes - at < chunk ? bytes - at : chunk, hipMemcpyHostToDevice, st));
    CK(hipStreamSynchronize(st));
    printf("upload straight from the registered page-cache mapping: %.2f GB/s\n", bytes / (now() - t0) / 1e9);
  }
  t0 = now(); CK(hipHostUnregister(map)); printf("hipHostUnregister: %.1f ms\n", (now() - t0) * 1e3);
  // (d) what the ingestion pipe would do: a mapping WITHOUT MAP_POPULATE, per 384 MiB window: make the (resident) pages present in the page
  // table (madvise MADV_POPULATE_READ, on T threads over disjoint sub-ranges), register, upload, unregister -- each phase timed
  munmap(map, bytes);
  for (int T : {1, 4, 8}) {
    void* m2 = mmap(nullptr, bytes, PROT_READ, MAP_SHARED, fd, 0);
    if (m2 == MAP_FAILED) { perror("mmap"); return 3; }
    double t_pop = 0, t_reg = 0, t_dma = 0, t_unreg = 0;
    for (size_t at = 0; at + chunk <= bytes; at += chunk) {
      char* w = (char*)m2 + at;
      double a = now();
      {
        std::vector<std::thread> th;
        for (int k = 0; k < T; ++k) th.emplace_back([=] { size_t lo = chunk * k / T / 4096 * 4096, hi = chunk * (k + 1) / T / 4096 * 4096; if (madvise(w + lo, hi - lo, 22 /* MADV_POPULATE_READ */) != 0) perror("madvise"); });
        for (auto& t : th) t.join();
      }
      double b = now();
      hipError_t e2 = hipHostRegister(w, chunk, hipHostRegisterDefault);
      if (e2 != hipSuccess) { printf("register: %s\n", hipGetErrorString(e2)); return 0; }
      double c = now();
      CK(hipMemcpyAsync(dev, w, chunk, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st));
      double d = now();
      CK(hipHostUnregister(w));
      double f = now();
      t_pop += b - a; t_reg += c - b; t_dma += d - c; t_unreg += f - d;
    }
    const double n = (double)(bytes / chunk);
    printf("per 384 MiB window, %d populate thread(s): populate %.2f ms, register %.2f ms, upload %.2f ms (%.1f GB/s), unregister %.2f ms\n", T, t_pop / n * 1e3, t_reg / n * 1e3,
           t_dma / n * 1e3, chunk / (t_dma / n) / 1e9, t_unreg / n * 1e3);
    munmap(m2, bytes);
  }
  // (e) no populate at all: register + upload of a window whose pages are only in the page cache, not in the page table
  {
    void* m2 = mmap(nullptr, bytes, PROT_READ, MAP_SHARED, fd, 0);
    double a = now();
    hipError_t e2 = hipHostRegister(m2, chunk, hipHostRegisterDefault);
    double b = now();
    if (e2 == hipSuccess) { CK(hipMemcpyAsync(dev, m2, chunk, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); }
    double c = now();
    printf("no populate: register %.2f ms (%s), upload %.2f ms (%.1f GB/s)\n", (b - a) * 1e3, hipGetErrorString(e2), (c - b) * 1e3, chunk / (c - b) / 1e9);
    if (e2 == hipSuccess) CK(hipHostUnregister(m2));
    munmap(m2, bytes);
  }
  // (f) do register / unregister wait for the DEVICE?  A kernel that spins for ~0.5 s runs on another stream meanwhile.
  {
    void* m2 = mmap(nullptr, bytes, PROT_READ, MAP_SHARED, fd, 0);
    hipStream_t busy; CK(hipStreamCreate(&busy));
    unsigned* sink = nullptr; CK(hipMalloc(&sink, 4));
    double t_start = now();
    hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, busy, 1000000000ull, sink);
    double a = now();
    hipError_t e2 = hipHostRegister(m2, chunk, hipHostRegisterDefault);
    double b = now();
    CK(hipMemcpyAsync(dev, m2, chunk, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st));
    double c = now();
    if (e2 == hipSuccess) CK(hipHostUnregister(m2));
    double d = now();
    CK(hipStreamSynchronize(busy));
    double t_end = now();
    printf("beside a kernel that spins for %.0f ms on another stream: register %.2f ms, upload %.2f ms, unregister %.2f ms\n", (t_end - t_start) * 1e3, (b - a) * 1e3, (c - b) * 1e3, (d - c) * 1e3);
    munmap(m2, bytes);
  }
  // (g) the runtime's pageable path from a plain mapping (no populate, no registration) beside the same spinning kernel: does the call
  // wait for the device, and for how long does it hold the host?
  {
    void* m2 = mmap(nullptr, bytes, PROT_READ, MAP_SHARED, fd, 0);
    hipStream_t busy; CK(hipStreamCreate(&busy));
    unsigned* sink = nullptr; CK(hipMalloc(&sink, 4));
    hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, busy, 1000000000ull, sink);
    double a = now();
    CK(hipMemcpyAsync(dev, m2, chunk, hipMemcpyHostToDevice, st));
    double b = now();
    CK(hipStreamSynchronize(st));
    double c = now();
    CK(hipMemcpyAsync(dev, (char*)m2 + chunk, chunk, hipMemcpyHostToDevice, st));
    double d = now();
    CK(hipStreamSynchronize(st));
    double e3 = now();
    CK(hipStreamSynchronize(busy));
    double f = now();
    printf("pageable upload from a plain mapping beside the spinning kernel: call returns after %.2f ms, landed after %.2f ms; second window %.2f / %.2f ms; the kernel ended %.0f ms after the first call\n",
           (b - a) * 1e3, (c - a) * 1e3, (d - c) * 1e3, (e3 - c) * 1e3, (f - a) * 1e3);
    munmap(m2, bytes);
  }
  map = mmap(nullptr, bytes, PROT_READ, MAP_SHARED | MAP_POPULATE, fd, 0);
  // (c) unregistered mapping (pageable path of the runtime)
  t0 = now();
  for (size_t at = 0; at < bytes; at += chunk) CK(hipMemcpyAsync(dev, (char*)map + at, bytes - at < chunk ? bytes - at : chunk, hipMemcpyHostToDevice, st));
  CK(hipStreamSynchronize(st));
  printf("upload from the UNregistered mapping (runtime's pageable path): %.2f GB/s\n", bytes / (now() - t0) / 1e9);
  return 0;
}

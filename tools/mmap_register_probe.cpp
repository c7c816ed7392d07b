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
#include <vector>
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
  // (c) unregistered mapping (pageable path of the runtime)
  t0 = now();
  for (size_t at = 0; at < bytes; at += chunk) CK(hipMemcpyAsync(dev, (char*)map + at, bytes - at < chunk ? bytes - at : chunk, hipMemcpyHostToDevice, st));
  CK(hipStreamSynchronize(st));
  printf("upload from the UNregistered mapping (runtime's pageable path): %.2f GB/s\n", bytes / (now() - t0) / 1e9);
  return 0;
}

// Calibrates rocprofv3's FETCH_SIZE for the access widths this repo uses (MI355X_MICROARCH.md: "other access
// widths are uncalibrated: calibrate on a known byte count in your own access pattern").
// Reads exactly 1 GiB once with 4, 8 and 16 bytes per lane, fully coalesced, and the k_hash_cells staging
// pattern (31 lanes x 4 B = 124 contiguous bytes per 2048-byte cell, per tile).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <typename T> __global__ void k_read(const T* __restrict__ p, size_t n, unsigned* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  unsigned acc = 0;
  for (; i < n; i += stride) { T v = p[i]; const unsigned* w = reinterpret_cast<const unsigned*>(&v); for (unsigned k = 0; k < sizeof(T) / 4; ++k) acc ^= w[k]; }
  if (acc == 0x12345678u) out[0] = acc;
}
// one pass of the staging pattern over n_cells cells of 2048 B: tile t reads bytes [124t, 124t+124) of each cell
__global__ void k_stage_pattern(const unsigned char* __restrict__ cells, size_t n_cells, unsigned* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t cell0 = (size_t)blockIdx.x * 256 + (size_t)wave * 64;
  unsigned acc = 0;
  for (int tile = 0; tile < 17; ++tile) {
    for (int k = 0; k < 31; ++k) {
      int idx = k * 64 + lane, c = idx / 31, w = idx - c * 31;
      size_t p0 = (size_t)tile * 124 + (size_t)w * 4;
      size_t cell = cell0 + c;
      if (cell < n_cells && p0 + 4 <= 2048) acc ^= *reinterpret_cast<const unsigned*>(cells + cell * 2048 + p0);
    }
    __syncthreads();
  }
  if (acc == 0x12345678u) out[0] = acc;
}
int main() {
  const size_t bytes = (size_t)1 << 30;
  void* p; unsigned* out;
  CK(hipMalloc(&p, bytes)); CK(hipMalloc(&out, 4));
  CK(hipMemset(p, 1, bytes));
  CK(hipDeviceSynchronize());
  k_read<unsigned><<<8192, 256>>>((const unsigned*)p, bytes / 4, out);
  k_read<uint2><<<8192, 256>>>((const uint2*)p, bytes / 8, out);
  k_read<uint4><<<8192, 256>>>((const uint4*)p, bytes / 16, out);
  k_stage_pattern<<<(unsigned)((bytes / 2048 + 255) / 256), 256>>>((const unsigned char*)p, bytes / 2048, out);
  CK(hipDeviceSynchronize());
  printf("done: each kernel read 1 GiB = 1048576 KiB once\n");
  return 0;
}

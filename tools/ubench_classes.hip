// Saturated issue cost of every VALU opcode that matters in k_permute_batch's stream (tools/valu_roof.py lists them: these
// eleven are 99.6 % of it), measured WITH THE SAME COUNTERS the kernel itself is measured with: run this binary under
//   rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d <dir> -- ./ubench_classes
// and tools/ubench_classes_summarize.py turns the counter CSV into cycles per wave-instruction per SIMD
// (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs / SQ_INSTS_VALU), one figure per kernel = per opcode.
// Every kernel runs a loop of 32 independent instructions of ONE kind (inline asm, 8 accumulators); the grid puts
// W waves on every SIMD (256 CUs x 4 SIMDs): W = 4 is k_permute_batch's own occupancy (102 VGPRs -> 4 waves: the compiler's
// kernel-resource-usage remark and SQ_WAVE_CYCLES agree; rounds 1-4 said 5), W = 3 k_hash_cells' (LDS ring: three workgroups per CU),
// W = 5 and 8 (the most a SIMD holds) for the trend.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITER = 1 << 15;   // x 32 instructions: about 1e6 wave-instructions per wave, 5-10 ms per launch

#define UB32(NAME, ASMSTR)                                                                      \
__global__ void __launch_bounds__(256) NAME(unsigned* out, unsigned a, unsigned b) {            \
  unsigned r[8]; unsigned x = a + threadIdx.x, y = b ^ threadIdx.x;                             \
  _Pragma("unroll") for (int i = 0; i < 8; ++i) r[i] = x * (i + 1) + y;                        \
  for (int it = 0; it < ITER; ++it) {                                                           \
    _Pragma("unroll") for (int u = 0; u < 4; ++u)                                               \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                               \
      asm volatile(ASMSTR : "+v"(r[i]) : "v"(x), "v"(y));                                       \
  }                                                                                             \
  unsigned s = 0; _Pragma("unroll") for (int i = 0; i < 8; ++i) s ^= r[i];                      \
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                               \
}
#define UB64(NAME, ASMSTR)                                                                      \
__global__ void __launch_bounds__(256) NAME(unsigned* out, unsigned a, unsigned b) {            \
  unsigned long long r[8]; unsigned x = a + threadIdx.x, y = b ^ threadIdx.x;                   \
  unsigned long long z = ((unsigned long long)x << 32) | y;                                     \
  _Pragma("unroll") for (int i = 0; i < 8; ++i) r[i] = (unsigned long long)x * (i + 1) + y;    \
  for (int it = 0; it < ITER; ++it) {                                                           \
    _Pragma("unroll") for (int u = 0; u < 4; ++u)                                               \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                               \
      asm volatile(ASMSTR : "+v"(r[i]) : "v"(x), "v"(y), "v"(z) : "vcc");                       \
  }                                                                                             \
  unsigned long long s = 0; _Pragma("unroll") for (int i = 0; i < 8; ++i) s ^= r[i];            \
  out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(s ^ (s >> 32));                       \
}

UB64(ub_v_mad_u64_u32,  "v_mad_u64_u32 %0, vcc, %1, %2, %0")
UB64(ub_v_lshrrev_b64,  "v_lshrrev_b64 %0, 29, %0")
UB64(ub_v_lshl_add_u64, "v_lshl_add_u64 %0, %3, 0, %0")
UB32(ub_v_and_b32,      "v_and_b32 %0, %1, %0")
UB32(ub_v_mul_lo_u32,   "v_mul_lo_u32 %0, %1, %0")
UB32(ub_v_lshlrev_b32,  "v_lshlrev_b32 %0, 3, %0")
UB32(ub_v_mov_b32,      "v_mov_b32 %0, %1")
UB32(ub_v_add_u32,      "v_add_u32 %0, %1, %0")
UB32(ub_v_lshrrev_b32,  "v_lshrrev_b32 %0, 29, %0")
UB32(ub_v_alignbit_b32, "v_alignbit_b32 %0, %1, %0, 7")
UB32(ub_v_add3_u32,     "v_add3_u32 %0, %1, %2, %0")

typedef void (*kern_t)(unsigned*, unsigned, unsigned);
struct Entry { const char* name; kern_t k; };

int main() {
  const Entry ks[] = {{"v_mad_u64_u32", ub_v_mad_u64_u32}, {"v_lshrrev_b64", ub_v_lshrrev_b64}, {"v_lshl_add_u64", ub_v_lshl_add_u64},
                      {"v_and_b32", ub_v_and_b32}, {"v_mul_lo_u32", ub_v_mul_lo_u32}, {"v_lshlrev_b32", ub_v_lshlrev_b32},
                      {"v_mov_b32", ub_v_mov_b32}, {"v_add_u32", ub_v_add_u32}, {"v_lshrrev_b32", ub_v_lshrrev_b32},
                      {"v_alignbit_b32", ub_v_alignbit_b32}, {"v_add3_u32", ub_v_add3_u32}};
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  unsigned* out = nullptr;
  CK(hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(unsigned)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("device %s  CUs=%d  clock=%d kHz; wall-derived cycles per wave-instruction per SIMD at that nominal clock (the PMC figures come from the profiler)\n",
         prop.gcnArchName, cus, prop.clockRate);
  const int waves[] = {3, 4, 5, 8};
  for (int w : waves) {                                 // grid = cus * w blocks of 4 waves: w waves on every SIMD
    for (const Entry& e : ks) {
      hipLaunchKernelGGL(e.k, dim3(cus * w), dim3(256), 0, 0, out, 3u, 5u);   // warm-up (also the first launch's code load)
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(e.k, dim3(cus * w), dim3(256), 0, 0, out, 3u, 5u);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double insts_per_simd = (double)w * ITER * 32;
      printf("W=%d  %-16s %8.3f ms  %.3f cycles per instruction (wall x nominal clock)\n", w, e.name, ms,
             ms * 1e-3 * prop.clockRate * 1e3 / insts_per_simd);
    }
  }
  CK(hipFree(out));
  return 0;
}

// VALU issue-rate microbenchmark for gfx950: decides the limb representation of the Fr multiplier.
// Each kernel runs ITER iterations of 8 independent instructions of one kind (inline asm, so the
// measured opcode is exactly the one named).  Reports wave-instructions / ns and cycles per
// wave-instruction per SIMD relative to a 2.4 GHz clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITER = 8192;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// ---- 32-bit accumulators -------------------------------------------------
#define KERNEL32(NAME, ASMSTR)                                                   \
__global__ void __launch_bounds__(256) NAME(unsigned* out, unsigned long long* cyc, unsigned a, unsigned b) { \
  unsigned r[8]; unsigned x = a + threadIdx.x, y = b ^ threadIdx.x;              \
  _Pragma("unroll") for (int i = 0; i < 8; ++i) r[i] = x * (i + 1) + y;         \
  unsigned long long t0 = __builtin_amdgcn_s_memtime();                          \
  for (int it = 0; it < ITER / 4; ++it) {                                        \
    _Pragma("unroll") for (int u = 0; u < 4; ++u)                                \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                \
      asm volatile(ASMSTR : "+v"(r[i]) : "v"(x), "v"(y) : "vcc");                \
  }                                                                              \
  unsigned long long t1 = __builtin_amdgcn_s_memtime();                          \
  unsigned s = 0; _Pragma("unroll") for (int i = 0; i < 8; ++i) s ^= r[i];       \
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                \
  if (threadIdx.x % 64 == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;   \
}

KERNEL32(k_fma_f32,        "v_fma_f32 %0, %1, %2, %0")
KERNEL32(k_add_u32,        "v_add_u32 %0, %1, %0")
KERNEL32(k_fma_f32_2op,    "v_fma_f32 %0, %1, %1, %0")
KERNEL32(k_fmac_f32,       "v_fmac_f32 %0, %1, %2")
KERNEL32(k_mul_lo_self,    "v_mul_lo_u32 %0, %0, %0")
KERNEL32(k_sub_u32,        "v_sub_u32 %0, %1, %0")
KERNEL32(k_xor,            "v_xor_b32 %0, %1, %0")
KERNEL32(k_lshlrev,        "v_lshlrev_b32 %0, 3, %0")
KERNEL32(k_cndmask,        "v_cndmask_b32 %0, %1, %0, vcc")
KERNEL32(k_and_b32,        "v_and_b32 %0, %1, %0")
KERNEL32(k_and_lit,        "v_and_b32 %0, 0x1fffffff, %0")
KERNEL32(k_or_b32,         "v_or_b32 %0, %1, %0")
KERNEL32(k_mov_b32,        "v_mov_b32 %0, %1")
KERNEL32(k_lshrrev_reg,    "v_lshrrev_b32 %0, %1, %0")
KERNEL32(k_lshrrev_imm,    "v_lshrrev_b32 %0, 29, %0")
KERNEL32(k_min_u32,        "v_min_u32 %0, %1, %0")
KERNEL32(k_bfe_u32,        "v_bfe_u32 %0, %0, 3, 20")
KERNEL32(k_lshl_add_u32,   "v_lshl_add_u32 %0, %1, 3, %0")
KERNEL32(k_add_f32,        "v_add_f32 %0, %1, %0")
KERNEL32(k_mul_f32,        "v_mul_f32 %0, %1, %0")
KERNEL32(k_subrev,         "v_subrev_u32 %0, %1, %0")
KERNEL32(k_add3_u32,       "v_add3_u32 %0, %1, %2, %0")
KERNEL32(k_add_co,         "v_add_co_u32 %0, vcc, %1, %0")
KERNEL32(k_addc_co,        "v_addc_co_u32 %0, vcc, %1, %0, vcc")
KERNEL32(k_mul_lo_u32,     "v_mul_lo_u32 %0, %1, %0")
KERNEL32(k_mul_hi_u32,     "v_mul_hi_u32 %0, %1, %0")
KERNEL32(k_mul_u32_u24,    "v_mul_u32_u24 %0, %1, %0")
KERNEL32(k_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %1, %0")
KERNEL32(k_mad_u32_u24,    "v_mad_u32_u24 %0, %1, %2, %0")
KERNEL32(k_mad_u32_u16,    "v_mad_u32_u16 %0, %1, %2, %0")
KERNEL32(k_pk_mad_u16,     "v_pk_mad_u16 %0, %1, %2, %0")
KERNEL32(k_pk_mul_lo_u16,  "v_pk_mul_lo_u16 %0, %1, %0")
KERNEL32(k_dot4_u32_u8,    "v_dot4_u32_u8 %0, %1, %2, %0")
KERNEL32(k_dot2_u32_u16,   "v_dot2_u32_u16 %0, %1, %2, %0")
KERNEL32(k_alignbit,       "v_alignbit_b32 %0, %1, %0, 7")
KERNEL32(k_lshl_or,        "v_lshl_or_b32 %0, %1, 3, %0")
KERNEL32(k_and_or,         "v_and_or_b32 %0, %1, %2, %0")
KERNEL32(k_xad,            "v_xad_u32 %0, %1, %2, %0")

// ---- 64-bit accumulators -------------------------------------------------
#define KERNEL64(NAME, ASMSTR)                                                   \
__global__ void __launch_bounds__(256) NAME(unsigned* out, unsigned long long* cyc, unsigned a, unsigned b) { \
  unsigned long long r[8]; unsigned x = a + threadIdx.x, y = b ^ threadIdx.x;    \
  unsigned long long z = ((unsigned long long)x << 32) | y;                      \
  _Pragma("unroll") for (int i = 0; i < 8; ++i) r[i] = (unsigned long long)x * (i + 1) + y; \
  unsigned long long t0 = __builtin_amdgcn_s_memtime();                          \
  for (int it = 0; it < ITER / 4; ++it) {                                        \
    _Pragma("unroll") for (int u = 0; u < 4; ++u)                                \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                \
      asm volatile(ASMSTR : "+v"(r[i]) : "v"(x), "v"(y), "v"(z) : "vcc");        \
  }                                                                              \
  unsigned long long t1 = __builtin_amdgcn_s_memtime();                          \
  unsigned long long s = 0; _Pragma("unroll") for (int i = 0; i < 8; ++i) s ^= r[i]; \
  out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(s ^ (s >> 32));        \
  if (threadIdx.x % 64 == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;   \
}

KERNEL64(k_mad_u64_u32,  "v_mad_u64_u32 %0, vcc, %1, %2, %0")
KERNEL64(k_mad_i64_i32,  "v_mad_i64_i32 %0, vcc, %1, %2, %0")
KERNEL64(k_lshl_add_u64, "v_lshl_add_u64 %0, %3, 0, %0")
KERNEL64(k_fma_f64,      "v_fma_f64 %0, %3, %3, %0")
KERNEL64(k_mul_f64,      "v_mul_f64 %0, %3, %0")
KERNEL64(k_add_f64,      "v_add_f64 %0, %3, %0")
KERNEL64(k_pk_fma_f32,   "v_pk_fma_f32 %0, %3, %3, %0")

// v_mad_u64_u32 operand-form variants: SGPR multiplicand, non-VCC carry-out pair, zero addend
__global__ void __launch_bounds__(256) k_mad_sgpr(unsigned* out, unsigned long long* cyc, unsigned a, unsigned b) {
  unsigned long long r[8]; unsigned x = a + threadIdx.x;
  _Pragma("unroll") for (int i = 0; i < 8; ++i) r[i] = (unsigned long long)x * (i + 1) + b;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER / 4; ++it) {
    _Pragma("unroll") for (int u = 0; u < 4; ++u)
    _Pragma("unroll") for (int i = 0; i < 8; ++i)
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r[i]) : "v"(x), "s"(b) : "vcc");
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long s = 0; _Pragma("unroll") for (int i = 0; i < 8; ++i) s ^= r[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(s ^ (s >> 32));
  if (threadIdx.x % 64 == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}
__global__ void __launch_bounds__(256) k_mad_sdst(unsigned* out, unsigned long long* cyc, unsigned a, unsigned b) {
  unsigned long long r[8]; unsigned x = a + threadIdx.x, y = b ^ threadIdx.x;
  _Pragma("unroll") for (int i = 0; i < 8; ++i) r[i] = (unsigned long long)x * (i + 1) + y;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER / 4; ++it) {
    _Pragma("unroll") for (int u = 0; u < 4; ++u)
    _Pragma("unroll") for (int i = 0; i < 8; ++i)
      asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(r[i]) : "v"(x), "v"(y) : "s20", "s21");
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long s = 0; _Pragma("unroll") for (int i = 0; i < 8; ++i) s ^= r[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(s ^ (s >> 32));
  if (threadIdx.x % 64 == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}
__global__ void __launch_bounds__(256) k_mad_chain(unsigned* out, unsigned long long* cyc, unsigned a, unsigned b) {
  // ONE dependent chain per wave (what the tie-ordered multiplier looks like): latency-bound unless enough waves
  unsigned long long r = a; unsigned x = a + threadIdx.x, y = b ^ threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER / 4; ++it) {
    _Pragma("unroll") for (int u = 0; u < 32; ++u)
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r) : "v"(x), "v"(y) : "vcc");
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(r ^ (r >> 32));
  if (threadIdx.x % 64 == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}

struct Entry { const char* name; void (*fn)(unsigned*, unsigned long long*, unsigned, unsigned); int instr_per_slot; };

int main(int argc, char** argv) {
  int dev = 0; CK(hipSetDevice(dev));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
  printf("device %s CUs=%d clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  const int cus = prop.multiProcessorCount;
  std::vector<Entry> es = {
    {"v_fma_f32", k_fma_f32, 1}, {"v_add_u32", k_add_u32, 1}, {"v_fma_f32 2op", k_fma_f32_2op, 1}, {"v_fmac_f32", k_fmac_f32, 1}, {"v_mul_lo_u32 self", k_mul_lo_self, 1}, {"v_sub_u32", k_sub_u32, 1}, {"v_xor_b32", k_xor, 1}, {"v_lshlrev_b32", k_lshlrev, 1}, {"v_cndmask_b32", k_cndmask, 1}, {"v_and_b32", k_and_b32, 1}, {"v_and_b32 literal", k_and_lit, 1}, {"v_or_b32", k_or_b32, 1}, {"v_mov_b32", k_mov_b32, 1}, {"v_lshrrev_b32 reg", k_lshrrev_reg, 1}, {"v_lshrrev_b32 imm", k_lshrrev_imm, 1}, {"v_min_u32", k_min_u32, 1}, {"v_bfe_u32", k_bfe_u32, 1}, {"v_lshl_add_u32", k_lshl_add_u32, 1}, {"v_add_f32", k_add_f32, 1}, {"v_mul_f32", k_mul_f32, 1}, {"v_subrev_u32", k_subrev, 1}, {"v_add3_u32", k_add3_u32, 1},
    {"v_add_co_u32", k_add_co, 1}, {"v_addc_co_u32", k_addc_co, 1},
    {"v_mul_lo_u32", k_mul_lo_u32, 1}, {"v_mul_hi_u32", k_mul_hi_u32, 1},
    {"v_mul_u32_u24", k_mul_u32_u24, 1}, {"v_mul_hi_u32_u24", k_mul_hi_u32_u24, 1},
    {"v_mad_u32_u24", k_mad_u32_u24, 1}, {"v_mad_u32_u16", k_mad_u32_u16, 1},
    {"v_pk_mad_u16", k_pk_mad_u16, 1}, {"v_pk_mul_lo_u16", k_pk_mul_lo_u16, 1},
    {"v_dot4_u32_u8", k_dot4_u32_u8, 1}, {"v_dot2_u32_u16", k_dot2_u32_u16, 1},
    {"v_alignbit_b32", k_alignbit, 1}, {"v_lshl_or_b32", k_lshl_or, 1}, {"v_and_or_b32", k_and_or, 1},
    {"v_xad_u32", k_xad, 1},
    {"v_mad_u64_u32", k_mad_u64_u32, 1}, {"v_mad_i64_i32", k_mad_i64_i32, 1},
    {"v_mad_u64 sgpr src", k_mad_sgpr, 1}, {"v_mad_u64 s[20:21]", k_mad_sdst, 1}, {"v_mad_u64 1 chain", k_mad_chain, 1}, {"v_lshl_add_u64", k_lshl_add_u64, 1}, {"v_fma_f64", k_fma_f64, 1}, {"v_mul_f64", k_mul_f64, 1},
    {"v_add_f64", k_add_f64, 1}, {"v_pk_fma_f32", k_pk_fma_f32, 1},
  };
  unsigned* out; CK(hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(unsigned)));
  unsigned long long* cyc; CK(hipMalloc(&cyc, (size_t)cus * 8 * 4 * sizeof(unsigned long long)));
  std::vector<unsigned long long> hc((size_t)cus * 8 * 4);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // warm the clocks: ~1 s of back-to-back work
  for (int i = 0; i < 400; ++i) k_fma_f32<<<cus * 8, 256>>>(out, cyc, 1, 2);
  CK(hipDeviceSynchronize());
  int occs[] = {1, 2, 4, 8};
  printf("cycles per wave-instruction per SIMD: in-kernel s_memtime / wall-derived at 2.4 GHz (cycle-count over wall, GHz)\n");
  printf("%-20s", "instr");
  for (int o : occs) printf("   w/SIMD=%d  (GHz)", o);
  printf("\n");
  for (auto& e : es) {
    printf("%-20s", e.name);
    for (int o : occs) {
      int grid = cus * o;
      for (int i = 0; i < 3; ++i) e.fn<<<grid, 256>>>(out, cyc, 1, 2);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      const int reps = 5;
      for (int i = 0; i < reps; ++i) e.fn<<<grid, 256>>>(out, cyc, 3, 5);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
      CK(hipMemcpy(hc.data(), cyc, (size_t)grid * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      double sum = 0; for (int i = 0; i < grid * 4; ++i) sum += (double)hc[i];
      double avg = sum / (grid * 4);                       // cycles per wave for ITER*8 instrs
      double per_simd = avg / ((double)ITER * 8) / o;      // o waves share a SIMD
      double ghz = avg / (ms * 1e-3) * 1e-9;               // approx: kernel wall ~ wave lifetime
      double wall_cyc = ms * 1e-3 * 2.4e9 / ((double)ITER * 8 * o); printf("   %5.2f/%5.2f(%4.2f)", per_simd, wall_cyc, ghz);
    }
    printf("\n");
    fflush(stdout);
  }
  return 0;
}

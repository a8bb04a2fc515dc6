// Microbenchmark: per-instruction VALU issue rates on gfx950 that decide the field-arithmetic
// representation (32-bit limb MADs vs 24-bit vs fp64). Not part of the product; measurement tool.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITERS = 2048;
constexpr int CHAINS = 8;   // independent dependency chains per lane
constexpr int UNROLL = 4;   // instrs per chain per loop iteration

// 32-bit in/out instruction:  d = op(d, a, b)
#define KERNEL32(NAME, ASM)                                                                 \
__global__ void __launch_bounds__(256) k_##NAME(uint32_t* out, uint32_t a0, uint32_t b0) {     \
  uint32_t r[CHAINS];                                                                        \
  uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;                                       \
  for (int c = 0; c < CHAINS; ++c) r[c] = a * (c + 1) + b;                                    \
  for (int it = 0; it < ITERS; ++it) {                                                       \
    _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                                     \
      _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) {                                   \
        asm volatile(ASM : "+v"(r[c]) : "v"(a), "v"(b));                                      \
      }                                                                                      \
    }                                                                                        \
  }                                                                                          \
  uint32_t s = 0;                                                                            \
  for (int c = 0; c < CHAINS; ++c) s ^= r[c];                                                \
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                            \
}

KERNEL32(add_u32,        "v_add_u32 %0, %0, %1")
KERNEL32(add3_u32,       "v_add3_u32 %0, %0, %1, %2")
KERNEL32(fma_f32,        "v_fma_f32 %0, %0, %1, %2")
KERNEL32(mul_lo_u32,     "v_mul_lo_u32 %0, %0, %1")
KERNEL32(mul_hi_u32,     "v_mul_hi_u32 %0, %0, %1")
KERNEL32(mad_u32_u24,    "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL32(mul_u32_u24,    "v_mul_u32_u24 %0, %0, %1")
KERNEL32(mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %0, %1")
KERNEL32(mad_u32_u16,    "v_mad_u32_u16 %0, %0, %1, %2")
KERNEL32(pk_mul_lo_u16,  "v_pk_mul_lo_u16 %0, %0, %1")
KERNEL32(pk_mad_u16,     "v_pk_mad_u16 %0, %0, %1, %2")
KERNEL32(dot4_u32_u8,    "v_dot4_u32_u8 %0, %0, %1, %2")
KERNEL32(dot2_u32_u16,   "v_dot2_u32_u16 %0, %0, %1, %2")
KERNEL32(alignbit,       "v_alignbit_b32 %0, %0, %1, 13")
KERNEL32(and_or,         "v_and_or_b32 %0, %0, %1, %2")
KERNEL32(lshl_add,       "v_lshl_add_u32 %0, %0, 3, %1")
KERNEL32(addc_pair,      "v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %2, vcc")

// 64-bit accumulate:  d(64) = op(a32, b32, d64)
#define KERNEL64(NAME, ASM)                                                                 \
__global__ void __launch_bounds__(256) k_##NAME(uint32_t* out, uint32_t a0, uint32_t b0) {     \
  unsigned long long r[CHAINS];                                                              \
  uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;                                       \
  unsigned long long a64 = ((unsigned long long)a << 32) | b;                                \
  for (int c = 0; c < CHAINS; ++c) r[c] = a64 * (c + 1);                                      \
  for (int it = 0; it < ITERS; ++it) {                                                       \
    _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                                     \
      _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) {                                   \
        asm volatile(ASM : "+v"(r[c]) : "v"(a), "v"(b), "v"(a64) : "vcc");                    \
      }                                                                                      \
    }                                                                                        \
  }                                                                                          \
  unsigned long long s = 0;                                                                  \
  for (int c = 0; c < CHAINS; ++c) s ^= r[c];                                                \
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(s ^ (s >> 32));                    \
}

KERNEL64(mad_u64_u32,  "v_mad_u64_u32 %0, vcc, %1, %2, %0")
KERNEL64(mad_i64_i32,  "v_mad_i64_i32 %0, vcc, %1, %2, %0")
KERNEL64(fma_f64,      "v_fma_f64 %0, %0, %3, %3")
KERNEL64(add_f64,      "v_add_f64 %0, %0, %3")
KERNEL64(mul_f64,      "v_mul_f64 %0, %0, %3")
KERNEL64(lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %3")
KERNEL64(lshlrev_b64,  "v_lshlrev_b64 %0, 1, %0")
KERNEL64(pk_fma_f32,   "v_pk_fma_f32 %0, %0, %3, %3")
KERNEL64(pk_mul_f32,   "v_pk_mul_f32 %0, %0, %3")
KERNEL64(pk_add_f32,   "v_pk_add_f32 %0, %0, %3")

typedef void (*kern_t)(uint32_t*, uint32_t, uint32_t);
struct Entry { const char* name; kern_t k; int instr_per_step; };

int main(int argc, char** argv) {
  int dev = 0; CK(hipSetDevice(dev));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
  int cus = prop.multiProcessorCount;
  printf("device %s CUs %d clock %d kHz\n", prop.name, cus, prop.clockRate);
  std::vector<Entry> es = {
    {"v_add_u32", k_add_u32, 1}, {"v_add3_u32", k_add3_u32, 1}, {"v_fma_f32", k_fma_f32, 1},
    {"v_mul_lo_u32", k_mul_lo_u32, 1}, {"v_mul_hi_u32", k_mul_hi_u32, 1},
    {"v_mad_u32_u24", k_mad_u32_u24, 1}, {"v_mul_u32_u24", k_mul_u32_u24, 1},
    {"v_mul_hi_u32_u24", k_mul_hi_u32_u24, 1}, {"v_mad_u32_u16", k_mad_u32_u16, 1},
    {"v_pk_mul_lo_u16", k_pk_mul_lo_u16, 1}, {"v_pk_mad_u16", k_pk_mad_u16, 1},
    {"v_dot4_u32_u8", k_dot4_u32_u8, 1}, {"v_dot2_u32_u16", k_dot2_u32_u16, 1},
    {"v_alignbit_b32", k_alignbit, 1}, {"v_and_or_b32", k_and_or, 1}, {"v_lshl_add_u32", k_lshl_add, 1},
    {"add_co+addc_co", k_addc_pair, 2},
    {"v_mad_u64_u32", k_mad_u64_u32, 1}, {"v_mad_i64_i32", k_mad_i64_i32, 1},
    {"v_fma_f64", k_fma_f64, 1}, {"v_add_f64", k_add_f64, 1}, {"v_mul_f64", k_mul_f64, 1},
    {"v_lshl_add_u64", k_lshl_add_u64, 1}, {"v_lshlrev_b64", k_lshlrev_b64, 1},
    {"v_pk_fma_f32", k_pk_fma_f32, 1}, {"v_pk_mul_f32", k_pk_mul_f32, 1}, {"v_pk_add_f32", k_pk_add_f32, 1},
  };
  int waves_per_simd_list[] = {1, 2, 4, 8};
  uint32_t* out; CK(hipMalloc(&out, sizeof(uint32_t) * cus * 8 * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%-18s", "instr");
  for (int w : waves_per_simd_list) printf("  w/simd=%d: Glane-op/s  cyc/wave-instr", w);
  printf("\n");
  for (auto& e : es) {
    printf("%-18s", e.name);
    for (int w : waves_per_simd_list) {
      // 256 threads = 4 waves = 1 wave per SIMD per block; w blocks per CU
      int blocks = cus * w;
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 12345u, 6789u);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      const int reps = 3;
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 12345u, 6789u);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
      double wave_instr_per_simd = (double)ITERS * UNROLL * CHAINS * e.instr_per_step * w;
      double lane_ops = wave_instr_per_simd * 64.0 * cus * 4;
      double glops = lane_ops / (ms * 1e-3) / 1e9;
      // cycles per wave-instruction per SIMD at nominal 2.4 GHz
      double cyc = (ms * 1e-3) * 2.4e9 / wave_instr_per_simd;
      printf("  %10.1f %8.2f        ", glops, cyc);
    }
    printf("\n");
  }
  return 0;
}

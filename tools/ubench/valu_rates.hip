// Microbenchmark (measurement tool, not product): VALU issue cost per wave64 instruction on gfx950, the denominator of the VALU
// roofline (bench.py MAD_PEAK_T).
// Round 3 rewrite ("What's weak" 2): the first version ran 0.1-0.3 ms kernels, 8 chains, and converted time to cycles at a
// NOMINAL 2.4 GHz; it read v_fma_f32 at 3.7 cycles where the architecture does 2.  This one
//   * runs >= 20 ms per kernel, 16 independent chains per lane (no dependency stalls at any tested occupancy),
//   * takes cycles from s_memtime inside the kernel (shader clock: immune to DVFS) and reports the clock the chip held
//     (s_memtime / s_memrealtime x 100 MHz),
//   * times every instruction in two operand shapes: all sources in VGPRs ("vvv") and one source an SGPR / inline constant ("vvs"),
//     because a three-VGPR-source VOP3 can cost a second operand-read cycle.
// Reported per instruction and occupancy: issue cycles per wave-instruction and SIMD FROM THE WALL CLOCK (kernel time x measured clock
// / instructions issued on a SIMD) and the chip-wide rate in T lane-ops/s; the median wave's own s_memtime span is printed beside it
// ("wave"): it reads LOWER than the wall-clock figure above two waves per SIMD because issue is arbitrated oldest-first, so older
// waves finish early and the kernel's tail runs at lower occupancy - the wall-clock figure is the throughput.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/valu_rates tools/ubench/valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <string>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef unsigned long long u64;
struct Stamp { u64 cyc, rt; };
constexpr int CHAINS = 16;
constexpr int UNROLL = 4;

#define PROLOGUE                                                                                   \
  const u64 c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#define EPILOGUE                                                                                   \
  const u64 c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();              \
  if ((threadIdx.x & 63) == 0) { Stamp s; s.cyc = c1 - c0; s.rt = r1 - r0; st[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = s; }

// One asm statement carries all 16 chains (hipcc puts an s_nop after EVERY inline-asm statement, 4 issue cycles each: with one
// instruction per statement the first version of this rewrite measured the s_nops, not the instructions).
#define R16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
#define OUT16(r) "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), \
                 "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
// operands: %0..%15 chains, %16 = a (VGPR), %17 = b (VGPR), %18 = s (SGPR), %19 = SGPR pair (32-bit kernels) / a64 (VGPR pair, 64-bit kernels)
#define KERNEL32(NAME, I)                                                                          \
__global__ void __launch_bounds__(256) k_##NAME(uint32_t* out, Stamp* st, uint32_t a0, uint32_t b0, int iters) { \
  uint32_t r[CHAINS];                                                                              \
  uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x; const uint32_t s = a0 | 3u;                 \
  const u64 m64 = 0x5555aaaa3333ccccull * (u64)(a0 | 1u);                                          \
  for (int c = 0; c < CHAINS; ++c) r[c] = a * (c + 1) + b;                                         \
  PROLOGUE                                                                                         \
  for (int it = 0; it < iters; ++it) {                                                             \
    _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                                           \
      asm volatile(R16(I) : OUT16(r) : "v"(a), "v"(b), "s"(s), "s"(m64) : "vcc");                  \
    }                                                                                              \
  }                                                                                                \
  EPILOGUE                                                                                         \
  uint32_t x = 0;                                                                                  \
  for (int c = 0; c < CHAINS; ++c) x ^= r[c];                                                      \
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;                                                  \
}
#define I_add_u32(n)      "v_add_u32 %" #n ", %" #n ", %16\n\t"
#define I_and_b32(n)      "v_and_b32 %" #n ", %" #n ", %16\n\t"
#define I_lshrrev_b32(n)  "v_lshrrev_b32 %" #n ", 3, %" #n "\n\t"
#define I_fmac_f32(n)     "v_fmac_f32 %" #n ", %16, %17\n\t"
#define I_fma_vvv(n)      "v_fma_f32 %" #n ", %" #n ", %16, %17\n\t"
#define I_fma_vvs(n)      "v_fma_f32 %" #n ", %" #n ", %16, %18\n\t"
#define I_add3_vvv(n)     "v_add3_u32 %" #n ", %" #n ", %16, %17\n\t"
#define I_add3_vvs(n)     "v_add3_u32 %" #n ", %" #n ", %16, %18\n\t"
#define I_mul_lo(n)       "v_mul_lo_u32 %" #n ", %" #n ", %16\n\t"
#define I_mul_lo_k(n)     "v_mul_lo_u32 %" #n ", %" #n ", 19\n\t"
#define I_mul_hi(n)       "v_mul_hi_u32 %" #n ", %" #n ", %16\n\t"
#define I_mad24_vvv(n)    "v_mad_u32_u24 %" #n ", %" #n ", %16, %17\n\t"
#define I_mad24_vkv(n)    "v_mad_u32_u24 %" #n ", %" #n ", 19, %16\n\t"
#define I_mul24(n)        "v_mul_u32_u24 %" #n ", %" #n ", %16\n\t"
#define I_alignbit(n)     "v_alignbit_b32 %" #n ", %" #n ", %16, 13\n\t"
#define I_lshl_add(n)     "v_lshl_add_u32 %" #n ", %" #n ", 4, %16\n\t"
#define I_and_or(n)       "v_and_or_b32 %" #n ", %" #n ", %16, %17\n\t"
#define I_bfe(n)          "v_bfe_u32 %" #n ", %" #n ", 3, 26\n\t"
#define I_cndmask(n)      "v_cndmask_b32 %" #n ", %" #n ", %16, vcc\n\t"
#define I_addc_pair(n)    "v_add_co_u32 %" #n ", vcc, %" #n ", %16\n\tv_addc_co_u32 %" #n ", vcc, %" #n ", %17, vcc\n\t"
KERNEL32(add_u32, I_add_u32)
KERNEL32(and_b32, I_and_b32)
KERNEL32(lshrrev_b32, I_lshrrev_b32)
KERNEL32(fmac_f32, I_fmac_f32)
KERNEL32(fma_f32_vvv, I_fma_vvv)
KERNEL32(fma_f32_vvs, I_fma_vvs)
KERNEL32(add3_u32_vvv, I_add3_vvv)
KERNEL32(add3_u32_vvs, I_add3_vvs)
KERNEL32(mul_lo_u32, I_mul_lo)
KERNEL32(mul_lo_u32_k, I_mul_lo_k)
KERNEL32(mul_hi_u32, I_mul_hi)
KERNEL32(mad_u32_u24_vvv, I_mad24_vvv)
KERNEL32(mad_u32_u24_vks, I_mad24_vkv)
KERNEL32(mul_u32_u24, I_mul24)
KERNEL32(alignbit_vvk, I_alignbit)
KERNEL32(lshl_add_u32, I_lshl_add)
KERNEL32(and_or_vvv, I_and_or)
KERNEL32(bfe_u32, I_bfe)
KERNEL32(cndmask, I_cndmask)
#define I_cndmask_e64(n)  "v_cndmask_b32_e64 %" #n ", %" #n ", %16, %19\n\t"
#define I_bfi(n)          "v_bfi_b32 %" #n ", %17, %16, %" #n "\n\t"
#define I_xor(n)          "v_xor_b32 %" #n ", %" #n ", %16\n\t"
#define I_sub(n)          "v_sub_u32 %" #n ", %" #n ", %16\n\t"
#define I_lshlrev(n)      "v_lshlrev_b32 %" #n ", 1, %" #n "\n\t"
#define I_mov(n)          "v_mov_b32 %" #n ", %16\n\t"
KERNEL32(cndmask_e64, I_cndmask_e64)
KERNEL32(bfi, I_bfi)

// Round 4: is the 22.9-cycle v_cndmask_b32 (VOP2, vcc) row real?  Variants: vcc set by an s_mov_b64 before the loop,
// the same select in VOP3 encoding reading vcc, a v_cmp in every block of 16, selects interleaved with plain adds, selects without a
// dependency chain, and the mask arithmetic that could replace a select.
#define KERNEL32P(NAME, I, PRE)                                                                    \
__global__ void __launch_bounds__(256) k_##NAME(uint32_t* out, Stamp* st, uint32_t a0, uint32_t b0, int iters) { \
  uint32_t r[CHAINS];                                                                              \
  uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x; const uint32_t s = a0 | 3u;                 \
  const u64 m64 = 0x5555aaaa3333ccccull * (u64)(a0 | 1u);                                          \
  for (int c = 0; c < CHAINS; ++c) r[c] = a * (c + 1) + b;                                         \
  PROLOGUE                                                                                         \
  for (int it = 0; it < iters; ++it) {                                                             \
    _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                                           \
      asm volatile(PRE R16(I) : OUT16(r) : "v"(a), "v"(b), "s"(s), "s"(m64) : "vcc");              \
    }                                                                                              \
  }                                                                                                \
  EPILOGUE                                                                                         \
  uint32_t x = 0;                                                                                  \
  for (int c = 0; c < CHAINS; ++c) x ^= r[c];                                                      \
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;                                                  \
}
#define I_cndmask_e64_vcc(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %16, vcc\n\t"
#define I_cndmask_nodep(n)   "v_cndmask_b32 %" #n ", %16, %17, vcc\n\t"
#define I_cnd_add_mix(n)     "v_cndmask_b32 %" #n ", %" #n ", %16, vcc\n\tv_add_u32 %" #n ", %" #n ", %17\n\t"
#define I_xor_and_xor(n)     "v_xor_b32 %" #n ", %" #n ", %16\n\tv_and_b32 %" #n ", %" #n ", %17\n\tv_xor_b32 %" #n ", %" #n ", %16\n\t"
KERNEL32P(cndmask_smov, I_cndmask, "s_mov_b64 vcc, %19\n\t")
KERNEL32P(cndmask_cmp, I_cndmask, "v_cmp_lt_u32 vcc, %16, %17\n\t")
KERNEL32P(cndmask_e64_vcc, I_cndmask_e64_vcc, "s_mov_b64 vcc, %19\n\t")
KERNEL32P(cndmask_nodep, I_cndmask_nodep, "s_mov_b64 vcc, %19\n\t")
KERNEL32P(cnd_add_mix, I_cnd_add_mix, "s_mov_b64 vcc, %19\n\t")
KERNEL32(xor_and_xor, I_xor_and_xor)
KERNEL32(xor_b32, I_xor)
KERNEL32(sub_u32, I_sub)
KERNEL32(lshlrev_b32, I_lshlrev)
KERNEL32(mov_b32, I_mov)
KERNEL32(addc_pair, I_addc_pair)

#define KERNEL64(NAME, I)                                                                          \
__global__ void __launch_bounds__(256) k_##NAME(uint32_t* out, Stamp* st, uint32_t a0, uint32_t b0, int iters) { \
  u64 r[CHAINS];                                                                                   \
  uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x; const uint32_t s = a0 | 3u;                 \
  u64 a64 = ((u64)a << 32) | b;                                                                    \
  for (int c = 0; c < CHAINS; ++c) r[c] = a64 * (c + 1);                                           \
  PROLOGUE                                                                                         \
  for (int it = 0; it < iters; ++it) {                                                             \
    _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                                           \
      asm volatile(R16(I) : OUT16(r) : "v"(a), "v"(b), "s"(s), "v"(a64) : "vcc");                  \
    }                                                                                              \
  }                                                                                                \
  EPILOGUE                                                                                         \
  u64 x = 0;                                                                                       \
  for (int c = 0; c < CHAINS; ++c) x ^= r[c];                                                      \
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(x ^ (x >> 32));                          \
}
#define I_mad64_vvv(n)    "v_mad_u64_u32 %" #n ", vcc, %16, %17, %" #n "\n\t"
#define I_mad64_vsv(n)    "v_mad_u64_u32 %" #n ", vcc, %16, %18, %" #n "\n\t"
#define I_mad64_vkv(n)    "v_mad_u64_u32 %" #n ", vcc, %16, 19, %" #n "\n\t"
#define I_fma_f64(n)      "v_fma_f64 %" #n ", %" #n ", %19, %19\n\t"
#define I_lshl_add_u64(n) "v_lshl_add_u64 %" #n ", %" #n ", 0, %19\n\t"
#define I_lshrrev_b64(n)  "v_lshrrev_b64 %" #n ", 26, %" #n "\n\t"
#define I_pk_fma_f32(n)   "v_pk_fma_f32 %" #n ", %" #n ", %19, %19\n\t"
// the mix of a field multiplication's inner code: 5 MADs, a 64-bit shift and a 64-bit add per chain
#define I_mix(n)          "v_mad_u64_u32 %" #n ", vcc, %16, %17, %" #n "\n\tv_mad_u64_u32 %" #n ", vcc, %17, %16, %" #n "\n\t" \
                          "v_mad_u64_u32 %" #n ", vcc, %16, %17, %" #n "\n\tv_mad_u64_u32 %" #n ", vcc, %17, %16, %" #n "\n\t" \
                          "v_mad_u64_u32 %" #n ", vcc, %16, %17, %" #n "\n\tv_lshrrev_b64 %" #n ", 1, %" #n "\n\t"            \
                          "v_lshl_add_u64 %" #n ", %" #n ", 0, %19\n\t"
KERNEL64(mad_u64_u32_vvv, I_mad64_vvv)
KERNEL64(mad_u64_u32_vsv, I_mad64_vsv)
KERNEL64(mad_u64_u32_vkv, I_mad64_vkv)
KERNEL64(fma_f64, I_fma_f64)
KERNEL64(lshl_add_u64, I_lshl_add_u64)
KERNEL64(lshrrev_b64, I_lshrrev_b64)
KERNEL64(pk_fma_f32, I_pk_fma_f32)
KERNEL64(mix_fmul, I_mix)

typedef void (*kern_t)(uint32_t*, Stamp*, uint32_t, uint32_t, int);
struct Entry { const char* name; kern_t k; int instr_per_step; };

int main(int argc, char** argv) {
  CK(hipSetDevice(0));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double target_ms = argc > 1 ? atof(argv[1]) : 25.0;
  printf("device %s, %d CUs; >= %.0f ms per kernel, %d chains per lane; cycles from s_memtime, clock from s_memrealtime\n",
         prop.name, cus, target_ms, CHAINS);
  std::vector<Entry> es = {
    {"v_add_u32 (VOP2)", k_add_u32, 1}, {"v_and_b32 (VOP2)", k_and_b32, 1}, {"v_lshrrev_b32 (VOP2)", k_lshrrev_b32, 1},
    {"v_fmac_f32 (VOP2)", k_fmac_f32, 1}, {"v_fma_f32 v,v,v", k_fma_f32_vvv, 1}, {"v_fma_f32 v,v,s", k_fma_f32_vvs, 1},
    {"v_add3_u32 v,v,v", k_add3_u32_vvv, 1}, {"v_add3_u32 v,v,s", k_add3_u32_vvs, 1},
    {"v_mul_lo_u32 v,v", k_mul_lo_u32, 1}, {"v_mul_lo_u32 v,19", k_mul_lo_u32_k, 1}, {"v_mul_hi_u32", k_mul_hi_u32, 1},
    {"v_mad_u32_u24 v,v,v", k_mad_u32_u24_vvv, 1}, {"v_mad_u32_u24 v,19,v", k_mad_u32_u24_vks, 1}, {"v_mul_u32_u24 (VOP2)", k_mul_u32_u24, 1},
    {"v_alignbit_b32 v,v,13", k_alignbit_vvk, 1}, {"v_lshl_add_u32 v,4,v", k_lshl_add_u32, 1}, {"v_and_or_b32 v,v,v", k_and_or_vvv, 1},
    {"v_bfe_u32 v,3,26", k_bfe_u32, 1}, {"v_cndmask_b32 vcc (VOP2)", k_cndmask, 1},
    {"v_cndmask_b32_e64 v,v,s[2]", k_cndmask_e64, 1},
    {"v_cndmask_b32 vcc<-s_mov per 16", k_cndmask_smov, 1}, {"v_cndmask_b32 vcc<-v_cmp per 16", k_cndmask_cmp, 1},
    {"v_cndmask_b32_e64 v,v,vcc", k_cndmask_e64_vcc, 1}, {"v_cndmask_b32 vcc, no dep chain", k_cndmask_nodep, 1},
    {"v_cndmask_b32 + v_add_u32 pairs", k_cnd_add_mix, 2}, {"v_xor + v_and + v_xor", k_xor_and_xor, 3}, {"v_bfi_b32 v,v,v", k_bfi, 1}, {"v_xor_b32 (VOP2)", k_xor_b32, 1}, {"v_sub_u32 (VOP2)", k_sub_u32, 1},
    {"v_lshlrev_b32 (VOP2)", k_lshlrev_b32, 1}, {"v_mov_b32 (VOP1)", k_mov_b32, 1}, {"v_add_co + v_addc_co", k_addc_pair, 2},
    {"v_mad_u64_u32 v,v,v64", k_mad_u64_u32_vvv, 1}, {"v_mad_u64_u32 v,s,v64", k_mad_u64_u32_vsv, 1}, {"v_mad_u64_u32 v,19,v64", k_mad_u64_u32_vkv, 1},
    {"v_fma_f64", k_fma_f64, 1}, {"v_lshl_add_u64", k_lshl_add_u64, 1}, {"v_lshrrev_b64", k_lshrrev_b64, 1}, {"v_pk_fma_f32", k_pk_fma_f32, 1},
    {"fmul mix (5 mad, 2 x64)", k_mix_fmul, 7},
  };
  uint32_t* out; Stamp* d_st;
  CK(hipMalloc(&out, sizeof(uint32_t) * cus * 8 * 256));
  CK(hipMalloc(&d_st, sizeof(Stamp) * cus * 8 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  if (argc > 1 && std::string(argv[1]) == "sustained") {
    // The table above comes from bursts of tens of milliseconds, which the power management lets run near the top clock.  The verifier
    // issues multiply-adds for seconds on end, so its roof is the rate the chip SUSTAINS: the same kernels launched back to back for
    // `seconds`, rate and clock taken over the second half (by then the clock has settled).
    const double seconds = argc > 2 ? atof(argv[2]) : 3.0;
    printf("sustained mode: each instruction back to back for %.1f s; rate and clock over the second half\n", seconds);
    printf("%-30s | w/SIMD | T-op/s first 50 ms (GHz) | T-op/s sustained (GHz) | cycles per wave64 instruction, sustained\n", "instruction");
    for (auto& e : es) {
      const std::string nm = e.name;
      if (nm != "v_mad_u64_u32 v,v,v64" && nm != "fmul mix (5 mad, 2 x64)" && nm != "v_add_u32 (VOP2)" && nm != "v_fma_f32 v,v,v") continue;
      for (int w : {2, 8}) {
        const int blocks = cus * w;
        const double per_iter = (double)UNROLL * CHAINS * e.instr_per_step * w * 4.0 / 2.0e9;
        const int iters = (int)(25e-3 / per_iter) + 1;
        const int waves = blocks * 4;
        const double n_instr = (double)iters * UNROLL * CHAINS * e.instr_per_step;
        auto run = [&](int launches, double& tops, double& ghz_med) {
          CK(hipEventRecord(e0));
          for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, d_st, 12345u, 6789u, iters);
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          std::vector<Stamp> st(waves);
          CK(hipMemcpy(st.data(), d_st, sizeof(Stamp) * waves, hipMemcpyDeviceToHost));     // the stamps of the last launch
          std::vector<double> ghz(waves);
          for (int i = 0; i < waves; ++i) ghz[i] = (double)st[i].cyc / ((double)st[i].rt * 10.0);
          std::sort(ghz.begin(), ghz.end());
          ghz_med = ghz[waves / 2];
          tops = n_instr * w * 64.0 * cus * 4 * launches / (ms * 1e-3) / 1e12;
        };
        CK(hipDeviceSynchronize());
        double t_first, g_first, t_half, g_half, t_sus, g_sus;
        run(2, t_first, g_first);
        const int half = (int)(seconds / 2 / 25e-3) + 1;
        run(half, t_half, g_half);
        run(half, t_sus, g_sus);
        printf("%-30s | %6d | %8.2f (%4.2f)          | %8.2f (%4.2f)        | %5.2f\n", e.name, w, t_first, g_first, t_sus, g_sus,
               (double)cus * 4 * 64 * g_sus * 1e9 / (t_sus * 1e12));
        fflush(stdout);
      }
    }
    return 0;
  }
  printf("%-30s", "instruction");
  const int wlist[] = {1, 2, 4, 8};
  for (int w : wlist) printf(" | w/SIMD=%d: cyc  T-op/s (wave cyc, GHz)", w);
  printf("\n");
  const char* only = argc > 2 ? argv[2] : nullptr;         // optional: run only the rows whose name contains this substring
  for (auto& e : es) {
    if (only && !strstr(e.name, only)) continue;
    printf("%-30s", e.name);
    for (int w : wlist) {
      const int blocks = cus * w;          // 256 threads = one wave per SIMD; w blocks per CU
      // size the loop for the target time assuming ~4 cycles per instruction at 2 GHz
      const double per_iter = (double)UNROLL * CHAINS * e.instr_per_step * w * 4.0 / 2.0e9;
      const int iters = (int)(target_ms * 1e-3 / per_iter) + 1;
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, d_st, 12345u, 6789u, iters / 16 + 1);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, d_st, 12345u, 6789u, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const int waves = blocks * 4;
      std::vector<Stamp> st(waves);
      CK(hipMemcpy(st.data(), d_st, sizeof(Stamp) * waves, hipMemcpyDeviceToHost));
      std::vector<double> cyc(waves), ghz(waves);
      for (int i = 0; i < waves; ++i) { cyc[i] = (double)st[i].cyc; ghz[i] = (double)st[i].cyc / ((double)st[i].rt * 10.0); }
      std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
      const double n_instr = (double)iters * UNROLL * CHAINS * e.instr_per_step;      // per wave
      const double clock = ghz[waves / 2] * 1e9;
      const double wall_cyc = ms * 1e-3 * clock / (n_instr * w);
      const double tops = n_instr * w * 64.0 * cus * 4 / (ms * 1e-3) / 1e12;
      printf(" | %6.2f %7.2f (%5.2f, %4.2f)       ", wall_cyc, tops, cyc[waves / 2] / n_instr / w, ghz[waves / 2]);
    }
    printf("\n");
  }
  return 0;
}

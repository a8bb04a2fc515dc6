// Microbenchmark (measurement tool, not product): field / point arithmetic of the shipped representation (9 limbs, radix 2^(255/9),
// elastic_elgamal_amd/csrc/fe25519.cuh) against the one of rounds 1-2 (10 limbs, radix 2^25.5, fe10.cuh), in the shape of the dominant loop:
// one comb column = doubling -> extended point -> addition of a cached table entry (ge_teeth_mul, ge25519.cuh).
// Cycles come from s_memtime inside the kernel (shader clock, so DVFS does not distort them); the clock itself from s_memrealtime
// (100 MHz).  Outputs of the two representations are compared word for word (canonical encodings), so the bench is also a check.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I elastic_elgamal_amd/csrc -I tools/ubench -o tools/ubench/field_bench tools/ubench/field_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <string>
#include <algorithm>
#include "ge25519.cuh"     // the shipped representation (namespace eg)
#include "fe10.cuh"       // the representation of rounds 1-2 (namespace eg10)
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef uint32_t u32; typedef uint64_t u64;

struct Stamp { u64 cyc, rt; };
__device__ __forceinline__ void stamp_begin(u64& c, u64& r) { c = __builtin_amdgcn_s_memtime(); r = __builtin_amdgcn_s_memrealtime(); }
__device__ __forceinline__ void stamp_end(Stamp* st, u64 c0, u64 r0) {
  const u64 c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) { Stamp s; s.cyc = c1 - c0; s.rt = r1 - r0; st[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = s; }
}
__device__ __forceinline__ void seed_words(u32 w[8], u32 seed, u32 salt) {
  u32 x = seed ^ (salt * 0x9e3779b9u) ^ ((blockIdx.x * blockDim.x + threadIdx.x) * 0x85ebca6bu);
#pragma unroll
  for (int i = 0; i < 8; ++i) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; w[i] = x; }
  w[7] &= 0x3fffffffu;
}

// ---- mul / sq chains ----------------------------------------------------------------------------------------------------
template <class FE, class OPS>
__global__ void __launch_bounds__(256) k_mul_chain(u32* out, Stamp* st, u32 seed, int iters) {
  u32 wa[8], wb[8]; seed_words(wa, seed, 1); seed_words(wb, seed, 2);
  FE x, y; OPS::from_words(x, wa); OPS::from_words(y, wb);
  u64 c0, r0; stamp_begin(c0, r0);
#pragma unroll 1
  for (int it = 0; it < iters; ++it) { OPS::mul(x, x, y); OPS::mul(y, y, x); }
  stamp_end(st, c0, r0);
  u32 o[8]; OPS::to_words(o, x);
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = 0; i < 8; ++i) out[g * 16 + i] = o[i];
  OPS::to_words(o, y);
  for (int i = 0; i < 8; ++i) out[g * 16 + 8 + i] = o[i];
}
template <class FE, class OPS>
__global__ void __launch_bounds__(256) k_sq_chain(u32* out, Stamp* st, u32 seed, int iters) {
  u32 wa[8], wb[8]; seed_words(wa, seed, 1); seed_words(wb, seed, 2);
  FE x, y; OPS::from_words(x, wa); OPS::from_words(y, wb);
  u64 c0, r0; stamp_begin(c0, r0);
#pragma unroll 1
  for (int it = 0; it < iters; ++it) { OPS::sq(x, x); OPS::sq(y, y); }
  stamp_end(st, c0, r0);
  u32 o[8]; OPS::to_words(o, x);
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = 0; i < 8; ++i) out[g * 16 + i] = o[i];
  OPS::to_words(o, y);
  for (int i = 0; i < 8; ++i) out[g * 16 + 8 + i] = o[i];
}
struct Ops10 {
  static __device__ __forceinline__ void from_words(eg10::fe& h, const u32 w[8]) { eg10::fe_from_words(h, w); }
  static __device__ __forceinline__ void to_words(u32 w[8], const eg10::fe& f) { eg10::fe_to_words(w, f); }
  static __device__ __forceinline__ void mul(eg10::fe& h, const eg10::fe& f, const eg10::fe& g) { eg10::fe_mul(h, f, g); }
  static __device__ __forceinline__ void sq(eg10::fe& h, const eg10::fe& f) { eg10::fe_sq(h, f); }
};
struct Ops9 {
  static __device__ __forceinline__ void from_words(eg::fe& h, const u32 w[8]) { eg::fe_from_words(h, w); }
  static __device__ __forceinline__ void to_words(u32 w[8], const eg::fe& f) { eg::fe_to_words(w, f); }
  static __device__ __forceinline__ void mul(eg::fe& h, const eg::fe& f, const eg::fe& g) { eg::fe_mul(h, f, g); }
  static __device__ __forceinline__ void sq(eg::fe& h, const eg::fe& f) { eg::fe_sq(h, f); }
};

// ---- comb columns: acc = 2 acc + (+-entry), as ge_teeth_mul does (entry kept in registers; sign from a per-lane word) ----------
template <int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_columns10(u32* out, Stamp* st, u32 seed, int iters) {
  using namespace eg10;
  u32 w[8];
  ge_cached e;
  seed_words(w, seed, 3); fe_from_words(e.YpX, w);
  seed_words(w, seed, 4); fe_from_words(e.YmX, w);
  seed_words(w, seed, 5); fe_from_words(e.Z2, w);
  seed_words(w, seed, 6); fe_from_words(e.T2d, w);
  ge acc;
  seed_words(w, seed, 7); fe_from_words(acc.X, w);
  seed_words(w, seed, 8); fe_from_words(acc.Y, w);
  seed_words(w, seed, 9); fe_from_words(acc.Z, w);
  u32 signs = w[0];
  u64 c0, r0; stamp_begin(c0, r0);
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    ge_cached cur = e;
    ge_p1p1 t;
    ge_dbl(t, acc.X, acc.Y, acc.Z);
    ge_dbl_to_p3(acc, t);
    ge_cached_cneg(cur, (signs >> (it & 31)) & 1u);
    ge_add(t, acc, cur);
    ge_p2 q; ge_add_to_p2(q, t);
    acc.X = q.X; acc.Y = q.Y; acc.Z = q.Z;
  }
  stamp_end(st, c0, r0);
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  u32 o[8];
  fe_to_words(o, acc.X); for (int i = 0; i < 8; ++i) out[g * 24 + i] = o[i];
  fe_to_words(o, acc.Y); for (int i = 0; i < 8; ++i) out[g * 24 + 8 + i] = o[i];
  fe_to_words(o, acc.Z); for (int i = 0; i < 8; ++i) out[g * 24 + 16 + i] = o[i];
}
template <int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_columns9(u32* out, Stamp* st, u32 seed, int iters) {
  using namespace eg;
  u32 w[8];
  ge_cached e;
  seed_words(w, seed, 3); fe_from_words(e.YpX, w);
  seed_words(w, seed, 4); fe_from_words(e.YmX, w);
  seed_words(w, seed, 5); fe_from_words(e.Z2, w);
  seed_words(w, seed, 6); fe_from_words(e.T2d, w);
  ge acc;
  seed_words(w, seed, 7); fe_from_words(acc.X, w);
  seed_words(w, seed, 8); fe_from_words(acc.Y, w);
  seed_words(w, seed, 9); fe_from_words(acc.Z, w);
  u32 signs = w[0];
  u64 c0, r0; stamp_begin(c0, r0);
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    ge_cached cur = e;
    ge_p1p1 t;
    ge_dbl(t, acc.X, acc.Y, acc.Z);
    ge_dbl_to_p3(acc, t);
    ge_cached_cneg(cur, (signs >> (it & 31)) & 1u);
    ge_add(t, acc, cur);
    ge_p2 q; ge_add_to_p2(q, t);
    acc.X = q.X; acc.Y = q.Y; acc.Z = q.Z;
  }
  stamp_end(st, c0, r0);
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  u32 o[8];
  fe_to_words(o, acc.X); for (int i = 0; i < 8; ++i) out[g * 24 + i] = o[i];
  fe_to_words(o, acc.Y); for (int i = 0; i < 8; ++i) out[g * 24 + 8 + i] = o[i];
  fe_to_words(o, acc.Z); for (int i = 0; i < 8; ++i) out[g * 24 + 16 + i] = o[i];
}
// doubling chains (the table build: 215 doublings per base)
template <int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_dbl10(u32* out, Stamp* st, u32 seed, int iters) {
  using namespace eg10;
  u32 w[8]; ge_p2 q;
  seed_words(w, seed, 7); fe_from_words(q.X, w);
  seed_words(w, seed, 8); fe_from_words(q.Y, w);
  seed_words(w, seed, 9); fe_from_words(q.Z, w);
  u64 c0, r0; stamp_begin(c0, r0);
#pragma unroll 1
  for (int it = 0; it < iters; ++it) { ge_p1p1 t; ge_dbl(t, q.X, q.Y, q.Z); ge_dbl_to_p2(q, t); }
  stamp_end(st, c0, r0);
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  u32 o[8];
  fe_to_words(o, q.X); for (int i = 0; i < 8; ++i) out[g * 24 + i] = o[i];
  fe_to_words(o, q.Y); for (int i = 0; i < 8; ++i) out[g * 24 + 8 + i] = o[i];
  fe_to_words(o, q.Z); for (int i = 0; i < 8; ++i) out[g * 24 + 16 + i] = o[i];
}
template <int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_dbl9(u32* out, Stamp* st, u32 seed, int iters) {
  using namespace eg;
  u32 w[8]; ge_p2 q;
  seed_words(w, seed, 7); fe_from_words(q.X, w);
  seed_words(w, seed, 8); fe_from_words(q.Y, w);
  seed_words(w, seed, 9); fe_from_words(q.Z, w);
  u64 c0, r0; stamp_begin(c0, r0);
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    ge_p1p1 t; ge_dbl(t, q.X, q.Y, q.Z);
    fe_mul(q.X, t.X, t.T); fe_mul(q.Y, t.Z, t.Y); fe_mul(q.Z, t.Z, t.T);
  }
  stamp_end(st, c0, r0);
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  u32 o[8];
  fe_to_words(o, q.X); for (int i = 0; i < 8; ++i) out[g * 24 + i] = o[i];
  fe_to_words(o, q.Y); for (int i = 0; i < 8; ++i) out[g * 24 + 8 + i] = o[i];
  fe_to_words(o, q.Z); for (int i = 0; i < 8; ++i) out[g * 24 + 16 + i] = o[i];
}

typedef void (*kern_t)(u32*, Stamp*, u32, int);
struct Result { double ms, cyc_med, cyc_max, ghz; std::vector<u32> out; };
// exactly w blocks per CU: each block asks for 1/w of the CU's 160 KiB of LDS (the dispatcher otherwise packs up to 8 blocks of a
// small kernel on some CUs and leaves others short: the first version of this bench measured that imbalance for w > 2)
static size_t lds_for(int w) { return ((size_t)160 * 1024 / w) / 1024 * 1024 - (w == 1 ? 0 : 0); }
static Result run(kern_t k, int blocks, int w, int iters, int words_per_lane, u32* d_out, Stamp* d_st) {
  const int waves = blocks * 4;
  const size_t lds = lds_for(w);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, d_out, d_st, 777u, iters / 8 + 1);     // warm-up
  CK(hipDeviceSynchronize());
  Result r; r.ms = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, d_out, d_st, 12345u, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms >= r.ms) continue;
    r.ms = ms;
    std::vector<Stamp> st(waves);
    CK(hipMemcpy(st.data(), d_st, sizeof(Stamp) * waves, hipMemcpyDeviceToHost));
    std::vector<double> cyc(waves), ghz(waves);
    for (int i = 0; i < waves; ++i) { cyc[i] = (double)st[i].cyc; ghz[i] = (double)st[i].cyc / ((double)st[i].rt * 10.0); }
    std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
    r.cyc_med = cyc[waves / 2]; r.cyc_max = cyc[waves - 1]; r.ghz = ghz[waves / 2];
  }
  r.out.resize((size_t)blocks * 256 * words_per_lane);
  CK(hipMemcpy(r.out.data(), d_out, r.out.size() * 4, hipMemcpyDeviceToHost));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return r;
}

int main(int argc, char** argv) {
  CK(hipSetDevice(0));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int scale = argc > 1 && atoi(argv[1]) > 0 ? atoi(argv[1]) : 1;
  printf("device %s, %d CUs; cycles from s_memtime (median over waves), clock from s_memrealtime\n", prop.name, cus);
  u32* d_out; Stamp* d_st;
  CK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * 24 * 4));
  CK(hipMalloc(&d_st, sizeof(Stamp) * cus * 8 * 4));
  struct Case { const char* name; kern_t k10, k9; int iters; int ops_per_iter; int words; int w; };
  const Case cases[] = {
    {"fe_mul chain", k_mul_chain<eg10::fe, Ops10>, k_mul_chain<eg::fe, Ops9>, 20000 * scale, 2, 16, 1},
    {"fe_mul chain", k_mul_chain<eg10::fe, Ops10>, k_mul_chain<eg::fe, Ops9>, 20000 * scale, 2, 16, 2},
    {"fe_mul chain", k_mul_chain<eg10::fe, Ops10>, k_mul_chain<eg::fe, Ops9>, 20000 * scale, 2, 16, 3},
    {"fe_mul chain", k_mul_chain<eg10::fe, Ops10>, k_mul_chain<eg::fe, Ops9>, 20000 * scale, 2, 16, 4},
    {"fe_mul chain", k_mul_chain<eg10::fe, Ops10>, k_mul_chain<eg::fe, Ops9>, 15000 * scale, 2, 16, 6},
    {"fe_mul chain", k_mul_chain<eg10::fe, Ops10>, k_mul_chain<eg::fe, Ops9>, 10000 * scale, 2, 16, 8},
    {"fe_sq chain", k_sq_chain<eg10::fe, Ops10>, k_sq_chain<eg::fe, Ops9>, 20000 * scale, 2, 16, 2},
    {"fe_sq chain", k_sq_chain<eg10::fe, Ops10>, k_sq_chain<eg::fe, Ops9>, 20000 * scale, 2, 16, 3},
    {"fe_sq chain", k_sq_chain<eg10::fe, Ops10>, k_sq_chain<eg::fe, Ops9>, 20000 * scale, 2, 16, 4},
    {"fe_sq chain", k_sq_chain<eg10::fe, Ops10>, k_sq_chain<eg::fe, Ops9>, 10000 * scale, 2, 16, 8},
    {"doubling (4S+3M) regs for 2", k_dbl10<2>, k_dbl9<2>, 6000 * scale, 1, 24, 2},
    {"doubling (4S+3M) regs for 3", k_dbl10<3>, k_dbl9<3>, 6000 * scale, 1, 24, 3},
    {"doubling (4S+3M) regs for 4", k_dbl10<4>, k_dbl9<4>, 6000 * scale, 1, 24, 4},
    {"comb column (4S+11M) regs for 2", k_columns10<2>, k_columns9<2>, 2500 * scale, 1, 24, 2},
    {"comb column (4S+11M) regs for 3", k_columns10<3>, k_columns9<3>, 2500 * scale, 1, 24, 3},
    {"comb column (4S+11M) regs for 4", k_columns10<4>, k_columns9<4>, 2500 * scale, 1, 24, 4},
  };
  if (argc > 1 && std::string(argv[1]) == "sustained") {
    // The comparison below uses bursts of 20-50 ms.  The verifier runs this arithmetic for seconds, and the clock the power management
    // settles on under it is lower than in a burst (and lower than under tools/ubench/valu_rates, whose multiplier operands never
    // change): the same kernels of the shipped representation launched back to back for `seconds`, rate and clock over the second half.
    const double seconds = argc > 2 ? atof(argv[2]) : 4.0;
    printf("sustained mode (shipped 9-limb representation): back to back for %.1f s; rate and clock over the second half\n", seconds);
    printf("%-34s | w/SIMD | Gop/s first launches (GHz) | Gop/s sustained (GHz)\n", "loop");
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const Case& c : cases) {
      if (c.w != 2 && c.w != 3) continue;
      const int w = c.w, blocks = cus * w, waves = blocks * 4;
      const size_t lds = lds_for(w);
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(c.k9), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      const double ops = (double)c.iters * c.ops_per_iter, lanes = (double)blocks * 256;
      auto burst = [&](int launches, double& gops, double& ghz_med) {
        CK(hipEventRecord(e0));
        for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(c.k9, dim3(blocks), dim3(256), lds, 0, d_out, d_st, 12345u + l, c.iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<Stamp> st(waves);
        CK(hipMemcpy(st.data(), d_st, sizeof(Stamp) * waves, hipMemcpyDeviceToHost));
        std::vector<double> ghz(waves);
        for (int i = 0; i < waves; ++i) ghz[i] = (double)st[i].cyc / ((double)st[i].rt * 10.0);
        std::sort(ghz.begin(), ghz.end());
        ghz_med = ghz[waves / 2]; gops = ops * lanes * launches / (ms * 1e6);
      };
      CK(hipDeviceSynchronize());
      double g1, c1, g2, c2, g3, c3;
      burst(1, g1, c1);
      const float one_ms = (float)(ops * lanes / (g1 * 1e6));
      const int half = (int)(seconds * 500.0 / one_ms) + 1;
      burst(half, g2, c2);
      burst(half, g3, c3);
      printf("%-34s | %6d | %9.2f (%4.2f)           | %9.2f (%4.2f)\n", c.name, w, g1, c1, g3, c3);
      fflush(stdout);
    }
    return 0;
  }
  printf("cyc/op = cycles of the LAST wave to finish / operations / waves per SIMD (issue cycles per operation and SIMD); Gop/s from the wall clock, best of 3\n");
  for (const Case& c : cases) {
    const int w = c.w;
    const int blocks = cus * w;
    Result a = run(c.k10, blocks, w, c.iters, c.words, d_out, d_st);
    Result b = run(c.k9, blocks, w, c.iters, c.words, d_out, d_st);
    Result a2 = run(c.k10, blocks, w, c.iters, c.words, d_out, d_st);    // A B A: the clock ramps up over the first launches
    if (a2.ms < a.ms) a = a2;
    const bool same = a.out == b.out;
    const double ops = (double)c.iters * c.ops_per_iter;
    const double lanes = (double)blocks * 256;
    printf("%-32s waves/SIMD=%d  10x25.5: %8.2f ms %8.1f cyc/op (%.2f GHz) %8.2f Gop/s | 9x28.3: %8.2f ms %8.1f cyc/op (%.2f GHz) %8.2f Gop/s | time ratio %.3f | outputs %s\n",
           c.name, w, a.ms, a.cyc_max / ops / w, a.ghz, ops * lanes / (a.ms * 1e6), b.ms, b.cyc_max / ops / w, b.ghz, ops * lanes / (b.ms * 1e6),
           b.ms / a.ms, same ? "IDENTICAL" : "DIFFER");
  }
  return 0;
}

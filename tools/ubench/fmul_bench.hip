// Microbenchmark: candidate GF(2^255-19) multiplication kernels on gfx950 (measurement tool only).
//  A: 8x32-bit saturated limbs, operand scanning with 64-bit temporaries (value kept < 2^256, 2^256 = 38)
//  B: 10 limbs radix 2^25.5 (26/25 bits alternating), 64-bit column accumulators, lazy carries
//  C: 8x32 saturated, product scanning through v_mad_u64_u32 carry-out (inline asm)
// Each thread runs a dependent chain x <- x*y, y <- y*x ... ; throughput in field-mul/s chip-wide.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef uint32_t u32; typedef uint64_t u64;

// ---------- A ----------
struct feA { u32 v[8]; };
__device__ __forceinline__ void mulA(feA& r, const feA& a, const feA& b) {
  u32 t[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    u32 carry = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      u64 p = (u64)a.v[i] * b.v[j] + t[i + j] + carry;
      t[i + j] = (u32)p; carry = (u32)(p >> 32);
    }
    t[i + 8] = carry;
  }
  // fold: lo + 38*hi
  u32 carry = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    u64 p = (u64)t[i + 8] * 38u + t[i] + carry;
    t[i] = (u32)p; carry = (u32)(p >> 32);
  }
  // carry < 39 ; fold again (carry*38 < 2^11)
  u64 p = (u64)carry * 38u + t[0];
  r.v[0] = (u32)p; u32 c = (u32)(p >> 32);
#pragma unroll
  for (int i = 1; i < 8; ++i) { u64 q = (u64)t[i] + c; r.v[i] = (u32)q; c = (u32)(q >> 32); }
  // c may be 1 only if value wrapped past 2^256: add 38 once more (cannot carry again)
  r.v[0] += c * 38u;
}

// ---------- B ----------
struct feB { u32 v[10]; };
__device__ __forceinline__ void mulB(feB& r, const feB& f, const feB& g) {
  u32 g19[10], f2[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) { g19[i] = 19u * g.v[i]; f2[i] = 2u * f.v[i]; }
  u64 h[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      int j = k - i; bool wrap = false;
      if (j < 0) { j += 10; wrap = true; }
      // odd*odd limbs carry an extra factor 2
      u32 fi = ((i & 1) && (j & 1)) ? f2[i] : f.v[i];
      u32 gj = wrap ? g19[j] : g.v[j];
      acc += (u64)fi * gj;
    }
    h[k] = acc;
  }
  // carry chain
  u64 c;
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    int bits = (k & 1) ? 25 : 26;
    c = h[k] >> bits; h[k] &= ((1ull << bits) - 1);
    if (k < 9) h[k + 1] += c; else h[0] += 19 * c;
  }
  c = h[0] >> 26; h[0] &= ((1ull << 26) - 1); h[1] += c;
#pragma unroll
  for (int k = 0; k < 10; ++k) r.v[k] = (u32)h[k];
}

// ---------- C ----------
// acc(96 bit: lo,hi in a 64-bit pair, ex) += a*b
__device__ __forceinline__ void mac96(u64& acc, u32& ex, u32 a, u32 b) {
  asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
               : "+v"(acc), "+v"(ex) : "v"(a), "v"(b) : "vcc");
}
__device__ __forceinline__ void mulC(feA& r, const feA& a, const feA& b) {
  u32 t[16];
  u64 acc = 0; u32 ex = 0;
#pragma unroll
  for (int k = 0; k < 15; ++k) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int j = k - i;
      if (j >= 0 && j < 8) mac96(acc, ex, a.v[i], b.v[j]);
    }
    t[k] = (u32)acc;
    acc = (acc >> 32) | ((u64)ex << 32); ex = 0;
  }
  t[15] = (u32)acc;
  u32 carry = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    u64 p = (u64)t[i + 8] * 38u + t[i] + carry;
    t[i] = (u32)p; carry = (u32)(p >> 32);
  }
  u64 p = (u64)carry * 38u + t[0];
  r.v[0] = (u32)p; u32 c = (u32)(p >> 32);
#pragma unroll
  for (int i = 1; i < 8; ++i) { u64 q = (u64)t[i] + c; r.v[i] = (u32)q; c = (u32)(q >> 32); }
  r.v[0] += c * 38u;
}

constexpr int ITERS = 4096;

template <int W>
__global__ void __launch_bounds__(256) kA(u32* out, u32 seed) {
  feA x, y;
  for (int i = 0; i < 8; ++i) { x.v[i] = seed * (i + 3) + threadIdx.x * 2654435761u; y.v[i] = seed * (i + 7) ^ (threadIdx.x * 40503u); }
  for (int it = 0; it < ITERS; ++it) { mulA(x, x, y); mulA(y, y, x); }
  u32 s = 0; for (int i = 0; i < 8; ++i) s ^= x.v[i] ^ y.v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) kB(u32* out, u32 seed) {
  feB x, y;
  for (int i = 0; i < 10; ++i) { x.v[i] = (seed * (i + 3) + threadIdx.x * 2654435761u) & 0x1ffffff; y.v[i] = (seed * (i + 7) ^ (threadIdx.x * 40503u)) & 0x1ffffff; }
  for (int it = 0; it < ITERS; ++it) { mulB(x, x, y); mulB(y, y, x); }
  u32 s = 0; for (int i = 0; i < 10; ++i) s ^= x.v[i] ^ y.v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) kC(u32* out, u32 seed) {
  feA x, y;
  for (int i = 0; i < 8; ++i) { x.v[i] = seed * (i + 3) + threadIdx.x * 2654435761u; y.v[i] = seed * (i + 7) ^ (threadIdx.x * 40503u); }
  for (int it = 0; it < ITERS; ++it) { mulC(x, x, y); mulC(y, y, x); }
  u32 s = 0; for (int i = 0; i < 8; ++i) s ^= x.v[i] ^ y.v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// two independent chains per thread (ILP across muls)
__global__ void __launch_bounds__(256) kA2(u32* out, u32 seed) {
  feA x, y, z, w;
  for (int i = 0; i < 8; ++i) { x.v[i] = seed * (i + 3) + threadIdx.x * 2654435761u; y.v[i] = seed * (i + 7) ^ (threadIdx.x * 40503u); z.v[i] = x.v[i] ^ 0x5555; w.v[i] = y.v[i] + 77; }
  for (int it = 0; it < ITERS / 2; ++it) { mulA(x, x, y); mulA(z, z, w); mulA(y, y, x); mulA(w, w, z); }
  u32 s = 0; for (int i = 0; i < 8; ++i) s ^= x.v[i] ^ y.v[i] ^ z.v[i] ^ w.v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef void (*kern_t)(u32*, u32);
int main() {
  CK(hipSetDevice(0));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  u32* out; CK(hipMalloc(&out, sizeof(u32) * cus * 8 * 256));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct { const char* n; kern_t k; } ks[] = {{"A sat32 operand-scan", kA<0>}, {"B radix-25.5", kB}, {"C sat32 carry-out asm", kC}, {"A x2 chains", kA2}};
  for (auto& e : ks) {
    for (int w : {1, 2, 4, 8}) {
      int blocks = cus * w;
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 12345u);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 12345u);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      double muls = (double)ITERS * 2 * blocks * 256;
      double cyc = ms * 1e-3 * 2.4e9 / ((double)ITERS * 2 * w);  // cycles per wave-fmul per SIMD
      printf("%-24s waves/simd=%d  %8.2f Gfmul/s   %7.1f cyc per wave-fmul\n", e.n, w, muls / (ms * 1e-3) / 1e9, cyc);
    }
  }
  return 0;
}

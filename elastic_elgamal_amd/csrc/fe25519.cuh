// fe25519.cuh -- GF(2^255-19) for gfx950 (CDNA4): 10 unsigned limbs, radix 2^25.5, 64-bit column sums.
//
// Why this representation (measured on MI355X, profiles/r01_ubench_*.txt): v_mad_u64_u32 issues at
// ~4.7 cycles per wave64 -- within 10 % of any other VOP3 op -- so a field multiply is priced by its
// total instruction count, not by its multiply count.  100 MADs that accumulate straight into 64-bit
// column sums (no per-product carry handling) beat 64 MADs on saturated 32-bit limbs + carry plumbing
// (256 vs 184..228 G field-mul/s chip-wide).  MFMA is not used: every product has two per-lane operands.
//
// Bound discipline.  "class c" means even limbs <= c*2^26, odd limbs <= c*2^25 (a hair above for c = 1).
//   fe_mul(h, f, g): needs class(g) <= 3.3 (19*g_i must fit 32 bits) and class(f)*class(g) <= 32
//                    (column sums < 2^64); output class 1.
//   fe_sq(h, f):     needs class(f) <= 3.3; output class 1.
//   fe_add:          class(f)+class(g).     fe_sub: class(f)+2 (g must be class 1).
//   fe_sub4:         class(f)+4 (g <= class 3.3).   fe_carry: any class <= 60 -> class 1.
// Compiled with -DEG_BOUNDCHECK on the host (tests/hostcheck) every fe carries its class and each
// operation asserts its precondition, so any executed code path is a proof of the discipline.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define EG_HD __host__ __device__ __forceinline__
#define EG_D __device__ __forceinline__
// Keeps the backend scheduler from interleaving independent field multiplications: one multiply already has
// 10 independent 10-MAD columns of ILP, while interleaving several multiplies only inflates live registers
// (256 VGPR + AGPR spills, 1 wave/SIMD).  See DESIGN.md section 9.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(EG_NO_SCHED_FENCE)
#define EG_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define EG_SCHED_FENCE() ((void)0)
#endif
#else
#define EG_SCHED_FENCE() ((void)0)
#define EG_HD inline
#define EG_D inline
#endif

#ifdef EG_BOUNDCHECK
#include <assert.h>
#include <stdio.h>
#include <stdlib.h>
#define EG_CLS(x) , (x)
#define EG_REQUIRE(cond, msg) do { if (!(cond)) { fprintf(stderr, "bound violation: %s (%s:%d)\n", msg, __FILE__, __LINE__); abort(); } } while (0)
#else
#define EG_REQUIRE(cond, msg) ((void)0)
#endif

namespace eg {

typedef uint32_t u32;
typedef uint64_t u64;

struct fe {
  u32 v[10];
#ifdef EG_BOUNDCHECK
  float cls;
#endif
};

#ifdef EG_BOUNDCHECK
#define EG_SETCLS(h, c) ((h).cls = (c))
#define EG_GETCLS(h) ((h).cls)
static inline void fe_check_values(const fe& f) {
  for (int i = 0; i < 10; ++i) {
    double nominal = (i & 1) ? 33554432.0 : 67108864.0;
    double lim = nominal * f.cls * 1.01 + 64.0;
    if ((double)f.v[i] > lim) { fprintf(stderr, "limb %d = %u exceeds class %.2f\n", i, f.v[i], f.cls); abort(); }
  }
}
// operation counters of the host check build (algorithmic work model)
static unsigned long long g_fe_mul_count = 0, g_fe_sq_count = 0;
#define EG_COUNT_MUL() (++g_fe_mul_count)
#define EG_COUNT_SQ() (++g_fe_sq_count)
#else
#define EG_COUNT_MUL() ((void)0)
#define EG_COUNT_SQ() ((void)0)
#define EG_SETCLS(h, c) ((void)0)
#define EG_GETCLS(h) (0.0f)
#define fe_check_values(f) ((void)0)
#endif

EG_HD void fe_0(fe& h) {
#pragma unroll
  for (int i = 0; i < 10; ++i) h.v[i] = 0;
  EG_SETCLS(h, 1.0f);
}
EG_HD void fe_1(fe& h) { fe_0(h); h.v[0] = 1; }

EG_HD void fe_add(fe& h, const fe& f, const fe& g) {
#pragma unroll
  for (int i = 0; i < 10; ++i) h.v[i] = f.v[i] + g.v[i];
  EG_SETCLS(h, EG_GETCLS(f) + EG_GETCLS(g));
  EG_REQUIRE(EG_GETCLS(h) <= 60.0f, "fe_add overflow");
}

// h = f + 2p - g ; g must be class 1
EG_HD void fe_sub(fe& h, const fe& f, const fe& g) {
  EG_REQUIRE(EG_GETCLS(g) <= 1.02f, "fe_sub: subtrahend must be class 1");
  float c = EG_GETCLS(f) + 2.0f; (void)c;
  h.v[0] = f.v[0] + 0x7ffffdau - g.v[0];
#pragma unroll
  for (int i = 1; i < 10; ++i) h.v[i] = f.v[i] + ((i & 1) ? 0x3fffffeu : 0x7fffffeu) - g.v[i];
  EG_SETCLS(h, c);
}

// h = f + 4p - g ; g up to class 3.3 (in fact < 4)
EG_HD void fe_sub4(fe& h, const fe& f, const fe& g) {
  EG_REQUIRE(EG_GETCLS(g) <= 3.9f, "fe_sub4: subtrahend class too large");
  float c = EG_GETCLS(f) + 4.0f; (void)c;
  h.v[0] = f.v[0] + 0xfffffb4u - g.v[0];
#pragma unroll
  for (int i = 1; i < 10; ++i) h.v[i] = f.v[i] + ((i & 1) ? 0x7fffffcu : 0xffffffcu) - g.v[i];
  EG_SETCLS(h, c);
}

// weak reduction to class 1 (one carry sweep + wrap)
EG_HD void fe_carry(fe& h) {
  EG_REQUIRE(EG_GETCLS(h) <= 60.0f, "fe_carry input too large");
  u32 c;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int bits = (i & 1) ? 25 : 26;
    c = h.v[i] >> bits; h.v[i] &= ((1u << bits) - 1); h.v[i + 1] += c;
  }
  c = h.v[9] >> 25; h.v[9] &= 0x1ffffffu; h.v[0] += 19u * c;
  c = h.v[0] >> 26; h.v[0] &= 0x3ffffffu; h.v[1] += c;
  EG_SETCLS(h, 1.0f);
}

EG_HD void fe_neg(fe& h, const fe& f) {  // class 1 in -> class 3 out (0 + 2p - f)
  fe z; fe_0(z);
  EG_SETCLS(z, 0.0f);
  fe_sub(h, z, f);
}

EG_HD void fe_reduce_columns(fe& h, u64 c[10]) {
  u64 t;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int bits = (i & 1) ? 25 : 26;
    t = c[i] >> bits; c[i] &= ((1ull << bits) - 1); c[i + 1] += t;
  }
  t = c[9] >> 25; c[9] &= 0x1ffffffull; c[0] += 19ull * t;
  t = c[0] >> 26; c[0] &= 0x3ffffffull; c[1] += t;
#pragma unroll
  for (int i = 0; i < 10; ++i) h.v[i] = (u32)c[i];
  EG_SETCLS(h, 1.0f);
}

EG_HD void fe_mul(fe& h, const fe& f, const fe& g) {
  EG_REQUIRE(EG_GETCLS(g) <= 3.31f, "fe_mul: g operand class > 3.3");
  EG_REQUIRE(EG_GETCLS(f) * EG_GETCLS(g) <= 32.0f, "fe_mul: class product > 32");
  fe_check_values(f); fe_check_values(g);
  EG_COUNT_MUL();
  EG_SCHED_FENCE();
  u32 g19[10], f2[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) { g19[i] = 19u * g.v[i]; f2[i] = 2u * f.v[i]; }
  u64 c[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      int j = k - i;
      bool wrap = false;
      if (j < 0) { j += 10; wrap = true; }
      const u32 fi = ((i & 1) && (j & 1)) ? f2[i] : f.v[i];
      const u32 gj = wrap ? g19[j] : g.v[j];
      acc += (u64)fi * gj;
    }
    c[k] = acc;
  }
  fe_reduce_columns(h, c);
  EG_SCHED_FENCE();
}

EG_HD void fe_sq(fe& h, const fe& f) {
  EG_REQUIRE(EG_GETCLS(f) <= 3.31f, "fe_sq: operand class > 3.3");
  fe_check_values(f);
  EG_COUNT_SQ();
  EG_SCHED_FENCE();
  const u32 f0 = f.v[0], f1 = f.v[1], f2 = f.v[2], f3 = f.v[3], f4 = f.v[4];
  const u32 f5 = f.v[5], f6 = f.v[6], f7 = f.v[7], f8 = f.v[8], f9 = f.v[9];
  const u32 f0_2 = 2 * f0, f1_2 = 2 * f1, f2_2 = 2 * f2, f3_2 = 2 * f3, f4_2 = 2 * f4;
  const u32 f5_2 = 2 * f5, f6_2 = 2 * f6, f7_2 = 2 * f7;
  const u32 f5_38 = 38 * f5, f6_19 = 19 * f6, f7_38 = 38 * f7, f8_19 = 19 * f8, f9_38 = 38 * f9;
  u64 c[10];
  c[0] = (u64)f0 * f0 + (u64)f1_2 * f9_38 + (u64)f2_2 * f8_19 + (u64)f3_2 * f7_38 + (u64)f4_2 * f6_19 + (u64)f5 * f5_38;
  c[1] = (u64)f0_2 * f1 + (u64)f2 * f9_38 + (u64)f3_2 * f8_19 + (u64)f4 * f7_38 + (u64)f5_2 * f6_19;
  c[2] = (u64)f0_2 * f2 + (u64)f1_2 * f1 + (u64)f3_2 * f9_38 + (u64)f4_2 * f8_19 + (u64)f5_2 * f7_38 + (u64)f6 * f6_19;
  c[3] = (u64)f0_2 * f3 + (u64)f1_2 * f2 + (u64)f4 * f9_38 + (u64)f5_2 * f8_19 + (u64)f6 * f7_38;
  c[4] = (u64)f0_2 * f4 + (u64)f1_2 * f3_2 + (u64)f2 * f2 + (u64)f5_2 * f9_38 + (u64)f6_2 * f8_19 + (u64)f7 * f7_38;
  c[5] = (u64)f0_2 * f5 + (u64)f1_2 * f4 + (u64)f2_2 * f3 + (u64)f6 * f9_38 + (u64)f7_2 * f8_19;
  c[6] = (u64)f0_2 * f6 + (u64)f1_2 * f5_2 + (u64)f2_2 * f4 + (u64)f3_2 * f3 + (u64)f7_2 * f9_38 + (u64)f8 * f8_19;
  c[7] = (u64)f0_2 * f7 + (u64)f1_2 * f6 + (u64)f2_2 * f5 + (u64)f3_2 * f4 + (u64)f8 * f9_38;
  c[8] = (u64)f0_2 * f8 + (u64)f1_2 * f7_2 + (u64)f2_2 * f6 + (u64)f3_2 * f5_2 + (u64)f4 * f4 + (u64)f9 * f9_38;
  c[9] = (u64)f0_2 * f9 + (u64)f1_2 * f8 + (u64)f2_2 * f7 + (u64)f3_2 * f6 + (u64)f4_2 * f5;
  fe_reduce_columns(h, c);
  EG_SCHED_FENCE();
}

EG_HD void fe_sqn(fe& h, const fe& f, int n) {
  fe_sq(h, f);
  for (int i = 1; i < n; ++i) fe_sq(h, h);
}

// ---- byte codec ---------------------------------------------------------------------------------
// w[0..7] = little-endian 32-bit words of the 32-byte encoding; bit 255 is ignored (as dalek does).
EG_HD void fe_from_words(fe& h, const u32 w[8]) {
  h.v[0] = w[0] & 0x3ffffffu;
  h.v[1] = ((w[0] >> 26) | (w[1] << 6)) & 0x1ffffffu;
  h.v[2] = ((w[1] >> 19) | (w[2] << 13)) & 0x3ffffffu;
  h.v[3] = ((w[2] >> 13) | (w[3] << 19)) & 0x1ffffffu;
  h.v[4] = (w[3] >> 6) & 0x3ffffffu;
  h.v[5] = w[4] & 0x1ffffffu;
  h.v[6] = ((w[4] >> 25) | (w[5] << 7)) & 0x3ffffffu;
  h.v[7] = ((w[5] >> 19) | (w[6] << 13)) & 0x1ffffffu;
  h.v[8] = ((w[6] >> 12) | (w[7] << 20)) & 0x3ffffffu;
  h.v[9] = (w[7] >> 6) & 0x1ffffffu;
  EG_SETCLS(h, 1.0f);
}

// canonical (fully reduced) words
EG_HD void fe_to_words(u32 w[8], const fe& f) {
  fe t = f;
  fe_carry(t);
  fe_carry(t);
  // t < 2^255 + small; q = 1 iff t >= p
  u32 q = (t.v[0] + 19) >> 26;
#pragma unroll
  for (int i = 1; i < 10; ++i) q = (t.v[i] + q) >> ((i & 1) ? 25 : 26);
  t.v[0] += 19 * q;
  u32 c;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int bits = (i & 1) ? 25 : 26;
    c = t.v[i] >> bits; t.v[i] &= ((1u << bits) - 1); t.v[i + 1] += c;
  }
  t.v[9] &= 0x1ffffffu;
  w[0] = t.v[0] | (t.v[1] << 26);
  w[1] = (t.v[1] >> 6) | (t.v[2] << 19);
  w[2] = (t.v[2] >> 13) | (t.v[3] << 13);
  w[3] = (t.v[3] >> 19) | (t.v[4] << 6);
  w[4] = t.v[5] | (t.v[6] << 25);
  w[5] = (t.v[6] >> 7) | (t.v[7] << 19);
  w[6] = (t.v[7] >> 13) | (t.v[8] << 12);
  w[7] = (t.v[8] >> 20) | (t.v[9] << 6);
}

EG_HD bool fe_isnegative(const fe& f) { u32 w[8]; fe_to_words(w, f); return w[0] & 1; }
EG_HD bool fe_iszero(const fe& f) {
  u32 w[8]; fe_to_words(w, f);
  u32 r = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r |= w[i];
  return r == 0;
}
EG_HD bool fe_eq(const fe& f, const fe& g) {
  u32 a[8], b[8]; fe_to_words(a, f); fe_to_words(b, g);
  u32 r = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r |= a[i] ^ b[i];
  return r == 0;
}
// h = flag ? g : h   (both must already be in comparable classes; class becomes the max)
EG_HD void fe_cmov(fe& h, const fe& g, bool flag) {
#pragma unroll
  for (int i = 0; i < 10; ++i) h.v[i] = flag ? g.v[i] : h.v[i];
#ifdef EG_BOUNDCHECK
  if (g.cls > h.cls) h.cls = g.cls;
#endif
}

// ---- 256-bit packing (table entries: device_io.cuh BaseTable) --------------------------------------------------------------------
// A class-1 element (every limb within its 26 / 25 bits, limb 1 a hair above: what fe_mul, fe_sq and fe_carry return) is < 2^256 as an
// integer: eight 32-bit words.  Limb offsets: 0 26 51 77 102 128 153 179 204 230.  Unpacking slices the integer again: limbs 0..8 within
// their widths, limb 9 <= 2^25 (the integer is < 2^255 + 2^40), i.e. class 1.
EG_HD void fe_pack8(u32 w[8], const fe& f) {
  EG_REQUIRE(EG_GETCLS(f) <= 1.02f, "fe_pack8: operand must be class 1");
  fe_check_values(f);
  u64 acc = (u64)f.v[0] + ((u64)f.v[1] << 26);
  w[0] = (u32)acc; acc >>= 32;
  acc += (u64)f.v[2] << 19; w[1] = (u32)acc; acc >>= 32;
  acc += (u64)f.v[3] << 13; w[2] = (u32)acc; acc >>= 32;
  acc += (u64)f.v[4] << 6;  w[3] = (u32)acc; acc >>= 32;
  acc += (u64)f.v[5] + ((u64)f.v[6] << 25); w[4] = (u32)acc; acc >>= 32;
  acc += (u64)f.v[7] << 19; w[5] = (u32)acc; acc >>= 32;
  acc += (u64)f.v[8] << 12; w[6] = (u32)acc; acc >>= 32;
  acc += (u64)f.v[9] << 6;  w[7] = (u32)acc;
  EG_REQUIRE((acc >> 32) == 0, "fe_pack8: value does not fit 256 bits");
}
EG_HD u32 eg_funnel(u32 hi, u32 lo, int s) { return (hi << (32 - s)) | (lo >> s); }      // v_alignbit_b32
EG_HD void fe_unpack8(fe& f, const u32 w[8]) {
  f.v[0] = w[0] & 0x3ffffffu;
  f.v[1] = eg_funnel(w[1], w[0], 26) & 0x1ffffffu;
  f.v[2] = eg_funnel(w[2], w[1], 19) & 0x3ffffffu;
  f.v[3] = eg_funnel(w[3], w[2], 13) & 0x1ffffffu;
  f.v[4] = w[3] >> 6;
  f.v[5] = w[4] & 0x1ffffffu;
  f.v[6] = eg_funnel(w[5], w[4], 25) & 0x3ffffffu;
  f.v[7] = eg_funnel(w[6], w[5], 19) & 0x1ffffffu;
  f.v[8] = eg_funnel(w[7], w[6], 12) & 0x3ffffffu;
  f.v[9] = w[7] >> 6;
  EG_SETCLS(f, 1.0f);
  fe_check_values(f);
}

// ---- constants (values checked against the oracle / SURVEY Appendix E in tests) --------------------
#ifdef EG_BOUNDCHECK
#define EG_FE_CONST(...) {{__VA_ARGS__}, 1.0f}
#else
#define EG_FE_CONST(...) {{__VA_ARGS__}}
#endif
#include "eg_constants.cuh"

// z^((p-5)/8) = z^(2^252-3)
EG_HD void fe_pow22523(fe& out, const fe& z) {
  fe t0, t1, t2;
  fe_sq(t0, z);
  fe_sqn(t1, t0, 2);
  fe_mul(t1, z, t1);
  fe_mul(t0, t0, t1);
  fe_sq(t0, t0);
  fe_mul(t0, t1, t0);
  fe_sqn(t1, t0, 5);
  fe_mul(t0, t1, t0);
  fe_sqn(t1, t0, 10);
  fe_mul(t1, t1, t0);
  fe_sqn(t2, t1, 20);
  fe_mul(t1, t2, t1);
  fe_sqn(t1, t1, 10);
  fe_mul(t0, t1, t0);
  fe_sqn(t1, t0, 50);
  fe_mul(t1, t1, t0);
  fe_sqn(t2, t1, 100);
  fe_mul(t1, t2, t1);
  fe_sqn(t1, t1, 50);
  fe_mul(t0, t1, t0);
  fe_sqn(t0, t0, 2);
  fe_mul(out, t0, z);
}

// z^(p-2)
EG_HD void fe_invert(fe& out, const fe& z) {
  fe t0, t1, t2, t3;
  fe_sq(t0, z);
  fe_sqn(t1, t0, 2);
  fe_mul(t1, z, t1);
  fe_mul(t0, t0, t1);
  fe_sq(t2, t0);
  fe_mul(t1, t1, t2);
  fe_sqn(t2, t1, 5);
  fe_mul(t1, t2, t1);
  fe_sqn(t2, t1, 10);
  fe_mul(t2, t2, t1);
  fe_sqn(t3, t2, 20);
  fe_mul(t2, t3, t2);
  fe_sqn(t2, t2, 10);
  fe_mul(t1, t2, t1);
  fe_sqn(t2, t1, 50);
  fe_mul(t2, t2, t1);
  fe_sqn(t3, t2, 100);
  fe_mul(t2, t3, t2);
  fe_sqn(t2, t2, 50);
  fe_mul(t1, t2, t1);
  fe_sqn(t1, t1, 5);
  fe_mul(out, t1, t0);
}

}  // namespace eg

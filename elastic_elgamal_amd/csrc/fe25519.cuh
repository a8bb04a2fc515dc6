// fe25519.cuh -- GF(2^255-19) for gfx950 (CDNA4): 9 unsigned limbs, radix 2^(255/9) (widths 29,28,28 repeating), 64-bit column sums.
//
// Why this representation (measured on MI355X, profiles/r03_ubench_valu_rates.txt, r03_ubench_field_bench.txt).  A wave64 VALU
// instruction costs ~2.25 cycles of a SIMD's issue when it is a plain 32-bit add / sub / and / xor / shift-right / move and ~4.2-4.7
// for everything else (v_mad_u64_u32, every VOP3, 64-bit shifts and adds, shift-left, v_cndmask), whatever the occupancy from two
// waves per SIMD up: a field operation is priced by its instruction count.  Rounds 1-2 used 10 limbs of 25.5 bits with the wrap
// constant pre-multiplied into one operand (100 multiply-adds + 10 v_mul_lo + an 11-step carry chain: 167 instructions per
// multiplication).  With 9 limbs 19 * limb no longer fits 32 bits, so the full 17-column product is accumulated and the eight high
// columns are folded in afterwards, each with two multiply-adds (low word x 19 into column k, high word x 19 * 2^(32 - w) into column
// k + 1): 81 + 16 + 1 multiply-adds, no pre-multiplications, a 9-step carry chain: 152 instructions, -3.5 % time per multiplication
// in the bench, squarings +2 % (62 multiply-adds against 56), and 10 % fewer registers and bytes per point, which is what lets the
// equation kernel hold three waves per SIMD without scratch (+5 % on the comb column in the bench).  Saturated 8 x 32 limbs and
// operand scanning were measured in round 1 and are slower (carry plumbing); MFMA does not apply: every product has two per-lane
// operands.  The north star's "4 x u64 limbs" do not exist on this ISA (no 64 x 64 multiply).
//
// Limb i sits at bit POS(i) = ceil(85 i / 3): 0 29 57 85 114 142 170 199 227.  A product of limbs i and j lands on limb i + j times
// 2^e, e = 1 when (i mod 3, j mod 3) is (1,1), (1,2) or (2,1) and 0 otherwise (fe_mul doubles limb i of the first operand there).
//
// Bound discipline.  "class c" means limb i <= c * 2^W(i) (limb 1 a hair above for c = 1).
//   fe_mul(h, f, g): needs class(f) * class(g) <= 12.5 (column sums < 2^64) and both classes < 7.9; output class 1.
//   fe_sq(h, f):     needs class(f) <= 3.5; output class 1.
//   fe_add:          class(f) + class(g) <= 7.9.     fe_sub: class(f) + 2 (g must be class 1).
//   fe_sub4:         class(f) + 4 (g < class 4).     fe_carry: any class <= 7.9 -> class 1.
// Compiled with -DEG_BOUNDCHECK on the host (tests/hostcheck) every fe carries its class and each
// operation asserts its precondition, so any executed code path is a proof of the discipline.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define EG_HD __host__ __device__ __forceinline__
#define EG_D __device__ __forceinline__
// Keeps the backend scheduler from interleaving independent field multiplications: one multiply already has
// independent columns of ILP, while interleaving several multiplies only inflates live registers
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

#define EG_NL 9      // limbs per field element

namespace eg {

typedef uint32_t u32;
typedef uint64_t u64;

struct fe {
  u32 v[EG_NL];
#ifdef EG_BOUNDCHECK
  float cls;
#endif
};

constexpr int fe_w(int i) { return (i % 3 == 0) ? 29 : 28; }            // width of limb i
constexpr int fe_pos(int i) { return (85 * i + 2) / 3; }                // its bit position
constexpr u32 fe_mask(int i) { return (1u << fe_w(i)) - 1u; }
// the product of limbs i and j carries an extra factor 2 (their positions add up to one bit more than the position of limb i + j)
constexpr bool fe_dbl(int i, int j) { return (i % 3 == 1 && j % 3 != 0) || (i % 3 == 2 && j % 3 == 1); }

// 2 x as an ADD: v_add_u32 issues in about half the cycles of the v_lshlrev_b32 hipcc picks for x << 1 (profiles/r03_ubench_valu_rates.txt)
EG_HD u32 fe_twice(u32 x) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(EG_NO_ADD_DOUBLING)
  u32 y;
  asm("v_add_u32 %0, %1, %1" : "=v"(y) : "v"(x));
  return y;
#else
  return 2u * x;
#endif
}

#ifdef EG_BOUNDCHECK
#define EG_SETCLS(h, c) ((h).cls = (c))
#define EG_GETCLS(h) ((h).cls)
static inline void fe_check_values(const fe& f) {
  for (int i = 0; i < EG_NL; ++i) {
    double nominal = (double)(1u << fe_w(i));
    double lim = nominal * f.cls * 1.001 + 4096.0;
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
  for (int i = 0; i < EG_NL; ++i) h.v[i] = 0;
  EG_SETCLS(h, 1.0f);
}
EG_HD void fe_1(fe& h) { fe_0(h); h.v[0] = 1; }

EG_HD void fe_add(fe& h, const fe& f, const fe& g) {
#pragma unroll
  for (int i = 0; i < EG_NL; ++i) h.v[i] = f.v[i] + g.v[i];
  EG_SETCLS(h, EG_GETCLS(f) + EG_GETCLS(g));
  EG_REQUIRE(EG_GETCLS(h) <= 7.9f, "fe_add overflow");
}

// h = f + 2p - g ; g must be class 1
EG_HD void fe_sub(fe& h, const fe& f, const fe& g) {
  EG_REQUIRE(EG_GETCLS(g) <= 1.02f, "fe_sub: subtrahend must be class 1");
  float c = EG_GETCLS(f) + 2.0f; (void)c;
  EG_REQUIRE(c <= 7.9f, "fe_sub overflow");
  h.v[0] = f.v[0] + ((2u << 29) - 38u) - g.v[0];
#pragma unroll
  for (int i = 1; i < EG_NL; ++i) h.v[i] = f.v[i] + ((2u << fe_w(i)) - 2u) - g.v[i];
  EG_SETCLS(h, c);
}

// h = f + 4p - g ; g below class 4
EG_HD void fe_sub4(fe& h, const fe& f, const fe& g) {
  EG_REQUIRE(EG_GETCLS(g) <= 3.9f, "fe_sub4: subtrahend class too large");
  float c = EG_GETCLS(f) + 4.0f; (void)c;
  EG_REQUIRE(c <= 7.9f, "fe_sub4 overflow");
  h.v[0] = f.v[0] + ((4u << 29) - 76u) - g.v[0];
#pragma unroll
  for (int i = 1; i < EG_NL; ++i) h.v[i] = f.v[i] + ((4u << fe_w(i)) - 4u) - g.v[i];
  EG_SETCLS(h, c);
}

// weak reduction to class 1 (one carry sweep + wrap)
EG_HD void fe_carry(fe& h) {
  EG_REQUIRE(EG_GETCLS(h) <= 7.9f, "fe_carry input too large");
  u32 c;
#pragma unroll
  for (int i = 0; i < EG_NL - 1; ++i) { c = h.v[i] >> fe_w(i); h.v[i] &= fe_mask(i); h.v[i + 1] += c; }
  c = h.v[8] >> 28; h.v[8] &= fe_mask(8); h.v[0] += 19u * c;
  c = h.v[0] >> 29; h.v[0] &= fe_mask(0); h.v[1] += c;
  EG_SETCLS(h, 1.0f);
}

EG_HD void fe_neg(fe& h, const fe& f) {  // class 1 in -> class 2 out (0 + 2p - f)
  fe z; fe_0(z);
  EG_SETCLS(z, 0.0f);
  fe_sub(h, z, f);
}

// One low column, finished: seed + its products are already in acc; adds the folds of the high columns, cuts the limb, returns the carry.
// The carry of column k is the SEED of column k + 1's multiply-add chain (the addend of its first v_mad_u64_u32), so the carry chain
// needs no 64-bit additions: per column one 64-bit shift and one mask.  (The independent-columns form it was measured against - 17 column
// sums, then a carry chain of 64-bit adds: -3.3 %, profiles/r03_ab_experiments.txt - is tools/variants/fe_unseeded.patch.)
// hipcc reassociates carry + sum of products into (sum of products) + carry, i.e. a chain seeded with 0 and a separate 64-bit add; the
// empty asm pins "seed + first product" as one value, which selects to ONE v_mad_u64_u32 with the carry as its addend.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(EG_NO_SEED_FENCE)
#define EG_SEED_FENCE(x) asm("" : "+v"(x))
#else
#define EG_SEED_FENCE(x) ((void)0)
#endif
#define EG_FE_COLUMNS_LOW(PRODUCTS)                                                                          \
  u64 carry = 0;                                                                                            \
  _Pragma("unroll") for (int k = 0; k < EG_NL; ++k) {                                                       \
    u64 acc = carry;                                                                                        \
    PRODUCTS                                                                                                \
    if (k + EG_NL < 2 * EG_NL - 1) { acc += (u64)(u32)hc[k] * 19u; EG_SEED_FENCE(acc); }                    \
    if (k >= 1) acc += (u64)(u32)(hc[k - 1] >> 32) * (19u << (32 - fe_w(k - 1)));                           \
    h.v[k] = (u32)acc & fe_mask(k);                                                                         \
    carry = acc >> fe_w(k);                                                                                 \
  }                                                                                                         \
  {                                                                                                         \
    u64 c0 = (u64)h.v[0] + 19ull * carry;          /* carry < 2^36: wraps around to limb 0 times 19 */      \
    h.v[0] = (u32)c0 & fe_mask(0);                                                                          \
    h.v[1] += (u32)(c0 >> 29);                                                                              \
  }                                                                                                         \
  EG_SETCLS(h, 1.0f);

EG_HD void fe_mul(fe& h, const fe& f, const fe& g) {
  EG_REQUIRE(EG_GETCLS(f) * EG_GETCLS(g) <= 12.5f, "fe_mul: class product > 12.5");
  EG_REQUIRE(EG_GETCLS(f) <= 7.9f && EG_GETCLS(g) <= 7.9f, "fe_mul: operand class > 7.9");
  fe_check_values(f); fe_check_values(g);
  EG_COUNT_MUL();
  EG_SCHED_FENCE();
  u32 f2[EG_NL];
#pragma unroll
  for (int i = 0; i < EG_NL; ++i) f2[i] = (i % 3 != 0) ? fe_twice(f.v[i]) : 0u;
  u64 hc[EG_NL - 1];                        // columns 9..16
#pragma unroll
  for (int k = EG_NL; k < 2 * EG_NL - 1; ++k) {
    u64 acc = 0;
#pragma unroll
    for (int i = k - EG_NL + 1; i < EG_NL; ++i) acc += (u64)(fe_dbl(i, k - i) ? f2[i] : f.v[i]) * g.v[k - i];
    hc[k - EG_NL] = acc;
  }
  fe r;                                     // h may alias f or g
  {
    fe& h = r;
    EG_FE_COLUMNS_LOW(
      _Pragma("unroll") for (int i = 0; i <= k; ++i) {
        acc += (u64)(fe_dbl(i, k - i) ? f2[i] : f.v[i]) * g.v[k - i];
        EG_SEED_FENCE(acc);
      }
    )
  }
  h = r;
  EG_SCHED_FENCE();
}

EG_HD void fe_sq(fe& h, const fe& f) {
  EG_REQUIRE(EG_GETCLS(f) <= 3.51f, "fe_sq: operand class > 3.5");
  fe_check_values(f);
  EG_COUNT_SQ();
  EG_SCHED_FENCE();
  u32 d[EG_NL];
#pragma unroll
  for (int i = 0; i < EG_NL; ++i) d[i] = fe_twice(f.v[i]);
#define EG_SQ_TERMS(K)                                                                                      \
  _Pragma("unroll") for (int i = 0; i < EG_NL; ++i) {                                                       \
    const int j = (K) - i;                                                                                  \
    if (j < i || j >= EG_NL) continue;                                                                      \
    if (i == j) acc += (u64)(fe_dbl(i, i) ? d[i] : f.v[i]) * f.v[i];                                        \
    else acc += (u64)d[i] * (fe_dbl(i, j) ? d[j] : f.v[j]);          /* cross terms count twice */          \
    EG_SEED_FENCE(acc);                                                                                     \
  }
  u64 hc[EG_NL - 1];
#pragma unroll
  for (int k = EG_NL; k < 2 * EG_NL - 1; ++k) {
    u64 acc = 0;
    EG_SQ_TERMS(k)
    hc[k - EG_NL] = acc;
  }
  fe r;
  {
    fe& h = r;
    EG_FE_COLUMNS_LOW(EG_SQ_TERMS(k))
  }
  h = r;
#undef EG_SQ_TERMS
  EG_SCHED_FENCE();
}

EG_HD void fe_sqn(fe& h, const fe& f, int n) {
  fe_sq(h, f);
  for (int i = 1; i < n; ++i) fe_sq(h, h);
}

// ---- 256-bit packing (wire encodings; table entries: device_io.cuh BaseTable) --------------------------------------------------------
// A class-1 element (every limb within its width, limb 1 a hair above: what fe_mul, fe_sq and fe_carry return) is < 2^255 + 2^41 as an
// integer: eight 32-bit words.  Unpacking slices the integer again: limbs 0..7 within their widths, limb 8 <= 2^28, i.e. class 1.
EG_HD void fe_pack8(u32 w[8], const fe& f) {
  EG_REQUIRE(EG_GETCLS(f) <= 1.02f, "fe_pack8: operand must be class 1");
  fe_check_values(f);
  u64 acc = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
#pragma unroll
    for (int i = 0; i < EG_NL; ++i)
      if (fe_pos(i) >= 32 * q && fe_pos(i) < 32 * (q + 1)) acc += (u64)f.v[i] << (fe_pos(i) - 32 * q);
    w[q] = (u32)acc; acc >>= 32;
  }
  EG_REQUIRE(acc == 0, "fe_pack8: value does not fit 256 bits");
}
EG_HD u32 eg_funnel(u32 hi, u32 lo, int s) { return (hi << (32 - s)) | (lo >> s); }      // v_alignbit_b32
EG_HD void fe_unpack_raw(fe& f, const u32 w[8]) {          // limbs 0..7 within their widths, limb 8 = bits 227..255
#pragma unroll
  for (int i = 0; i < EG_NL; ++i) {
    const int q = fe_pos(i) >> 5, s = fe_pos(i) & 31;
    u32 v;
    if (s == 0) v = w[q];
    else if (s + fe_w(i) > 32 && q + 1 < 8) v = eg_funnel(w[q + 1], w[q], s);
    else v = w[q] >> s;
    f.v[i] = (i == EG_NL - 1) ? v : (v & fe_mask(i));
  }
}
EG_HD void fe_unpack8(fe& f, const u32 w[8]) {             // a packed class-1 element: the top limb is <= 2^28, class 1 again
  fe_unpack_raw(f, w);
  EG_SETCLS(f, 1.0f);
  fe_check_values(f);
}

// ---- byte codec ---------------------------------------------------------------------------------
// w[0..7] = little-endian 32-bit words of the 32-byte encoding; bit 255 is ignored (as dalek does).
EG_HD void fe_from_words(fe& h, const u32 w[8]) {
  fe_unpack_raw(h, w);
  h.v[EG_NL - 1] &= fe_mask(EG_NL - 1);
  EG_SETCLS(h, 1.0f);
}

// canonical (fully reduced) words
EG_HD void fe_to_words(u32 w[8], const fe& f) {
  fe t = f;
  fe_carry(t);
  fe_carry(t);
  // t < 2^255 + small; q = 1 iff t >= p
  u32 q = (t.v[0] + 19u) >> 29;
#pragma unroll
  for (int i = 1; i < EG_NL; ++i) q = (t.v[i] + q) >> fe_w(i);
  t.v[0] += 19u * q;
  u32 c;
#pragma unroll
  for (int i = 0; i < EG_NL - 1; ++i) { c = t.v[i] >> fe_w(i); t.v[i] &= fe_mask(i); t.v[i + 1] += c; }
  t.v[EG_NL - 1] &= fe_mask(EG_NL - 1);
  fe_pack8(w, t);
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
  for (int i = 0; i < EG_NL; ++i) h.v[i] = flag ? g.v[i] : h.v[i];
#ifdef EG_BOUNDCHECK
  if (g.cls > h.cls) h.cls = g.cls;
#endif
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

// fe10.cuh -- the field representation of rounds 1-2 (10 limbs, radix 2^25.5, wrap constant pre-multiplied), frozen here as the
// baseline of tools/ubench/field_bench.hip after elastic_elgamal_amd/csrc/fe25519.cuh moved to 9 limbs in round 3.  Measurement tool only.
#pragma once
#include <stdint.h>
namespace eg10 {
typedef uint32_t u32;
typedef uint64_t u64;
#define E10 __device__ __forceinline__
#define E10_FENCE() __builtin_amdgcn_sched_barrier(0)
struct fe { u32 v[10]; };
E10 void fe_0(fe& h) {
#pragma unroll
  for (int i = 0; i < 10; ++i) h.v[i] = 0;
}
E10 void fe_1(fe& h) { fe_0(h); h.v[0] = 1; }

E10 void fe_add(fe& h, const fe& f, const fe& g) {
#pragma unroll
  for (int i = 0; i < 10; ++i) h.v[i] = f.v[i] + g.v[i];
}

// h = f + 2p - g ; g must be class 1
E10 void fe_sub(fe& h, const fe& f, const fe& g) {
  h.v[0] = f.v[0] + 0x7ffffdau - g.v[0];
#pragma unroll
  for (int i = 1; i < 10; ++i) h.v[i] = f.v[i] + ((i & 1) ? 0x3fffffeu : 0x7fffffeu) - g.v[i];
}

// h = f + 4p - g ; g up to class 3.3 (in fact < 4)
E10 void fe_sub4(fe& h, const fe& f, const fe& g) {
  h.v[0] = f.v[0] + 0xfffffb4u - g.v[0];
#pragma unroll
  for (int i = 1; i < 10; ++i) h.v[i] = f.v[i] + ((i & 1) ? 0x7fffffcu : 0xffffffcu) - g.v[i];
}

// weak reduction to class 1 (one carry sweep + wrap)
E10 void fe_carry(fe& h) {
  u32 c;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int bits = (i & 1) ? 25 : 26;
    c = h.v[i] >> bits; h.v[i] &= ((1u << bits) - 1); h.v[i + 1] += c;
  }
  c = h.v[9] >> 25; h.v[9] &= 0x1ffffffu; h.v[0] += 19u * c;
  c = h.v[0] >> 26; h.v[0] &= 0x3ffffffu; h.v[1] += c;
}

E10 void fe_neg(fe& h, const fe& f) {  // class 1 in -> class 3 out (0 + 2p - f)
  fe z; fe_0(z);
  fe_sub(h, z, f);
}

E10 void fe_reduce_columns(fe& h, u64 c[10]) {
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
}

E10 void fe_mul(fe& h, const fe& f, const fe& g) {
  
  E10_FENCE();
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
  E10_FENCE();
}

E10 void fe_sq(fe& h, const fe& f) {
  
  E10_FENCE();
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
  E10_FENCE();
}

E10 void fe_sqn(fe& h, const fe& f, int n) {
  fe_sq(h, f);
  for (int i = 1; i < n; ++i) fe_sq(h, h);
}

// ---- byte codec ---------------------------------------------------------------------------------
// w[0..7] = little-endian 32-bit words of the 32-byte encoding; bit 255 is ignored (as dalek does).
E10 void fe_from_words(fe& h, const u32 w[8]) {
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
}

// canonical (fully reduced) words
E10 void fe_to_words(u32 w[8], const fe& f) {
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

E10 bool fe_isnegative(const fe& f) { u32 w[8]; fe_to_words(w, f); return w[0] & 1; }
E10 bool fe_iszero(const fe& f) {
  u32 w[8]; fe_to_words(w, f);
  u32 r = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r |= w[i];
  return r == 0;
}
E10 bool fe_eq(const fe& f, const fe& g) {
  u32 a[8], b[8]; fe_to_words(a, f); fe_to_words(b, g);
  u32 r = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r |= a[i] ^ b[i];
  return r == 0;
}
// h = flag ? g : h   (both must already be in comparable classes; class becomes the max)
E10 void fe_cmov(fe& h, const fe& g, bool flag) {
#pragma unroll
  for (int i = 0; i < 10; ++i) h.v[i] = flag ? g.v[i] : h.v[i];
#ifdef EG_BOUNDCHECK
  if (g.cls > h.cls) h.cls = g.cls;
#endif
}

// ---- points ------------------------------------------------------------------------------------------------------------
struct ge { fe X, Y, Z, T; };
struct ge_p2 { fe X, Y, Z; };
struct ge_p1p1 { fe X, Y, Z, T; };
struct ge_cached { fe YpX, YmX, Z2, T2d; };
E10 void ge_dbl(ge_p1p1& r, const fe& X, const fe& Y, const fe& Z) {
  fe xx, yy, b2, a;
  fe_sq(xx, X);
  fe_sq(yy, Y);
  fe_sq(b2, Z); fe_add(b2, b2, b2);
  fe_add(a, X, Y);
  fe_sq(a, a);
  fe_add(r.Y, yy, xx);
  fe_sub(r.Z, yy, xx);
  fe_sub4(r.X, a, r.Y);
  fe_sub4(r.T, b2, r.Z);
  fe_carry(r.T);
}
E10 void ge_dbl_to_p2(ge_p2& r, const ge_p1p1& p) {
  fe_mul(r.X, p.X, p.T);
  fe_mul(r.Y, p.Z, p.Y);
  fe_mul(r.Z, p.Z, p.T);
}
E10 void ge_dbl_to_p3(ge& r, const ge_p1p1& p) {
  fe_mul(r.X, p.X, p.T);
  fe_mul(r.Y, p.Z, p.Y);
  fe_mul(r.Z, p.Z, p.T);
  fe_mul(r.T, p.X, p.Y);
}
E10 void ge_add(ge_p1p1& r, const ge& p, const ge_cached& q) {
  fe a, b, t0;
  fe_add(a, p.Y, p.X);
  fe_sub(b, p.Y, p.X);
  fe_mul(r.Z, a, q.YpX);
  fe_mul(r.Y, b, q.YmX);
  fe_mul(r.T, p.T, q.T2d);
  fe_mul(t0, p.Z, q.Z2);
  fe_sub(r.X, r.Z, r.Y);
  fe_add(r.Y, r.Z, r.Y);
  fe_add(r.Z, t0, r.T);
  fe_sub(r.T, t0, r.T);
}
E10 void ge_add_to_p2(ge_p2& r, const ge_p1p1& p) {
  fe_mul(r.X, p.T, p.X);
  fe_mul(r.Y, p.Z, p.Y);
  fe_mul(r.Z, p.T, p.Z);
}
E10 void ge_cached_cneg(ge_cached& c, bool neg) {
  fe t = c.YpX; fe_cmov(c.YpX, c.YmX, neg); fe_cmov(c.YmX, t, neg);
  fe n; fe_neg(n, c.T2d);
  fe_cmov(c.T2d, n, neg);
}
}  // namespace eg10

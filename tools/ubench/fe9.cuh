// fe9.cuh -- candidate GF(2^255-19) representation for gfx950: 9 unsigned limbs, radix 2^(255/9) (widths 29,28,28 repeating),
// 64-bit column sums over the full 17-column product, high columns folded with the wrap constant 19 AFTER accumulation.
// Measurement tool (tools/ubench/field_bench.hip); the winner of the bench becomes elastic_elgamal_amd/csrc/fe25519.cuh.
//   multiply: 81 + 16 (fold) + 1 (top carry) v_mad_u64_u32, no x19 pre-multiplications, 9-step carry chain
//   square:   45 + 16 + 1
// Class c: limb i <= c * 2^w(i).  fe_mul needs class(f) * class(g) <= 12.5 (column sums < 2^64), fe_sq class <= 3.5.
#pragma once
#include <stdint.h>
namespace eg9 {
typedef uint32_t u32;
typedef uint64_t u64;
#define E9 __device__ __forceinline__
#define E9_FENCE() __builtin_amdgcn_sched_barrier(0)
struct fe { u32 v[9]; };
constexpr int W(int i) { return (i % 3 == 0) ? 29 : 28; }
constexpr u32 MASK(int i) { return (1u << W(i)) - 1u; }
constexpr bool DBL(int i, int j) { return (i % 3 == 1 && j % 3 != 0) || (i % 3 == 2 && j % 3 == 1); }

E9 void fe_0(fe& h) {
#pragma unroll
  for (int i = 0; i < 9; ++i) h.v[i] = 0;
}
E9 void fe_1(fe& h) { fe_0(h); h.v[0] = 1; }
E9 void fe_add(fe& h, const fe& f, const fe& g) {
#pragma unroll
  for (int i = 0; i < 9; ++i) h.v[i] = f.v[i] + g.v[i];
}
// h = f + 2p - g  (g class 1)
E9 void fe_sub(fe& h, const fe& f, const fe& g) {
  h.v[0] = f.v[0] + ((2u << 29) - 38u) - g.v[0];
#pragma unroll
  for (int i = 1; i < 9; ++i) h.v[i] = f.v[i] + ((2u << W(i)) - 2u) - g.v[i];
}
// h = f + 4p - g  (g class < 4)
E9 void fe_sub4(fe& h, const fe& f, const fe& g) {
  h.v[0] = f.v[0] + ((4u << 29) - 76u) - g.v[0];
#pragma unroll
  for (int i = 1; i < 9; ++i) h.v[i] = f.v[i] + ((4u << W(i)) - 4u) - g.v[i];
}
E9 void fe_carry(fe& h) {
  u32 c;
#pragma unroll
  for (int i = 0; i < 8; ++i) { c = h.v[i] >> W(i); h.v[i] &= MASK(i); h.v[i + 1] += c; }
  c = h.v[8] >> 28; h.v[8] &= MASK(8); h.v[0] += 19u * c;
  c = h.v[0] >> 29; h.v[0] &= MASK(0); h.v[1] += c;
}
E9 void fe_neg(fe& h, const fe& f) { fe z; fe_0(z); fe_sub(h, z, f); }
E9 void fe_cmov(fe& h, const fe& g, bool flag) {
#pragma unroll
  for (int i = 0; i < 9; ++i) h.v[i] = flag ? g.v[i] : h.v[i];
}

// c[0..16] -> h (class 1)
E9 void fe_reduce_columns(fe& h, u64 c[17]) {
#pragma unroll
  for (int k = 9; k < 17; ++k) {
    const u32 lo = (u32)c[k], hi = (u32)(c[k] >> 32);
    c[k - 9] += (u64)lo * 19u;
    c[k - 8] += (u64)hi * (19u << (32 - W(k - 9)));
  }
  u64 t;
#pragma unroll
  for (int i = 0; i < 8; ++i) { t = c[i] >> W(i); c[i] &= MASK(i); c[i + 1] += t; }
  t = c[8] >> 28; c[8] &= MASK(8); c[0] += 19ull * t;
  t = c[0] >> 29; c[0] &= MASK(0); c[1] += t;
#pragma unroll
  for (int i = 0; i < 9; ++i) h.v[i] = (u32)c[i];
}

E9 void fe_mul(fe& h, const fe& f, const fe& g) {
  E9_FENCE();
  u32 f2[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) f2[i] = 2u * f.v[i];
  u64 c[17];
#pragma unroll
  for (int k = 0; k < 17; ++k) {
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int j = k - i;
      if (j < 0 || j > 8) continue;
      acc += (u64)(DBL(i, j) ? f2[i] : f.v[i]) * g.v[j];
    }
    c[k] = acc;
  }
  fe_reduce_columns(h, c);
  E9_FENCE();
}

E9 void fe_sq(fe& h, const fe& f) {
  E9_FENCE();
  u32 d[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) d[i] = 2u * f.v[i];
  u64 c[17];
#pragma unroll
  for (int k = 0; k < 17; ++k) {
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int j = k - i;
      if (j < i || j > 8) continue;
      if (i == j) acc += (u64)(DBL(i, i) ? d[i] : f.v[i]) * f.v[i];
      else acc += (u64)d[i] * (DBL(i, j) ? d[j] : f.v[j]);
    }
    c[k] = acc;
  }
  fe_reduce_columns(h, c);
  E9_FENCE();
}

// limbs <-> 256-bit little-endian words
constexpr int POS(int i) { return (85 * i + 2) / 3; }     // ceil(85 i / 3): 0 29 57 85 114 142 170 199 227
E9 void fe_from_words(fe& h, const u32 w[8]) {
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int q = POS(i) >> 5, s = POS(i) & 31;
    u32 v = w[q] >> s;
    if (s + W(i) > 32 && q + 1 < 8) v |= w[q + 1] << (32 - s);
    h.v[i] = v & MASK(i);
  }
}
E9 void fe_pack8(u32 w[8], const fe& f) {       // class 1 in
  u64 acc = 0;
  int have = 0, limb = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    // add every limb whose position starts below bit 32 (q + 1)
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      if (POS(i) >= 32 * q && POS(i) < 32 * (q + 1)) acc += (u64)f.v[i] << (POS(i) - 32 * q);
    }
    w[q] = (u32)acc; acc >>= 32;
  }
  (void)have; (void)limb;
}

// ---- points ------------------------------------------------------------------------------------------------------------
struct ge { fe X, Y, Z, T; };
struct ge_p2 { fe X, Y, Z; };
struct ge_p1p1 { fe X, Y, Z, T; };
struct ge_cached { fe YpX, YmX, Z2, T2d; };
E9 void ge_dbl(ge_p1p1& r, const fe& X, const fe& Y, const fe& Z) {
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
E9 void ge_dbl_to_p3(ge& r, const ge_p1p1& p) {
  fe_mul(r.X, p.X, p.T);
  fe_mul(r.Y, p.Z, p.Y);
  fe_mul(r.Z, p.Z, p.T);
  fe_mul(r.T, p.X, p.Y);
}
E9 void ge_add(ge_p1p1& r, const ge& p, const ge_cached& q) {
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
E9 void ge_add_to_p2(ge_p2& r, const ge_p1p1& p) {
  fe_mul(r.X, p.T, p.X);
  fe_mul(r.Y, p.Z, p.Y);
  fe_mul(r.Z, p.T, p.Z);
}
E9 void ge_cached_cneg(ge_cached& c, bool neg) {
  fe t = c.YpX; fe_cmov(c.YpX, c.YmX, neg); fe_cmov(c.YmX, t, neg);
  fe n; fe_neg(n, c.T2d);
  fe_cmov(c.T2d, n, neg);
}
}  // namespace eg9
namespace eg9 {
// canonical little-endian words (fully reduced)
E9 void fe_to_words(u32 w[8], const fe& f) {
  fe t = f;
  fe_carry(t);
  fe_carry(t);
  u32 q = (t.v[0] + 19u) >> 29;
#pragma unroll
  for (int i = 1; i < 9; ++i) q = (t.v[i] + q) >> W(i);
  t.v[0] += 19u * q;
  u32 c;
#pragma unroll
  for (int i = 0; i < 8; ++i) { c = t.v[i] >> W(i); t.v[i] &= MASK(i); t.v[i + 1] += c; }
  t.v[8] &= MASK(8);
  fe_pack8(w, t);
}
}  // namespace eg9

// sc25519.cuh -- arithmetic modulo the group order l = 2^252 + 27742317777372353535851937790883648493.
//
// Replaces the dalek Scalar operations the reference reaches through src/group/ristretto.rs:
// from_bytes_mod_order_wide (:34-38, every Fiat-Shamir challenge), from_canonical_bytes (:59-62),
// Neg (ring.rs:339, log_equality.rs:160), and a*b+c for the prover (ring.rs:192-193).
// Method: 21-bit signed limbs in 64-bit lanes; 2^252 is folded with the signed-digit expansion of -c
// (the classic ed25519 "sc_reduce" schedule).  All loops are compile-time unrolled so the 24 limbs stay
// in registers.  Scalars travel as 8 little-endian 32-bit words.
#pragma once
#include "fe25519.cuh"

namespace eg {

typedef int64_t i64;

// 2^252 = sum M21[k] * 2^(21k)  (mod l)
#define EG_SC_M0 666643
#define EG_SC_M1 470296
#define EG_SC_M2 654183
#define EG_SC_M3 (-997805)
#define EG_SC_M4 136657
#define EG_SC_M5 (-683901)

EG_HD void sc_fold(i64 s[24], int i) {
  s[i - 12] += s[i] * EG_SC_M0;
  s[i - 11] += s[i] * EG_SC_M1;
  s[i - 10] += s[i] * EG_SC_M2;
  s[i - 9] += s[i] * EG_SC_M3;
  s[i - 8] += s[i] * EG_SC_M4;
  s[i - 7] += s[i] * EG_SC_M5;
  s[i] = 0;
}
EG_HD void sc_carry_centered(i64 s[24], int i) {
  const i64 c = (s[i] + (1 << 20)) >> 21;
  s[i + 1] += c;
  s[i] -= c * (1 << 21);
}
EG_HD void sc_carry_floor(i64 s[24], int i) {
  const i64 c = s[i] >> 21;
  s[i + 1] += c;
  s[i] -= c * (1 << 21);
}

// s[0..23] (|s[i]| small enough, s[23] may be wide) -> canonical scalar words
EG_HD void sc_reduce_limbs(u32 out[8], i64 s[24]) {
#pragma unroll
  for (int i = 23; i >= 18; --i) sc_fold(s, i);
#pragma unroll
  for (int i = 6; i <= 16; i += 2) sc_carry_centered(s, i);
#pragma unroll
  for (int i = 7; i <= 15; i += 2) sc_carry_centered(s, i);
#pragma unroll
  for (int i = 17; i >= 12; --i) sc_fold(s, i);
#pragma unroll
  for (int i = 0; i <= 10; i += 2) sc_carry_centered(s, i);
#pragma unroll
  for (int i = 1; i <= 11; i += 2) sc_carry_centered(s, i);
  sc_fold(s, 12);
#pragma unroll
  for (int i = 0; i <= 11; ++i) sc_carry_floor(s, i);
  sc_fold(s, 12);
#pragma unroll
  for (int i = 0; i <= 10; ++i) sc_carry_floor(s, i);
  // pack 12 x 21 bits
  u64 acc = 0;
  int bits = 0, wi = 0;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    acc |= (u64)s[i] << bits;
    bits += 21;
    if (bits >= 32) {
      out[wi++] = (u32)acc;
      acc >>= 32;
      bits -= 32;
    }
  }
  out[7] = (u32)acc;
}

// bits [off, off+21) of a little-endian word array
template <int NW>
EG_HD i64 sc_bits21(const u32* w, int off) {
  const int wi = off >> 5, sh = off & 31;
  u64 v = w[wi];
  if (wi + 1 < NW) v |= (u64)w[wi + 1] << 32;
  return (i64)((v >> sh) & 0x1fffffu);
}

// Scalar::from_bytes_mod_order_wide: 64 bytes (16 LE words) -> canonical scalar
EG_HD void sc_from_wide(u32 out[8], const u32 w[16]) {
  i64 s[24];
#pragma unroll
  for (int i = 0; i < 23; ++i) s[i] = sc_bits21<16>(w, 21 * i);
  s[23] = (i64)(w[15] >> 3);   // bits 483..511
  sc_reduce_limbs(out, s);
}

// a*b + c mod l  (all canonical or at least < 2^256)
EG_HD void sc_muladd(u32 out[8], const u32 a[8], const u32 b[8], const u32 c[8]) {
  i64 al[12], bl[12], s[24];
#pragma unroll
  for (int i = 0; i < 11; ++i) { al[i] = sc_bits21<8>(a, 21 * i); bl[i] = sc_bits21<8>(b, 21 * i); }
  al[11] = (i64)(a[7] >> 7); bl[11] = (i64)(b[7] >> 7);
#pragma unroll
  for (int i = 0; i < 11; ++i) s[i] = sc_bits21<8>(c, 21 * i);
  s[11] = (i64)(c[7] >> 7);
#pragma unroll
  for (int i = 12; i < 24; ++i) s[i] = 0;
#pragma unroll
  for (int i = 0; i < 12; ++i)
#pragma unroll
    for (int j = 0; j < 12; ++j) s[i + j] += al[i] * bl[j];
#pragma unroll
  for (int i = 0; i <= 22; i += 2) sc_carry_centered(s, i);
#pragma unroll
  for (int i = 1; i <= 21; i += 2) sc_carry_centered(s, i);
  sc_reduce_limbs(out, s);
}

EG_HD void sc_mul(u32 out[8], const u32 a[8], const u32 b[8]) {
  const u32 z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  sc_muladd(out, a, b, z);
}
EG_HD void sc_add(u32 out[8], const u32 a[8], const u32 b[8]) {
  const u32 one[8] = {1, 0, 0, 0, 0, 0, 0, 0};
  sc_muladd(out, a, one, b);
}

// l - a (0 for a = 0); a canonical
EG_HD void sc_neg(u32 out[8], const u32 a[8]) {
  const u32 l[8] = EG_L_WORDS;
  u32 nz = 0;
  i64 borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    nz |= a[i];
    const i64 t = (i64)l[i] - (i64)a[i] + borrow;
    out[i] = (u32)t;
    borrow = t >> 32;   // 0 or -1
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = nz ? out[i] : 0u;
}

// h with 2h = s (mod l): (s + (s odd ? l : 0)) >> 1.  The result is < 2^252 + 2^251 and is used only as a
// multiplier of points (it is not a canonical scalar when s is odd: it represents (s + l) / 2).
EG_HD void sc_halve(u32 out[8], const u32 s[8]) {
  const u32 l[8] = EG_L_WORDS;
  const u32 odd = s[0] & 1u;
  u64 carry = 0;
  u32 t[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const u64 v = (u64)s[i] + (odd ? l[i] : 0u) + carry;
    t[i] = (u32)v;
    carry = v >> 32;
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) out[i] = (t[i] >> 1) | (t[i + 1] << 31);
  out[7] = (t[7] >> 1) | ((u32)carry << 31);
}

// a^(l-2) mod l = 1/a (0 for a = 0, as Scalar::invert gives in release builds): ScalarOps::invert_scalar
// (group/mod.rs:104-107 -> ristretto.rs:40-42).  Left-to-right square and multiply over the constant exponent.
EG_HD void sc_invert(u32 out[8], const u32 a[8]) {
  const u32 l[8] = EG_L_WORDS;
  u32 e[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) e[i] = l[i];
  e[0] -= 2u;   // l is odd and l[0] >= 2: no borrow
  u32 r[8] = {1, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 252; i >= 0; --i) {
    sc_mul(r, r, r);
    u32 t[8];
    sc_mul(t, r, a);
    const bool bit = (e[i >> 5] >> (i & 31)) & 1u;
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = bit ? t[k] : r[k];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) out[k] = r[k];
}

// Scalar::from_canonical_bytes: value < l
EG_HD bool sc_is_canonical(const u32 a[8]) {
  const u32 l[8] = EG_L_WORDS;
  bool lt = false, decided = false;
#pragma unroll
  for (int i = 7; i >= 0; --i) {
    const bool less = a[i] < l[i], greater = a[i] > l[i];
    lt = decided ? lt : less;
    decided = decided | less | greater;
  }
  return lt;   // equal -> false
}

EG_HD bool sc_eq(const u32 a[8], const u32 b[8]) {
  u32 r = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r |= a[i] ^ b[i];
  return r == 0;
}

EG_HD void sc_from_u64(u32 out[8], u64 x) {
  out[0] = (u32)x; out[1] = (u32)(x >> 32);
#pragma unroll
  for (int i = 2; i < 8; ++i) out[i] = 0;
}

}  // namespace eg

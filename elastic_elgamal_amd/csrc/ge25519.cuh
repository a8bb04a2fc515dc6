// ge25519.cuh -- twisted Edwards (a = -1) point arithmetic, ristretto255 codec and scalar
// multiplication for gfx950.
//
// Replaces what src/group/ristretto.rs:65-146 of the reference delegates to curve25519-dalek:
// identity/generator/add/sub (ristretto.rs:76-86), compress (:88-90), decompress (:93-95),
// mul_generator (:105-107), vartime_double_scalar_mul_basepoint (:131-137),
// vartime_multiscalar_mul (:139-145).
// Formulas: hwcd-2008 unified addition / dedicated doubling in extended coordinates; RFC 9496
// 4.3.1 / 4.3.2 / 4.2 codec.  Ristretto encodings are canonical, so any correct scalar-multiplication
// schedule is bit-exact with dalek's.  The schedule here is a UNIFORM signed radix-16
// ladder: every lane of a wavefront runs the same instruction stream; digit signs and zero digits are
// handled with selects instead of branches (dalek's NAF is data dependent and would diverge).
//
// [n] annotations are limb classes, see the bound discipline in fe25519.cuh.
#pragma once
#include "fe25519.cuh"

namespace eg {

struct ge { fe X, Y, Z, T; };              // extended, all [1]
struct ge_p2 { fe X, Y, Z; };              // projective, all [1]
struct ge_p1p1 { fe X, Y, Z, T; };         // completed: x = X/Z, y = Y/T
struct ge_cached { fe YpX, YmX, Z2, T2d; };  // (Y+X, Y-X, 2Z, 2dT), all [1]
struct ge_niels { fe ypx, ymx, xy2d; };    // affine (y+x, y-x, 2dxy), all [1]

EG_HD void ge_identity(ge& p) { fe_0(p.X); fe_1(p.Y); fe_1(p.Z); fe_0(p.T); }
EG_HD void ge_generator(ge& p) {
  const fe bx = EG_FE_BASE_X, by = EG_FE_BASE_Y, bt = EG_FE_BASE_T;
  p.X = bx; p.Y = by; fe_1(p.Z); p.T = bt;
}

// ---- conversions out of the completed form ---------------------------------------------------------
// result of ge_dbl: {X [5], Y [2], Z [3], T [1]}
EG_HD void ge_dbl_to_p2(ge_p2& r, const ge_p1p1& p) {
  fe_mul(r.X, p.X, p.T);
  fe_mul(r.Y, p.Z, p.Y);
  fe_mul(r.Z, p.Z, p.T);
}
EG_HD void ge_dbl_to_p3(ge& r, const ge_p1p1& p) {
  fe_mul(r.X, p.X, p.T);
  fe_mul(r.Y, p.Z, p.Y);
  fe_mul(r.Z, p.Z, p.T);
  fe_mul(r.T, p.X, p.Y);
}
// result of ge_add / ge_madd: {X [3], Y [2], Z [2..3], T [3..4]}
EG_HD void ge_add_to_p2(ge_p2& r, const ge_p1p1& p) {
  fe_mul(r.X, p.T, p.X);
  fe_mul(r.Y, p.Z, p.Y);
  fe_mul(r.Z, p.T, p.Z);
}
EG_HD void ge_add_to_p3(ge& r, const ge_p1p1& p) {
  fe_mul(r.X, p.T, p.X);
  fe_mul(r.Y, p.Z, p.Y);
  fe_mul(r.Z, p.T, p.Z);
  fe_mul(r.T, p.X, p.Y);
}

EG_HD void ge_to_cached(ge_cached& c, const ge& p) {
  const fe d2 = EG_FE_2D;
  fe_add(c.YpX, p.Y, p.X); fe_carry(c.YpX);
  fe_sub(c.YmX, p.Y, p.X); fe_carry(c.YmX);
  fe_add(c.Z2, p.Z, p.Z); fe_carry(c.Z2);
  fe_mul(c.T2d, p.T, d2);
}
// the same without the three carries: (Y+X [2], Y-X [3], 2Z [2], 2dT [1]).  ge_add takes these classes as they are (the
// second operand of fe_mul may be class <= 3.3), so per-ballot table entries are stored uncarried (measured +1.3 %).
EG_HD void ge_to_cached_lazy(ge_cached& c, const ge& p) {
  const fe d2 = EG_FE_2D;
  fe_add(c.YpX, p.Y, p.X);
  fe_sub(c.YmX, p.Y, p.X);
  fe_add(c.Z2, p.Z, p.Z);
  fe_mul(c.T2d, p.T, d2);
}
EG_HD void ge_cached_identity(ge_cached& c) { fe_1(c.YpX); fe_1(c.YmX); fe_0(c.Z2); c.Z2.v[0] = 2; fe_0(c.T2d); }
EG_HD void ge_niels_identity(ge_niels& c) { fe_1(c.ypx); fe_1(c.ymx); fe_0(c.xy2d); }

// conditional negation of an addend (branch-free: lanes of one wave differ in digit sign)
EG_HD void ge_cached_cneg(ge_cached& c, bool neg) {
  fe t = c.YpX; fe_cmov(c.YpX, c.YmX, neg); fe_cmov(c.YmX, t, neg);
  fe n; fe_neg(n, c.T2d);          // [3]
  fe_cmov(c.T2d, n, neg);
}
EG_HD void ge_niels_cneg(ge_niels& c, bool neg) {
  fe t = c.ypx; fe_cmov(c.ypx, c.ymx, neg); fe_cmov(c.ymx, t, neg);
  fe n; fe_neg(n, c.xy2d);
  fe_cmov(c.xy2d, n, neg);
}

// r = p + q   (q.T2d may be [3] after a conditional negation)
EG_HD void ge_add(ge_p1p1& r, const ge& p, const ge_cached& q) {
  fe a, b, t0;
  fe_add(a, p.Y, p.X);             // [2]
  fe_sub(b, p.Y, p.X);             // [3]
  fe_mul(r.Z, a, q.YpX);           // PP
  fe_mul(r.Y, b, q.YmX);           // MM
  fe_mul(r.T, p.T, q.T2d);         // TT
  fe_mul(t0, p.Z, q.Z2);           // D = 2 Z1 Z2 [1]
  fe_sub(r.X, r.Z, r.Y);           // E = PP - MM [3]
  fe_add(r.Y, r.Z, r.Y);           // H = PP + MM [2]
  fe_add(r.Z, t0, r.T);            // G = D + TT  [2]
  fe_sub(r.T, t0, r.T);            // F = D - TT  [3]
}
// mixed addition with an affine table entry (Z2 = 1)
EG_HD void ge_madd(ge_p1p1& r, const ge& p, const ge_niels& q) {
  fe a, b, t0;
  fe_add(a, p.Y, p.X);
  fe_sub(b, p.Y, p.X);
  fe_mul(r.Z, a, q.ypx);
  fe_mul(r.Y, b, q.ymx);
  fe_mul(r.T, p.T, q.xy2d);
  fe_add(t0, p.Z, p.Z);            // D = 2 Z1 [2]
  fe_sub(r.X, r.Z, r.Y);           // E [3]
  fe_add(r.Y, r.Z, r.Y);           // H [2]
  fe_add(r.Z, t0, r.T);            // G [3]
  fe_sub(r.T, t0, r.T);            // F [4]
}
// r = 2p  (needs X, Y, Z only)
EG_HD void ge_dbl(ge_p1p1& r, const fe& X, const fe& Y, const fe& Z) {
  fe xx, yy, b2, a;
  fe_sq(xx, X);
  fe_sq(yy, Y);
  fe_sq(b2, Z); fe_add(b2, b2, b2);   // [2]
  fe_add(a, X, Y);                    // [2]
  fe_sq(a, a);                        // AA [1]
  fe_add(r.Y, yy, xx);                // H [2]
  fe_sub(r.Z, yy, xx);                // G [3]
  fe_sub4(r.X, a, r.Y);               // E = AA - H [5]
  fe_sub4(r.T, b2, r.Z);              // F = 2ZZ - G [6]
  fe_carry(r.T);                      // [1]
}

// full-width helpers (not on the hot loop)
EG_HD void ge_add_full(ge& r, const ge& p, const ge& q) {
  ge_cached c; ge_p1p1 t;
  ge_to_cached(c, q);
  ge_add(t, p, c);
  ge_add_to_p3(r, t);
}
EG_HD void ge_neg(ge& r, const ge& p) {
  r = p;
  fe_neg(r.X, p.X); fe_carry(r.X);
  fe_neg(r.T, p.T); fe_carry(r.T);
}
EG_HD void ge_sub_full(ge& r, const ge& p, const ge& q) {
  ge n; ge_neg(n, q);
  ge_add_full(r, p, n);
}
EG_HD void ge_dbl_full(ge& r, const ge& p) {
  ge_p1p1 t;
  ge_dbl(t, p.X, p.Y, p.Z);
  ge_dbl_to_p3(r, t);
}

// ---- SQRT_RATIO_M1 (RFC 9496 4.2) ----------------------------------------------------------------------
// u, v must be [1].  r = non-negative sqrt(u/v) or sqrt(i*u/v); returns was_square.
EG_HD bool fe_sqrt_ratio_m1(fe& r, const fe& u, const fe& v) {
  const fe sqrtm1 = EG_FE_SQRTM1;
  fe v3, v7, t, check, neg_u, neg_u_i, ri;
  fe_sq(v3, v); fe_mul(v3, v3, v);
  fe_sq(v7, v3); fe_mul(v7, v7, v);
  fe_mul(t, u, v7);
  fe_pow22523(t, t);
  fe_mul(r, u, v3);
  fe_mul(r, r, t);
  fe_sq(check, r);
  fe_mul(check, check, v);
  fe_neg(neg_u, u);                 // [3]
  fe_mul(neg_u_i, neg_u, sqrtm1);   // [1]
  const bool correct = fe_eq(check, u);
  const bool flipped = fe_eq(check, neg_u);
  const bool flipped_i = fe_eq(check, neg_u_i);
  fe_mul(ri, r, sqrtm1);
  fe_cmov(r, ri, flipped | flipped_i);
  fe n; fe_neg(n, r); fe_carry(n);
  fe_cmov(r, n, fe_isnegative(r));
  return correct | flipped;
}

// ---- ristretto255 decode (deserialize_element, ristretto.rs:93-95) ------------------------------------------
// w = little-endian words of the 32-byte encoding.  Returns false for an invalid encoding (p is then the
// identity so that downstream arithmetic stays well defined).
EG_HD bool ristretto_decode(ge& p, const u32 w[8]) {
  const fe d = EG_FE_D;
  fe s, one;
  fe_1(one);
  fe_from_words(s, w);
  u32 chk[8];
  fe_to_words(chk, s);
  u32 diff = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) diff |= chk[i] ^ w[i];
  bool ok = (diff == 0) & ((w[0] & 1u) == 0);   // canonical and non-negative
  fe ss, u1, u2, u2s, v, t, inv, dx, dy, x, y, tt;
  fe_sq(ss, s);
  fe_sub(u1, one, ss); fe_carry(u1);
  fe_add(u2, one, ss); fe_carry(u2);
  fe_sq(u2s, u2);
  fe_sq(t, u1);
  fe_mul(t, t, d);
  fe_add(t, t, u2s);                // d*u1^2 + u2^2 [2]
  fe_carry(t);
  fe_neg(v, t); fe_carry(v);        // v = -(d*u1^2) - u2^2
  fe_mul(t, v, u2s);
  const bool was_square = fe_sqrt_ratio_m1(inv, one, t);
  fe_mul(dx, inv, u2);
  fe_mul(dy, inv, dx);
  fe_mul(dy, dy, v);
  fe_mul(x, s, dx);
  fe_add(x, x, x); fe_carry(x);
  fe n; fe_neg(n, x); fe_carry(n);
  fe_cmov(x, n, fe_isnegative(x));
  fe_mul(y, u1, dy);
  fe_mul(tt, x, y);
  ok = ok & was_square & !fe_isnegative(tt) & !fe_iszero(y);
  ge id; ge_identity(id);
  p.X = x; p.Y = y; fe_1(p.Z); p.T = tt;
  fe_cmov(p.X, id.X, !ok); fe_cmov(p.Y, id.Y, !ok); fe_cmov(p.T, id.T, !ok);
  return ok;
}

// ---- ristretto255 encode (serialize_element, ristretto.rs:88-90) ------------------------------------------------
EG_HD void ristretto_encode(u32 w[8], const ge& p) {
  const fe sqrtm1 = EG_FE_SQRTM1, invsqrt_amd = EG_FE_INVSQRT_A_MINUS_D;
  fe one; fe_1(one);
  fe u1, u2, t0, t1, inv, d1, d2, zinv, x, y, dinv, s;
  fe_add(t0, p.Z, p.Y);             // [2]
  fe_sub(t1, p.Z, p.Y);             // [3]
  fe_mul(u1, t0, t1);
  fe_mul(u2, p.X, p.Y);
  fe_sq(t0, u2);
  fe_mul(t0, t0, u1);
  fe_sqrt_ratio_m1(inv, one, t0);
  fe_mul(d1, inv, u1);
  fe_mul(d2, inv, u2);
  fe_mul(zinv, d1, d2);
  fe_mul(zinv, zinv, p.T);
  fe_mul(t0, p.T, zinv);
  const bool rotate = fe_isnegative(t0);
  fe ix, iy, dr;
  fe_mul(ix, p.Y, sqrtm1);
  fe_mul(iy, p.X, sqrtm1);
  fe_mul(dr, d1, invsqrt_amd);
  x = p.X; y = p.Y; dinv = d2;
  fe_cmov(x, ix, rotate); fe_cmov(y, iy, rotate); fe_cmov(dinv, dr, rotate);
  fe_mul(t0, x, zinv);
  fe n; fe_neg(n, y); fe_carry(n);
  fe_cmov(y, n, fe_isnegative(t0));
  fe_sub(t0, p.Z, y);               // [3]
  fe_mul(s, t0, dinv);
  fe_neg(n, s); fe_carry(n);
  fe_cmov(s, n, fe_isnegative(s));
  fe_to_words(w, s);
}

// ---- encoding of a DOUBLED point with a batched inversion --------------------------------------------------------------------
// For Q = 2P the inverse square root of the encoder is rational in P: with the doubling's intermediates
// E = 2XY, H = Y^2 + X^2, G = Y^2 - X^2, F = 2Z^2 - G  (Q = (EF : GH : GF : EH)) one has
//   u1 u2^2 = (Z0^2 - Y0^2)(X0 Y0)^2 = (E F G^2 H)^2 (F^2 - H^2)   and   F^2 - H^2 = 4 (Z^2 - Y^2)(Z^2 + X^2) = (a - d) E^2
// (the last step is the curve equation), hence 1/sqrt(u1 u2^2) = +-(1/sqrt(a-d)) / N with N = E^2 F G^2 H.
// Inversions batch (Montgomery), exponentiations do not: k_encode_batch encodes all commitments of a ballot and stage
// with ONE field inversion instead of one 255-squaring chain each.  The equations are evaluated with halved scalars
// (sc_halve) so that the commitment is 2P; [l]P is in the 4-torsion that ristretto quotients out, so adding l to an odd
// scalar before halving does not change the encoding.
// prepare: N (class 1, 0 replaced by 1 with `zero` set) from P.
EG_HD void ge_double_encode_prepare(fe& n, bool& zero, const ge& p) {
  ge_p1p1 t;
  ge_dbl(t, p.X, p.Y, p.Z);           // t.X = E [5], t.Y = H [2], t.Z = G [3], t.T = F [1]
  fe e = t.X; fe_carry(e);
  fe e2, g2, a, b;
  fe_sq(e2, e);
  fe_sq(g2, t.Z);
  fe_mul(a, e2, t.T);
  fe_mul(b, t.Y, g2);                 // H [2] as f operand, G^2 [1] as g
  fe_mul(n, a, b);
  zero = fe_iszero(n);
  fe one; fe_1(one);
  fe_cmov(n, one, zero);
}
// finish: w = encode(2P) given inv_n = 1/N (ignored when zero)
EG_HD void ge_double_encode_finish(u32 w[8], const ge& p, const fe& inv_n, bool zero) {
  const fe sqrtm1 = EG_FE_SQRTM1, invsqrt_amd = EG_FE_INVSQRT_A_MINUS_D;
  ge_p1p1 t;
  ge_dbl(t, p.X, p.Y, p.Z);
  ge q;
  ge_dbl_to_p3(q, t);
  fe inv, z; fe_0(z);
  fe_mul(inv, inv_n, invsqrt_amd);
  fe_cmov(inv, z, zero);              // SQRT_RATIO_M1(1, 0) = 0
  fe n; fe_neg(n, inv); fe_carry(n);
  fe_cmov(inv, n, fe_isnegative(inv)); // the non-negative root
  // from here on identical to ristretto_encode(q)
  fe u1, u2, t0, t1, d1, d2, zinv, x, y, dinv, s2;
  fe_add(t0, q.Z, q.Y);
  fe_sub(t1, q.Z, q.Y);
  fe_mul(u1, t0, t1);
  fe_mul(u2, q.X, q.Y);
  fe_mul(d1, inv, u1);
  fe_mul(d2, inv, u2);
  fe_mul(zinv, d1, d2);
  fe_mul(zinv, zinv, q.T);
  fe_mul(t0, q.T, zinv);
  const bool rotate = fe_isnegative(t0);
  fe ix, iy, dr;
  fe_mul(ix, q.Y, sqrtm1);
  fe_mul(iy, q.X, sqrtm1);
  fe_mul(dr, d1, invsqrt_amd);
  x = q.X; y = q.Y; dinv = d2;
  fe_cmov(x, ix, rotate); fe_cmov(y, iy, rotate); fe_cmov(dinv, dr, rotate);
  fe_mul(t0, x, zinv);
  fe_neg(n, y); fe_carry(n);
  fe_cmov(y, n, fe_isnegative(t0));
  fe_sub(t0, q.Z, y);
  fe_mul(s2, t0, dinv);
  fe_neg(n, s2); fe_carry(n);
  fe_cmov(s2, n, fe_isnegative(s2));
  fe_to_words(w, s2);
}

// ---- scalar recoding -------------------------------------------------------------------------------------
// 256-bit scalar (< 2^253) -> 64 signed radix-16 digits in [-8, 7], packed as nibbles (two's complement).
EG_HD void sc_recode_radix16(u32 out[8], const u32 k[8]) {
  u32 carry = 0;
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    u32 o = 0;
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      u32 dgt = ((k[w] >> (4 * n)) & 15u) + carry;   // 0..16
      carry = (dgt + 8u) >> 4;
      o |= (dgt & 15u) << (4 * n);
    }
    out[w] = o;
  }
  // carry out of digit 63 is 0 for scalars < 2^255 - 2^251 (all canonical scalars are < 2^253)
}
// signed value of digit i (0..63)
EG_HD int sc_digit16(const u32 d[8], int i) {
  u32 w = d[0];
#pragma unroll
  for (int j = 1; j < 8; ++j) w = ((i >> 3) == j) ? d[j] : w;
  const int nib = (int)((w >> (4 * (i & 7))) & 15u);
  return nib >= 8 ? nib - 16 : nib;
}

// ---- variable-base scalar multiplication ------------------------------------------------------------------------
// The per-lane table {1P..8P} (cached form, 160 words per entry) does not fit registers or LDS at useful
// occupancy (1.25 KiB per lane), so it lives in a per-lane slice of a global workspace through the TableIO
// policy: device code uses a coalesced [entry][word4][lane] layout (kernels.hip); host tests use an array.
template <class TableIO>
EG_HD void ge_var_table_build(TableIO& io, const ge& p) {
  // entries k = 1..8 by repeated addition of P (one rolled add body; no indexed point arrays, which
  // hipcc would place in scratch)
  ge_cached pc; ge_to_cached_lazy(pc, p);
  io.store(0, pc);
  ge cur = p;
#pragma unroll 1
  for (int k = 2; k <= 8; ++k) {
    ge_p1p1 t;
    ge_add(t, cur, pc);
    ge_add_to_p3(cur, t);
    ge_cached c; ge_to_cached_lazy(c, cur);
    io.store(k - 1, c);
  }
}

// acc = [k]P with the table already built; digits = sc_recode_radix16(k)
template <class TableIO>
EG_HD void ge_var_mul(ge& acc, TableIO& io, const u32 digits[8]) {
  ge_cached ident; ge_cached_identity(ident);
  // top digit
  {
    const int d = sc_digit16(digits, 63);
    const int ad = d < 0 ? -d : d;
    ge_cached c; io.load(c, ad == 0 ? 0 : ad - 1);
    fe_cmov(c.YpX, ident.YpX, ad == 0); fe_cmov(c.YmX, ident.YmX, ad == 0);
    fe_cmov(c.Z2, ident.Z2, ad == 0); fe_cmov(c.T2d, ident.T2d, ad == 0);
    ge_cached_cneg(c, d < 0);
    ge id; ge_identity(id);
    ge_p1p1 t; ge_add(t, id, c);
    ge_add_to_p3(acc, t);
  }
#pragma unroll 1
  for (int i = 62; i >= 0; --i) {
    const int d = sc_digit16(digits, i);
    const int ad = d < 0 ? -d : d;
    ge_cached c; io.load(c, ad == 0 ? 0 : ad - 1);   // issued before the doublings: latency hidden
    ge_p1p1 t; ge_p2 q;
    q.X = acc.X; q.Y = acc.Y; q.Z = acc.Z;
#pragma unroll 1
    for (int r = 0; r < 3; ++r) {
      ge_dbl(t, q.X, q.Y, q.Z);
      ge_dbl_to_p2(q, t);
    }
    ge_dbl(t, q.X, q.Y, q.Z);
    ge_dbl_to_p3(acc, t);
    fe_cmov(c.YpX, ident.YpX, ad == 0); fe_cmov(c.YmX, ident.YmX, ad == 0);
    fe_cmov(c.Z2, ident.Z2, ad == 0); fe_cmov(c.T2d, ident.T2d, ad == 0);
    ge_cached_cneg(c, d < 0);
    ge_add(t, acc, c);
    if (i > 0) {                      // next operation is a doubling: T is not needed
      ge_p2 q2; ge_add_to_p2(q2, t);
      acc.X = q2.X; acc.Y = q2.Y; acc.Z = q2.Z;
    } else {
      ge_add_to_p3(acc, t);
    }
  }
}

// ---- variable-base multiplication with a per-base signed comb ("teeth" tables) -----------------------------------------
// A ring base (R or B of one ciphertext) is multiplied by one challenge per equation of its ring (ring.rs:333-361), and
// equation j+1 cannot start before equation j is hashed.  The doublings are therefore amortised ACROSS equations with a
// per-base table, arranged as a signed Lim-Lee comb of t teeth x c columns (Teeth<T> below; numbers in brackets: 6 x 43): with
// P_j = [2^(c j)] P, j = 0..t-1, the table holds the 2^(t-1) [32] points P_(t-1) +- .. +- P_1 +- P_0 (entry index = bitmask of the '+'
// signs of the lower teeth), and an odd multiplier k < 2^(t c) is written with t c [258] signed bits s_i = +-1 (k = sum s_i 2^i:
// s_i = 2 bit_(i+1)(k) - 1, top sign +1).  Column i = (s_i, s_(c+i), ..) selects +-entry, so [k]P = sum_i 2^i D_i costs c - 1 [42]
// doublings + c [43] additions (every digit is non-zero: no identity select) instead of 252 + 64 for a fresh ladder; the table costs
// c (t - 1) [215] doublings + 2^(t-1) + t - 1 [37] additions (Gray-code walk, each step adds +-2 P_j), 128-byte cached entries, packed
// (device_io.cuh).  (Round 1 first
// used four radix-16 tables of P, 2^64 P, 2^128 P, 2^192 P: 192 + 28 for the tables but 60 + 64 per product; the comb
// measured +10 % on 2-equation rings and +23 % on the QV ballot.)  Even multipliers use k + l, which changes the
// product by the 4-torsion point [l]P only - invisible to the Ristretto encoding, like the halving in sc_halve.
// The shape is a property of the PLAN (template parameter T = teeth): a table that serves two products (the rings of two of a choice
// ballot) is better off with 5 teeth x 51 columns - 16 entries, 2 KiB, 204 doublings + 21 additions to build, 50 + 51 per product -
// than with 6 x 43; rings of 3 .. 7 members (range proofs) pay the larger table back.  Same-call A/B (r03_ab_experiments.txt, block 9):
// 5 x 51 +2.4 % on single- and multi-choice ballots and -5.1 % on quadratic voting; 7 x 37: -10 % / -8.6 % / +0.5 %.
template <int T> struct Teeth {
  static_assert(T == 5 || T == 6 || T == 7, "comb shapes: 5 x 51, 6 x 43, 7 x 37");
  static constexpr int N = T;
  static constexpr int COLS = T == 5 ? 51 : T == 6 ? 43 : 37;       // N x COLS >= 255 signed bits
  static constexpr int ENTRIES = 1 << (T - 1);
  static constexpr int TOP = T * COLS - 1;                          // position of the sign that is always +1
};

// sg = the N x COLS sign bits of the multiplier (sg bit i <=> s_i = +1): (k_odd >> 1) with the top bit set, k_odd = k or k + l
template <int T>
EG_HD void sc_teeth_signs(u32 sg[9], const u32 s[8]) {
  const u32 l[8] = EG_L_WORDS;
  const bool even = (s[0] & 1u) == 0;
  u32 t[8];
  u64 carry = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const u64 v = (u64)s[i] + (even ? l[i] : 0u) + carry;
    t[i] = (u32)v;
    carry = v >> 32;          // no carry out of word 7: s < 2^254
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) sg[i] = (t[i] >> 1) | (t[i + 1] << 31);
  sg[7] = t[7] >> 1;
  sg[8] = 0u;
  sg[Teeth<T>::TOP >> 5] |= 1u << (Teeth<T>::TOP & 31);     // the top sign (6 x 43: bit 257; k_odd < 2^254 leaves it free in every shape)
}
// rows[j] = the COLS sign bits of tooth j, left-aligned (bit 63 = the highest column)
template <int T>
EG_HD void sc_recode_teeth(u64 rows[T], const u32 s[8]) {
  u32 sg[9];
  sc_teeth_signs<T>(sg, s);
#pragma unroll
  for (int j = 0; j < T; ++j) {
    const int off = Teeth<T>::COLS * j, wi = off >> 5, sh = off & 31;
    u64 v = ((u64)sg[wi] | ((u64)sg[wi + 1] << 32)) >> sh;
    if (sh + Teeth<T>::COLS > 64) v |= (u64)sg[wi + 2] << (64 - sh);
    rows[j] = (v & ((1ull << Teeth<T>::COLS) - 1ull)) << (64 - Teeth<T>::COLS);
  }
}
// column c of a sign vector read word by word through `word(i)` (the multi-term kernel keeps the vectors in LDS): table entry
// index and whether the entry is negated.  Bit positions are the same in every lane, so the word index is wave-uniform.
template <int T, class WordFn>
EG_HD void sc_teeth_column(WordFn word, int c, int& idx, bool& neg) {
  u32 m = 0;
#pragma unroll
  for (int j = 0; j < T - 1; ++j) {
    const int q = Teeth<T>::COLS * j + c;
    m |= ((word(q >> 5) >> (q & 31)) & 1u) << j;
  }
  const int qt = Teeth<T>::COLS * (T - 1) + c;
  const bool top = ((word(qt >> 5) >> (qt & 31)) & 1u) != 0;
  idx = (int)(top ? m : (m ^ (Teeth<T>::ENTRIES - 1)));
  neg = !top;
}
// pops the next column (from the highest downwards): table entry index and whether the entry is negated
template <int T>
EG_HD void sc_teeth_next(u64 rows[T], int& idx, bool& neg) {
  u32 m = 0;
#pragma unroll
  for (int j = 0; j < T - 1; ++j) m |= (u32)(rows[j] >> 63) << j;
  const bool top = (rows[T - 1] >> 63) != 0;
#pragma unroll
  for (int j = 0; j < T; ++j) rows[j] <<= 1;
  idx = (int)(top ? m : (m ^ (Teeth<T>::ENTRIES - 1)));
  neg = !top;
}

// io: the table of this base (2^(T-1) entries); tmp: scratch for the T - 1 cached points 2 P_j
template <int T, class TableIO, class TmpIO>
EG_HD void ge_teeth_tables_build(TableIO& io, TmpIO& tmp, const ge& p) {
  ge cur = p, sum;
  ge_identity(sum);
#pragma unroll 1
  for (int j = 0; j < T - 1; ++j) {
    ge_p1p1 t;
    {
      ge_cached pc; ge_to_cached_lazy(pc, cur);
      ge_cached_cneg(pc, true);
      ge_add(t, sum, pc);                 // sum -= P_j
      ge_add_to_p3(sum, t);
    }
    ge q3;
    ge_dbl(t, cur.X, cur.Y, cur.Z);
    ge_dbl_to_p3(q3, t);                  // 2 P_j, the step of the Gray-code walk for tooth j
    {
      ge_cached qc; ge_to_cached_lazy(qc, q3);
      tmp.store(j, qc);
    }
    ge_p2 q;
    q.X = q3.X; q.Y = q3.Y; q.Z = q3.Z;
#pragma unroll 1
    for (int r = 0; r < Teeth<T>::COLS - 2; ++r) { ge_dbl(t, q.X, q.Y, q.Z); ge_dbl_to_p2(q, t); }
    ge_dbl(t, q.X, q.Y, q.Z);
    ge_dbl_to_p3(cur, t);                 // P_(j+1)
  }
  {
    ge_cached pc; ge_to_cached_lazy(pc, cur);
    ge_p1p1 t; ge_add(t, sum, pc);        // sum = P_5 - P_4 - .. - P_0 = entry 0
    ge_add_to_p3(sum, t);
    ge_cached e; ge_to_cached_lazy(e, sum);
    io.store(0, e);
  }
#pragma unroll 1
  for (int i = 1; i < Teeth<T>::ENTRIES; ++i) {
    int j = 0;
    while (((i >> j) & 1) == 0) ++j;      // Gray code: step i flips tooth ctz(i)
    const int g = i ^ (i >> 1);
    ge_cached qc; tmp.load(qc, j);
    ge_cached_cneg(qc, ((g >> j) & 1) == 0);
    ge_p1p1 t; ge_add(t, sum, qc);
    ge_add_to_p3(sum, t);
    ge_cached e; ge_to_cached_lazy(e, sum);
    io.store(g, e);
  }
}

// (Y+X, Y-X, 2Z, 2dT) -> the same point as (2X : 2Y : 2Z : 2T); accepts the lazily stored classes of table entries.
// Result {X [1], Y [1], Z [2], T [1]}.
EG_HD void ge_cached_to_p3(ge& p, const ge_cached& c) {
  const fe dinv = EG_FE_DINV;
  fe_sub4(p.X, c.YpX, c.YmX); fe_carry(p.X);
  fe_add(p.Y, c.YpX, c.YmX); fe_carry(p.Y);
  p.Z = c.Z2;
  fe_mul(p.T, c.T2d, dinv);
}

// Comb table of S = B_1 + .. + B_m when every B_i already has its table: the table is LINEAR in the base (entry g of S is the
// sum of the members' entries g), so no doubling is needed at all.  Only the T entries where the Gray-code walk of
// ge_teeth_tables_build flips a tooth for the first time (steps 0, 1, 2, 4, ..) are summed over the members; their differences
// to the walk's previous entry are the steps +-2 P_j of S, and the other entries follow with one addition each (6 teeth:
// 6 (m - 1) + 5 + 26 additions against 215 doublings + 37 additions for a table built from S itself, or 252 doublings + 71
// additions for a ladder over S).  Used for the log-equality proof over the sum of a ballot's ciphertexts (choice.rs:363,
// log_equality.rs:160-164), whose two bases are the sums of the ring bases.
// src(k, g, entry) loads entry g of member k; io: the table of S; tmp: scratch for the T - 1 cached steps.
// (Requesting a member's entry one addition ahead of its use, with the walk re-reading its own output, measured slower:
// the kernel is bound by the scattered two-line reads themselves, not by their latency.)
template <int T, class SrcFn, class TableIO, class TmpIO>
EG_HD void ge_teeth_tables_sum(TableIO& io, TmpIO& tmp, int m, SrcFn src) {
  ge sum;
  ge_identity(sum);
#pragma unroll 1
  for (int i = 0; i < Teeth<T>::ENTRIES; ++i) {
    const int g = i ^ (i >> 1);
    int j = 0;
    while (i != 0 && ((i >> j) & 1) == 0) ++j;      // Gray code: step i flips tooth ctz(i)
    ge_p1p1 t;
    if ((i & (i - 1)) == 0) {                       // first flip of tooth j (or the start): entry g summed over the members
      ge acc;
      {
        ge_cached c; src(0, g, c);
        ge_cached_to_p3(acc, c);
      }
#pragma unroll 1
      for (int k = 1; k < m; ++k) {
        ge_cached c; src(k, g, c);
        ge_add(t, acc, c);
        ge_add_to_p3(acc, t);
      }
      if (i != 0) {                                 // step of tooth j: 2 P_j(S) = entry g - previous entry
        ge_cached pc; ge_to_cached_lazy(pc, sum);
        ge_cached_cneg(pc, true);
        ge d;
        ge_add(t, acc, pc);
        ge_add_to_p3(d, t);
        ge_cached dc; ge_to_cached_lazy(dc, d);
        tmp.store(j, dc);
      }
      sum = acc;
      fe_carry(sum.Z);                              // [2] after ge_cached_to_p3 when m == 1
    } else {
      ge_cached qc; tmp.load(qc, j);
      ge_cached_cneg(qc, ((g >> j) & 1) == 0);
      ge_add(t, sum, qc);
      ge_add_to_p3(sum, t);
    }
    ge_cached e; ge_to_cached_lazy(e, sum);
    io.store(g, e);
  }
}

// ---- the same sum of tables when the members' tables do NOT exist at the same time (ring-group walk) ----------------
// ge_teeth_tables_sum reads only T entries of every member: the ones where the Gray-code walk flips a tooth for the first time.  They
// can be summed up group by group into T accumulator entries per sum (ge_teeth_sum_accumulate), and once every member has been added
// the table of the sum is made from the accumulator alone (ge_teeth_tables_sum with one pseudo-member that answers from it).
template <int T> EG_HD int teeth_first_flip_entry(int t) { const int i = t == 0 ? 0 : 1 << (t - 1); return i ^ (i >> 1); }   // 0, 1, 3, 6, 12, 24
template <int T> EG_HD int teeth_first_flip_index(int g) { int t = 0; while ((g >> t) != 0) ++t; return t; }                // its inverse
// accumulator entry t (+)= sum over the m members of this group of their entries first_flip(t); first: the sum has no contribution yet
template <int T, class AccIO, class SrcFn>
EG_HD void ge_teeth_sum_accumulate(AccIO& acc_io, int t, bool first, int m, SrcFn src) {
  const int g = teeth_first_flip_entry<T>(t);
  ge acc;
  int k0 = 0;
  {
    ge_cached c;
    if (first) { src(0, g, c); k0 = 1; } else acc_io.load(c, t);
    ge_cached_to_p3(acc, c);
  }
#pragma unroll 1
  for (int k = k0; k < m; ++k) {
    ge_cached c; src(k, g, c);
    ge_p1p1 r; ge_add(r, acc, c);
    ge_add_to_p3(acc, r);
  }
  ge_cached e; ge_to_cached_lazy(e, acc);       // Z may be [2] when nothing was added: 2Z [4], carried by the store
  acc_io.store(t, e);
}

// acc = [k]P from the teeth table; rows = sc_recode_teeth(k) (consumed).  A column's entry is requested before the doubling and
// used after it, which hides the load without a second entry buffer (requesting it a whole column ahead, in a second register
// buffer, measured -0.4 % in round 1 and +-0.2 % = nothing in round 2, when the kernel had the 40 registers to spare).
// The first column is not added to the identity: +-entry = (Y+X, Y-X, 2Z, ..) IS the point (2X : 2Y : 2Z) in projective
// coordinates, and the operation that follows is a doubling, which does not read T (saves one 8-multiplication addition).
template <int T, class TableIO>
EG_HD void ge_teeth_mul(ge& acc, TableIO& io, u64 rows[T]) {
  {
    int idx; bool neg;
    sc_teeth_next<T>(rows, idx, neg);
    ge_cached cur;
    io.load(cur, idx);
    fe t = cur.YpX; fe_cmov(cur.YpX, cur.YmX, neg); fe_cmov(cur.YmX, t, neg);   // -(x, y) = (-x, y)
    fe_sub4(acc.X, cur.YpX, cur.YmX); fe_carry(acc.X);    // 2X
    fe_add(acc.Y, cur.YpX, cur.YmX); fe_carry(acc.Y);     // 2Y
    acc.Z = cur.Z2;                                       // 2Z [2]
  }
#pragma unroll 1
  for (int c = Teeth<T>::COLS - 2; c >= 0; --c) {
    int idx; bool neg;
    sc_teeth_next<T>(rows, idx, neg);
    ge_cached cur;
    io.load(cur, idx);
    ge_p1p1 t;
    ge_dbl(t, acc.X, acc.Y, acc.Z);
    ge_dbl_to_p3(acc, t);
    ge_cached_cneg(cur, neg);
    ge_add(t, acc, cur);
    if (c > 0) {                          // next operation is a doubling: T is not needed (saves one multiplication)
      ge_p2 q; ge_add_to_p2(q, t);
      acc.X = q.X; acc.Y = q.Y; acc.Z = q.Z;
    } else {
      ge_add_to_p3(acc, t);
    }
  }
}

// acc = sum_t [k_t]P_t for bases that all have teeth tables, with ONE chain of c - 1 doublings shared by every term (Straus /
// interleaved evaluation, the structure dalek uses for vartime_multi_mul, ristretto.rs:139-145): per column one doubling and
// one addition per term.  column(t, c, idx, neg) yields term t's entry for column c; load(t, idx, entry) fetches it.
template <int T, class ColumnFn, class LoadFn>
EG_HD void ge_teeth_mul_multi(ge& acc, int n_terms, ColumnFn column, LoadFn load) {
  ge_identity(acc);
#pragma unroll 1
  for (int c = Teeth<T>::COLS - 1; c >= 0; --c) {
    ge_p1p1 t;
    if (c != Teeth<T>::COLS - 1) {
      ge_dbl(t, acc.X, acc.Y, acc.Z);
      ge_dbl_to_p3(acc, t);
    }
#pragma unroll 1
    for (int k = 0; k < n_terms; ++k) {
      int idx; bool neg;
      column(k, c, idx, neg);
      ge_cached cur;
      load(k, idx, cur);
      ge_cached_cneg(cur, neg);
      ge_add(t, acc, cur);
      if (k + 1 == n_terms && c > 0) {    // a doubling follows: T is not needed
        ge_p2 q; ge_add_to_p2(q, t);
        acc.X = q.X; acc.Y = q.Y; acc.Z = q.Z;
      } else {
        ge_add_to_p3(acc, t);
      }
    }
  }
}

// ---- fixed-base scalar multiplication -----------------------------------------------------------------------------
// Signed radix-2^B comb: ceil(254 / B) windows x 2^(B-1) affine-Niels entries of 128 B per base, built on the device
// (k_build_fixed_table).  B is a property of the TABLE (the table I/O policy carries it), so tables of different widths coexist:
//   EG_COMB_BITS (20)      13 windows, 832 MiB per base: built when the parameters are created;
//   EG_COMB_BITS_BIG (24)  11 windows, 11 GiB per base: built the first time a large batch arrives (engine, eg_hip.hip) - HBM
//                          traded for two additions per comb (measured +1.0 % single-choice, +1.9 % quadratic voting).
// Entries are read from HBM one addition ahead of their use.  Measured in one call against 15 bits (17 windows, 34 MiB,
// Infinity-Cache resident): 20 bits +1.5 % single-choice, +2.8 % quadratic voting (round 1: 8 -> 13 bits +3 %, 13 -> 15 +0.9 %).
// Table index = window * 2^(B-1) + (|digit| - 1).  acc += [k]Base with one mixed addition (7M) per window and no doublings.
#ifndef EG_COMB_BITS
#define EG_COMB_BITS 20
#endif
#ifndef EG_COMB_BITS_BIG
#define EG_COMB_BITS_BIG 24
#endif
#define EG_COMB_WORDS 8
EG_HD int comb_windows(int bits) { return (254 + bits - 1) / bits; }      // scalars (also halved ones) are < 2^254
EG_HD int comb_entries(int bits) { return 1 << (bits - 1); }
// The signed digits are cut from the scalar on the fly, lowest window first (a comb has no doublings, so its windows can be
// summed in any order): no digit array, only a running carry.  sc_recode_comb is kept as the (now trivial) hand-over of the scalar.
EG_HD void sc_recode_comb(u32 out[EG_COMB_WORDS], const u32 k[8]) {
#pragma unroll
  for (int w = 0; w < 8; ++w) out[w] = k[w];
}
// signed digit in [-2^(B-1), 2^(B-1)) of window i of a scalar < 2^254, given the carry of window i - 1
EG_HD int sc_comb_digit(const u32 k[8], int i, u32& carry, int bits) {
  const int off = i * bits, wi = off >> 5, sh = off & 31;
  u32 lo = 0, hi = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) { lo = (wi == j) ? k[j] : lo; hi = (wi + 1 == j) ? k[j] : hi; }
  const u64 v = (u64)lo | ((u64)hi << 32);
  const u32 raw = ((u32)(v >> sh) & ((1u << bits) - 1u)) + carry;      // 0 .. 2^B
  carry = (raw + (1u << (bits - 1))) >> bits;
  return (int)raw - (int)(carry << bits);
}
EG_HD int ge_fixed_index(int i, int d, int bits) {
  const int ad = d < 0 ? -d : d;
  return i * comb_entries(bits) + (ad == 0 ? 0 : ad - 1);
}
// io.bits = window width of the table behind io
template <class NielsIO>
EG_HD void ge_fixed_mul_add(ge& acc, NielsIO& io, const u32 k[EG_COMB_WORDS]) {
  ge_niels ident; ge_niels_identity(ident);
  ge_niels nxt;
  const int bits = io.bits, windows = comb_windows(bits);
  u32 carry = 0;
  int d_nxt = sc_comb_digit(k, 0, carry, bits);
  io.load(nxt, ge_fixed_index(0, d_nxt, bits));
#pragma unroll 1
  for (int i = 0; i < windows; ++i) {
    ge_niels c = nxt;
    const int d = d_nxt;
    if (i + 1 < windows) {                              // one addition ahead of its use
      d_nxt = sc_comb_digit(k, i + 1, carry, bits);
      io.load(nxt, ge_fixed_index(i + 1, d_nxt, bits));
    }
    fe_cmov(c.ypx, ident.ypx, d == 0); fe_cmov(c.ymx, ident.ymx, d == 0); fe_cmov(c.xy2d, ident.xy2d, d == 0);
    ge_niels_cneg(c, d < 0);
    ge_p1p1 t; ge_madd(t, acc, c);
    ge_add_to_p3(acc, t);
  }
}

// affine niels entry from a projective point (one inversion; table construction only)
EG_HD void ge_to_niels(ge_niels& n, const ge& p) {
  const fe d2 = EG_FE_2D;
  fe zi, x, y;
  fe_invert(zi, p.Z);
  fe_mul(x, p.X, zi);
  fe_mul(y, p.Y, zi);
  fe_add(n.ypx, y, x); fe_carry(n.ypx);
  fe_sub(n.ymx, y, x); fe_carry(n.ymx);
  fe_mul(n.xy2d, x, y);
  fe_mul(n.xy2d, n.xy2d, d2);
}

}  // namespace eg

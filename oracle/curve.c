/*
 * curve.c -- CPU ORACLE (test infrastructure, not the product): GF(2^255-19), scalars mod l,
 * extended twisted Edwards arithmetic and the ristretto255 codec.
 *
 * Restates what the reference's Ristretto backend delegates to curve25519-dalek =5.0.0-rc.0
 * (reference call sites: src/group/ristretto.rs:23-146).  Algorithms: RFC 9496 (ristretto255
 * decode/encode/SQRT_RATIO_M1), RFC 8032 5.1 (curve constants), hwcd-2008 extended coordinates.
 * Pinned by the reference's golden snapshots through tests/test_oracle_golden.py.
 */
#include "eg_oracle.h"

#include <string.h>

typedef unsigned __int128 u128;
#define M51 0x7ffffffffffffULL

/* ------------------------------------------------------------------ field */

static void fe_0(fe *h) { memset(h, 0, sizeof *h); }
static void fe_1(fe *h) { fe_0(h); h->v[0] = 1; }

static void fe_carry(fe *h) {
  uint64_t c;
  c = h->v[0] >> 51; h->v[0] &= M51; h->v[1] += c;
  c = h->v[1] >> 51; h->v[1] &= M51; h->v[2] += c;
  c = h->v[2] >> 51; h->v[2] &= M51; h->v[3] += c;
  c = h->v[3] >> 51; h->v[3] &= M51; h->v[4] += c;
  c = h->v[4] >> 51; h->v[4] &= M51; h->v[0] += 19 * c;
  c = h->v[0] >> 51; h->v[0] &= M51; h->v[1] += c;
}

void or_fe_add(fe *h, const fe *f, const fe *g) {
  for (int i = 0; i < 5; i++) h->v[i] = f->v[i] + g->v[i];
  fe_carry(h);
}

void or_fe_sub(fe *h, const fe *f, const fe *g) {
  /* f + 4p - g, limbs of g are < 2^52 after a carry pass */
  h->v[0] = f->v[0] + 0x1fffffffffffb4ULL - g->v[0];
  for (int i = 1; i < 5; i++) h->v[i] = f->v[i] + 0x1ffffffffffffcULL - g->v[i];
  fe_carry(h);
}

static void fe_neg(fe *h, const fe *f) {
  fe z; fe_0(&z);
  or_fe_sub(h, &z, f);
}

void or_fe_mul(fe *h, const fe *f, const fe *g) {
  uint64_t f0 = f->v[0], f1 = f->v[1], f2 = f->v[2], f3 = f->v[3], f4 = f->v[4];
  uint64_t g0 = g->v[0], g1 = g->v[1], g2 = g->v[2], g3 = g->v[3], g4 = g->v[4];
  uint64_t g1_19 = 19 * g1, g2_19 = 19 * g2, g3_19 = 19 * g3, g4_19 = 19 * g4;
  u128 r0 = (u128)f0 * g0 + (u128)f1 * g4_19 + (u128)f2 * g3_19 + (u128)f3 * g2_19 + (u128)f4 * g1_19;
  u128 r1 = (u128)f0 * g1 + (u128)f1 * g0 + (u128)f2 * g4_19 + (u128)f3 * g3_19 + (u128)f4 * g2_19;
  u128 r2 = (u128)f0 * g2 + (u128)f1 * g1 + (u128)f2 * g0 + (u128)f3 * g4_19 + (u128)f4 * g3_19;
  u128 r3 = (u128)f0 * g3 + (u128)f1 * g2 + (u128)f2 * g1 + (u128)f3 * g0 + (u128)f4 * g4_19;
  u128 r4 = (u128)f0 * g4 + (u128)f1 * g3 + (u128)f2 * g2 + (u128)f3 * g1 + (u128)f4 * g0;
  uint64_t c;
  uint64_t h0, h1, h2, h3, h4;
  h0 = (uint64_t)r0 & M51; r1 += (uint64_t)(r0 >> 51);
  h1 = (uint64_t)r1 & M51; r2 += (uint64_t)(r1 >> 51);
  h2 = (uint64_t)r2 & M51; r3 += (uint64_t)(r2 >> 51);
  h3 = (uint64_t)r3 & M51; r4 += (uint64_t)(r3 >> 51);
  h4 = (uint64_t)r4 & M51; c = (uint64_t)(r4 >> 51);
  h0 += 19 * c;
  c = h0 >> 51; h0 &= M51; h1 += c;
  h->v[0] = h0; h->v[1] = h1; h->v[2] = h2; h->v[3] = h3; h->v[4] = h4;
}

void or_fe_sq(fe *h, const fe *f) { or_fe_mul(h, f, f); }

static void fe_sqn(fe *h, const fe *f, int n) {
  or_fe_sq(h, f);
  for (int i = 1; i < n; i++) or_fe_sq(h, h);
}

void or_fe_frombytes(fe *h, const uint8_t s[32]) {
  uint64_t w[4];
  for (int i = 0; i < 4; i++) {
    w[i] = 0;
    for (int j = 0; j < 8; j++) w[i] |= (uint64_t)s[8 * i + j] << (8 * j);
  }
  h->v[0] = w[0] & M51;
  h->v[1] = ((w[0] >> 51) | (w[1] << 13)) & M51;
  h->v[2] = ((w[1] >> 38) | (w[2] << 26)) & M51;
  h->v[3] = ((w[2] >> 25) | (w[3] << 39)) & M51;
  h->v[4] = (w[3] >> 12) & M51; /* drops bit 255 */
}

void or_fe_tobytes(uint8_t s[32], const fe *f) {
  fe t = *f;
  fe_carry(&t);
  fe_carry(&t);
  /* t < 2^255 + small: subtract p iff t >= p */
  uint64_t q = (t.v[0] + 19) >> 51;
  q = (t.v[1] + q) >> 51;
  q = (t.v[2] + q) >> 51;
  q = (t.v[3] + q) >> 51;
  q = (t.v[4] + q) >> 51;
  t.v[0] += 19 * q;
  uint64_t c;
  c = t.v[0] >> 51; t.v[0] &= M51; t.v[1] += c;
  c = t.v[1] >> 51; t.v[1] &= M51; t.v[2] += c;
  c = t.v[2] >> 51; t.v[2] &= M51; t.v[3] += c;
  c = t.v[3] >> 51; t.v[3] &= M51; t.v[4] += c;
  t.v[4] &= M51;
  uint64_t w[4];
  w[0] = t.v[0] | (t.v[1] << 51);
  w[1] = (t.v[1] >> 13) | (t.v[2] << 38);
  w[2] = (t.v[2] >> 26) | (t.v[3] << 25);
  w[3] = (t.v[3] >> 39) | (t.v[4] << 12);
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 8; j++) s[8 * i + j] = (uint8_t)(w[i] >> (8 * j));
}

static int fe_isnegative(const fe *f) {
  uint8_t s[32];
  or_fe_tobytes(s, f);
  return s[0] & 1;
}

static int fe_iszero(const fe *f) {
  uint8_t s[32];
  or_fe_tobytes(s, f);
  uint8_t r = 0;
  for (int i = 0; i < 32; i++) r |= s[i];
  return r == 0;
}

static int fe_eq(const fe *f, const fe *g) {
  uint8_t a[32], b[32];
  or_fe_tobytes(a, f);
  or_fe_tobytes(b, g);
  return memcmp(a, b, 32) == 0;
}

/* z^(2^252 - 3) = z^((p-5)/8) */
static void fe_pow22523(fe *out, const fe *z) {
  fe t0, t1, t2;
  or_fe_sq(&t0, z);
  fe_sqn(&t1, &t0, 2);
  or_fe_mul(&t1, z, &t1);
  or_fe_mul(&t0, &t0, &t1);
  or_fe_sq(&t0, &t0);
  or_fe_mul(&t0, &t1, &t0);
  fe_sqn(&t1, &t0, 5);
  or_fe_mul(&t0, &t1, &t0);
  fe_sqn(&t1, &t0, 10);
  or_fe_mul(&t1, &t1, &t0);
  fe_sqn(&t2, &t1, 20);
  or_fe_mul(&t1, &t2, &t1);
  fe_sqn(&t1, &t1, 10);
  or_fe_mul(&t0, &t1, &t0);
  fe_sqn(&t1, &t0, 50);
  or_fe_mul(&t1, &t1, &t0);
  fe_sqn(&t2, &t1, 100);
  or_fe_mul(&t1, &t2, &t1);
  fe_sqn(&t1, &t1, 50);
  or_fe_mul(&t0, &t1, &t0);
  fe_sqn(&t0, &t0, 2);
  or_fe_mul(out, &t0, z);
}

/* z^(p-2) */
void or_fe_invert(fe *out, const fe *z) {
  fe t0, t1, t2, t3;
  or_fe_sq(&t0, z);
  fe_sqn(&t1, &t0, 2);
  or_fe_mul(&t1, z, &t1);
  or_fe_mul(&t0, &t0, &t1);
  or_fe_sq(&t2, &t0);
  or_fe_mul(&t1, &t1, &t2);
  fe_sqn(&t2, &t1, 5);
  or_fe_mul(&t1, &t2, &t1);
  fe_sqn(&t2, &t1, 10);
  or_fe_mul(&t2, &t2, &t1);
  fe_sqn(&t3, &t2, 20);
  or_fe_mul(&t2, &t3, &t2);
  fe_sqn(&t2, &t2, 10);
  or_fe_mul(&t1, &t2, &t1);
  fe_sqn(&t2, &t1, 50);
  or_fe_mul(&t2, &t2, &t1);
  fe_sqn(&t3, &t2, 100);
  or_fe_mul(&t2, &t3, &t2);
  fe_sqn(&t2, &t2, 50);
  or_fe_mul(&t1, &t2, &t1);
  fe_sqn(&t1, &t1, 5);
  or_fe_mul(out, &t1, &t0);
}

/* ------------------------------------------------------------------ constants */

static int g_init_done = 0;
static fe C_D, C_2D, C_SQRTM1, C_INVSQRT_A_MINUS_D, C_ONE, C_ZERO;
static ge C_BASE;

#define FIXED_WINDOWS 64
typedef struct { ge t[FIXED_WINDOWS][8]; } fixed_table; /* t[i][j] = (j+1) * 16^i * P */
static fixed_table g_base_table;
static void fixed_table_init(fixed_table *ft, const ge *p);
static void fixed_table_mul(ge *r, const fixed_table *ft, const sc *k);

static void fe_from_u64(fe *h, uint64_t x) {
  fe_0(h);
  h->v[0] = x & M51;
  h->v[1] = x >> 51;
}

int or_fe_sqrt_ratio_m1(fe *out, const fe *u, const fe *v) {
  fe v3, v7, r, check, t, neg_u, neg_u_i;
  or_fe_sq(&v3, v);
  or_fe_mul(&v3, &v3, v);
  or_fe_sq(&v7, &v3);
  or_fe_mul(&v7, &v7, v);
  or_fe_mul(&t, u, &v7);
  fe_pow22523(&t, &t);
  or_fe_mul(&r, u, &v3);
  or_fe_mul(&r, &r, &t);
  or_fe_sq(&check, &r);
  or_fe_mul(&check, &check, v);
  fe_neg(&neg_u, u);
  or_fe_mul(&neg_u_i, &neg_u, &C_SQRTM1);
  int correct = fe_eq(&check, u);
  int flipped = fe_eq(&check, &neg_u);
  int flipped_i = fe_eq(&check, &neg_u_i);
  if (flipped | flipped_i) or_fe_mul(&r, &r, &C_SQRTM1);
  if (fe_isnegative(&r)) fe_neg(&r, &r);
  *out = r;
  return correct | flipped;
}

void or_init(void) {
  if (g_init_done) return;
  fe_0(&C_ZERO);
  fe_1(&C_ONE);
  /* sqrt(-1) = 2^((p-1)/4);  (p-1)/4 = 2^253 - 5 = 2*(2^252-3) + 1 */
  fe two, t;
  fe_from_u64(&two, 2);
  fe_pow22523(&t, &two);
  or_fe_sq(&t, &t);
  or_fe_mul(&C_SQRTM1, &t, &two);
  /* d = -121665/121666 */
  fe a, b;
  fe_from_u64(&a, 121665);
  fe_from_u64(&b, 121666);
  or_fe_invert(&b, &b);
  or_fe_mul(&C_D, &a, &b);
  fe_neg(&C_D, &C_D);
  or_fe_add(&C_2D, &C_D, &C_D);
  /* 1/sqrt(a-d) with a = -1 */
  fe amd;
  fe_neg(&amd, &C_ONE);
  or_fe_sub(&amd, &amd, &C_D);
  g_init_done = 1; /* sqrt_ratio below needs C_SQRTM1 only */
  or_fe_sqrt_ratio_m1(&C_INVSQRT_A_MINUS_D, &C_ONE, &amd);
  /* base point: y = 4/5, x even (RFC 8032) */
  fe y, y2, num, den, x, five;
  fe_from_u64(&y, 4);
  fe_from_u64(&five, 5);
  or_fe_invert(&five, &five);
  or_fe_mul(&y, &y, &five);
  or_fe_sq(&y2, &y);
  or_fe_sub(&num, &y2, &C_ONE);   /* y^2 - 1 */
  or_fe_mul(&den, &C_D, &y2);
  or_fe_add(&den, &den, &C_ONE);  /* d y^2 + 1 */
  or_fe_sqrt_ratio_m1(&x, &num, &den); /* non-negative root = even x */
  C_BASE.X = x;
  C_BASE.Y = y;
  C_BASE.Z = C_ONE;
  or_fe_mul(&C_BASE.T, &x, &y);
  fixed_table_init(&g_base_table, &C_BASE);
}

static const uint8_t L_BYTES[32] = {0xed, 0xd3, 0xf5, 0x5c, 0x1a, 0x63, 0x12, 0x58, 0xd6, 0x9c, 0xf7,
                                    0xa2, 0xde, 0xf9, 0xde, 0x14, 0,    0,    0,    0,    0,    0,
                                    0,    0,    0,    0,    0,    0,    0,    0,    0,    0x10};

void or_const_bytes(int which, uint8_t out[32]) {
  or_init();
  switch (which) {
    case 0: or_fe_tobytes(out, &C_D); break;
    case 1: or_fe_tobytes(out, &C_SQRTM1); break;
    case 2: or_fe_tobytes(out, &C_INVSQRT_A_MINUS_D); break;
    case 3: memcpy(out, L_BYTES, 32); break;
    default: or_ristretto_encode(out, &C_BASE); break;
  }
}

/* ------------------------------------------------------------------ scalars mod l */

static const int64_t L64[32] = {0xed, 0xd3, 0xf5, 0x5c, 0x1a, 0x63, 0x12, 0x58, 0xd6, 0x9c, 0xf7,
                                0xa2, 0xde, 0xf9, 0xde, 0x14, 0,    0,    0,    0,    0,    0,
                                0,    0,    0,    0,    0,    0,    0,    0,    0,    0x10};

/* byte-radix reduction mod l (the TweetNaCl modL algorithm, public domain) */
static void modL(uint8_t *r, int64_t x[64]) {
  int64_t carry;
  int i, j;
  for (i = 63; i >= 32; --i) {
    carry = 0;
    for (j = i - 32; j < i - 12; ++j) {
      x[j] += carry - 16 * x[i] * L64[j - (i - 32)];
      carry = (x[j] + 128) >> 8;
      x[j] -= carry * 256;
    }
    x[j] += carry;
    x[i] = 0;
  }
  carry = 0;
  for (j = 0; j < 32; ++j) {
    x[j] += carry - (x[31] >> 4) * L64[j];
    carry = x[j] >> 8;
    x[j] &= 255;
  }
  for (j = 0; j < 32; ++j) x[j] -= carry * L64[j];
  for (i = 0; i < 32; ++i) {
    x[i + 1] += x[i] >> 8;
    r[i] = (uint8_t)(x[i] & 255);
  }
}

void or_sc_from_wide(sc *r, const uint8_t wide[64]) {
  int64_t x[64];
  for (int i = 0; i < 64; i++) x[i] = wide[i];
  modL(r->b, x);
}

int or_sc_is_canonical(const uint8_t s[32]) {
  for (int i = 31; i >= 0; i--) {
    if (s[i] < L_BYTES[i]) return 1;
    if (s[i] > L_BYTES[i]) return 0;
  }
  return 0; /* == l */
}

void or_sc_from_u64(sc *r, uint64_t x) {
  memset(r->b, 0, 32);
  for (int i = 0; i < 8; i++) r->b[i] = (uint8_t)(x >> (8 * i));
}

void or_sc_add(sc *r, const sc *a, const sc *b) {
  int64_t x[64] = {0};
  for (int i = 0; i < 32; i++) x[i] = (int64_t)a->b[i] + b->b[i];
  modL(r->b, x);
}

void or_sc_sub(sc *r, const sc *a, const sc *b) {
  int64_t x[64] = {0};
  for (int i = 0; i < 32; i++) x[i] = (int64_t)a->b[i] + L64[i] - b->b[i];
  modL(r->b, x);
}

void or_sc_neg(sc *r, const sc *a) {
  int64_t x[64] = {0};
  for (int i = 0; i < 32; i++) x[i] = L64[i] - a->b[i];
  modL(r->b, x);
}

void or_sc_muladd(sc *r, const sc *a, const sc *b, const sc *c) {
  int64_t x[64] = {0};
  for (int i = 0; i < 32; i++) x[i] = c->b[i];
  for (int i = 0; i < 32; i++)
    for (int j = 0; j < 32; j++) x[i + j] += (int64_t)a->b[i] * b->b[j];
  modL(r->b, x);
}

void or_sc_mul(sc *r, const sc *a, const sc *b) {
  sc z;
  memset(&z, 0, sizeof z);
  or_sc_muladd(r, a, b, &z);
}

void or_sc_invert(sc *r, const sc *a) {
  /* a^(l-2) by square-and-multiply */
  uint8_t e[32];
  memcpy(e, L_BYTES, 32);
  e[0] -= 2; /* 0xed - 2, no borrow */
  sc acc;
  or_sc_from_u64(&acc, 1);
  for (int i = 255; i >= 0; i--) {
    or_sc_mul(&acc, &acc, &acc);
    if ((e[i >> 3] >> (i & 7)) & 1) or_sc_mul(&acc, &acc, a);
  }
  *r = acc;
}

int or_sc_eq(const sc *a, const sc *b) { return memcmp(a->b, b->b, 32) == 0; }

/* ------------------------------------------------------------------ group */

void or_ge_identity(ge *p) {
  or_init();
  p->X = C_ZERO; p->Y = C_ONE; p->Z = C_ONE; p->T = C_ZERO;
}

void or_ge_generator(ge *p) {
  or_init();
  *p = C_BASE;
}

void or_ge_add(ge *r, const ge *p, const ge *q) {
  or_init();
  fe a, b, c, d, e, f, g, h, t0, t1;
  or_fe_sub(&t0, &p->Y, &p->X);
  or_fe_sub(&t1, &q->Y, &q->X);
  or_fe_mul(&a, &t0, &t1);
  or_fe_add(&t0, &p->Y, &p->X);
  or_fe_add(&t1, &q->Y, &q->X);
  or_fe_mul(&b, &t0, &t1);
  or_fe_mul(&c, &p->T, &q->T);
  or_fe_mul(&c, &c, &C_2D);
  or_fe_mul(&d, &p->Z, &q->Z);
  or_fe_add(&d, &d, &d);
  or_fe_sub(&e, &b, &a);
  or_fe_sub(&f, &d, &c);
  or_fe_add(&g, &d, &c);
  or_fe_add(&h, &b, &a);
  or_fe_mul(&r->X, &e, &f);
  or_fe_mul(&r->Y, &g, &h);
  or_fe_mul(&r->Z, &f, &g);
  or_fe_mul(&r->T, &e, &h);
}

void or_ge_neg(ge *r, const ge *p) {
  ge t = *p;
  fe_neg(&t.X, &p->X);
  fe_neg(&t.T, &p->T);
  *r = t;
}

void or_ge_sub(ge *r, const ge *p, const ge *q) {
  ge n;
  or_ge_neg(&n, q);
  or_ge_add(r, p, &n);
}

void or_ge_double(ge *r, const ge *p) {
  fe a, b, c, d, e, f, g, h, t0;
  or_fe_sq(&a, &p->X);
  or_fe_sq(&b, &p->Y);
  or_fe_sq(&c, &p->Z);
  or_fe_add(&c, &c, &c);
  fe_neg(&d, &a);
  or_fe_add(&t0, &p->X, &p->Y);
  or_fe_sq(&e, &t0);
  or_fe_sub(&e, &e, &a);
  or_fe_sub(&e, &e, &b);
  or_fe_add(&g, &d, &b);
  or_fe_sub(&f, &g, &c);
  or_fe_sub(&h, &d, &b);
  or_fe_mul(&r->X, &e, &f);
  or_fe_mul(&r->Y, &g, &h);
  or_fe_mul(&r->Z, &f, &g);
  or_fe_mul(&r->T, &e, &h);
}

int or_ge_eq(const ge *p, const ge *q) {
  fe a, b;
  or_fe_mul(&a, &p->X, &q->Y);
  or_fe_mul(&b, &p->Y, &q->X);
  if (fe_eq(&a, &b)) return 1;
  or_fe_mul(&a, &p->Y, &q->Y);
  or_fe_mul(&b, &p->X, &q->X);
  return fe_eq(&a, &b);
}

int or_ge_is_identity(const ge *p) {
  ge id;
  or_ge_identity(&id);
  return or_ge_eq(p, &id);
}

int or_ristretto_decode(ge *p, const uint8_t s_bytes[32]) {
  or_init();
  fe s;
  uint8_t chk[32];
  or_fe_frombytes(&s, s_bytes);
  or_fe_tobytes(chk, &s);
  if (memcmp(chk, s_bytes, 32) != 0) return 0; /* non-canonical */
  if (s_bytes[0] & 1) return 0;                /* negative */
  fe ss, u1, u2, u2s, v, t, inv, dx, dy, x, y, tt;
  or_fe_sq(&ss, &s);
  or_fe_sub(&u1, &C_ONE, &ss);
  or_fe_add(&u2, &C_ONE, &ss);
  or_fe_sq(&u2s, &u2);
  or_fe_sq(&t, &u1);
  or_fe_mul(&t, &t, &C_D);
  fe_neg(&t, &t);
  or_fe_sub(&v, &t, &u2s); /* -d*u1^2 - u2^2 */
  or_fe_mul(&t, &v, &u2s);
  int ok = or_fe_sqrt_ratio_m1(&inv, &C_ONE, &t);
  or_fe_mul(&dx, &inv, &u2);
  or_fe_mul(&dy, &inv, &dx);
  or_fe_mul(&dy, &dy, &v);
  or_fe_mul(&x, &s, &dx);
  or_fe_add(&x, &x, &x);
  if (fe_isnegative(&x)) fe_neg(&x, &x);
  or_fe_mul(&y, &u1, &dy);
  or_fe_mul(&tt, &x, &y);
  if (!ok || fe_isnegative(&tt) || fe_iszero(&y)) return 0;
  p->X = x; p->Y = y; p->Z = C_ONE; p->T = tt;
  return 1;
}

void or_ristretto_encode(uint8_t out[32], const ge *p) {
  or_init();
  fe u1, u2, t0, t1, inv, d1, d2, zinv, x, y, dinv, s;
  or_fe_add(&t0, &p->Z, &p->Y);
  or_fe_sub(&t1, &p->Z, &p->Y);
  or_fe_mul(&u1, &t0, &t1);
  or_fe_mul(&u2, &p->X, &p->Y);
  or_fe_sq(&t0, &u2);
  or_fe_mul(&t0, &t0, &u1);
  or_fe_sqrt_ratio_m1(&inv, &C_ONE, &t0);
  or_fe_mul(&d1, &inv, &u1);
  or_fe_mul(&d2, &inv, &u2);
  or_fe_mul(&zinv, &d1, &d2);
  or_fe_mul(&zinv, &zinv, &p->T);
  or_fe_mul(&t0, &p->T, &zinv);
  if (fe_isnegative(&t0)) {
    or_fe_mul(&x, &p->Y, &C_SQRTM1);
    or_fe_mul(&y, &p->X, &C_SQRTM1);
    or_fe_mul(&dinv, &d1, &C_INVSQRT_A_MINUS_D);
  } else {
    x = p->X;
    y = p->Y;
    dinv = d2;
  }
  or_fe_mul(&t0, &x, &zinv);
  if (fe_isnegative(&t0)) fe_neg(&y, &y);
  or_fe_sub(&t0, &p->Z, &y);
  or_fe_mul(&s, &dinv, &t0);
  if (fe_isnegative(&s)) fe_neg(&s, &s);
  or_fe_tobytes(out, &s);
}

/* signed radix-16 recoding: 64 digits in [-8, 8) (top digit <= 8 since scalars are < 2^253) */
static void sc_to_radix16(int8_t e[64], const sc *k) {
  for (int i = 0; i < 32; i++) {
    e[2 * i] = k->b[i] & 15;
    e[2 * i + 1] = (k->b[i] >> 4) & 15;
  }
  int8_t carry = 0;
  for (int i = 0; i < 63; i++) {
    e[i] += carry;
    carry = (int8_t)((e[i] + 8) >> 4);
    e[i] -= (int8_t)(carry * 16);
  }
  e[63] += carry;
}

static void table8(ge t[8], const ge *p) {
  t[0] = *p;
  for (int i = 1; i < 8; i++) {
    if (i & 1) or_ge_double(&t[i], &t[i / 2]);
    else or_ge_add(&t[i], &t[i - 1], p);
  }
}

static void add_digit(ge *acc, const ge t[8], int d) {
  if (d > 0) or_ge_add(acc, acc, &t[d - 1]);
  else if (d < 0) or_ge_sub(acc, acc, &t[-d - 1]);
}

/* Straus interleaved signed 4-bit windows; vartime (skips zero digits) */
static void ge_multi_mul_block(ge *out, size_t n, const sc *k, const ge *p) {   /* n <= 64: one shared doubling chain (Straus) */
  enum { MAXN = 64 };
  ge tabs[MAXN][8];
  int8_t digs[MAXN][64];
  ge acc;
  or_ge_identity(&acc);
  for (size_t j = 0; j < n; j++) {
    table8(tabs[j], &p[j]);
    sc_to_radix16(digs[j], &k[j]);
  }
  for (int i = 63; i >= 0; i--) {
    if (i != 63)
      for (int d = 0; d < 4; d++) or_ge_double(&acc, &acc);
    for (size_t j = 0; j < n; j++) add_digit(&acc, tabs[j], digs[j][i]);
  }
  *out = acc;
}
void or_ge_multi_mul(ge *out, size_t n, const sc *k, const ge *p) {   /* any n: blocks of 64 terms, summed (no recursion: 2^16 terms) */
  ge acc, part;
  or_ge_identity(&acc);
  for (size_t off = 0; off < n || off == 0; off += 64) {
    const size_t m = n - off < 64 ? n - off : 64;
    ge_multi_mul_block(&part, m, k + off, p + off);
    or_ge_add(&acc, &acc, &part);
    if (n == 0) break;
  }
  *out = acc;
}

void or_ge_scalarmult(ge *r, const sc *k, const ge *p) { or_ge_multi_mul(r, 1, k, p); }

static void fixed_table_init(fixed_table *ft, const ge *p) {
  ge base = *p;
  for (int i = 0; i < FIXED_WINDOWS; i++) {
    table8(ft->t[i], &base);
    for (int d = 0; d < 4; d++) or_ge_double(&base, &base);
  }
}

static void fixed_table_mul(ge *r, const fixed_table *ft, const sc *k) {
  int8_t e[64];
  sc_to_radix16(e, k);
  ge acc;
  or_ge_identity(&acc);
  for (int i = 0; i < 64; i++) add_digit(&acc, ft->t[i], e[i]);
  *r = acc;
}

void or_ge_mul_generator(ge *r, const sc *k) {
  or_init();
  fixed_table_mul(r, &g_base_table, k);
}

void or_ge_double_mul_generator(ge *out, const sc *k, const ge *p, const sc *r) {
  ge a, b;
  or_ge_scalarmult(&a, k, p);
  or_ge_mul_generator(&b, r);
  or_ge_add(out, &a, &b);
}

/* fixed-base tables for the election key (prover speed; not on the verify path) */
#include <stdlib.h>
void *or_fixed_table_new(const ge *p) {
  fixed_table *ft = (fixed_table *)malloc(sizeof(fixed_table));
  fixed_table_init(ft, p);
  return ft;
}
void or_fixed_table_free(void *ft) { free(ft); }
void or_fixed_table_mul(ge *r, const void *ft, const sc *k) { fixed_table_mul(r, (const fixed_table *)ft, k); }

/*
 * transcript.c -- CPU ORACLE (test infrastructure, not the product): Keccak-f[1600], STROBE-128,
 * Merlin transcripts and the ChaCha20 RNG.
 *
 * Restates merlin 3.0.0 / keccak 0.1.6 (reference call sites: src/proofs/mod.rs:39-57,
 * src/group/mod.rs:37-62) and rand_chacha 0.10.0's ChaCha20Rng + rand_core's
 * SeedableRng::seed_from_u64 (reference call site: tests/snapshots.rs:32).  Framing follows the
 * published Merlin/STROBE-128 specification; pinned by the upstream Merlin
 * known-answer test and by the reference's snapshots (tests/test_oracle_golden.py).
 */
#include "eg_oracle.h"

#include <string.h>

/* ------------------------------------------------------------------ Keccak-f[1600] */

static const uint64_t KECCAK_RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int KECCAK_ROT[24] = {1,  3,  6,  10, 15, 21, 28, 36, 45, 55, 2,  14,
                                   27, 41, 56, 8,  25, 43, 62, 18, 39, 61, 20, 44};
static const int KECCAK_PI[24] = {10, 7,  11, 17, 18, 3, 5,  16, 8,  21, 24, 4,
                                  15, 23, 19, 13, 12, 2, 20, 14, 22, 9,  6,  1};

static __thread uint64_t g_keccak_calls = 0;
uint64_t or_keccak_calls(void) { return g_keccak_calls; }

#define ROTL64(x, n) (((x) << (n)) | ((x) >> (64 - (n))))

void or_keccak_f1600(uint64_t st[25]) {
  uint64_t bc[5], t;
  g_keccak_calls++;
  for (int round = 0; round < 24; round++) {
    for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
    for (int i = 0; i < 5; i++) {
      t = bc[(i + 4) % 5] ^ ROTL64(bc[(i + 1) % 5], 1);
      for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
    }
    t = st[1];
    for (int i = 0; i < 24; i++) {
      int j = KECCAK_PI[i];
      bc[0] = st[j];
      st[j] = ROTL64(t, KECCAK_ROT[i]);
      t = bc[0];
    }
    for (int j = 0; j < 25; j += 5) {
      for (int i = 0; i < 5; i++) bc[i] = st[j + i];
      for (int i = 0; i < 5; i++) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
    }
    st[0] ^= KECCAK_RC[round];
  }
}

/* ------------------------------------------------------------------ STROBE-128 (Merlin subset) */

#define STROBE_R 166
#define FLAG_I 1
#define FLAG_A 2
#define FLAG_C 4
#define FLAG_T 8
#define FLAG_M 16
#define FLAG_K 32

static void strobe_run_f(merlin_t *t) {
  t->st[t->pos] ^= t->pos_begin;
  t->st[t->pos + 1] ^= 0x04;
  t->st[STROBE_R + 1] ^= 0x80;
  uint64_t lanes[25];
  for (int i = 0; i < 25; i++) {
    uint64_t w = 0;
    for (int j = 0; j < 8; j++) w |= (uint64_t)t->st[8 * i + j] << (8 * j);
    lanes[i] = w;
  }
  or_keccak_f1600(lanes);
  for (int i = 0; i < 25; i++)
    for (int j = 0; j < 8; j++) t->st[8 * i + j] = (uint8_t)(lanes[i] >> (8 * j));
  t->pos = 0;
  t->pos_begin = 0;
}

static void strobe_absorb(merlin_t *t, const uint8_t *data, size_t len) {
  for (size_t i = 0; i < len; i++) {
    t->st[t->pos] ^= data[i];
    t->pos++;
    if (t->pos == STROBE_R) strobe_run_f(t);
  }
}

static void strobe_squeeze(merlin_t *t, uint8_t *data, size_t len) {
  for (size_t i = 0; i < len; i++) {
    data[i] = t->st[t->pos];
    t->st[t->pos] = 0;
    t->pos++;
    if (t->pos == STROBE_R) strobe_run_f(t);
  }
}

static void strobe_begin_op(merlin_t *t, uint8_t flags, int more) {
  if (more) return; /* continuation of the current operation */
  uint8_t old_begin = t->pos_begin;
  t->pos_begin = (uint8_t)(t->pos + 1);
  t->cur_flags = flags;
  uint8_t hdr[2] = {old_begin, flags};
  strobe_absorb(t, hdr, 2);
  if ((flags & (FLAG_C | FLAG_K)) && t->pos != 0) strobe_run_f(t);
}

static void strobe_meta_ad(merlin_t *t, const uint8_t *d, size_t n, int more) {
  strobe_begin_op(t, FLAG_M | FLAG_A, more);
  strobe_absorb(t, d, n);
}
static void strobe_ad(merlin_t *t, const uint8_t *d, size_t n, int more) {
  strobe_begin_op(t, FLAG_A, more);
  strobe_absorb(t, d, n);
}
static void strobe_prf(merlin_t *t, uint8_t *d, size_t n, int more) {
  strobe_begin_op(t, FLAG_I | FLAG_A | FLAG_C, more);
  strobe_squeeze(t, d, n);
}

static void le32(uint8_t out[4], uint32_t x) {
  for (int i = 0; i < 4; i++) out[i] = (uint8_t)(x >> (8 * i));
}

void or_merlin_append(merlin_t *t, const char *label, const uint8_t *msg, size_t len) {
  uint8_t l4[4];
  le32(l4, (uint32_t)len);
  strobe_meta_ad(t, (const uint8_t *)label, strlen(label), 0);
  strobe_meta_ad(t, l4, 4, 1);
  strobe_ad(t, msg, len, 0);
}

void or_merlin_init(merlin_t *t, const char *label) {
  memset(t, 0, sizeof *t);
  static const uint8_t head[6] = {1, STROBE_R + 2, 1, 0, 1, 96};
  memcpy(t->st, head, 6);
  memcpy(t->st + 6, "STROBEv1.0.2", 12);
  uint64_t lanes[25];
  for (int i = 0; i < 25; i++) {
    uint64_t w = 0;
    for (int j = 0; j < 8; j++) w |= (uint64_t)t->st[8 * i + j] << (8 * j);
    lanes[i] = w;
  }
  or_keccak_f1600(lanes);
  for (int i = 0; i < 25; i++)
    for (int j = 0; j < 8; j++) t->st[8 * i + j] = (uint8_t)(lanes[i] >> (8 * j));
  strobe_meta_ad(t, (const uint8_t *)"Merlin v1.0", 11, 0);
  or_merlin_append(t, "dom-sep", (const uint8_t *)label, strlen(label));
}

void or_merlin_append_u64(merlin_t *t, const char *label, uint64_t x) {
  uint8_t b[8];
  for (int i = 0; i < 8; i++) b[i] = (uint8_t)(x >> (8 * i));
  or_merlin_append(t, label, b, 8);
}

void or_merlin_challenge(merlin_t *t, const char *label, uint8_t *out, size_t len) {
  uint8_t l4[4];
  le32(l4, (uint32_t)len);
  strobe_meta_ad(t, (const uint8_t *)label, strlen(label), 0);
  strobe_meta_ad(t, l4, 4, 1);
  strobe_prf(t, out, len, 0);
}

/* reference glue: src/proofs/mod.rs:39-57 */
void or_t_start_proof(merlin_t *t, const char *label) {
  or_merlin_append(t, "dom-sep", (const uint8_t *)label, strlen(label));
}
void or_t_append_element(merlin_t *t, const char *label, const ge *p) {
  uint8_t b[32];
  or_ristretto_encode(b, p);
  or_merlin_append(t, label, b, 32);
}
void or_t_challenge_scalar(merlin_t *t, const char *label, sc *out) {
  uint8_t wide[64];
  or_merlin_challenge(t, label, wide, 64);
  or_sc_from_wide(out, wide);
}

/* ------------------------------------------------------------------ ChaCha20 RNG */

#define ROTL32(x, n) (((x) << (n)) | ((x) >> (32 - (n))))
#define QR(a, b, c, d)        \
  a += b; d ^= a; d = ROTL32(d, 16); \
  c += d; b ^= c; b = ROTL32(b, 12); \
  a += b; d ^= a; d = ROTL32(d, 8);  \
  c += d; b ^= c; b = ROTL32(b, 7);

static void chacha20_block(const uint32_t key[8], uint64_t counter, uint8_t out[64]) {
  uint32_t s[16], x[16];
  s[0] = 0x61707865; s[1] = 0x3320646e; s[2] = 0x79622d32; s[3] = 0x6b206574;
  for (int i = 0; i < 8; i++) s[4 + i] = key[i];
  s[12] = (uint32_t)counter;
  s[13] = (uint32_t)(counter >> 32);
  s[14] = 0;
  s[15] = 0;
  memcpy(x, s, sizeof x);
  for (int i = 0; i < 10; i++) {
    QR(x[0], x[4], x[8], x[12]) QR(x[1], x[5], x[9], x[13]) QR(x[2], x[6], x[10], x[14]) QR(x[3], x[7], x[11], x[15])
    QR(x[0], x[5], x[10], x[15]) QR(x[1], x[6], x[11], x[12]) QR(x[2], x[7], x[8], x[13]) QR(x[3], x[4], x[9], x[14])
  }
  for (int i = 0; i < 16; i++) {
    uint32_t v = x[i] + s[i];
    out[4 * i] = (uint8_t)v;
    out[4 * i + 1] = (uint8_t)(v >> 8);
    out[4 * i + 2] = (uint8_t)(v >> 16);
    out[4 * i + 3] = (uint8_t)(v >> 24);
  }
}

void or_rng_from_seed(chacha_rng *r, const uint8_t seed[32]) {
  for (int i = 0; i < 8; i++)
    r->key[i] = (uint32_t)seed[4 * i] | ((uint32_t)seed[4 * i + 1] << 8) |
                ((uint32_t)seed[4 * i + 2] << 16) | ((uint32_t)seed[4 * i + 3] << 24);
  r->counter = 0;
}

void or_rng_seed_from_u64(chacha_rng *r, uint64_t state) {
  /* rand_core SeedableRng::seed_from_u64: PCG32 expands the u64 into the 32-byte seed */
  uint8_t seed[32];
  for (int i = 0; i < 8; i++) {
    state = state * 6364136223846793005ULL + 11634580027462260723ULL;
    uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
    uint32_t rot = (uint32_t)(state >> 59);
    uint32_t x = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
    seed[4 * i] = (uint8_t)x;
    seed[4 * i + 1] = (uint8_t)(x >> 8);
    seed[4 * i + 2] = (uint8_t)(x >> 16);
    seed[4 * i + 3] = (uint8_t)(x >> 24);
  }
  or_rng_from_seed(r, seed);
}

void or_rng_fill64(chacha_rng *r, uint8_t out[64]) {
  chacha20_block(r->key, r->counter, out);
  r->counter++;
}

void or_rng_scalar(chacha_rng *r, sc *out) {
  uint8_t wide[64];
  or_rng_fill64(r, wide);
  or_sc_from_wide(out, wide);
}

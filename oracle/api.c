/*
 * api.c -- CPU ORACLE (test infrastructure, not the product): heap constructors so that the
 * Python ctypes wrapper (oracle/oracle.py) does not have to mirror struct layouts.
 */
#include "eg_oracle.h"

#include <stdlib.h>
#include <string.h>

or_choice_params *or_choice_params_new(const uint8_t pk[32], int n_options, int single) {
  /* ChoiceParams::single / ::multi (choice.rs:160-196) */
  if (n_options < 1 || n_options > 256) return NULL; /* batch generator keeps 256 selection flags */
  or_choice_params *p = (or_choice_params *)calloc(1, sizeof *p);
  if (or_pubkey_from_bytes(&p->pk, pk) != 0) { free(p); return NULL; }
  p->n_options = n_options;
  p->single = single;
  return p;
}

or_qv_params *or_qv_params_new(const uint8_t pk[32], int n_options, uint64_t credits) {
  or_pubkey t;
  if (or_pubkey_from_bytes(&t, pk) != 0) return NULL;
  if (n_options < 1 || n_options > 256) return NULL;
  or_qv_params *p = (or_qv_params *)calloc(1, sizeof *p);
  or_qv_params_init(p, pk, n_options, credits);
  return p;
}

or_prepared_range *or_prepared_range_new(uint64_t upper_bound) {
  or_prepared_range *r = (or_prepared_range *)calloc(1, sizeof *r);
  range_decomp d;
  or_range_optimal(&d, upper_bound);
  or_prepared_range_init(r, &d);
  return r;
}

or_pubkey *or_pubkey_new(const uint8_t pk[32]) {
  or_pubkey *p = (or_pubkey *)calloc(1, sizeof *p);
  if (or_pubkey_from_bytes(p, pk) != 0) { free(p); return NULL; }
  return p;
}

const or_pubkey *or_choice_params_pk(const or_choice_params *p) { return &p->pk; }
const or_pubkey *or_qv_params_pk(const or_qv_params *p) { return &p->pk; }
const or_prepared_range *or_qv_vote_range(const or_qv_params *p) { return &p->vote_range; }
const or_prepared_range *or_qv_credit_range(const or_qv_params *p) { return &p->credit_range; }
int or_prepared_range_name(const or_prepared_range *r, char *buf, size_t cap) {
  size_t n = (size_t)r->name_len < cap - 1 ? (size_t)r->name_len : cap - 1;
  memcpy(buf, r->name, n);
  buf[n] = 0;
  return (int)n;
}
int or_prepared_range_rings(const or_prepared_range *r, uint64_t *sizes, uint64_t *steps) {
  for (int i = 0; i < r->n_rings; i++) { sizes[i] = r->d.size[i]; steps[i] = r->d.step[i]; }
  return r->n_rings;
}
/* admissible value table entry as a compressed point (for building GPU-side tables in tests) */
void or_prepared_range_table(const or_prepared_range *r, int ring, int j, uint8_t out[32]) {
  or_ristretto_encode(out, &r->table[ring][j]);
}

void or_free(void *p) { free(p); }

/* keypair from a u64 seed exactly as tests/snapshots.rs:32-33 */
void or_keypair_from_seed(uint64_t seed, uint8_t sk_out[32], uint8_t pk_out[32], chacha_rng *rng_out) {
  chacha_rng rng;
  or_rng_seed_from_u64(&rng, seed);
  sc sk;
  or_pubkey pk;
  or_keypair_generate(&rng, &sk, &pk);
  memcpy(sk_out, sk.b, 32);
  memcpy(pk_out, pk.bytes, 32);
  free(pk.ktable);
  if (rng_out) *rng_out = rng;
}

/* byte-level wrappers of the Group primitives (for parity tests against the HIP primitive tier) */
int or_point_double_mul_generator(const uint8_t k[32], const uint8_t p[32], const uint8_t r[32], uint8_t out[32]) {
  ge P, Q;
  if (!or_ristretto_decode(&P, p)) return -1;
  or_ge_double_mul_generator(&Q, (const sc *)k, &P, (const sc *)r);
  or_ristretto_encode(out, &Q);
  return 0;
}
int or_point_multi_mul(size_t n, const uint8_t *ks, const uint8_t *ps, uint8_t out[32]) {
  ge *P = (ge *)malloc(sizeof(ge) * (n ? n : 1));
  for (size_t i = 0; i < n; i++)
    if (!or_ristretto_decode(&P[i], ps + 32 * i)) { free(P); return -1; }
  ge Q;
  or_ge_multi_mul(&Q, n, (const sc *)ks, P);
  or_ristretto_encode(out, &Q);
  free(P);
  return 0;
}
void or_point_mul_generator(const uint8_t k[32], uint8_t out[32]) {
  ge Q;
  or_ge_mul_generator(&Q, (const sc *)k);
  or_ristretto_encode(out, &Q);
}
int or_point_add(const uint8_t a[32], const uint8_t b[32], int sub, uint8_t out[32]) {
  ge A, B, C;
  if (!or_ristretto_decode(&A, a) || !or_ristretto_decode(&B, b)) return -1;
  if (sub) or_ge_sub(&C, &A, &B); else or_ge_add(&C, &A, &B);
  or_ristretto_encode(out, &C);
  return 0;
}
int or_point_roundtrip(const uint8_t a[32], uint8_t out[32]) {
  ge A;
  if (!or_ristretto_decode(&A, a)) return -1;
  or_ristretto_encode(out, &A);
  return 0;
}

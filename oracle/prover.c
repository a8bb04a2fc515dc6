/*
 * prover.c -- CPU ORACLE (test infrastructure, not the product): the reference's provers, restated
 * with the exact RNG draw order so that ChaChaRng::seed_from_u64(12345) reproduces the reference's
 * golden snapshots byte-for-byte, plus threaded batch drivers used to
 * synthesise test/benchmark ballots and to time the CPU baseline.
 *
 * Follows: ring.rs:54-195,440-506 (Ring::new / aggregate / finalize, RingProofBuilder),
 * log_equality.rs:114-139, choice.rs:313-349, range.rs:462-534, mul.rs:107-181,
 * quadratic_voting.rs:234-284, encryption.rs:310-327,403-407, keys/impls.rs:16-51,77-91,120-129.
 */
#include "eg_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

void *or_fixed_table_new(const ge *p);
void or_fixed_table_mul(ge *r, const void *ft, const sc *k);

static void mul_k(ge *r, const or_pubkey *pk, const sc *k) {
  if (pk->ktable) or_fixed_table_mul(r, pk->ktable, k);
  else or_ge_scalarmult(r, k, &pk->element);
}

static void ensure_ktable(or_pubkey *pk) {
  if (!pk->ktable) pk->ktable = or_fixed_table_new(&pk->element);
}

void or_keypair_generate(chacha_rng *rng, sc *sk, or_pubkey *pk) {
  /* Keypair::generate -> SecretKey::generate (one generate_scalar draw), pk = [sk]G */
  or_rng_scalar(rng, sk);
  or_ge_mul_generator(&pk->element, sk);
  or_ristretto_encode(pk->bytes, &pk->element);
  pk->ktable = NULL;
  ensure_ktable(pk);
}

/* ExtendedCiphertext (encryption.rs:303-327) */
typedef struct { ge R, B; sc r; } ext_ct;

static void ext_ct_new(ext_ct *c, const ge *value, const or_pubkey *pk, chacha_rng *rng) {
  or_rng_scalar(rng, &c->r);
  or_ge_mul_generator(&c->R, &c->r);
  ge dh;
  mul_k(&dh, pk, &c->r);
  or_ge_add(&c->B, value, &dh);
}

/* ------------------------------------------------------------------ rings */

typedef struct {
  int index, size, value_index;
  const ge *admissible;
  ge R, B;
  merlin_t transcript;
  sc *responses;
  ge term_g, term_k;
  sc discrete_log, random_scalar;
} ring_t;

static void ring_commitments(ge *cg, ge *ck, const ring_t *ring, const or_pubkey *pk, int eq,
                             const sc *response, const sc *challenge) {
  /* ([s]G - [e]R, [s]K - [e](B - x_eq))  ring.rs:113-117,176-180 */
  ge dh, t0, t1;
  sc neg_e;
  or_sc_neg(&neg_e, challenge);
  or_ge_sub(&dh, &ring->B, &ring->admissible[eq]);
  or_ge_mul_generator(&t0, response);
  or_ge_scalarmult(&t1, &neg_e, &ring->R);
  or_ge_add(cg, &t0, &t1);
  mul_k(&t0, pk, response);
  or_ge_scalarmult(&t1, &neg_e, &dh);
  or_ge_add(ck, &t0, &t1);
}

static void ring_new(ring_t *ring, int index, const or_pubkey *pk, const ext_ct *ct,
                     const ge *admissible, int size, int value_index, const merlin_t *transcript,
                     sc *responses, chacha_rng *rng) {
  /* ring.rs:54-131 */
  ring->index = index;
  ring->size = size;
  ring->value_index = value_index;
  ring->admissible = admissible;
  ring->R = ct->R;
  ring->B = ct->B;
  ring->responses = responses;
  ring->discrete_log = ct->r;
  ring->transcript = *transcript;
  or_t_start_proof(&ring->transcript, "ring_enc");
  uint8_t enc[64];
  or_ristretto_encode(enc, &ct->R);
  or_ristretto_encode(enc + 32, &ct->B);
  or_merlin_append(&ring->transcript, "enc", enc, 64);
  or_merlin_append_u64(&ring->transcript, "i", (uint64_t)index);

  or_rng_scalar(rng, &ring->random_scalar);
  ge cg, ck;
  or_ge_mul_generator(&cg, &ring->random_scalar);
  mul_k(&ck, pk, &ring->random_scalar);
  for (int eq = value_index + 1; eq < size; eq++) {
    merlin_t et = ring->transcript;
    or_merlin_append_u64(&et, "j", (uint64_t)(eq - 1));
    or_t_append_element(&et, "R_G", &cg);
    or_t_append_element(&et, "R_K", &ck);
    sc challenge;
    or_t_challenge_scalar(&et, "c", &challenge);
    or_rng_scalar(rng, &responses[eq]);
    ring_commitments(&cg, &ck, ring, pk, eq, &responses[eq], &challenge);
  }
  ring->term_g = cg;
  ring->term_k = ck;
}

static void ring_finalize(ring_t *ring, const or_pubkey *pk, const sc *common, chacha_rng *rng) {
  /* ring.rs:162-194 */
  sc challenge = *common;
  for (int eq = 0; eq < ring->value_index; eq++) {
    or_rng_scalar(rng, &ring->responses[eq]);
    ge cg, ck;
    ring_commitments(&cg, &ck, ring, pk, eq, &ring->responses[eq], &challenge);
    merlin_t et = ring->transcript;
    or_merlin_append_u64(&et, "j", (uint64_t)eq);
    or_t_append_element(&et, "R_G", &cg);
    or_t_append_element(&et, "R_K", &ck);
    or_t_challenge_scalar(&et, "c", &challenge);
  }
  or_sc_muladd(&ring->responses[ring->value_index], &challenge, &ring->discrete_log,
               &ring->random_scalar);
}

typedef struct {
  const or_pubkey *pk;
  merlin_t *transcript;
  ring_t *rings; /* one per ciphertext of the proof */
  int n_rings;
  sc *responses;
  size_t used;
  chacha_rng *rng;
} builder_t;

static void builder_init(builder_t *b, const or_pubkey *pk, sc *responses, merlin_t *t, chacha_rng *rng, int max_rings) {
  /* ring.rs:442-457 */
  or_t_start_proof(t, "multi_ring_enc");
  or_merlin_append(t, "K", pk->bytes, 32);
  b->pk = pk; b->transcript = t; b->n_rings = 0; b->responses = responses; b->used = 0; b->rng = rng;
  b->rings = (ring_t *)malloc(sizeof(ring_t) * (size_t)(max_rings > 0 ? max_rings : 1));
}

static void builder_add_precomputed(builder_t *b, const ext_ct *ct, const ge *adm, int size, int vi) {
  /* ring.rs:471-492 */
  ring_new(&b->rings[b->n_rings], b->n_rings, b->pk, ct, adm, size, vi, b->transcript,
           b->responses + b->used, b->rng);
  b->used += (size_t)size;
  b->n_rings++;
}

static void builder_add_value(builder_t *b, const ge *adm, int size, int vi, ext_ct *out) {
  /* ring.rs:460-469 */
  ext_ct_new(out, &adm[vi], b->pk, b->rng);
  builder_add_precomputed(b, out, adm, size, vi);
}

static void builder_build(builder_t *b, sc *common) {
  /* Ring::aggregate, ring.rs:138-160 */
  for (int i = 0; i < b->n_rings; i++) {
    or_t_append_element(b->transcript, "R_G", &b->rings[i].term_g);
    or_t_append_element(b->transcript, "R_K", &b->rings[i].term_k);
  }
  or_t_challenge_scalar(b->transcript, "c", common);
  for (int i = 0; i < b->n_rings; i++) ring_finalize(&b->rings[i], b->pk, common, b->rng);
}

/* ------------------------------------------------------------------ LogEqualityProof::new */

static void logeq_new(const or_pubkey *pk, const sc *secret, const ge *p0, const ge *p1,
                      merlin_t *t, chacha_rng *rng, sc *challenge, sc *response) {
  /* log_equality.rs:114-139 */
  or_t_start_proof(t, "log_eq");
  or_merlin_append(t, "K", pk->bytes, 32);
  or_t_append_element(t, "[r]G", p0);
  or_t_append_element(t, "[r]K", p1);
  sc x;
  or_rng_scalar(rng, &x);
  ge xg, xk;
  or_ge_mul_generator(&xg, &x);
  mul_k(&xk, pk, &x);
  or_t_append_element(t, "[x]G", &xg);
  or_t_append_element(t, "[x]K", &xk);
  or_t_challenge_scalar(t, "c", challenge);
  or_sc_muladd(response, challenge, secret, &x);
}

/* ------------------------------------------------------------------ simple encryptions */

static void put_ct(uint8_t *out, const ge *R, const ge *B) {
  or_ristretto_encode(out, R);
  or_ristretto_encode(out + 32, B);
}

void or_encrypt_u64(const or_pubkey *pk, uint64_t value, chacha_rng *rng, uint8_t out[64]) {
  sc v;
  ge vg;
  ext_ct c;
  or_sc_from_u64(&v, value);
  or_ge_mul_generator(&vg, &v);
  ext_ct_new(&c, &vg, pk, rng);
  put_ct(out, &c.R, &c.B);
}

void or_encrypt_zero(const or_pubkey *pk, chacha_rng *rng, uint8_t out[128]) {
  /* keys/impls.rs:30-51 */
  sc r;
  ge R, B;
  or_rng_scalar(rng, &r);
  or_ge_mul_generator(&R, &r);
  mul_k(&B, pk, &r);
  put_ct(out, &R, &B);
  merlin_t t;
  or_merlin_init(&t, "zero_encryption");
  logeq_new(pk, &r, &R, &B, &t, rng, (sc *)(out + 64), (sc *)(out + 96));
}

void or_encrypt_bool(const or_pubkey *pk, int value, chacha_rng *rng, uint8_t out[160]) {
  /* keys/impls.rs:77-91 */
  merlin_t t;
  or_merlin_init(&t, "bool_encryption");
  ge adm[2];
  or_ge_identity(&adm[0]);
  or_ge_generator(&adm[1]);
  sc responses[2];
  memset(responses, 0, sizeof responses);
  builder_t *b = (builder_t *)malloc(sizeof(builder_t));
  builder_init(b, pk, responses, &t, rng, 1);
  ext_ct c;
  builder_add_value(b, adm, 2, value ? 1 : 0, &c);
  builder_build(b, (sc *)(out + 64));
  put_ct(out, &c.R, &c.B);
  memcpy(out + 96, responses, 64);
  free(b->rings); free(b);
}

/* ------------------------------------------------------------------ EncryptedChoice::new */

void or_choice_new(const or_choice_params *p, const uint8_t *flags, chacha_rng *rng, uint8_t *out) {
  /* choice.rs:313-349 */
  int n = p->n_options;
  ge adm[2];
  or_ge_identity(&adm[0]);
  or_ge_generator(&adm[1]);
  sc *responses = (sc *)calloc((size_t)(2 * n), sizeof(sc));
  merlin_t t;
  or_merlin_init(&t, "encrypted_choice_ranges");
  builder_t *b = (builder_t *)malloc(sizeof(builder_t));
  builder_init(b, &p->pk, responses, &t, rng, n);
  ge sum_r, sum_b;
  sc sum_rand;
  or_ge_identity(&sum_r);
  or_ge_identity(&sum_b);
  or_sc_from_u64(&sum_rand, 0);
  for (int i = 0; i < n; i++) {
    ext_ct c;
    builder_add_value(b, adm, 2, flags[i] ? 1 : 0, &c);
    put_ct(out + 64 * (size_t)i, &c.R, &c.B);
    or_ge_add(&sum_r, &sum_r, &c.R);
    or_ge_add(&sum_b, &sum_b, &c.B);
    or_sc_add(&sum_rand, &sum_rand, &c.r);
  }
  uint8_t *ring_proof = out + 64 * (size_t)n;
  builder_build(b, (sc *)ring_proof);
  memcpy(ring_proof + 32, responses, 32 * (size_t)(2 * n));
  if (p->single) { /* SingleChoice::prove, choice.rs:59-75 */
    ge g, p1;
    or_ge_generator(&g);
    or_ge_sub(&p1, &sum_b, &g);
    merlin_t ts;
    or_merlin_init(&ts, "choice_encryption_sum");
    uint8_t *sum_proof = ring_proof + 32 * (size_t)(1 + 2 * n);
    logeq_new(&p->pk, &sum_rand, &sum_r, &p1, &ts, rng, (sc *)sum_proof, (sc *)(sum_proof + 32));
  }
  free(b->rings); free(b);
  free(responses);
}

/* ------------------------------------------------------------------ RangeProof::new */

/* returns the value ciphertext (with randomness) and writes ct || partials || ring proof */
static void range_new(const or_pubkey *pk, const or_prepared_range *r, uint64_t value,
                      const char *label, chacha_rng *rng, uint8_t *out, ext_ct *out_ct) {
  /* range.rs:462-534 */
  sc v;
  ge vg;
  or_sc_from_u64(&v, value);
  or_ge_mul_generator(&vg, &v);
  ext_ct ct;
  ext_ct_new(&ct, &vg, pk, rng); /* CiphertextWithValue::new, encryption.rs:403-407 */
  int idx[OR_MAX_RINGS];
  or_range_decompose(&r->d, value, idx);
  merlin_t t;
  or_merlin_init(&t, label);
  or_t_start_proof(&t, "encryption_range_proof");
  or_merlin_append(&t, "range", (const uint8_t *)r->name, (size_t)r->name_len);
  sc *responses = (sc *)calloc((size_t)r->total_size, sizeof(sc));
  builder_t *b = (builder_t *)malloc(sizeof(builder_t));
  builder_init(b, pk, responses, &t, rng, OR_MAX_RINGS);
  ext_ct cum;
  or_ge_identity(&cum.R);
  or_ge_identity(&cum.B);
  or_sc_from_u64(&cum.r, 0);
  int nr = r->n_rings;
  put_ct(out, &ct.R, &ct.B);
  for (int i = 0; i < nr - 1; i++) {
    ext_ct c;
    builder_add_value(b, r->table[i], (int)r->d.size[i], idx[i], &c);
    put_ct(out + 64 + 64 * (size_t)i, &c.R, &c.B);
    or_ge_add(&cum.R, &cum.R, &c.R);
    or_ge_add(&cum.B, &cum.B, &c.B);
    or_sc_add(&cum.r, &cum.r, &c.r);
  }
  ext_ct last;
  or_ge_sub(&last.R, &ct.R, &cum.R);
  or_ge_sub(&last.B, &ct.B, &cum.B);
  or_sc_sub(&last.r, &ct.r, &cum.r);
  builder_add_precomputed(b, &last, r->table[nr - 1], (int)r->d.size[nr - 1], idx[nr - 1]);
  uint8_t *proof = out + 64 + 64 * (size_t)(nr - 1);
  builder_build(b, (sc *)proof);
  memcpy(proof + 32, responses, 32 * (size_t)r->total_size);
  free(b->rings); free(b);
  free(responses);
  if (out_ct) *out_ct = ct;
}

void or_encrypt_range(const or_pubkey *pk, const or_prepared_range *r, uint64_t value,
                      chacha_rng *rng, uint8_t *out) {
  /* keys/impls.rs:120-129 */
  range_new(pk, r, value, "ciphertext_range", rng, out, NULL);
}

/* ------------------------------------------------------------------ SumOfSquaresProof::new */

static void sumsq_new(const or_pubkey *pk, int n, const ext_ct *cts, const uint64_t *values,
                      const ext_ct *sum_ct, const char *label, chacha_rng *rng, uint8_t *out) {
  /* mul.rs:107-181 */
  merlin_t t;
  or_merlin_init(&t, label);
  or_t_start_proof(&t, "sum_of_squares");
  or_merlin_append(&t, "K", pk->bytes, 32);
  sc e_z, sum_rand = sum_ct->r;
  or_rng_scalar(rng, &e_z); /* :116 */
  sc *e_r = (sc *)malloc(sizeof(sc) * (size_t)n), *e_x = (sc *)malloc(sizeof(sc) * (size_t)n);
  for (int i = 0; i < n; i++) { /* :119-139 */
    or_t_append_element(&t, "R_x", &cts[i].R);
    or_t_append_element(&t, "X", &cts[i].B);
    or_rng_scalar(rng, &e_r[i]);
    ge c0, c1, c2;
    or_ge_mul_generator(&c0, &e_r[i]);
    or_t_append_element(&t, "[e_r]G", &c0);
    or_rng_scalar(rng, &e_x[i]);
    or_ge_mul_generator(&c1, &e_x[i]);
    mul_k(&c2, pk, &e_r[i]);
    or_ge_add(&c1, &c1, &c2);
    or_t_append_element(&t, "[e_x]G + [e_r]K", &c1);
    sc x, neg_x;
    or_sc_from_u64(&x, values[i]);
    or_sc_neg(&neg_x, &x);
    or_sc_muladd(&sum_rand, &cts[i].r, &neg_x, &sum_rand);
  }
  sc *ks = (sc *)malloc(sizeof(sc) * (size_t)(n + 1));
  ge *ps = (ge *)malloc(sizeof(ge) * (size_t)(n + 1));
  for (int i = 0; i < n; i++) ks[i] = e_x[i];
  ks[n] = e_z;
  ge rsum, vsum;
  for (int i = 0; i < n; i++) ps[i] = cts[i].R;
  or_ge_generator(&ps[n]);
  or_ge_multi_mul(&rsum, (size_t)(n + 1), ks, ps);
  for (int i = 0; i < n; i++) ps[i] = cts[i].B;
  ps[n] = pk->element;
  or_ge_multi_mul(&vsum, (size_t)(n + 1), ks, ps);
  or_t_append_element(&t, "R_z", &sum_ct->R);
  or_t_append_element(&t, "Z", &sum_ct->B);
  or_t_append_element(&t, "[e_x]R_x + [e_z]G", &rsum);
  or_t_append_element(&t, "[e_x]X + [e_z]K", &vsum);
  sc c;
  or_t_challenge_scalar(&t, "c", &c);
  memcpy(out, c.b, 32);
  for (int i = 0; i < n; i++) { /* :163-172 */
    sc x;
    or_sc_from_u64(&x, values[i]);
    or_sc_muladd((sc *)(out + 32 + 64 * (size_t)i), &c, &cts[i].r, &e_r[i]);
    or_sc_muladd((sc *)(out + 64 + 64 * (size_t)i), &c, &x, &e_x[i]);
  }
  or_sc_muladd((sc *)(out + 32 + 64 * (size_t)n), &c, &sum_rand, &e_z);
  free(e_r); free(e_x); free(ks); free(ps);
}

void or_sumsq_snapshot(const or_pubkey *pk, int n, const uint64_t *values, chacha_rng *rng,
                       uint8_t *out_cts, uint8_t *out_proof) {
  /* tests/snapshots.rs:128-150 */
  uint64_t ss = 0;
  for (int i = 0; i < n; i++) ss += values[i] * values[i];
  ext_ct sum_ct, *cts = (ext_ct *)malloc(sizeof(ext_ct) * (size_t)n);
  sc v;
  ge vg;
  or_sc_from_u64(&v, ss);
  or_ge_mul_generator(&vg, &v);
  ext_ct_new(&sum_ct, &vg, pk, rng);
  put_ct(out_cts, &sum_ct.R, &sum_ct.B);
  for (int i = 0; i < n; i++) {
    or_sc_from_u64(&v, values[i]);
    or_ge_mul_generator(&vg, &v);
    ext_ct_new(&cts[i], &vg, pk, rng);
    put_ct(out_cts + 64 * (size_t)(i + 1), &cts[i].R, &cts[i].B);
  }
  sumsq_new(pk, n, cts, values, &sum_ct, "test", rng, out_proof);
  free(cts);
}

/* ------------------------------------------------------------------ QuadraticVotingBallot::new */

void or_qv_new(const or_qv_params *p, const uint64_t *votes, chacha_rng *rng, uint8_t *out) {
  /* quadratic_voting.rs:234-284 */
  int n = p->n_options;
  size_t vote_sz = 64 + or_range_proof_size(&p->vote_range);
  size_t credit_sz = 64 + or_range_proof_size(&p->credit_range);
  ext_ct *cts = (ext_ct *)malloc(sizeof(ext_ct) * (size_t)n);
  uint64_t credit = 0;
  for (int i = 0; i < n; i++) credit += votes[i] * votes[i];
  for (int i = 0; i < n; i++)
    range_new(&p->pk, &p->vote_range, votes[i], "quadratic_voting_variant", rng,
              out + (size_t)i * vote_sz, &cts[i]);
  ext_ct credit_ct;
  range_new(&p->pk, &p->credit_range, credit, "quadratic_voting_credit_range", rng,
            out + (size_t)n * vote_sz, &credit_ct);
  sumsq_new(&p->pk, n, cts, votes, &credit_ct, "quadratic_voting_credit_equiv", rng,
            out + (size_t)n * vote_sz + credit_sz);
  free(cts);
}

/* ------------------------------------------------------------------ decryption shares (tally stage) */

static void share_transcript(merlin_t *t, uint64_t shares, uint64_t threshold, const uint8_t shared_key[32], uint64_t index) {
  /* PublicKeySet::verify_share / ActiveParticipant::decrypt_share: sharing/key_set.rs:209-228,167-171 */
  or_merlin_init(t, "elgamal_decryption_share");
  or_merlin_append_u64(t, "n", shares);
  or_merlin_append_u64(t, "t", threshold);
  or_merlin_append(t, "K", shared_key, 32);
  or_merlin_append_u64(t, "i", index);
}

/* out = dh(32) || challenge || response : dh = [sk_share]R with a log-equality proof (participant.rs:163-186) */
int or_decryption_share_new(const uint8_t sk_share[32], const uint8_t ct_random[32], uint64_t shares, uint64_t threshold,
                            const uint8_t shared_key[32], uint64_t index, chacha_rng *rng, uint8_t out[96]) {
  or_pubkey base;   /* log_base = PublicKey::from_element(ciphertext.random_element) */
  if (!or_ristretto_decode(&base.element, ct_random)) return -1;
  memcpy(base.bytes, ct_random, 32);
  base.ktable = NULL;
  ge dh, ks;
  or_ge_scalarmult(&dh, (const sc *)sk_share, &base.element);
  or_ge_mul_generator(&ks, (const sc *)sk_share);
  or_ristretto_encode(out, &dh);
  merlin_t t;
  share_transcript(&t, shares, threshold, shared_key, index);
  logeq_new(&base, (const sc *)sk_share, &ks, &dh, &t, rng, (sc *)(out + 32), (sc *)(out + 64));
  return 0;
}

/* item = R(32) || dh(32) || challenge || response */
uint32_t or_decryption_share_verify(const uint8_t key_share[32], uint64_t shares, uint64_t threshold,
                                    const uint8_t shared_key[32], uint64_t index, const uint8_t item[128]) {
  or_pubkey base;
  ge ks, dh;
  if (!or_ristretto_decode(&base.element, item)) return OR_STATUS(OR_BAD_POINT, 0);
  if (!or_ristretto_decode(&dh, item + 32)) return OR_STATUS(OR_BAD_POINT, 1);
  if (!or_sc_is_canonical(item + 64)) return OR_STATUS(OR_BAD_SCALAR, 2);
  if (!or_sc_is_canonical(item + 96)) return OR_STATUS(OR_BAD_SCALAR, 3);
  if (!or_ristretto_decode(&ks, key_share)) return OR_STATUS(OR_BAD_POINT, 0xffff);
  memcpy(base.bytes, item, 32);
  base.ktable = NULL;
  merlin_t t;
  share_transcript(&t, shares, threshold, shared_key, index);
  return or_logeq_verify(&base, &ks, &dh, (const sc *)(item + 64), (const sc *)(item + 96), &t) ? OR_OK : OR_SUM_CHALLENGE;
}

/* ------------------------------------------------------------------ selection streams */

static uint32_t sel_next(chacha_rng *r, uint8_t buf[64], int *pos) {
  if (*pos >= 64) { or_rng_fill64(r, buf); *pos = 0; }
  uint32_t x = (uint32_t)buf[*pos] | ((uint32_t)buf[*pos + 1] << 8) | ((uint32_t)buf[*pos + 2] << 16) |
               ((uint32_t)buf[*pos + 3] << 24);
  *pos += 4;
  return x;
}

static void sel_init(chacha_rng *r, uint64_t ballot_seed) { or_rng_seed_from_u64(r, ~ballot_seed); }

void or_select_single(uint64_t ballot_seed, int n, uint8_t *flags) {
  chacha_rng r; uint8_t buf[64]; int pos = 64;
  sel_init(&r, ballot_seed);
  memset(flags, 0, (size_t)n);
  flags[sel_next(&r, buf, &pos) % (uint32_t)n] = 1;
}

void or_select_multi(uint64_t ballot_seed, int n, int k, uint8_t *flags) {
  chacha_rng r; uint8_t buf[64]; int pos = 64, got = 0;
  sel_init(&r, ballot_seed);
  memset(flags, 0, (size_t)n);
  while (got < k) {
    uint32_t i = sel_next(&r, buf, &pos) % (uint32_t)n;
    if (!flags[i]) { flags[i] = 1; got++; }
  }
}

void or_select_qv(uint64_t ballot_seed, int n, uint64_t credits, uint64_t *votes) {
  /* vote drawing as in tests/integration/sharing.rs:135-147 (geometric, p = 0.8) */
  chacha_rng r; uint8_t buf[64]; int pos = 64;
  sel_init(&r, ballot_seed);
  memset(votes, 0, sizeof(uint64_t) * (size_t)n);
  for (;;) {
    if (sel_next(&r, buf, &pos) % 10 >= 8) break;
    uint32_t i = sel_next(&r, buf, &pos) % (uint32_t)n;
    uint64_t c = 0;
    for (int j = 0; j < n; j++) { uint64_t v = votes[j] + (j == (int)i); c += v * v; }
    if (c > credits) break;
    votes[i]++;
  }
}

/* ------------------------------------------------------------------ batch drivers */

typedef struct {
  int kind; /* 0 choice gen, 1 qv gen, 2 choice verify, 3 qv verify */
  const void *params;
  uint64_t base_seed;
  size_t first, begin, end;
  int n_selected;
  uint8_t *buf;
  uint32_t *status;
} job_t;

static void *worker(void *arg) {
  job_t *j = (job_t *)arg;
  if (j->kind == 0) {
    const or_choice_params *p = (const or_choice_params *)j->params;
    size_t sz = or_choice_ballot_size(p->n_options, p->single);
    uint8_t flags[256];
    for (size_t b = j->begin; b < j->end; b++) {
      uint64_t seed = j->base_seed + j->first + b;
      chacha_rng rng;
      or_rng_seed_from_u64(&rng, seed);
      if (p->single) or_select_single(seed, p->n_options, flags);
      else or_select_multi(seed, p->n_options, j->n_selected, flags);
      or_choice_new(p, flags, &rng, j->buf + b * sz);
    }
  } else if (j->kind == 1) {
    const or_qv_params *p = (const or_qv_params *)j->params;
    size_t sz = or_qv_ballot_size(p);
    uint64_t votes[256];
    for (size_t b = j->begin; b < j->end; b++) {
      uint64_t seed = j->base_seed + j->first + b;
      chacha_rng rng;
      or_rng_seed_from_u64(&rng, seed);
      or_select_qv(seed, p->n_options, p->credits, votes);
      or_qv_new(p, votes, &rng, j->buf + b * sz);
    }
  } else if (j->kind == 2) {
    const or_choice_params *p = (const or_choice_params *)j->params;
    size_t sz = or_choice_ballot_size(p->n_options, p->single);
    for (size_t b = j->begin; b < j->end; b++) j->status[b] = or_choice_verify(p, j->buf + b * sz);
  } else {
    const or_qv_params *p = (const or_qv_params *)j->params;
    size_t sz = or_qv_ballot_size(p);
    for (size_t b = j->begin; b < j->end; b++) j->status[b] = or_qv_verify(p, j->buf + b * sz);
  }
  return NULL;
}

static void run_jobs(job_t proto, size_t n, int n_threads) {
  or_init();
  if (n_threads < 1) n_threads = 1;
  if ((size_t)n_threads > n) n_threads = n ? (int)n : 1;
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
  job_t *jobs = (job_t *)malloc(sizeof(job_t) * (size_t)n_threads);
  for (int t = 0; t < n_threads; t++) {
    jobs[t] = proto;
    jobs[t].begin = n * (size_t)t / (size_t)n_threads;
    jobs[t].end = n * (size_t)(t + 1) / (size_t)n_threads;
    if (n_threads == 1) worker(&jobs[t]);
    else pthread_create(&th[t], NULL, worker, &jobs[t]);
  }
  if (n_threads > 1)
    for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
  free(th);
  free(jobs);
}

void or_choice_generate_batch(const or_choice_params *p, uint64_t base_seed, size_t first, size_t n,
                              int n_selected, uint8_t *out, int n_threads) {
  ensure_ktable((or_pubkey *)&p->pk);
  job_t j = {0, p, base_seed, first, 0, 0, n_selected, out, NULL};
  run_jobs(j, n, n_threads);
}

void or_qv_generate_batch(const or_qv_params *p, uint64_t base_seed, size_t first, size_t n,
                          uint8_t *out, int n_threads) {
  ensure_ktable((or_pubkey *)&p->pk);
  job_t j = {1, p, base_seed, first, 0, 0, 0, out, NULL};
  run_jobs(j, n, n_threads);
}

void or_choice_verify_batch(const or_choice_params *p, size_t n, const uint8_t *ballots,
                            uint32_t *status, int n_threads) {
  job_t j = {2, p, 0, 0, 0, 0, 0, (uint8_t *)ballots, status};
  run_jobs(j, n, n_threads);
}

void or_qv_verify_batch(const or_qv_params *p, size_t n, const uint8_t *ballots, uint32_t *status,
                        int n_threads) {
  job_t j = {3, p, 0, 0, 0, 0, 0, (uint8_t *)ballots, status};
  run_jobs(j, n, n_threads);
}

void or_tally(int n_options, size_t stride, size_t n, const uint8_t *ballots, const uint32_t *status,
              size_t ct_offset, size_t ct_stride, uint8_t *out) {
  /* examples/voting.rs:199-203: totals[k] += vote[k] for verified ballots */
  ge *acc = (ge *)malloc(sizeof(ge) * (size_t)(2 * n_options));
  for (int i = 0; i < 2 * n_options; i++) or_ge_identity(&acc[i]);
  for (size_t b = 0; b < n; b++) {
    if (status[b] != OR_OK) continue;
    for (int k = 0; k < n_options; k++) {
      const uint8_t *ct = ballots + b * stride + ct_offset + (size_t)k * ct_stride;
      ge R, B;
      or_ristretto_decode(&R, ct);
      or_ristretto_decode(&B, ct + 32);
      or_ge_add(&acc[2 * k], &acc[2 * k], &R);
      or_ge_add(&acc[2 * k + 1], &acc[2 * k + 1], &B);
    }
  }
  for (int i = 0; i < 2 * n_options; i++) or_ristretto_encode(out + 32 * (size_t)i, &acc[i]);
  free(acc);
}

/*
 * proofs.c -- CPU ORACLE (test infrastructure, not the product): the reference's verifiers.
 *
 * Each function follows the control flow, transcript labels, ordering and error precedence of the
 * reference function it cites.  Packed layouts are the concatenation of the reference's own
 * to_bytes formats (encryption.rs:155-160, ring.rs:383-392, log_equality.rs:184-189); for
 * RangeProof / SumOfSquaresProof (no to_bytes in the reference) the serde field order is used
 * (range.rs:446-450, mul.rs:86-93).
 */
#include "eg_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ keys */

int or_pubkey_from_bytes(or_pubkey *pk, const uint8_t b[32]) {
  /* keys/mod.rs:161-176 */
  if (!or_ristretto_decode(&pk->element, b)) return -1;
  if (or_ge_is_identity(&pk->element)) return -2;
  memcpy(pk->bytes, b, 32);
  pk->ktable = NULL;
  return 0;
}

/* ------------------------------------------------------------------ LogEqualityProof::verify */

int or_logeq_verify(const or_pubkey *pk, const ge *p0, const ge *p1, const sc *challenge,
                    const sc *response, merlin_t *t) {
  /* log_equality.rs:153-180 */
  sc neg_c;
  or_sc_neg(&neg_c, challenge);
  ge c0, c1;
  or_ge_double_mul_generator(&c0, &neg_c, p0, response); /* :160 */
  sc ks[2] = {neg_c, *response};
  ge ps[2] = {*p1, pk->element};
  or_ge_multi_mul(&c1, 2, ks, ps); /* :161-164 */

  or_t_start_proof(t, "log_eq");
  or_merlin_append(t, "K", pk->bytes, 32);
  or_t_append_element(t, "[r]G", p0);
  or_t_append_element(t, "[r]K", p1);
  or_t_append_element(t, "[x]G", &c0);
  or_t_append_element(t, "[x]K", &c1);
  sc expected;
  or_t_challenge_scalar(t, "c", &expected);
  return or_sc_eq(&expected, challenge);
}

/* ------------------------------------------------------------------ RingProof::verify */

int or_ring_verify(const or_pubkey *pk, int n_rings, const int *sizes, const ge *const *admissible,
                   const ge *cts, const sc *common_challenge, size_t n_responses,
                   const sc *responses, merlin_t *t) {
  /* ring.rs:302-374 */
  size_t total = 0;
  for (int i = 0; i < n_rings; i++) total += (size_t)sizes[i];
  if (total != n_responses) return OR_RANGE_LEN; /* :310-315 */

  or_t_start_proof(t, "multi_ring_enc"); /* :290-293 */
  or_merlin_append(t, "K", pk->bytes, 32);
  merlin_t initial = *t; /* :320 */

  size_t start = 0;
  for (int ring = 0; ring < n_rings; ring++) {
    const ge *R = &cts[2 * ring], *B = &cts[2 * ring + 1];
    sc challenge = *common_challenge;
    ge cg, ck;
    or_ge_generator(&cg);
    ck = cg;

    merlin_t rt = initial;
    or_t_start_proof(&rt, "ring_enc");
    uint8_t enc[64];
    or_ristretto_encode(enc, R);
    or_ristretto_encode(enc + 32, B);
    or_merlin_append(&rt, "enc", enc, 64);
    or_merlin_append_u64(&rt, "i", (uint64_t)ring);

    for (int eq = 0; eq < sizes[ring]; eq++) {
      const sc *response = &responses[start + (size_t)eq];
      ge dh;
      or_ge_sub(&dh, B, &admissible[ring][eq]); /* :338 */
      sc neg_e;
      or_sc_neg(&neg_e, &challenge);
      or_ge_double_mul_generator(&cg, &neg_e, R, response); /* :342-346 */
      sc ks[2] = {*response, neg_e};
      ge ps[2] = {pk->element, dh};
      or_ge_multi_mul(&ck, 2, ks, ps); /* :347-350 */
      if (eq + 1 < sizes[ring]) { /* :354-360 */
        merlin_t et = rt;
        or_merlin_append_u64(&et, "j", (uint64_t)eq);
        or_t_append_element(&et, "R_G", &cg);
        or_t_append_element(&et, "R_K", &ck);
        or_t_challenge_scalar(&et, "c", &challenge);
      }
    }
    start += (size_t)sizes[ring];
    or_t_append_element(t, "R_G", &cg); /* :364-365 */
    or_t_append_element(t, "R_K", &ck);
  }
  sc expected;
  or_t_challenge_scalar(t, "c", &expected);
  return or_sc_eq(&expected, common_challenge) ? OR_OK : OR_RANGE_CHALLENGE;
}

/* ------------------------------------------------------------------ packed-ballot helpers */

size_t or_choice_ballot_size(int n, int single) {
  return (size_t)n * 64 + 32 * (size_t)(1 + 2 * n) + (single ? 64 : 0);
}

/* scan a packed blob described by a type string: 'P' = point, 'S' = scalar (32 B each).
 * Mirrors deserialisation-time rejection (serde.rs:191-206,254-269): first bad item wins. */
static uint32_t scan_items(const uint8_t *blob, size_t n_items, const char *kinds, ge *points_out,
                           size_t *n_points) {
  size_t np = 0;
  for (size_t i = 0; i < n_items; i++) {
    const uint8_t *item = blob + 32 * i;
    if (kinds[i] == 'P') {
      if (!or_ristretto_decode(&points_out[np], item)) return OR_STATUS(OR_BAD_POINT, i);
      np++;
    } else {
      if (!or_sc_is_canonical(item)) return OR_STATUS(OR_BAD_SCALAR, i);
    }
  }
  if (n_points) *n_points = np;
  return OR_OK;
}

static const sc *as_sc(const uint8_t *p) { return (const sc *)p; }

/* ------------------------------------------------------------------ EncryptedChoice::verify */

uint32_t or_choice_verify(const or_choice_params *p, const uint8_t *ballot) {
  /* choice.rs:358-380.  check_options_count (:149-158) is implied by the packed stride. */
  int n = p->n_options;
  size_t n_items = or_choice_ballot_size(n, p->single) / 32;
  char *kinds = (char *)malloc(n_items);
  ge *pts = (ge *)malloc(sizeof(ge) * (size_t)(2 * n));
  memset(kinds, 'S', n_items);
  memset(kinds, 'P', (size_t)(2 * n));
  uint32_t st = scan_items(ballot, n_items, kinds, pts, NULL);
  free(kinds);
  if (st != OR_OK) { free(pts); return st; }

  const uint8_t *ring_proof = ballot + 64 * (size_t)n;
  const uint8_t *sum_proof = ring_proof + 32 * (size_t)(1 + 2 * n);

  if (p->single) {
    /* sum of ciphertexts (:363), SingleChoice::verify (:77-94) */
    ge sr = pts[0], sb = pts[1], g;
    for (int i = 1; i < n; i++) {
      or_ge_add(&sr, &sr, &pts[2 * i]);
      or_ge_add(&sb, &sb, &pts[2 * i + 1]);
    }
    or_ge_generator(&g);
    or_ge_sub(&sb, &sb, &g);
    merlin_t t;
    or_merlin_init(&t, "choice_encryption_sum");
    if (!or_logeq_verify(&p->pk, &sr, &sb, as_sc(sum_proof), as_sc(sum_proof + 32), &t)) {
      free(pts);
      return OR_SUM_CHALLENGE;
    }
  }

  ge adm[2];
  or_ge_identity(&adm[0]);
  or_ge_generator(&adm[1]);
  int *sizes = (int *)malloc(sizeof(int) * (size_t)n);
  const ge **tables = (const ge **)malloc(sizeof(ge *) * (size_t)n);
  for (int i = 0; i < n; i++) { sizes[i] = 2; tables[i] = adm; }
  merlin_t t;
  or_merlin_init(&t, "encrypted_choice_ranges");
  int r = or_ring_verify(&p->pk, n, sizes, tables, pts, as_sc(ring_proof), (size_t)(2 * n),
                         as_sc(ring_proof + 32), &t);
  free(sizes); free(tables); free(pts);
  return (uint32_t)r;
}

/* ------------------------------------------------------------------ PreparedRange / RangeProof */

void or_prepared_range_init(or_prepared_range *r, const range_decomp *d) {
  /* range.rs:341-355 */
  r->d = *d;
  r->n_rings = d->n_rings;
  r->total_size = 0;
  for (int i = 0; i < d->n_rings; i++) {
    for (uint64_t j = 0; j < d->size[i]; j++) {
      sc k;
      or_sc_from_u64(&k, j * d->step[i]);
      or_ge_mul_generator(&r->table[i][j], &k);
    }
    r->total_size += (int)d->size[i];
  }
  r->name_len = or_range_to_string(d, r->name, sizeof r->name);
}

size_t or_range_proof_size(const or_prepared_range *r) {
  return 64 * (size_t)(r->n_rings - 1) + 32 * (size_t)(1 + r->total_size);
}

/* RangeProof::verify with already-decoded inputs (range.rs:547-577) */
static int range_verify_decoded(const or_pubkey *pk, const or_prepared_range *r, const ge *ct,
                                const ge *partials, const sc *common_challenge,
                                const sc *responses, const char *label) {
  merlin_t t;
  or_merlin_init(&t, label);
  or_t_start_proof(&t, "encryption_range_proof"); /* :561 */
  or_merlin_append(&t, "range", (const uint8_t *)r->name, (size_t)r->name_len); /* :562 */
  int nr = r->n_rings;
  ge cts[2 * OR_MAX_RINGS];
  ge sum_r, sum_b;
  or_ge_identity(&sum_r);
  or_ge_identity(&sum_b);
  for (int i = 0; i < nr - 1; i++) { /* :564-567 */
    cts[2 * i] = partials[2 * i];
    cts[2 * i + 1] = partials[2 * i + 1];
    or_ge_add(&sum_r, &sum_r, &partials[2 * i]);
    or_ge_add(&sum_b, &sum_b, &partials[2 * i + 1]);
  }
  or_ge_sub(&cts[2 * (nr - 1)], &ct[0], &sum_r); /* :572 */
  or_ge_sub(&cts[2 * (nr - 1) + 1], &ct[1], &sum_b);
  int sizes[OR_MAX_RINGS];
  const ge *tables[OR_MAX_RINGS];
  for (int i = 0; i < nr; i++) { sizes[i] = (int)r->d.size[i]; tables[i] = r->table[i]; }
  return or_ring_verify(pk, nr, sizes, tables, cts, common_challenge, (size_t)r->total_size,
                        responses, &t);
}

uint32_t or_range_verify(const or_pubkey *pk, const or_prepared_range *r, const uint8_t *in,
                         const char *label) {
  size_t n_pts = 2 * (size_t)r->n_rings;
  size_t n_items = n_pts + 1 + (size_t)r->total_size;
  char kinds[2 * OR_MAX_RINGS + 1 + OR_MAX_RINGS * OR_MAX_RING_SIZE];
  memset(kinds, 'S', n_items);
  memset(kinds, 'P', n_pts);
  ge pts[2 * OR_MAX_RINGS];
  uint32_t st = scan_items(in, n_items, kinds, pts, NULL);
  if (st != OR_OK) return st;
  const uint8_t *proof = in + 32 * n_pts;
  return (uint32_t)range_verify_decoded(pk, r, pts, pts + 2, as_sc(proof), as_sc(proof + 32), label);
}

/* ------------------------------------------------------------------ SumOfSquaresProof::verify */

static int sumsq_verify_decoded(const or_pubkey *pk, int n, const ge *cts, const ge *sum_ct,
                                const sc *challenge, const sc *responses, const sc *sum_response,
                                const char *label) {
  /* mul.rs:190-260 (length check :197-202 is implied by the packed layout) */
  merlin_t t;
  or_merlin_init(&t, label);
  or_t_start_proof(&t, "sum_of_squares"); /* :96-99 */
  or_merlin_append(&t, "K", pk->bytes, 32);
  sc neg_c;
  or_sc_neg(&neg_c, challenge);
  ge g;
  or_ge_generator(&g);
  for (int i = 0; i < n; i++) { /* :207-230 */
    const ge *Rx = &cts[2 * i], *X = &cts[2 * i + 1];
    const sc *s_r = &responses[2 * i], *s_x = &responses[2 * i + 1];
    or_t_append_element(&t, "R_x", Rx);
    or_t_append_element(&t, "X", X);
    ge e_r, e_x;
    or_ge_double_mul_generator(&e_r, &neg_c, Rx, s_r);
    or_t_append_element(&t, "[e_r]G", &e_r);
    sc ks[3] = {*s_x, *s_r, neg_c};
    ge ps[3] = {g, pk->element, *X};
    or_ge_multi_mul(&e_x, 3, ks, ps);
    or_t_append_element(&t, "[e_x]G + [e_r]K", &e_x);
  }
  /* :232-247 */
  sc *ks = (sc *)malloc(sizeof(sc) * (size_t)(n + 2));
  ge *ps = (ge *)malloc(sizeof(ge) * (size_t)(n + 2));
  for (int i = 0; i < n; i++) ks[i] = responses[2 * i + 1];
  ks[n] = *sum_response;
  ks[n + 1] = neg_c;
  ge e_rz, e_z;
  for (int i = 0; i < n; i++) ps[i] = cts[2 * i];
  ps[n] = g;
  ps[n + 1] = sum_ct[0];
  or_ge_multi_mul(&e_rz, (size_t)(n + 2), ks, ps);
  for (int i = 0; i < n; i++) ps[i] = cts[2 * i + 1];
  ps[n] = pk->element;
  ps[n + 1] = sum_ct[1];
  or_ge_multi_mul(&e_z, (size_t)(n + 2), ks, ps);
  free(ks); free(ps);

  or_t_append_element(&t, "R_z", &sum_ct[0]); /* :249-253 */
  or_t_append_element(&t, "Z", &sum_ct[1]);
  or_t_append_element(&t, "[e_x]R_x + [e_z]G", &e_rz);
  or_t_append_element(&t, "[e_x]X + [e_z]K", &e_z);
  sc expected;
  or_t_challenge_scalar(&t, "c", &expected);
  return or_sc_eq(&expected, challenge);
}

uint32_t or_sumsq_verify(const or_pubkey *pk, int n, const uint8_t *cts_b, const uint8_t *sum_ct_b,
                         const uint8_t *proof, const char *label) {
  ge *cts = (ge *)malloc(sizeof(ge) * (size_t)(2 * n + 2));
  for (int i = 0; i < 2 * n; i++)
    if (!or_ristretto_decode(&cts[i], cts_b + 32 * i)) { free(cts); return OR_STATUS(OR_BAD_POINT, i); }
  for (int i = 0; i < 2; i++)
    if (!or_ristretto_decode(&cts[2 * n + i], sum_ct_b + 32 * i)) { free(cts); return OR_STATUS(OR_BAD_POINT, 2 * n + i); }
  for (int i = 0; i < 2 * n + 2; i++)
    if (!or_sc_is_canonical(proof + 32 * i)) { free(cts); return OR_STATUS(OR_BAD_SCALAR, 2 * n + 2 + i); }
  int ok = sumsq_verify_decoded(pk, n, cts, cts + 2 * n, as_sc(proof), as_sc(proof + 32),
                                as_sc(proof + 32 * (size_t)(1 + 2 * n)), label);
  free(cts);
  return ok ? OR_OK : OR_QV_CREDIT_EQUIV_CHALLENGE;
}

/* ------------------------------------------------------------------ QuadraticVotingBallot::verify */

void or_qv_params_init(or_qv_params *p, const uint8_t pk[32], int n_options, uint64_t credits) {
  /* quadratic_voting.rs:63-76 */
  or_pubkey_from_bytes(&p->pk, pk);
  p->n_options = n_options;
  p->credits = credits;
  range_decomp dv, dc;
  or_range_optimal(&dv, or_isqrt(credits) + 1);
  or_range_optimal(&dc, credits + 1);
  or_prepared_range_init(&p->vote_range, &dv);
  or_prepared_range_init(&p->credit_range, &dc);
}

size_t or_qv_ballot_size(const or_qv_params *p) {
  int n = p->n_options;
  return (size_t)n * (64 + or_range_proof_size(&p->vote_range)) + 64 +
         or_range_proof_size(&p->credit_range) + 32 * (size_t)(2 * n + 2);
}

uint32_t or_qv_verify(const or_qv_params *p, const uint8_t *ballot) {
  /* quadratic_voting.rs:291-329; deserialisation first (wire order), then the checks in order */
  int n = p->n_options;
  size_t vote_sz = 64 + or_range_proof_size(&p->vote_range);
  size_t credit_sz = 64 + or_range_proof_size(&p->credit_range);
  size_t total_items = or_qv_ballot_size(p) / 32;
  char *kinds = (char *)malloc(total_items);
  memset(kinds, 'S', total_items);
  for (int i = 0; i < n; i++) memset(kinds + (size_t)i * vote_sz / 32, 'P', 2 * (size_t)p->vote_range.n_rings);
  memset(kinds + (size_t)n * vote_sz / 32, 'P', 2 * (size_t)p->credit_range.n_rings);
  size_t n_vote_pts = 2 * (size_t)p->vote_range.n_rings, n_credit_pts = 2 * (size_t)p->credit_range.n_rings;
  ge *pts = (ge *)malloc(sizeof(ge) * ((size_t)n * n_vote_pts + n_credit_pts));
  uint32_t st = scan_items(ballot, total_items, kinds, pts, NULL);
  free(kinds);
  if (st != OR_OK) { free(pts); return st; }

  for (int i = 0; i < n; i++) { /* :297-307 */
    const uint8_t *v = ballot + (size_t)i * vote_sz;
    const ge *vp = pts + (size_t)i * n_vote_pts;
    const uint8_t *proof = v + 32 * n_vote_pts;
    int r = range_verify_decoded(&p->pk, &p->vote_range, vp, vp + 2, as_sc(proof), as_sc(proof + 32),
                                 "quadratic_voting_variant");
    if (r != OR_OK) {
      free(pts);
      return OR_STATUS(r == OR_RANGE_LEN ? OR_QV_VARIANT_LEN : OR_QV_VARIANT_CHALLENGE, i);
    }
  }
  const uint8_t *c = ballot + (size_t)n * vote_sz;
  const ge *cp = pts + (size_t)n * n_vote_pts;
  {
    const uint8_t *proof = c + 32 * n_credit_pts; /* :309-317 */
    int r = range_verify_decoded(&p->pk, &p->credit_range, cp, cp + 2, as_sc(proof), as_sc(proof + 32),
                                 "quadratic_voting_credit_range");
    if (r != OR_OK) {
      free(pts);
      return r == OR_RANGE_LEN ? OR_QV_CREDIT_RANGE_LEN : OR_QV_CREDIT_RANGE_CHALLENGE;
    }
  }
  /* :319-326 */
  ge *vcts = (ge *)malloc(sizeof(ge) * (size_t)(2 * n));
  for (int i = 0; i < n; i++) {
    vcts[2 * i] = pts[(size_t)i * n_vote_pts];
    vcts[2 * i + 1] = pts[(size_t)i * n_vote_pts + 1];
  }
  const uint8_t *sq = c + credit_sz;
  int ok = sumsq_verify_decoded(&p->pk, n, vcts, cp, as_sc(sq), as_sc(sq + 32),
                                as_sc(sq + 32 * (size_t)(1 + 2 * n)), "quadratic_voting_credit_equiv");
  free(vcts); free(pts);
  return ok ? OR_OK : OR_QV_CREDIT_EQUIV_CHALLENGE;
}

/* ------------------------------------------------------------------ PublicKey::verify_zero / verify_bool */

uint32_t or_verify_zero(const or_pubkey *pk, const uint8_t in[128]) {
  /* keys/impls.rs:59-69 */
  ge pts[2];
  for (int i = 0; i < 2; i++)
    if (!or_ristretto_decode(&pts[i], in + 32 * i)) return OR_STATUS(OR_BAD_POINT, i);
  for (int i = 2; i < 4; i++)
    if (!or_sc_is_canonical(in + 32 * i)) return OR_STATUS(OR_BAD_SCALAR, i);
  merlin_t t;
  or_merlin_init(&t, "zero_encryption");
  return or_logeq_verify(pk, &pts[0], &pts[1], as_sc(in + 64), as_sc(in + 96), &t) ? OR_OK : OR_SUM_CHALLENGE;
}

uint32_t or_verify_bool(const or_pubkey *pk, const uint8_t in[160]) {
  /* keys/impls.rs:100-112 */
  ge pts[2];
  for (int i = 0; i < 2; i++)
    if (!or_ristretto_decode(&pts[i], in + 32 * i)) return OR_STATUS(OR_BAD_POINT, i);
  for (int i = 2; i < 5; i++)
    if (!or_sc_is_canonical(in + 32 * i)) return OR_STATUS(OR_BAD_SCALAR, i);
  ge adm[2];
  or_ge_identity(&adm[0]);
  or_ge_generator(&adm[1]);
  int sizes[1] = {2};
  const ge *tables[1] = {adm};
  merlin_t t;
  or_merlin_init(&t, "bool_encryption");
  return (uint32_t)or_ring_verify(pk, 1, sizes, tables, pts, as_sc(in + 64), 2, as_sc(in + 96), &t);
}

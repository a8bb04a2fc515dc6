/* objects.c -- TEST INFRASTRUCTURE (part of the CPU oracle; never used by the product).
 * EncryptedChoice::verify and QuadraticVotingBallot::verify restated for ballot OBJECTS of arbitrary shape (any number of
 * choices / responses / partial ciphertexts), i.e. including the checks that a packed batch cannot exercise:
 *   deserialisation in document order (serde.rs:191-206,254-269; first invalid element wins),
 *   ChoiceParams::check_options_count (choice.rs:149-158), VerificationError::check_lengths (proofs/mod.rs:73-99) at
 *   ring.rs:310-315 ("items in all rings"), range.rs:555-559 ("admissible values"), mul.rs:197-202 ("ciphertext responses"),
 * in the order the reference performs them (choice.rs:358-380, quadratic_voting.rs:291-329).  Pins the *_LEN verdicts of the
 * product's object-ingest layer independently of the hand-written expectation table (tests/ingest_cases.py).
 *
 * Flat calling convention (ctypes): `items` = every 32-byte element of the object in document order; `shape` says how many. */
#include <stdlib.h>
#include <string.h>
#include "eg_oracle.h"

static const sc *as_sc(const uint8_t *p) { return (const sc *)p; }

/* document-order deserialisation: kinds[i] 'P' | 'S'; decoded points are written to pts in order */
static uint32_t scan(const uint8_t *items, size_t n_items, const char *kinds, ge *pts) {
  size_t np = 0;
  for (size_t i = 0; i < n_items; i++) {
    if (kinds[i] == 'P') { if (!or_ristretto_decode(&pts[np++], items + 32 * i)) return OR_STATUS(OR_BAD_POINT, i); }
    else if (!or_sc_is_canonical(items + 32 * i)) return OR_STATUS(OR_BAD_SCALAR, i);
  }
  return OR_OK;
}

/* EncryptedChoice { choices[n_choices], range_proof { common_challenge, ring_responses[n_responses] }, sum_proof? }
 * items = choices (R, B each) || common_challenge || ring_responses || [challenge, response] */
uint32_t or_choice_verify_object(const or_choice_params *p, int n_choices, int n_responses, const uint8_t *items) {
  const size_t n_items = 2 * (size_t)n_choices + 1 + (size_t)n_responses + (p->single ? 2 : 0);
  char *kinds = (char *)malloc(n_items + 1);
  ge *pts = (ge *)malloc(sizeof(ge) * (2 * (size_t)n_choices + 2));
  memset(kinds, 'S', n_items);
  memset(kinds, 'P', 2 * (size_t)n_choices);
  uint32_t st = scan(items, n_items, kinds, pts);
  free(kinds);
  if (st != OR_OK) { free(pts); return st; }
  if (n_choices != p->n_options) { free(pts); return OR_OPTIONS_LEN; }                 /* choice.rs:365 */
  const int n = p->n_options;
  const uint8_t *ring = items + 64 * (size_t)n, *sum = ring + 32 * (size_t)(1 + n_responses);
  if (p->single) {                                                                     /* choice.rs:363-366, 77-94 */
    ge sr = pts[0], sb = pts[1], g;
    for (int i = 1; i < n; i++) { or_ge_add(&sr, &sr, &pts[2 * i]); or_ge_add(&sb, &sb, &pts[2 * i + 1]); }
    or_ge_generator(&g);
    or_ge_sub(&sb, &sb, &g);
    merlin_t t;
    or_merlin_init(&t, "choice_encryption_sum");
    if (!or_logeq_verify(&p->pk, &sr, &sb, as_sc(sum), as_sc(sum + 32), &t)) { free(pts); return OR_SUM_CHALLENGE; }
  }
  ge adm[2];
  or_ge_identity(&adm[0]);
  or_ge_generator(&adm[1]);
  int *sizes = (int *)malloc(sizeof(int) * (size_t)n);
  const ge **tables = (const ge **)malloc(sizeof(ge *) * (size_t)n);
  for (int i = 0; i < n; i++) { sizes[i] = 2; tables[i] = adm; }
  merlin_t t;
  or_merlin_init(&t, "encrypted_choice_ranges");
  const int r = or_ring_verify(&p->pk, n, sizes, tables, pts, as_sc(ring), (size_t)n_responses, as_sc(ring + 32), &t);
  free(sizes); free(tables); free(pts);
  return (uint32_t)r;                                                                  /* OR_RANGE_LEN | OR_RANGE_CHALLENGE | OR_OK */
}

/* RangeProof::verify on an object block (range.rs:547-577): pts = ciphertext (2) then partials (2 each) */
static int range_object(const or_pubkey *pk, const or_prepared_range *r, const ge *pts, int n_partials, const sc *challenge,
                        int n_responses, const sc *responses, const char *label) {
  if (n_partials + 1 != r->n_rings) return OR_RANGE_LEN;                                /* :555-559 "admissible values" */
  merlin_t t;
  or_merlin_init(&t, label);
  or_t_start_proof(&t, "encryption_range_proof");
  or_merlin_append(&t, "range", (const uint8_t *)r->name, (size_t)r->name_len);
  const int nr = r->n_rings;
  ge cts[2 * OR_MAX_RINGS], sum_r, sum_b;
  or_ge_identity(&sum_r);
  or_ge_identity(&sum_b);
  for (int i = 0; i < nr - 1; i++) {
    cts[2 * i] = pts[2 + 2 * i];
    cts[2 * i + 1] = pts[2 + 2 * i + 1];
    or_ge_add(&sum_r, &sum_r, &pts[2 + 2 * i]);
    or_ge_add(&sum_b, &sum_b, &pts[2 + 2 * i + 1]);
  }
  or_ge_sub(&cts[2 * (nr - 1)], &pts[0], &sum_r);
  or_ge_sub(&cts[2 * (nr - 1) + 1], &pts[1], &sum_b);
  int sizes[OR_MAX_RINGS];
  const ge *tables[OR_MAX_RINGS];
  for (int i = 0; i < nr; i++) { sizes[i] = (int)r->d.size[i]; tables[i] = r->table[i]; }
  return or_ring_verify(pk, nr, sizes, tables, cts, challenge, (size_t)n_responses, responses, &t);   /* ring.rs:310-315 inside */
}

/* QuadraticVotingBallot { votes[n_votes] { ciphertext, range_proof { partial_ciphertexts[], common_challenge, ring_responses[] } },
 *                         credit { same }, credit_equivalence_proof { challenge, ciphertext_responses[], sum_response } }
 * shape = [n_votes, (n_partials, n_responses) x n_votes, n_partials_credit, n_responses_credit, n_ciphertext_responses] */
uint32_t or_qv_verify_object(const or_qv_params *p, const int *shape, const uint8_t *items) {
  const int n_votes = shape[0];
  const int *blk = shape + 1;
  const int n_sq = shape[1 + 2 * (n_votes + 1)];
  size_t n_items = 0, n_pts = 0;
  for (int b = 0; b <= n_votes; b++) { n_items += 2 + 2 * (size_t)blk[2 * b] + 1 + (size_t)blk[2 * b + 1]; n_pts += 2 + 2 * (size_t)blk[2 * b]; }
  n_items += 2 + (size_t)n_sq;
  char *kinds = (char *)malloc(n_items + 1);
  ge *pts = (ge *)malloc(sizeof(ge) * (n_pts + 2));
  size_t it = 0;
  for (int b = 0; b <= n_votes; b++) {
    const size_t np = 2 + 2 * (size_t)blk[2 * b];
    memset(kinds + it, 'P', np); it += np;
    memset(kinds + it, 'S', 1 + (size_t)blk[2 * b + 1]); it += 1 + (size_t)blk[2 * b + 1];
  }
  memset(kinds + it, 'S', 2 + (size_t)n_sq);
  uint32_t st = scan(items, n_items, kinds, pts);
  free(kinds);
  if (st != OR_OK) { free(pts); return st; }
  if (n_votes != p->n_options) { free(pts); return OR_OPTIONS_LEN; }                   /* quadratic_voting.rs:295 */
  size_t item_off = 0, pt_off = 0;
  ge *vote_cts = (ge *)malloc(sizeof(ge) * (2 * (size_t)n_votes + 2));
  uint32_t out = OR_OK;
  for (int b = 0; b <= n_votes && out == OR_OK; b++) {                                 /* :297-317 */
    const int n_partials = blk[2 * b], n_resp = blk[2 * b + 1];
    const size_t np = 2 + 2 * (size_t)n_partials;
    const uint8_t *scalars = items + 32 * (item_off + np);
    const int vote = b < n_votes;
    const int r = range_object(&p->pk, vote ? &p->vote_range : &p->credit_range, pts + pt_off, n_partials, as_sc(scalars), n_resp,
                               as_sc(scalars + 32), vote ? "quadratic_voting_variant" : "quadratic_voting_credit_range");
    vote_cts[2 * b] = pts[pt_off];
    vote_cts[2 * b + 1] = pts[pt_off + 1];
    if (r == OR_RANGE_LEN) out = vote ? OR_STATUS(OR_QV_VARIANT_LEN, b) : OR_QV_CREDIT_RANGE_LEN;
    else if (r != OR_OK) out = vote ? OR_STATUS(OR_QV_VARIANT_CHALLENGE, b) : OR_QV_CREDIT_RANGE_CHALLENGE;
    item_off += np + 1 + (size_t)n_resp;
    pt_off += np;
  }
  if (out == OR_OK) {                                                                  /* :319-326, mul.rs:197-202 */
    const uint8_t *sq = items + 32 * item_off;
    if (n_sq != 2 * n_votes) out = OR_QV_CREDIT_EQUIV_LEN;
    else {
      /* SumOfSquaresProof::verify through the packed entry: ciphertexts re-encoded (canonical, so identical bytes) */
      uint8_t *enc = (uint8_t *)malloc(64 * ((size_t)n_votes + 1));
      for (int i = 0; i < 2 * (n_votes + 1); i++) or_ristretto_encode(enc + 32 * (size_t)i, &vote_cts[i]);
      const uint32_t r = or_sumsq_verify(&p->pk, n_votes, enc, enc + 64 * (size_t)n_votes, sq, "quadratic_voting_credit_equiv");
      free(enc);
      out = r == OR_OK ? OR_OK : OR_QV_CREDIT_EQUIV_CHALLENGE;
    }
  }
  free(pts); free(vote_cts);
  return out;
}

/*
 * range.c -- CPU ORACLE (test infrastructure, not the product): RangeDecomposition and isqrt.
 *
 * Follows src/proofs/range.rs:148-305 (optimal / optimize / lower_len_estimate, std flavour with the
 * f64 log2 bound), :110-124 (Display -- the string is hashed into the transcript at :562),
 * :179-185 (upper_bound), :199-210 (decompose) and src/app/quadratic_voting.rs:127-143 (isqrt).
 * Pinned by the known-answer decompositions in range.rs:592-706 (tests/test_oracle_golden.py).
 */
#include "eg_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  uint64_t key;
  uint64_t len;
  range_decomp d;
} memo_entry;

typedef struct {
  memo_entry *e;
  size_t n, cap;
} memo_t;

static memo_entry *memo_get(memo_t *m, uint64_t key) {
  for (size_t i = 0; i < m->n; i++)
    if (m->e[i].key == key) return &m->e[i];
  return NULL;
}

static void memo_put(memo_t *m, const memo_entry *e) {
  if (m->n == m->cap) {
    m->cap = m->cap ? 2 * m->cap : 64;
    m->e = (memo_entry *)realloc(m->e, m->cap * sizeof(memo_entry));
  }
  m->e[m->n++] = *e;
}

static uint64_t lower_len_estimate(uint64_t ub) { /* range.rs:302-305 */
  return (uint64_t)ceil(log2((double)ub) * 3.0);
}

static memo_entry optimize(uint64_t ub, memo_t *m) { /* range.rs:238-300 */
  memo_entry *hit = memo_get(m, ub);
  if (hit) return *hit;
  memo_entry opt;
  memset(&opt, 0, sizeof opt);
  opt.key = ub;
  opt.len = ub + 2;
  opt.d.n_rings = 1;
  opt.d.size[0] = ub;
  opt.d.step[0] = 1;
  for (uint64_t first = 2;; first++) {
    if (first + 2 > opt.len) break;
    uint64_t remaining = ub - first;
    for (uint64_t mult = 2; mult <= first; mult++) {
      if (remaining % mult != 0) continue;
      uint64_t inner_ub = remaining / mult + 1;
      if (inner_ub < 2) break;
      uint64_t best = first + 2 + lower_len_estimate(inner_ub);
      if (best > opt.len) continue;
      memo_entry inner = optimize(inner_ub, m);
      uint64_t cand_len = first + 2 + inner.len;
      int cand_rings = 1 + inner.d.n_rings;
      if (cand_len < opt.len || (cand_len == opt.len && cand_rings < opt.d.n_rings)) {
        if (cand_rings > OR_MAX_RINGS) continue; /* beyond this oracle's static capacity */
        opt.len = cand_len;
        opt.d = inner.d; /* combine_mul, range.rs:163-177 */
        for (int i = 0; i < opt.d.n_rings; i++) opt.d.step[i] *= mult;
        opt.d.size[opt.d.n_rings] = first;
        opt.d.step[opt.d.n_rings] = 1;
        opt.d.n_rings++;
      }
    }
  }
  memo_put(m, &opt);
  return opt;
}

void or_range_optimal(range_decomp *d, uint64_t upper_bound) {
  memo_t m = {0};
  memo_entry e = optimize(upper_bound, &m);
  *d = e.d;
  free(m.e);
}

int or_range_to_string(const range_decomp *d, char *buf, size_t cap) {
  size_t off = 0;
  for (int i = 0; i < d->n_rings; i++) {
    if (d->step[i] > 1) off += (size_t)snprintf(buf + off, cap - off, "%llu * ", (unsigned long long)d->step[i]);
    off += (size_t)snprintf(buf + off, cap - off, "0..%llu", (unsigned long long)d->size[i]);
    if (i + 1 < d->n_rings) off += (size_t)snprintf(buf + off, cap - off, " + ");
  }
  return (int)off;
}

uint64_t or_range_upper_bound(const range_decomp *d) {
  uint64_t s = 0;
  for (int i = 0; i < d->n_rings; i++) s += (d->size[i] - 1) * d->step[i];
  return s + 1;
}

void or_range_decompose(const range_decomp *d, uint64_t value, int *idx) {
  for (int i = 0; i < d->n_rings; i++) {
    uint64_t vi = value / d->step[i];
    if (vi > d->size[i] - 1) vi = d->size[i] - 1;
    idx[i] = (int)vi;
    value -= vi * d->step[i];
  }
}

uint64_t or_isqrt(uint64_t x) {
  uint64_t root = 0, p4 = 1ULL << 62;
  while (p4 > x) p4 /= 4;
  while (p4 > 0) {
    if (x >= root + p4) {
      x -= root + p4;
      root = root / 2 + p4;
    } else {
      root /= 2;
    }
    p4 /= 4;
  }
  return root;
}

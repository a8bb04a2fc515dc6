// hostcheck.cpp -- TEST-ONLY host build of the device arithmetic headers with -DEG_BOUNDCHECK.
// Every fe carries its limb class and each operation asserts its precondition, so running the
// code paths below proves the bound discipline of fe25519.cuh/ge25519.cuh; results are compared with
// the oracle by tests/test_hostcheck.py.  This library is never loaded by the product.
#include <string.h>
#include <map>
#include <vector>
#include "../../elastic_elgamal_amd/csrc/ge25519.cuh"
#include "../../elastic_elgamal_amd/csrc/sc25519.cuh"
#include "../../elastic_elgamal_amd/csrc/merlin.cuh"

using namespace eg;

struct ArrTable {
  ge_cached e[8];
  void store(int i, const ge_cached& c) { e[i] = c; }
  void load(ge_cached& c, int i) const { c = e[i]; }
};
// the comb table of a base, stored the way the device stores it (device_io.cuh BaseTable): packed, 4 x 256 bits per entry
struct ArrBase {
  u32 w[64][32];          // up to 2^6 entries (7 teeth)
  void store(int i, const ge_cached& c) {
    fe a = c.YpX, b = c.YmX, z = c.Z2;
    fe_carry(a); fe_carry(b); fe_carry(z);
    fe_pack8(w[i], a); fe_pack8(w[i] + 8, b); fe_pack8(w[i] + 16, z); fe_pack8(w[i] + 24, c.T2d);
  }
  void load(ge_cached& c, int i) const {
    fe_unpack8(c.YpX, w[i]); fe_unpack8(c.YmX, w[i] + 8); fe_unpack8(c.Z2, w[i] + 16); fe_unpack8(c.T2d, w[i] + 24);
  }
};
// fixed-base comb table with entries computed on demand (the product's table has millions of entries; a comb touches one per window)
struct ArrNiels {
  ge base;
  int bits = EG_COMB_BITS;
  bool ready = false;
  mutable std::map<int, ge_niels> cache;
  void load(ge_niels& c, int idx) const {
    auto it = cache.find(idx);
    if (it == cache.end()) {
      const unsigned long long m0 = g_fe_mul_count, s0 = g_fe_sq_count;     // table construction is not the comb's work
      const int w = idx / comb_entries(bits), k = idx % comb_entries(bits) + 1;    // entry = [k * 2^(B w)] base
      ge p = base;
      for (int i = 0; i < bits * w; ++i) { ge d; ge_dbl_full(d, p); p = d; }
      ge q; ge_identity(q);
      for (int bit = bits - 1; bit >= 0; --bit) {
        ge d; ge_dbl_full(d, q); q = d;
        if ((k >> bit) & 1) { ge t; ge_add_full(t, q, p); q = t; }
      }
      ge_niels n; ge_to_niels(n, q);
      it = cache.emplace(idx, n).first;
      g_fe_mul_count = m0; g_fe_sq_count = s0;
    }
    c = it->second;
  }
};
struct ArrState {
  u32 w[50];
  u32 rd(int i) const { return w[i]; }
  void wr(int i, u32 v) { w[i] = v; }
};

static void words_from_bytes(u32* w, const uint8_t* b, int nwords) {
  for (int i = 0; i < nwords; ++i) w[i] = (u32)b[4 * i] | ((u32)b[4 * i + 1] << 8) | ((u32)b[4 * i + 2] << 16) | ((u32)b[4 * i + 3] << 24);
}
static void bytes_from_words(uint8_t* b, const u32* w, int nwords) {
  for (int i = 0; i < nwords; ++i) for (int j = 0; j < 4; ++j) b[4 * i + j] = (uint8_t)(w[i] >> (8 * j));
}

static ArrNiels g_base_table;
static void build_fixed(ArrNiels& t, const ge& base) { t.base = base; t.ready = true; }

static int g_teeth = 6;      // comb shape the table-backed checks run with (hc_set_teeth): every shape the product instantiates, and 7 x 37
extern "C" void hc_set_teeth(int t) { g_teeth = t; }

extern "C" int hc_point_roundtrip(const uint8_t in[32], uint8_t out[32]) {
  u32 w[8], o[8]; words_from_bytes(w, in, 8);
  ge p; bool ok = ristretto_decode(p, w);
  ristretto_encode(o, p);
  bytes_from_words(out, o, 8);
  return ok ? 1 : 0;
}

// the PREPARED form of a point (kernels.cuh: k_prim_points_prepare / prepared_load): decode, carry the three affine coordinates to class 1,
// pack them to 3 x 256 bits, unpack, Z = 1, encode again - the limb-class preconditions of fe_pack8 / fe_unpack8 are asserted on the way
extern "C" int hc_prepared_roundtrip(const uint8_t in[32], uint8_t out[32], uint8_t prepared[96]) {
  u32 w[8], o[8]; words_from_bytes(w, in, 8);
  ge p; const bool ok = ristretto_decode(p, w);
  fe_carry(p.X); fe_carry(p.Y); fe_carry(p.T);
  u32 pw[24];
  fe_pack8(pw, p.X); fe_pack8(pw + 8, p.Y); fe_pack8(pw + 16, p.T);
  bytes_from_words(prepared, pw, 24);
  ge q;
  fe_unpack8(q.X, pw); fe_unpack8(q.Y, pw + 8); fe_unpack8(q.T, pw + 16); fe_1(q.Z);
  ristretto_encode(o, q);
  bytes_from_words(out, o, 8);
  return ok ? 1 : 0;
}

// out = enc([k]P + [r]G)   (Group::vartime_double_mul_generator)
extern "C" int hc_double_mul_generator(const uint8_t k[32], const uint8_t p_enc[32], const uint8_t r[32], uint8_t out[32]) {
  if (!g_base_table.ready) { ge g; ge_generator(g); build_fixed(g_base_table, g); }
  u32 kw[8], rw[8], pw[8], o[8];
  words_from_bytes(kw, k, 8); words_from_bytes(rw, r, 8); words_from_bytes(pw, p_enc, 8);
  ge p; if (!ristretto_decode(p, pw)) return 0;
  ArrTable tab; ge_var_table_build(tab, p);
  u32 dk[8], dr[EG_COMB_WORDS]; sc_recode_radix16(dk, kw); sc_recode_comb(dr, rw);
  ge acc; ge_var_mul(acc, tab, dk);
  ge_fixed_mul_add(acc, g_base_table, dr);
  ristretto_encode(o, acc);
  bytes_from_words(out, o, 8);
  return 1;
}

// same as hc_double_mul_generator but through the per-base comb table (ge_teeth_tables_build / ge_teeth_mul)
template <int T>
static int hc_double_mul_generator_teeth_t(const uint8_t k[32], const uint8_t p_enc[32], const uint8_t r[32], uint8_t out[32]) {
  if (!g_base_table.ready) { ge g; ge_generator(g); build_fixed(g_base_table, g); }
  u32 kw[8], rw[8], pw[8], o[8];
  words_from_bytes(kw, k, 8); words_from_bytes(rw, r, 8); words_from_bytes(pw, p_enc, 8);
  ge p; if (!ristretto_decode(p, pw)) return 0;
  ArrBase tab; ArrTable tmp; ge_teeth_tables_build<T>(tab, tmp, p);
  u64 rows[T]; u32 dr[EG_COMB_WORDS]; sc_recode_teeth<T>(rows, kw); sc_recode_comb(dr, rw);
  ge acc; ge_teeth_mul<T>(acc, tab, rows);
  ge_fixed_mul_add(acc, g_base_table, dr);
  ristretto_encode(o, acc);
  bytes_from_words(out, o, 8);
  return 1;
}

// out = enc(sum_i [k_i]P_i + [r]G) with every base behind a teeth table and ONE shared doubling chain (ge_teeth_mul_multi, what
// k_eq_table<true> runs); the sign vectors are read word by word, as the kernel reads them from LDS
template <int T>
static int hc_multi_mul_teeth_t(int n, const uint8_t* ks, const uint8_t* ps, const uint8_t r[32], uint8_t out[32]) {
  if (!g_base_table.ready) { ge g; ge_generator(g); build_fixed(g_base_table, g); }
  std::vector<ArrBase> tabs(n);
  std::vector<u32> sg(9 * (size_t)n);
  for (int i = 0; i < n; ++i) {
    u32 kw[8], pw[8];
    words_from_bytes(kw, ks + 32 * i, 8); words_from_bytes(pw, ps + 32 * i, 8);
    ge p; if (!ristretto_decode(p, pw)) return 0;
    ArrTable tmp; ge_teeth_tables_build<T>(tabs[i], tmp, p);
    sc_teeth_signs<T>(&sg[9 * i], kw);
  }
  ge acc;
  ge_teeth_mul_multi<T>(acc, n,
      [&](int t, int c, int& idx, bool& neg) { sc_teeth_column<T>([&](int w) { return sg[9 * t + w]; }, c, idx, neg); },
      [&](int t, int idx, ge_cached& e) { tabs[t].load(e, idx); });
  u32 rw[8], dr[EG_COMB_WORDS], o[8];
  words_from_bytes(rw, r, 8); sc_recode_comb(dr, rw);
  ge_fixed_mul_add(acc, g_base_table, dr);
  ristretto_encode(o, acc);
  bytes_from_words(out, o, 8);
  return 1;
}

// out = enc([k](P_1 + .. + P_n) + [r]G) through the table that ge_teeth_tables_sum makes from the members' tables (what k_sum_tables +
// k_eq_table<false> run for the log-equality proof over the sum of the ciphertexts).  Returns 1 when, in addition, every entry of
// that table is the same curve point as the entry of a table built from the sum itself; 0 when one differs; -1 for a bad encoding.
template <int T>
static int hc_sum_table_mul_t(int n, const uint8_t* ps, const uint8_t k[32], const uint8_t r[32], uint8_t out[32]) {
  if (!g_base_table.ready) { ge g; ge_generator(g); build_fixed(g_base_table, g); }
  std::vector<ArrBase> tabs(n);
  ge total; ge_identity(total);
  for (int i = 0; i < n; ++i) {
    u32 pw[8]; words_from_bytes(pw, ps + 32 * i, 8);
    ge p; if (!ristretto_decode(p, pw)) return -1;
    ArrTable tmp; ge_teeth_tables_build<T>(tabs[i], tmp, p);
    ge t; ge_add_full(t, total, p); total = t;
  }
  ArrBase sum_tab, ref_tab; ArrTable tmp, tmp2;
  ge_teeth_tables_sum<T>(sum_tab, tmp, n, [&](int t, int g, ge_cached& e) { tabs[t].load(e, g); });
  ge_teeth_tables_build<T>(ref_tab, tmp2, total);
  int same = 1;
  for (int g = 0; g < Teeth<T>::ENTRIES; ++g) {
    ge_cached ea, eb; sum_tab.load(ea, g); ref_tab.load(eb, g);
    ge a, b; ge_cached_to_p3(a, ea); ge_cached_to_p3(b, eb);
    fe az, bz, l, rr;
    az = a.Z; fe_carry(az); bz = b.Z; fe_carry(bz);
    fe_mul(l, a.X, bz); fe_mul(rr, b.X, az); if (!fe_eq(l, rr)) same = 0;
    fe_mul(l, a.Y, bz); fe_mul(rr, b.Y, az); if (!fe_eq(l, rr)) same = 0;
    fe_mul(l, a.T, bz); fe_mul(rr, b.T, az); if (!fe_eq(l, rr)) same = 0;
  }
  u32 kw[8], rw[8], o[8];
  words_from_bytes(kw, k, 8); words_from_bytes(rw, r, 8);
  u64 rows[T]; u32 dr[EG_COMB_WORDS]; sc_recode_teeth<T>(rows, kw); sc_recode_comb(dr, rw);
  ge acc; ge_teeth_mul<T>(acc, sum_tab, rows);
  ge_fixed_mul_add(acc, g_base_table, dr);
  ristretto_encode(o, acc);
  bytes_from_words(out, o, 8);
  return same;
}

// encode(2P) through the batched-inversion path vs the plain encoder; returns 1 when they agree
extern "C" int hc_double_encode(const uint8_t p_enc[32], uint8_t out[32]) {
  u32 pw[8], o[8], ref[8]; words_from_bytes(pw, p_enc, 8);
  ge p; if (!ristretto_decode(p, pw)) return -1;
  fe n; bool zero;
  ge_double_encode_prepare(n, zero, p);
  fe inv; fe_invert(inv, n);
  ge_double_encode_finish(o, p, inv, zero);
  ge q; ge_dbl_full(q, p);
  ristretto_encode(ref, q);
  bytes_from_words(out, o, 8);
  return memcmp(o, ref, 32) == 0 ? 1 : 0;
}
// [k]P + [r]G evaluated as 2 * ([k/2]P + [r/2]G) with the doubled encoder (what k_msm_jobs + k_encode_batch do)
template <int T>
static int hc_double_mul_generator_halved_t(const uint8_t k[32], const uint8_t p_enc[32], const uint8_t r[32], uint8_t out[32]) {
  if (!g_base_table.ready) { ge g; ge_generator(g); build_fixed(g_base_table, g); }
  u32 kw[8], rw[8], pw[8], o[8], kh[8], rh[8];
  words_from_bytes(kw, k, 8); words_from_bytes(rw, r, 8); words_from_bytes(pw, p_enc, 8);
  ge p; if (!ristretto_decode(p, pw)) return 0;
  sc_halve(kh, kw); sc_halve(rh, rw);
  ArrBase tab; ArrTable tmp; ge_teeth_tables_build<T>(tab, tmp, p);
  u64 rows[T]; u32 dr[EG_COMB_WORDS]; sc_recode_teeth<T>(rows, kh); sc_recode_comb(dr, rh);
  ge acc; ge_teeth_mul<T>(acc, tab, rows);
  ge_fixed_mul_add(acc, g_base_table, dr);
  fe n, inv; bool zero;
  ge_double_encode_prepare(n, zero, acc);
  fe_invert(inv, n);
  ge_double_encode_finish(o, acc, inv, zero);
  bytes_from_words(out, o, 8);
  return 1;
}

extern "C" int hc_point_add(const uint8_t a[32], const uint8_t b[32], int sub, uint8_t out[32]) {
  u32 aw[8], bw[8], o[8]; words_from_bytes(aw, a, 8); words_from_bytes(bw, b, 8);
  ge p, q, r; if (!ristretto_decode(p, aw) || !ristretto_decode(q, bw)) return 0;
  if (sub) ge_sub_full(r, p, q); else ge_add_full(r, p, q);
  ge d; ge_dbl_full(d, r); (void)d;
  ristretto_encode(o, r); bytes_from_words(out, o, 8);
  return 1;
}

extern "C" void hc_sc_from_wide(const uint8_t in[64], uint8_t out[32]) {
  u32 w[16], o[8]; words_from_bytes(w, in, 16); sc_from_wide(o, w); bytes_from_words(out, o, 8);
}
extern "C" void hc_sc_muladd(const uint8_t a[32], const uint8_t b[32], const uint8_t c[32], uint8_t out[32]) {
  u32 aw[8], bw[8], cw[8], o[8]; words_from_bytes(aw, a, 8); words_from_bytes(bw, b, 8); words_from_bytes(cw, c, 8);
  sc_muladd(o, aw, bw, cw); bytes_from_words(out, o, 8);
}
extern "C" void hc_sc_neg(const uint8_t a[32], uint8_t out[32]) {
  u32 aw[8], o[8]; words_from_bytes(aw, a, 8); sc_neg(o, aw); bytes_from_words(out, o, 8);
}
extern "C" void hc_sc_invert(const uint8_t a[32], uint8_t out[32]) {
  u32 aw[8], o[8]; words_from_bytes(aw, a, 8); sc_invert(o, aw); bytes_from_words(out, o, 8);
}
extern "C" int hc_sc_is_canonical(const uint8_t a[32]) { u32 aw[8]; words_from_bytes(aw, a, 8); return sc_is_canonical(aw) ? 1 : 0; }

// transcript: new(label); append(l1, m1); append_u64(l2, x); challenge(l3) -> 64 bytes; also exports pos
extern "C" int hc_merlin(const char* label, const char* l1, const uint8_t* m1, int m1_len, const char* l2, uint64_t x,
              const char* l3, uint8_t out[64]) {
  Transcript<ArrState> t;
  merlin_init(t, label, (int)strlen(label));
  std::vector<u32> mw((m1_len + 3) / 4 + 1, 0);
  std::vector<uint8_t> pad(mw.size() * 4, 0); memcpy(pad.data(), m1, m1_len);
  words_from_bytes(mw.data(), pad.data(), (int)mw.size());
  merlin_append_words(t, l1, (int)strlen(l1), mw.data(), m1_len);
  if (l2) merlin_append_u64(t, l2, (int)strlen(l2), x);
  Transcript<ArrState> c; merlin_clone(c, t);
  u32 o[16]; merlin_challenge64(c, l3, (int)strlen(l3), o);
  bytes_from_words(out, o, 16);
  return (int)t.pos;
}

// field operation counts of the hot-path building blocks: out[2*i], out[2*i+1] = (fe_mul, fe_sq) calls of
// 0: ristretto_decode  1: direct table build  2: direct variable-base multiply  3: fixed-base comb
// 4: ristretto_encode  5: comb-table build (per base)  6: comb multiply (per equation)
// 9: shared-chain product of ONE term (ge_teeth_mul_multi)   10: every further term of it
// 11: table of a sum base with ONE member (ge_teeth_tables_sum)   12: every further member of it
// 13: fixed-base comb over a wide (EG_COMB_BITS_BIG) table
template <int T>
static void hc_op_counts_t(unsigned long long out[28]) {
  if (!g_base_table.ready) { ge g; ge_generator(g); build_fixed(g_base_table, g); }
  u32 gw[8] = {0x0aaef2e2u, 0x714ebc6au, 0x61a984a8u, 0x5f5100c5u, 0x6a0be358u, 0x8ddd82a5u, 0x4559a6b6u, 0x762d8de0u};
  u32 k[8] = {0x12345678u, 0x9abcdef0u, 0x0fedcba9u, 0x87654321u, 0x11111111u, 0x22222222u, 0x33333333u, 0x04444444u};
  auto snap = [&](int i, unsigned long long m0, unsigned long long s0) { out[2 * i] = g_fe_mul_count - m0; out[2 * i + 1] = g_fe_sq_count - s0; };
  unsigned long long m0 = g_fe_mul_count, s0 = g_fe_sq_count;
  ge p; ristretto_decode(p, gw); snap(0, m0, s0);
  m0 = g_fe_mul_count; s0 = g_fe_sq_count;
  ArrTable tab; ge_var_table_build(tab, p); snap(1, m0, s0);
  u32 dg[8]; sc_recode_radix16(dg, k);
  m0 = g_fe_mul_count; s0 = g_fe_sq_count;
  ge acc; ge_var_mul(acc, tab, dg); snap(2, m0, s0);
  u32 dg8[EG_COMB_WORDS]; sc_recode_comb(dg8, k);
  m0 = g_fe_mul_count; s0 = g_fe_sq_count;
  ge_fixed_mul_add(acc, g_base_table, dg8); snap(3, m0, s0);
  m0 = g_fe_mul_count; s0 = g_fe_sq_count;
  u32 o[8]; ristretto_encode(o, acc); snap(4, m0, s0);
  m0 = g_fe_mul_count; s0 = g_fe_sq_count;
  ArrBase st; ArrTable tmp; ge_teeth_tables_build<T>(st, tmp, p); snap(5, m0, s0);
  u64 rows[T]; sc_recode_teeth<T>(rows, k);
  m0 = g_fe_mul_count; s0 = g_fe_sq_count;
  ge_teeth_mul<T>(acc, st, rows); snap(6, m0, s0);
  // 7: doubled encoder per commitment (prepare + prefix/backward products + finish)   8: the shared field inversion
  m0 = g_fe_mul_count; s0 = g_fe_sq_count;
  fe n, inv, t1, t2; bool zero;
  ge_double_encode_prepare(n, zero, acc);
  fe_mul(t1, n, n); fe_mul(t2, n, n); fe_mul(t1, t1, t2);     // the three bookkeeping multiplications of the batch
  snap(7, m0, s0);
  unsigned long long m1 = g_fe_mul_count, s1 = g_fe_sq_count;
  fe_invert(inv, n);
  out[16] = g_fe_mul_count - m1; out[17] = g_fe_sq_count - s1;
  m0 = g_fe_mul_count; s0 = g_fe_sq_count;
  ge_double_encode_finish(o, acc, inv, zero);
  out[14] += g_fe_mul_count - m0; out[15] += g_fe_sq_count - s0;
  u32 sg[9]; sc_teeth_signs<T>(sg, k);
  auto column = [&](int, int c, int& idx, bool& neg) { sc_teeth_column<T>([&](int w) { return sg[w]; }, c, idx, neg); };
  auto load = [&](int, int idx, ge_cached& e) { st.load(e, idx); };
  m0 = g_fe_mul_count; s0 = g_fe_sq_count;
  ge_teeth_mul_multi<T>(acc, 1, column, load); snap(9, m0, s0);
  m0 = g_fe_mul_count; s0 = g_fe_sq_count;
  ge_teeth_mul_multi<T>(acc, 2, column, load);
  out[20] = g_fe_mul_count - m0 - out[18]; out[21] = g_fe_sq_count - s0 - out[19];
  ArrBase sum_tab;
  auto src = [&](int, int g, ge_cached& e) { st.load(e, g); };
  m0 = g_fe_mul_count; s0 = g_fe_sq_count;
  ge_teeth_tables_sum<T>(sum_tab, tmp, 1, src); snap(11, m0, s0);
  m0 = g_fe_mul_count; s0 = g_fe_sq_count;
  ge_teeth_tables_sum<T>(sum_tab, tmp, 2, src);
  out[24] = g_fe_mul_count - m0 - out[22]; out[25] = g_fe_sq_count - s0 - out[23];
  static ArrNiels wide;
  if (!wide.ready) { ge g; ge_generator(g); wide.bits = EG_COMB_BITS_BIG; build_fixed(wide, g); }
  m0 = g_fe_mul_count; s0 = g_fe_sq_count;
  ge_fixed_mul_add(acc, wide, dg8); snap(13, m0, s0);
}
// enc([r]G) through a comb of the given window width (the wide tables of large batches use EG_COMB_BITS_BIG)
extern "C" void hc_mul_generator_bits(int bits, const uint8_t r[32], uint8_t out[32]) {
  static std::map<int, ArrNiels> tabs;
  ArrNiels& t = tabs[bits];
  if (!t.ready) { ge g; ge_generator(g); t.bits = bits; build_fixed(t, g); }
  u32 rw[8], dr[EG_COMB_WORDS], o[8];
  words_from_bytes(rw, r, 8); sc_recode_comb(dr, rw);
  ge acc; ge_identity(acc);
  ge_fixed_mul_add(acc, t, dr);
  ristretto_encode(o, acc);
  bytes_from_words(out, o, 8);
}

// pack / unpack of a field element given as EG_NL raw limbs (any class-1 representation, not only the canonical one): returns 1 when
// the unpacked element equals the input as a field element; out = its canonical bytes
extern "C" int hc_fe_pack_roundtrip(const uint32_t limbs[EG_NL], uint8_t out[32]) {
  fe f; for (int i = 0; i < EG_NL; ++i) f.v[i] = limbs[i];
  EG_SETCLS(f, 1.0f);
  u32 w[8], o[8]; fe_pack8(w, f);
  fe g; fe_unpack8(g, w);
  fe_to_words(o, g); bytes_from_words(out, o, 8);
  return fe_eq(f, g) ? 1 : 0;
}
extern "C" void hc_fe_roundtrip(const uint8_t in[32], uint8_t out[32]) {
  u32 w[8], o[8]; words_from_bytes(w, in, 8); fe f; fe_from_words(f, w); fe_to_words(o, f); bytes_from_words(out, o, 8);
}
// (a*b, a^2, 1/a, a+b, a-b) canonical bytes
extern "C" void hc_fe_ops(const uint8_t a[32], const uint8_t b[32], uint8_t out[160]) {
  u32 aw[8], bw[8], o[8]; words_from_bytes(aw, a, 8); words_from_bytes(bw, b, 8);
  fe x, y, r; fe_from_words(x, aw); fe_from_words(y, bw);
  fe_mul(r, x, y); fe_to_words(o, r); bytes_from_words(out, o, 8);
  fe_sq(r, x); fe_to_words(o, r); bytes_from_words(out + 32, o, 8);
  fe_invert(r, x); fe_to_words(o, r); bytes_from_words(out + 64, o, 8);
  fe_add(r, x, y); fe_to_words(o, r); bytes_from_words(out + 96, o, 8);
  fe_sub(r, x, y); fe_to_words(o, r); bytes_from_words(out + 128, o, 8);
}
extern "C" int hc_double_mul_generator_teeth(const uint8_t k[32], const uint8_t p_enc[32], const uint8_t r[32], uint8_t out[32]) { return g_teeth == 5 ? hc_double_mul_generator_teeth_t<5>(k, p_enc, r, out) : g_teeth == 7 ? hc_double_mul_generator_teeth_t<7>(k, p_enc, r, out) : hc_double_mul_generator_teeth_t<6>(k, p_enc, r, out); }
extern "C" int hc_multi_mul_teeth(int n, const uint8_t* ks, const uint8_t* ps, const uint8_t r[32], uint8_t out[32]) { return g_teeth == 5 ? hc_multi_mul_teeth_t<5>(n, ks, ps, r, out) : g_teeth == 7 ? hc_multi_mul_teeth_t<7>(n, ks, ps, r, out) : hc_multi_mul_teeth_t<6>(n, ks, ps, r, out); }
// the same table of a sum made the way the ring-group walk makes it: members in groups of `group`, whose tables exist only while their
// group is accumulated (ge_teeth_sum_accumulate), then ge_teeth_tables_sum from the accumulator alone.  Every entry must equal the
// entry of the table summed directly from all members' tables (same curve point), and the product must be the same encoding.
template <int T>
static int hc_sum_table_grouped_t(int n, int group, const uint8_t* ps, const uint8_t k[32], const uint8_t r[32], uint8_t out[32]) {
  if (!g_base_table.ready) { ge g; ge_generator(g); build_fixed(g_base_table, g); }
  std::vector<ArrBase> tabs(n);
  for (int i = 0; i < n; ++i) {
    u32 pw[8]; words_from_bytes(pw, ps + 32 * i, 8);
    ge p; if (!ristretto_decode(p, pw)) return -1;
    ArrTable tmp; ge_teeth_tables_build<T>(tabs[i], tmp, p);
  }
  ArrBase direct, acc, viaacc; ArrTable tmp, tmp2;
  ge_teeth_tables_sum<T>(direct, tmp, n, [&](int t, int g, ge_cached& e) { tabs[t].load(e, g); });
  for (int g0 = 0; g0 < n; g0 += group) {
    const int m = std::min(group, n - g0);
    for (int t = 0; t < T; ++t)
      ge_teeth_sum_accumulate<T>(acc, t, g0 == 0, m, [&](int i, int g, ge_cached& e) { tabs[g0 + i].load(e, g); });
  }
  ge_teeth_tables_sum<T>(viaacc, tmp2, 1, [&](int, int g, ge_cached& e) { acc.load(e, teeth_first_flip_index<T>(g)); });
  int same = 1;
  for (int t = 0; t < T; ++t) if (teeth_first_flip_index<T>(teeth_first_flip_entry<T>(t)) != t) same = 0;
  for (int g = 0; g < Teeth<T>::ENTRIES; ++g) {
    ge_cached ea, eb; direct.load(ea, g); viaacc.load(eb, g);
    ge a, b; ge_cached_to_p3(a, ea); ge_cached_to_p3(b, eb);
    fe az, bz, l, rr;
    az = a.Z; fe_carry(az); bz = b.Z; fe_carry(bz);
    fe_mul(l, a.X, bz); fe_mul(rr, b.X, az); if (!fe_eq(l, rr)) same = 0;
    fe_mul(l, a.Y, bz); fe_mul(rr, b.Y, az); if (!fe_eq(l, rr)) same = 0;
    fe_mul(l, a.T, bz); fe_mul(rr, b.T, az); if (!fe_eq(l, rr)) same = 0;
  }
  u32 kw[8], rw[8], o[8];
  words_from_bytes(kw, k, 8); words_from_bytes(rw, r, 8);
  u64 rows[T]; u32 dr[EG_COMB_WORDS]; sc_recode_teeth<T>(rows, kw); sc_recode_comb(dr, rw);
  ge res; ge_teeth_mul<T>(res, viaacc, rows);
  ge_fixed_mul_add(res, g_base_table, dr);
  ristretto_encode(o, res);
  bytes_from_words(out, o, 8);
  return same;
}
extern "C" int hc_sum_table_grouped(int n, int group, const uint8_t* ps, const uint8_t k[32], const uint8_t r[32], uint8_t out[32]) { return g_teeth == 5 ? hc_sum_table_grouped_t<5>(n, group, ps, k, r, out) : g_teeth == 7 ? hc_sum_table_grouped_t<7>(n, group, ps, k, r, out) : hc_sum_table_grouped_t<6>(n, group, ps, k, r, out); }
extern "C" int hc_sum_table_mul(int n, const uint8_t* ps, const uint8_t k[32], const uint8_t r[32], uint8_t out[32]) { return g_teeth == 5 ? hc_sum_table_mul_t<5>(n, ps, k, r, out) : g_teeth == 7 ? hc_sum_table_mul_t<7>(n, ps, k, r, out) : hc_sum_table_mul_t<6>(n, ps, k, r, out); }
extern "C" int hc_double_mul_generator_halved(const uint8_t k[32], const uint8_t p_enc[32], const uint8_t r[32], uint8_t out[32]) { return g_teeth == 5 ? hc_double_mul_generator_halved_t<5>(k, p_enc, r, out) : g_teeth == 7 ? hc_double_mul_generator_halved_t<7>(k, p_enc, r, out) : hc_double_mul_generator_halved_t<6>(k, p_enc, r, out); }
extern "C" void hc_op_counts(unsigned long long out[28]) { if (g_teeth == 5) hc_op_counts_t<5>(out); else if (g_teeth == 7) hc_op_counts_t<7>(out); else hc_op_counts_t<6>(out); }

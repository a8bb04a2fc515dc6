// prover_kernels.cuh -- synthetic-ballot generation on the GPU: EncryptedChoice::new for one voter per lane
// (SURVEY.md 8f row 1).  Follows the reference's construction and RNG draw order exactly
//   choice.rs:313-349 (EncryptedChoice::new), ring.rs:54-194 (Ring::new / aggregate / finalize),
//   ring.rs:440-506 (RingProofBuilder), encryption.rs:310-327 (ExtendedCiphertext::new),
//   log_equality.rs:114-139 (LogEqualityProof::new), ristretto.rs:28-32 (generate_scalar)
// so that ballot i equals what the reference produces from ChaChaRng::seed_from_u64(seed0 + i) given the
// same selection; the selection itself comes from a second ChaCha stream seeded with the complemented seed.
// Because the prover knows r with R = [r]G and B = x_v + [r]K, every simulated commitment
//   R_G = [s]G - [e]R = [s - e r]G          R_K = [s]K - [e](B - x_j) = [s - e r]K - [e (v - j) step]G
// is computed with fixed-base tables only; group elements are equal, hence the canonical encodings that are
// hashed and the resulting proof bytes are identical to the reference's.
#pragma once
#include "device_io.cuh"

namespace eg {

// ---- ChaCha20 block RNG as rand_chacha::ChaCha20Rng + SeedableRng::seed_from_u64 (tests/snapshots.rs:32) ----
struct ChaChaRng {
  u32 key[8];
  u64 counter;
};
__device__ __forceinline__ u32 rotl32(u32 x, int n) { return (x << n) | (x >> (32 - n)); }
#define EG_QR(a, b, c, d) a += b; d ^= a; d = rotl32(d, 16); c += d; b ^= c; b = rotl32(b, 12); a += b; d ^= a; d = rotl32(d, 8); c += d; b ^= c; b = rotl32(b, 7);
__device__ __forceinline__ void chacha_seed_from_u64(ChaChaRng& r, u64 state) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {   // PCG32 expansion of the u64 seed
    state = state * 6364136223846793005ULL + 11634580027462260723ULL;
    const u32 xorshifted = (u32)(((state >> 18) ^ state) >> 27);
    const u32 rot = (u32)(state >> 59);
    r.key[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
  }
  r.counter = 0;
}
__device__ __forceinline__ void chacha_block(ChaChaRng& r, u32 out[16]) {
  u32 s[16], x[16];
  s[0] = 0x61707865u; s[1] = 0x3320646eu; s[2] = 0x79622d32u; s[3] = 0x6b206574u;
#pragma unroll
  for (int i = 0; i < 8; ++i) s[4 + i] = r.key[i];
  s[12] = (u32)r.counter; s[13] = (u32)(r.counter >> 32); s[14] = 0; s[15] = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = s[i];
#pragma unroll 1
  for (int i = 0; i < 10; ++i) {
    EG_QR(x[0], x[4], x[8], x[12]) EG_QR(x[1], x[5], x[9], x[13]) EG_QR(x[2], x[6], x[10], x[14]) EG_QR(x[3], x[7], x[11], x[15])
    EG_QR(x[0], x[5], x[10], x[15]) EG_QR(x[1], x[6], x[11], x[12]) EG_QR(x[2], x[7], x[8], x[13]) EG_QR(x[3], x[4], x[9], x[14])
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) out[i] = x[i] + s[i];
  r.counter++;
}
// ScalarOps::generate_scalar: 64 random bytes -> from_bytes_mod_order_wide
__device__ __noinline__ void rng_scalar(ChaChaRng& r, u32 out[8]) {
  u32 w[16];
  chacha_block(r, w);
  sc_from_wide(out, w);
}

// out = encode([a]G + [b]K)   (b may be null)
__device__ __noinline__ void fixed2_encode(u32 out[8], const FixedTable& tg, const u32* a, const FixedTable& tk, const u32* b) {
  ge acc;
  ge_identity(acc);
  u32 dg[EG_COMB_WORDS];
  if (a) { sc_recode_comb(dg, a); ge_fixed_mul_add(acc, tg, dg); }
  if (b) { sc_recode_comb(dg, b); ge_fixed_mul_add(acc, tk, dg); }
  ristretto_encode(out, acc);
}

// out-of-line transcript helpers: the generator is not throughput critical, keep its code (and compile time) small
__device__ __noinline__ void gen_append32(Transcript<LdsState>& t, const char* label, int label_len, const u32* words) {
  merlin_append_words(t, label, label_len, words, 32);
}
__device__ __noinline__ void gen_append_u64(Transcript<LdsState>& t, const char* label, int label_len, u64 x) {
  merlin_append_u64(t, label, label_len, x);
}
__device__ __noinline__ void gen_append_ct(Transcript<LdsState>& t, const u32* encR, const u32* encB) {
  merlin_frame(t, "enc", 3, 64u);
  strobe_begin_op(t, EG_FLAG_AD);
  strobe_absorb_words(t, encR, 32);
  strobe_absorb_words(t, encB, 32);
}
__device__ __noinline__ void gen_challenge(Transcript<LdsState>& t, u32 e[8]) {
  u32 wide[16];
  merlin_challenge64(t, "c", 1, wide);
  sc_from_wide(e, wide);
}
__device__ __noinline__ void gen_import(Transcript<LdsState>& t, const u32* prefix) { merlin_import(t, prefix); }
__device__ __noinline__ void gen_muladd(u32 out[8], const u32* a, const u32* b, const u32* c) { sc_muladd(out, a, b, c); }

// ---- per-lane scratch of the generators, in HBM -----------------------------------------------------------------------------
// A ballot's secrets (per option: r, x, the terminal commitments, a simulated response; per ring of a range proof the same plus
// every response) do not fit registers for large elections, so they live in a workspace slice per lane, [word][lane] so that the
// accesses of a wave coalesce.  The slice is sized from the election's shape; nothing caps the number of options or rings.
struct LaneWs {
  u32* p;          // ws + lane
  size_t lanes;
  __device__ __forceinline__ u32 ld(u32 w) const { return p[(size_t)w * lanes]; }
  __device__ __forceinline__ void st(u32 w, u32 v) const { p[(size_t)w * lanes] = v; }
  __device__ __forceinline__ void ld8(u32 out[8], u32 w) const {
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = ld(w + i);
  }
  __device__ __forceinline__ void st8(u32 w, const u32 in[8]) const {
#pragma unroll
    for (int i = 0; i < 8; ++i) st(w + i, in[i]);
  }
};

// choice ballots: selection bits [0, SW), then 40 words per option: r +0, x +8, terminal R_G +16, R_K +24, simulated s_1 +32
__host__ __device__ inline u32 gen_choice_ws_words(int n_options) { return (u32)((n_options + 31) / 32 + 40 * n_options); }

// One lane = one voter.  out: packed ballot (choice wire layout of eg_hip.h).
// selection: null = the voter's choice comes from the second stream; else ceil(n_options / 32) bitmask words per voter (bit k =
// option k chosen: EncryptedChoice::single(params, choice, rng) / ::new(params, &[bool], rng)).  rng_skip = 64-byte draws the
// voter's RNG has served before the ballot (tests/snapshots.rs:107-131 draws the keypair first: 1).
__global__ void __launch_bounds__(NT) k_choice_encrypt(u64 seed0, size_t n, int n_options, int single, int n_selected,
                                                       const u32* selection, u64 rng_skip,
                                                       const uint4* tabG, const uint4* tabK, const u32* prefixes,
                                                       int pre_main, int pre_ring, int pre_logeq, u32* out, u32 stride_words,
                                                       u32* gws) {
  __shared__ u32 lds[50 * NT];
  const FixedTable tg(tabG), tk(tabK);
  const LaneWs ws{gws + ((size_t)blockIdx.x * NT + threadIdx.x), (size_t)gridDim.x * NT};
  const u32 SW = (u32)((n_options + 31) / 32);
  for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
    const u64 seed = seed0 + i;
    ChaChaRng rng;
    chacha_seed_from_u64(rng, seed);
    rng.counter = rng_skip;
    u32* ob = out + i * stride_words;
    // ---- voter selection (given, or drawn from a second stream) ----
    if (selection) {
      for (u32 w = 0; w < SW; ++w) ws.st(w, selection[i * SW + w]);
    } else {
      for (u32 w = 0; w < SW; ++w) ws.st(w, 0u);
      ChaChaRng sel;
      chacha_seed_from_u64(sel, ~seed);
      u32 buf[16];
      int pos = 16, got = 0;
      const int want = single ? 1 : n_selected;
#pragma unroll 1
      while (got < want) {
        if (pos >= 16) { chacha_block(sel, buf); pos = 0; }
        u32 v = buf[0];
#pragma unroll
        for (int q = 1; q < 16; ++q) v = (pos == q) ? buf[q] : v;
        ++pos;
        const u32 k = v % (u32)n_options;
        const u32 f = ws.ld(k >> 5);
        if (!((f >> (k & 31)) & 1u)) { ws.st(k >> 5, f | (1u << (k & 31))); ++got; }
      }
    }
    Transcript<LdsState> t;
    t.st.base = lds + threadIdx.x;
    u32 sum_r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const u32 one[8] = {1, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 1
    for (int k = 0; k < n_options; ++k) {
      const u32 wb = SW + 40u * (u32)k;
      const bool vi = (ws.ld((u32)k >> 5) >> (k & 31)) & 1u;
      // ExtendedCiphertext::new (encryption.rs:310-327): r, R = [r]G, B = value + [r]K
      u32 r[8], x[8], encR[8], encB[8];
      rng_scalar(rng, r);
      fixed2_encode(encR, tg, r, tk, nullptr);
      fixed2_encode(encB, tg, vi ? one : nullptr, tk, r);
#pragma unroll
      for (int w = 0; w < 8; ++w) { ob[(2 * k) * 8 + w] = encR[w]; ob[(2 * k + 1) * 8 + w] = encB[w]; }
      ws.st8(wb, r);
      gen_muladd(sum_r, sum_r, one, r);
      // Ring::new (ring.rs:54-131)
      gen_import(t, prefixes + (size_t)pre_ring * 52);
      gen_append_ct(t, encR, encB);
      gen_append_u64(t, "i", 1, (u64)k);
      rng_scalar(rng, x);
      u32 cg[8], ck[8];
      fixed2_encode(cg, tg, x, tk, nullptr);
      fixed2_encode(ck, tg, nullptr, tk, x);
      ws.st8(wb + 8, x);
      if (!vi) {
        // equation 1 is simulated now (ring.rs:103-118): challenge from ([x]G, [x]K), random response
        gen_append_u64(t, "j", 1, 0);
        gen_append32(t, "R_G", 3, cg);
        gen_append32(t, "R_K", 3, ck);
        u32 e[8], s1[8], ne[8], tt[8];
        gen_challenge(t, e);
        rng_scalar(rng, s1);
        sc_neg(ne, e);
        gen_muladd(tt, ne, r, s1);                // s - e r
        fixed2_encode(cg, tg, tt, tk, nullptr);   // [s]G - [e]R
        fixed2_encode(ck, tg, e, tk, tt);         // [s]K - [e](B - G),  B - G = [r]K - G
        ws.st8(wb + 32, s1);
      }
      ws.st8(wb + 16, cg);
      ws.st8(wb + 24, ck);
    }
    // Ring::aggregate (ring.rs:138-160)
    u32 e0[8];
    {
      gen_import(t, prefixes + (size_t)pre_main * 52);
#pragma unroll 1
      for (int k = 0; k < n_options; ++k) {
        u32 cg[8], ck[8];
        ws.ld8(cg, SW + 40u * (u32)k + 16);
        ws.ld8(ck, SW + 40u * (u32)k + 24);
        gen_append32(t, "R_G", 3, cg);
        gen_append32(t, "R_K", 3, ck);
      }
      gen_challenge(t, e0);
    }
    u32* proof = ob + (size_t)(2 * n_options) * 8;
#pragma unroll
    for (int w = 0; w < 8; ++w) proof[w] = e0[w];
    // Ring::finalize (ring.rs:162-194)
#pragma unroll 1
    for (int k = 0; k < n_options; ++k) {
      const u32 wb = SW + 40u * (u32)k;
      const bool vi = (ws.ld((u32)k >> 5) >> (k & 31)) & 1u;
      u32 r[8], x[8], s0[8], s1[8];
      ws.ld8(r, wb);
      ws.ld8(x, wb + 8);
      if (vi) {
        u32 ne[8], tt[8], cg[8], ck[8], encR[8], encB[8];
        rng_scalar(rng, s0);
        sc_neg(ne, e0);
        gen_muladd(tt, ne, r, s0);
        fixed2_encode(cg, tg, tt, tk, nullptr);   // [s0]G - [e0]R
        fixed2_encode(ck, tg, ne, tk, tt);        // [s0]K - [e0](B - O),  B = G + [r]K
#pragma unroll
        for (int w = 0; w < 8; ++w) { encR[w] = ob[(2 * k) * 8 + w]; encB[w] = ob[(2 * k + 1) * 8 + w]; }
        gen_import(t, prefixes + (size_t)pre_ring * 52);
        gen_append_ct(t, encR, encB);
        gen_append_u64(t, "i", 1, (u64)k);
        gen_append_u64(t, "j", 1, 0);
        gen_append32(t, "R_G", 3, cg);
        gen_append32(t, "R_K", 3, ck);
        u32 e1[8];
        gen_challenge(t, e1);
        gen_muladd(s1, e1, r, x);                 // trapdoor response (ring.rs:192-193)
      } else {
        gen_muladd(s0, e0, r, x);
        ws.ld8(s1, wb + 32);
      }
#pragma unroll
      for (int w = 0; w < 8; ++w) { proof[(1 + 2 * k) * 8 + w] = s0[w]; proof[(2 + 2 * k) * 8 + w] = s1[w]; }
    }
    if (single) {
      // SingleChoice::prove (choice.rs:59-75) -> LogEqualityProof::new (log_equality.rs:114-139)
      // powers = (sum R, sum B - G) = ([sum r]G, [sum r]K + (count - 1)G); count = 1 for a single choice
      u32 p0[8], p1[8], x[8], xg[8], xk[8];
      fixed2_encode(p0, tg, sum_r, tk, nullptr);
      fixed2_encode(p1, tg, nullptr, tk, sum_r);
      gen_import(t, prefixes + (size_t)pre_logeq * 52);
      gen_append32(t, "[r]G", 4, p0);
      gen_append32(t, "[r]K", 4, p1);
      rng_scalar(rng, x);
      fixed2_encode(xg, tg, x, tk, nullptr);
      fixed2_encode(xk, tg, nullptr, tk, x);
      gen_append32(t, "[x]G", 4, xg);
      gen_append32(t, "[x]K", 4, xk);
      u32 c[8], s[8];
      gen_challenge(t, c);
      gen_muladd(s, c, sum_r, x);
      u32* sp = proof + (size_t)(1 + 2 * n_options) * 8;
#pragma unroll
      for (int w = 0; w < 8; ++w) { sp[w] = c[w]; sp[8 + w] = s[w]; }
    }
  }
}


// =====================================================================================================================
// QuadraticVotingBallot::new (quadratic_voting.rs:234-284) = RangeProof::new per option + credit (range.rs:462-534)
// + SumOfSquaresProof::new (mul.rs:107-181), one voter per lane, fixed-base arithmetic only (see the file header).
// =====================================================================================================================
struct GenRange {
  int n_rings;
  const u32* desc;           // device memory: size_0, step_0, size_1, step_1, ..
  int pre_main, pre_ring;    // hoisted transcript prefixes of this (label, range) pair
  __device__ __forceinline__ u32 size(int i) const { return desc[2 * i]; }
  __device__ __forceinline__ u32 step(int i) const { return desc[2 * i + 1]; }
};

__device__ __noinline__ void sc_from_small(u32 out[8], long long m) {   // m mod l for a small signed integer
  u32 a[8];
  sc_from_u64(a, (u64)(m < 0 ? -m : m));
  if (m < 0) sc_neg(out, a);
  else {
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = a[i];
  }
}

// commitments of a simulated equation: R_G = [s - e r]G, R_K = [s - e r]K - [e * delta]G  with delta = (v - eq) * step
__device__ __noinline__ void gen_sim_commitments(u32 cg[8], u32 ck[8], const FixedTable& tg, const FixedTable& tk, const u32* s,
                                                 const u32* e, const u32* r, long long delta) {
  u32 ne[8], tt[8], dl[8], gcoef[8];
  sc_neg(ne, e);
  gen_muladd(tt, ne, r, s);               // s - e r
  sc_from_small(dl, delta);
  const u32 zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  gen_muladd(gcoef, ne, dl, zero);        // -e * delta
  fixed2_encode(cg, tg, tt, tk, nullptr);
  fixed2_encode(ck, tg, gcoef, tk, tt);
}

// workspace of one range proof at word offset rb: per ring 33 words (value index +0, r +1, x +9, terminal R_G +17, R_K +25),
// then 8 words per response in ring order
__host__ __device__ inline u32 gen_range_ws_words(u32 n_rings, u32 total_responses) { return 33u * n_rings + 8u * total_responses; }

// RangeProof::new for `value` with randomness drawn from rng; writes ct || partials || e0 || responses at `ob`
// (words) and returns the ciphertext randomness in r_out.
__device__ __noinline__ void gen_range_proof(Transcript<LdsState>& t, ChaChaRng& rng, const GenRange& R, u64 value,
                                             const FixedTable& tg, const FixedTable& tk, const u32* prefixes, u32* ob,
                                             u32 r_out[8], const LaneWs& ws, u32 rb) {
  const int nr = R.n_rings;
  const u32 resp0 = rb + 33u * (u32)nr;
  // CiphertextWithValue::new (encryption.rs:403-407)
  u32 r[8], vsc[8];
  rng_scalar(rng, r);
  sc_from_u64(vsc, value);
  u32 enc[8];
  fixed2_encode(enc, tg, r, tk, nullptr);
#pragma unroll
  for (int w = 0; w < 8; ++w) ob[w] = enc[w];
  fixed2_encode(enc, tg, vsc, tk, r);
#pragma unroll
  for (int w = 0; w < 8; ++w) { ob[8 + w] = enc[w]; r_out[w] = r[w]; }
  // decompose (range.rs:199-210)
  int total = 0;
  {
    u64 rem = value;
    for (int i = 0; i < nr; ++i) {
      u64 q = rem / R.step(i);
      if (q > R.size(i) - 1) q = R.size(i) - 1;
      ws.st(rb + 33u * (u32)i, (u32)q);
      rem -= q * R.step(i);
      total += (int)R.size(i);
    }
  }
  u32 cum_r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const u32 one[8] = {1, 0, 0, 0, 0, 0, 0, 0};
  u64 cum_v = 0;
  u32* partials = ob + 16;
  u32* proof = partials + (size_t)(nr - 1) * 16;    // e0 then responses
  u32 off = 0;                                      // index of the ring's first response
#pragma unroll 1
  for (int i = 0; i < nr; ++i) {
    const u32 rw = rb + 33u * (u32)i;
    const int vi = (int)ws.ld(rw);
    const u64 mval = (u64)vi * R.step(i);
    u32 rr[8], encR[8], encB[8], msc[8];
    if (i + 1 < nr) {   // add_value (ring.rs:460-469): fresh randomness
      rng_scalar(rng, rr);
      gen_muladd(cum_r, cum_r, one, rr);
      cum_v += mval;
    } else if (nr > 1) {   // last ring: ciphertext - sum(partials)  (range.rs:520-528)
      u32 nc[8];
      sc_neg(nc, cum_r);
      gen_muladd(rr, r, one, nc);
    } else {
#pragma unroll
      for (int w = 0; w < 8; ++w) rr[w] = r[w];
    }
    sc_from_u64(msc, (i + 1 < nr || nr == 1) ? mval : value - cum_v);
    fixed2_encode(encR, tg, rr, tk, nullptr);
    fixed2_encode(encB, tg, msc, tk, rr);
    if (i + 1 < nr) {
#pragma unroll
      for (int w = 0; w < 8; ++w) { partials[i * 16 + w] = encR[w]; partials[i * 16 + 8 + w] = encB[w]; }
    }
    // Ring::new (ring.rs:54-131)
    gen_import(t, prefixes + (size_t)R.pre_ring * 52);
    gen_append_ct(t, encR, encB);
    gen_append_u64(t, "i", 1, (u64)i);
    u32 x[8], cg[8], ck[8];
    rng_scalar(rng, x);
    fixed2_encode(cg, tg, x, tk, nullptr);
    fixed2_encode(ck, tg, nullptr, tk, x);
    ws.st8(rw + 1, rr);
    ws.st8(rw + 9, x);
#pragma unroll 1
    for (int eq = vi + 1; eq < (int)R.size(i); ++eq) {
      // fork of the ring transcript: state is rebuilt per equation (cheap) to avoid keeping a second LDS column
      gen_import(t, prefixes + (size_t)R.pre_ring * 52);
      gen_append_ct(t, encR, encB);
      gen_append_u64(t, "i", 1, (u64)i);
      gen_append_u64(t, "j", 1, (u64)(eq - 1));
      gen_append32(t, "R_G", 3, cg);
      gen_append32(t, "R_K", 3, ck);
      u32 e[8], sq[8];
      gen_challenge(t, e);
      rng_scalar(rng, sq);
      ws.st8(resp0 + 8u * (off + (u32)eq), sq);
      gen_sim_commitments(cg, ck, tg, tk, sq, e, rr, ((long long)vi - eq) * (long long)R.step(i));
    }
    ws.st8(rw + 17, cg);
    ws.st8(rw + 25, ck);
    off += R.size(i);
  }
  // Ring::aggregate (ring.rs:138-160)
  u32 e0[8];
  gen_import(t, prefixes + (size_t)R.pre_main * 52);
#pragma unroll 1
  for (int i = 0; i < nr; ++i) {
    u32 cg[8], ck[8];
    ws.ld8(cg, rb + 33u * (u32)i + 17);
    ws.ld8(ck, rb + 33u * (u32)i + 25);
    gen_append32(t, "R_G", 3, cg);
    gen_append32(t, "R_K", 3, ck);
  }
  gen_challenge(t, e0);
#pragma unroll
  for (int w = 0; w < 8; ++w) proof[w] = e0[w];
  // Ring::finalize (ring.rs:162-194)
  off = 0;
#pragma unroll 1
  for (int i = 0; i < nr; ++i) {
    const u32 rw = rb + 33u * (u32)i;
    const int vi = (int)ws.ld(rw);
    u32 rr[8], x[8], ch[8], encR[8], encB[8];
    ws.ld8(rr, rw + 1);
    ws.ld8(x, rw + 9);
#pragma unroll
    for (int w = 0; w < 8; ++w) ch[w] = e0[w];
    const u32* src = (i + 1 < nr) ? partials + i * 16 : nullptr;
    if (src) {
#pragma unroll
      for (int w = 0; w < 8; ++w) { encR[w] = src[w]; encB[w] = src[8 + w]; }
    } else {   // last ring: re-encode its ciphertext ([r_last]G, [m_last]G + [r_last]K)
      u32 msc[8];
      sc_from_u64(msc, nr == 1 ? (u64)vi * R.step(i) : value - cum_v);
      fixed2_encode(encR, tg, rr, tk, nullptr);
      fixed2_encode(encB, tg, msc, tk, rr);
    }
#pragma unroll 1
    for (int eq = 0; eq < vi; ++eq) {
      u32 sq[8], cg[8], ck[8];
      rng_scalar(rng, sq);
      ws.st8(resp0 + 8u * (off + (u32)eq), sq);
      gen_sim_commitments(cg, ck, tg, tk, sq, ch, rr, ((long long)vi - eq) * (long long)R.step(i));
      gen_import(t, prefixes + (size_t)R.pre_ring * 52);
      gen_append_ct(t, encR, encB);
      gen_append_u64(t, "i", 1, (u64)i);
      gen_append_u64(t, "j", 1, (u64)eq);
      gen_append32(t, "R_G", 3, cg);
      gen_append32(t, "R_K", 3, ck);
      gen_challenge(t, ch);
    }
    u32 sv[8];
    gen_muladd(sv, ch, rr, x);
    ws.st8(resp0 + 8u * (off + (u32)vi), sv);
#pragma unroll 1
    for (int eq = 0; eq < (int)R.size(i); ++eq) {
      u32 sq[8];
      ws.ld8(sq, resp0 + 8u * (off + (u32)eq));
#pragma unroll
      for (int w = 0; w < 8; ++w) proof[(size_t)(1 + off + eq) * 8 + w] = sq[w];
    }
    off += R.size(i);
  }
  (void)total;
}

// quadratic voting: votes [0, n), then 24 words per option (ciphertext randomness +0, e_r +8, e_x +16), then one range-proof area
__host__ __device__ inline u32 gen_qv_ws_words(int n_options, u32 max_rings, u32 max_responses) {
  return (u32)n_options * 25u + gen_range_ws_words(max_rings, max_responses);
}

// votes_in: null = votes drawn from the second stream; else n_options words per voter (QuadraticVotingBallot::new(params, votes,
// rng), quadratic_voting.rs:234-284; the caller keeps sum(v^2) <= credits and v <= isqrt(credits), as the reference asserts)
__global__ void __launch_bounds__(NT) k_qv_encrypt(u64 seed0, size_t n, int n_options, u64 credits, const u32* votes_in, u64 rng_skip,
                                                   GenRange vote_range,
                                                   GenRange credit_range, int pre_sumsq, const uint4* tabG, const uint4* tabK,
                                                   const u32* prefixes, u32* out, u32 stride_words, u32 vote_words,
                                                   u32 credit_words, u32* gws) {
  __shared__ u32 lds[50 * NT];
  const FixedTable tg(tabG), tk(tabK);
  const LaneWs ws{gws + ((size_t)blockIdx.x * NT + threadIdx.x), (size_t)gridDim.x * NT};
  const u32 NO = (u32)n_options, per = NO, rb = NO * 25u;     // votes at [0, NO); option k's scalars at per + 24 k; range area at rb
  for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
    const u64 seed = seed0 + i;
    ChaChaRng rng;
    chacha_seed_from_u64(rng, seed);
    rng.counter = rng_skip;
    u32* ob = out + i * stride_words;
    // votes: given, or as in tests/integration/sharing.rs:135-147 (geometric, p = 0.8) from a second stream
    for (u32 k = 0; k < NO; ++k) ws.st(k, votes_in ? votes_in[i * (size_t)n_options + k] : 0u);
    if (!votes_in) {
      ChaChaRng sel;
      chacha_seed_from_u64(sel, ~seed);
      u32 buf[16];
      int pos = 16;
      auto next = [&]() -> u32 {
        if (pos >= 16) { chacha_block(sel, buf); pos = 0; }
        u32 v = buf[0];
#pragma unroll
        for (int q = 1; q < 16; ++q) v = (pos == q) ? buf[q] : v;
        ++pos;
        return v;
      };
#pragma unroll 1
      for (;;) {
        if (next() % 10u >= 8u) break;
        const u32 k = next() % NO;
        u64 c = 0;
        for (u32 j = 0; j < NO; ++j) { const u64 v = (u64)ws.ld(j) + (j == k ? 1 : 0); c += v * v; }
        if (c > credits) break;
        ws.st(k, ws.ld(k) + 1u);
      }
    }
    Transcript<LdsState> t;
    t.st.base = lds + threadIdx.x;
    u32 credit_r[8];
    u64 credit = 0;
#pragma unroll 1
    for (u32 k = 0; k < NO; ++k) {
      const u64 v = ws.ld(k);
      u32 vr[8];
      gen_range_proof(t, rng, vote_range, v, tg, tk, prefixes, ob + (size_t)k * vote_words, vr, ws, rb);
      ws.st8(per + 24u * k, vr);
      credit += v * v;
    }
    u32* cb = ob + (size_t)n_options * vote_words;
    gen_range_proof(t, rng, credit_range, credit, tg, tk, prefixes, cb, credit_r, ws, rb);
    // SumOfSquaresProof::new (mul.rs:107-181)
    u32* sp = cb + credit_words;
    gen_import(t, prefixes + (size_t)pre_sumsq * 52);
    u32 e_z[8], sum_rand[8];
    rng_scalar(rng, e_z);
#pragma unroll
    for (int w = 0; w < 8; ++w) sum_rand[w] = credit_r[w];
    u32 acc_g[8] = {0, 0, 0, 0, 0, 0, 0, 0}, acc_k[8];   // sum e_x v  and  sum e_x r + e_z
#pragma unroll
    for (int w = 0; w < 8; ++w) acc_k[w] = e_z[w];
#pragma unroll 1
    for (u32 k = 0; k < NO; ++k) {
      const u32* vct = ob + (size_t)k * vote_words;
      gen_append32(t, "R_x", 3, vct);
      gen_append32(t, "X", 1, vct + 8);
      u32 c0[8], c1[8], xs[8], nx[8], er[8], ex[8], vr[8];
      ws.ld8(vr, per + 24u * k);
      rng_scalar(rng, er);
      fixed2_encode(c0, tg, er, tk, nullptr);
      gen_append32(t, "[e_r]G", 6, c0);
      rng_scalar(rng, ex);
      fixed2_encode(c1, tg, ex, tk, er);
      gen_append32(t, "[e_x]G + [e_r]K", 15, c1);
      ws.st8(per + 24u * k + 8, er);
      ws.st8(per + 24u * k + 16, ex);
      sc_from_u64(xs, (u64)ws.ld(k));
      sc_neg(nx, xs);
      gen_muladd(sum_rand, vr, nx, sum_rand);      // sum_random_scalar += r_x * (-x)
      gen_muladd(acc_g, ex, xs, acc_g);
      gen_muladd(acc_k, ex, vr, acc_k);
    }
    u32 rsum[8], vsum[8];
    fixed2_encode(rsum, tg, acc_k, tk, nullptr);        // sum e_x R_x + e_z G
    fixed2_encode(vsum, tg, acc_g, tk, acc_k);          // sum e_x X + e_z K
    gen_append32(t, "R_z", 3, cb);
    gen_append32(t, "Z", 1, cb + 8);
    gen_append32(t, "[e_x]R_x + [e_z]G", 17, rsum);
    gen_append32(t, "[e_x]X + [e_z]K", 15, vsum);
    u32 c[8];
    gen_challenge(t, c);
#pragma unroll
    for (int w = 0; w < 8; ++w) sp[w] = c[w];
#pragma unroll 1
    for (u32 k = 0; k < NO; ++k) {
      u32 s_r[8], s_x[8], xs[8], er[8], ex[8], vr[8];
      ws.ld8(vr, per + 24u * k);
      ws.ld8(er, per + 24u * k + 8);
      ws.ld8(ex, per + 24u * k + 16);
      sc_from_u64(xs, (u64)ws.ld(k));
      gen_muladd(s_r, c, vr, er);
      gen_muladd(s_x, c, xs, ex);
#pragma unroll
      for (int w = 0; w < 8; ++w) { sp[(size_t)(1 + 2 * k) * 8 + w] = s_r[w]; sp[(size_t)(2 + 2 * k) * 8 + w] = s_x[w]; }
    }
    u32 s_z[8];
    gen_muladd(s_z, c, sum_rand, e_z);
#pragma unroll
    for (int w = 0; w < 8; ++w) sp[(size_t)(1 + 2 * n_options) * 8 + w] = s_z[w];
  }
}

}  // namespace eg

// merlin.cuh -- Keccak-f[1600], STROBE-128 and the Merlin transcript framing for gfx950.
//
// Replaces merlin 3.0.0 / keccak 0.1.6 as used through the reference's TranscriptForGroup glue
// (src/proofs/mod.rs:39-57: start_proof = append_message("dom-sep", ..), append_element,
// challenge_scalar = 64-byte PRF -> wide reduce; src/group/mod.rs:37-62).  Framing per the published
// Merlin / STROBE-128 specification.
//
// Every lane of a wavefront runs the same transcript program, so the STROBE position is wave-uniform
// and the 200-byte state is kept outside registers behind a storage policy S:
//   S::rd(word) / S::wr(word, value) for 50 little-endian 32-bit words
// (device: word-interleaved LDS, conflict-free for uniform positions; host tests: a plain array).
// Keccak-f pulls the state into registers, runs 24 rolled rounds and writes it back.
#pragma once
#include "fe25519.cuh"

namespace eg {

#define EG_STROBE_R 166

EG_HD u64 eg_rotl64(u64 x, int n) { return (x << n) | (x >> (64 - n)); }

EG_HD u64 keccak_rc(int round) {
  // LFSR-free closed table; a switch keeps it in SGPR/literals instead of a memory table
  switch (round) {
    case 0: return 0x0000000000000001ULL; case 1: return 0x0000000000008082ULL;
    case 2: return 0x800000000000808aULL; case 3: return 0x8000000080008000ULL;
    case 4: return 0x000000000000808bULL; case 5: return 0x0000000080000001ULL;
    case 6: return 0x8000000080008081ULL; case 7: return 0x8000000000008009ULL;
    case 8: return 0x000000000000008aULL; case 9: return 0x0000000000000088ULL;
    case 10: return 0x0000000080008009ULL; case 11: return 0x000000008000000aULL;
    case 12: return 0x000000008000808bULL; case 13: return 0x800000000000008bULL;
    case 14: return 0x8000000000008089ULL; case 15: return 0x8000000000008003ULL;
    case 16: return 0x8000000000008002ULL; case 17: return 0x8000000000000080ULL;
    case 18: return 0x000000000000800aULL; case 19: return 0x800000008000000aULL;
    case 20: return 0x8000000080008081ULL; case 21: return 0x8000000000008080ULL;
    case 22: return 0x0000000080000001ULL; default: return 0x8000000080008008ULL;
  }
}

// 64-bit lanes as two 32-bit words for the device: gfx950 has a three-input boolean instruction (v_bitop3_b32: a five-way XOR is two
// instructions, chi's a ^ (~b & c) one) and a funnel shift (v_alignbit_b32: a 64-bit rotation is two), which hipcc does not find from
// 64-bit C code (it emitted two 64-bit shifts and an OR per rotation, and two-input XORs: ~300 instructions per round, now ~190).
#if defined(__HIP_DEVICE_COMPILE__)
struct kw { u32 lo, hi; };
__device__ __forceinline__ u32 eg_xor3(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
__device__ __forceinline__ u32 eg_chi(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xd2); }    // a ^ (~b & c)
template <int N>
__device__ __forceinline__ kw eg_rol(kw x) {      // rotate the 64-bit lane left by N (0 < N < 64)
  kw r;
  if (N == 32) { r.lo = x.hi; r.hi = x.lo; }
  else if (N < 32) { r.lo = __builtin_amdgcn_alignbit(x.lo, x.hi, 32 - N); r.hi = __builtin_amdgcn_alignbit(x.hi, x.lo, 32 - N); }
  else { r.lo = __builtin_amdgcn_alignbit(x.hi, x.lo, 64 - N); r.hi = __builtin_amdgcn_alignbit(x.lo, x.hi, 64 - N); }
  return r;
}
__device__ __forceinline__ void keccak_f1600(u64 state[25]) {
  kw a[25];
#pragma unroll
  for (int i = 0; i < 25; ++i) { a[i].lo = (u32)state[i]; a[i].hi = (u32)(state[i] >> 32); }
#pragma unroll 1
  for (int round = 0; round < 24; ++round) {
    kw c[5], d[5];
#pragma unroll
    for (int x = 0; x < 5; ++x) {
      c[x].lo = eg_xor3(eg_xor3(a[x].lo, a[x + 5].lo, a[x + 10].lo), a[x + 15].lo, a[x + 20].lo);
      c[x].hi = eg_xor3(eg_xor3(a[x].hi, a[x + 5].hi, a[x + 10].hi), a[x + 15].hi, a[x + 20].hi);
    }
#pragma unroll
    for (int x = 0; x < 5; ++x) {
      const kw r = eg_rol<1>(c[(x + 1) % 5]);
      d[x].lo = c[(x + 4) % 5].lo ^ r.lo;
      d[x].hi = c[(x + 4) % 5].hi ^ r.hi;
    }
#pragma unroll
    for (int i = 0; i < 25; ++i) { a[i].lo ^= d[i % 5].lo; a[i].hi ^= d[i % 5].hi; }
    kw b[25];
    b[0] = a[0];
    b[10] = eg_rol<1>(a[1]);   b[20] = eg_rol<62>(a[2]);  b[5] = eg_rol<28>(a[3]);   b[15] = eg_rol<27>(a[4]);
    b[16] = eg_rol<36>(a[5]);  b[1] = eg_rol<44>(a[6]);   b[11] = eg_rol<6>(a[7]);   b[21] = eg_rol<55>(a[8]);
    b[6] = eg_rol<20>(a[9]);   b[7] = eg_rol<3>(a[10]);   b[17] = eg_rol<10>(a[11]); b[2] = eg_rol<43>(a[12]);
    b[12] = eg_rol<25>(a[13]); b[22] = eg_rol<39>(a[14]); b[23] = eg_rol<41>(a[15]); b[8] = eg_rol<45>(a[16]);
    b[18] = eg_rol<15>(a[17]); b[3] = eg_rol<21>(a[18]);  b[13] = eg_rol<8>(a[19]);  b[14] = eg_rol<18>(a[20]);
    b[24] = eg_rol<2>(a[21]);  b[9] = eg_rol<61>(a[22]);  b[19] = eg_rol<56>(a[23]); b[4] = eg_rol<14>(a[24]);
#pragma unroll
    for (int y = 0; y < 25; y += 5)
#pragma unroll
      for (int x = 0; x < 5; ++x) {
        a[y + x].lo = eg_chi(b[y + x].lo, b[y + (x + 1) % 5].lo, b[y + (x + 2) % 5].lo);
        a[y + x].hi = eg_chi(b[y + x].hi, b[y + (x + 1) % 5].hi, b[y + (x + 2) % 5].hi);
      }
    const u64 rc = keccak_rc(round);
    a[0].lo ^= (u32)rc; a[0].hi ^= (u32)(rc >> 32);
  }
#pragma unroll
  for (int i = 0; i < 25; ++i) state[i] = (u64)a[i].lo | ((u64)a[i].hi << 32);
}
#else
EG_HD void keccak_f1600(u64 a[25]) {
#pragma unroll 1
  for (int round = 0; round < 24; ++round) {
    u64 c[5], d[5];
#pragma unroll
    for (int x = 0; x < 5; ++x) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
#pragma unroll
    for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ eg_rotl64(c[(x + 1) % 5], 1);
#pragma unroll
    for (int i = 0; i < 25; ++i) a[i] ^= d[i % 5];
    // rho + pi : b[y, 2x+3y] = rot(a[x, y], r[x, y])
    u64 b[25];
    b[0] = a[0];
    b[10] = eg_rotl64(a[1], 1);   b[20] = eg_rotl64(a[2], 62);  b[5] = eg_rotl64(a[3], 28);   b[15] = eg_rotl64(a[4], 27);
    b[16] = eg_rotl64(a[5], 36);  b[1] = eg_rotl64(a[6], 44);   b[11] = eg_rotl64(a[7], 6);   b[21] = eg_rotl64(a[8], 55);
    b[6] = eg_rotl64(a[9], 20);   b[7] = eg_rotl64(a[10], 3);   b[17] = eg_rotl64(a[11], 10); b[2] = eg_rotl64(a[12], 43);
    b[12] = eg_rotl64(a[13], 25); b[22] = eg_rotl64(a[14], 39); b[23] = eg_rotl64(a[15], 41); b[8] = eg_rotl64(a[16], 45);
    b[18] = eg_rotl64(a[17], 15); b[3] = eg_rotl64(a[18], 21);  b[13] = eg_rotl64(a[19], 8);  b[14] = eg_rotl64(a[20], 18);
    b[24] = eg_rotl64(a[21], 2);  b[9] = eg_rotl64(a[22], 61);  b[19] = eg_rotl64(a[23], 56); b[4] = eg_rotl64(a[24], 14);
#pragma unroll
    for (int y = 0; y < 25; y += 5)
#pragma unroll
      for (int x = 0; x < 5; ++x) a[y + x] = b[y + x] ^ (~b[y + (x + 1) % 5] & b[y + (x + 2) % 5]);
    a[0] ^= keccak_rc(round);
  }
}
#endif

template <class S>
struct Transcript {
  S st;
  u32 pos, pos_begin, cur_flags;
};

template <class S>
EG_HD void strobe_xor_byte(S& st, u32 pos, u32 byte) {
  const u32 w = pos >> 2, sh = (pos & 3u) * 8u;
  st.wr(w, st.rd(w) ^ (byte << sh));
}

// (Kept inline: an out-of-line permutation - one copy instead of ~90 - measured -1.5 % on single-choice and -3 % on QV ballots;
// the call ABI costs more than the instruction-cache misses of the 420 KB kernel.)
template <class S>
EG_HD void strobe_run_f(Transcript<S>& t) {
  strobe_xor_byte(t.st, t.pos, t.pos_begin);
  strobe_xor_byte(t.st, t.pos + 1, 0x04u);
  strobe_xor_byte(t.st, EG_STROBE_R + 1, 0x80u);
  u64 a[25];
#pragma unroll
  for (int i = 0; i < 25; ++i) a[i] = (u64)t.st.rd(2 * i) | ((u64)t.st.rd(2 * i + 1) << 32);
  keccak_f1600(a);
#pragma unroll
  for (int i = 0; i < 25; ++i) { t.st.wr(2 * i, (u32)a[i]); t.st.wr(2 * i + 1, (u32)(a[i] >> 32)); }
  t.pos = 0;
  t.pos_begin = 0;
}

template <class S>
EG_HD void strobe_absorb_byte(Transcript<S>& t, u32 byte) {
  strobe_xor_byte(t.st, t.pos, byte);
  t.pos++;
  if (t.pos == EG_STROBE_R) strobe_run_f(t);
}

template <class S>
EG_HD u32 strobe_squeeze_byte(Transcript<S>& t) {
  const u32 w = t.pos >> 2, sh = (t.pos & 3u) * 8u;
  const u32 word = t.st.rd(w);
  const u32 out = (word >> sh) & 0xffu;
  t.st.wr(w, word & ~(0xffu << sh));
  t.pos++;
  if (t.pos == EG_STROBE_R) strobe_run_f(t);
  return out;
}

// 4 message bytes at once (little-endian word).  The STROBE position is wave-uniform, so the branch is too.
template <class S>
EG_HD void strobe_absorb_word(Transcript<S>& t, u32 w) {
  if (t.pos + 4 <= EG_STROBE_R) {
    const u32 wi = t.pos >> 2, a = (t.pos & 3u) * 8u;
    if (a == 0) {
      t.st.wr(wi, t.st.rd(wi) ^ w);
    } else {
      t.st.wr(wi, t.st.rd(wi) ^ (w << a));
      t.st.wr(wi + 1, t.st.rd(wi + 1) ^ (w >> (32u - a)));
    }
    t.pos += 4;
    if (t.pos == EG_STROBE_R) strobe_run_f(t);
  } else {
#pragma unroll 1
    for (int i = 0; i < 4; ++i) strobe_absorb_byte(t, (w >> (8 * i)) & 0xffu);
  }
}
// n_bytes (multiple of 4 for the fast path; any tail is handled bytewise) from a little-endian word array
template <class S>
EG_HD void strobe_absorb_words(Transcript<S>& t, const u32* words, int n_bytes) {
  const int nw = n_bytes >> 2;
  for (int i = 0; i < nw; ++i) strobe_absorb_word(t, words[i]);
  for (int i = nw * 4; i < n_bytes; ++i) strobe_absorb_byte(t, (words[i >> 2] >> (8 * (i & 3))) & 0xffu);
}
template <class S>
EG_HD u32 strobe_squeeze_word(Transcript<S>& t) {
  if (t.pos + 4 <= EG_STROBE_R) {
    const u32 wi = t.pos >> 2, a = (t.pos & 3u) * 8u;
    u32 out;
    if (a == 0) {
      out = t.st.rd(wi);
      t.st.wr(wi, 0u);
    } else {
      const u32 lo = t.st.rd(wi), hi = t.st.rd(wi + 1);
      out = (lo >> a) | (hi << (32u - a));
      t.st.wr(wi, lo & ((1u << a) - 1u));
      t.st.wr(wi + 1, hi & ~((1u << a) - 1u));
    }
    t.pos += 4;
    if (t.pos == EG_STROBE_R) strobe_run_f(t);
    return out;
  }
  u32 v = 0;
#pragma unroll 1
  for (int b = 0; b < 4; ++b) v |= strobe_squeeze_byte(t) << (8 * b);
  return v;
}

template <class S>
EG_HD void strobe_begin_op(Transcript<S>& t, u32 flags) {
  const u32 old_begin = t.pos_begin;
  t.pos_begin = t.pos + 1;
  t.cur_flags = flags;
  strobe_absorb_byte(t, old_begin);
  strobe_absorb_byte(t, flags);
  if ((flags & (4u | 32u)) && t.pos != 0) strobe_run_f(t);   // C or K flag forces a permutation
}

#define EG_FLAG_META_AD (16u | 2u)
#define EG_FLAG_AD 2u
#define EG_FLAG_PRF (1u | 2u | 4u)

// label + LE32(len) as meta-AD (the common prefix of append_message and challenge_bytes)
template <class S>
EG_HD void merlin_frame(Transcript<S>& t, const char* label, int label_len, u32 len) {
  strobe_begin_op(t, EG_FLAG_META_AD);
#pragma unroll 1
  for (int i = 0; i < label_len; ++i) strobe_absorb_byte(t, (u32)(unsigned char)label[i]);
#pragma unroll 1
  for (int i = 0; i < 4; ++i) strobe_absorb_byte(t, (len >> (8 * i)) & 0xffu);   // "more" continuation
}

// Transcript::append_message with the message given as little-endian 32-bit words
template <class S>
EG_HD void merlin_append_words(Transcript<S>& t, const char* label, int label_len, const u32* words, int n_bytes) {
  merlin_frame(t, label, label_len, (u32)n_bytes);
  strobe_begin_op(t, EG_FLAG_AD);
  strobe_absorb_words(t, words, n_bytes);
}
template <class S>
EG_HD void merlin_append_bytes(Transcript<S>& t, const char* label, int label_len, const char* msg, int n_bytes) {
  merlin_frame(t, label, label_len, (u32)n_bytes);
  strobe_begin_op(t, EG_FLAG_AD);
#pragma unroll 1
  for (int i = 0; i < n_bytes; ++i) strobe_absorb_byte(t, (u32)(unsigned char)msg[i]);
}
template <class S>
EG_HD void merlin_append_u64(Transcript<S>& t, const char* label, int label_len, u64 x) {
  const u32 w[2] = {(u32)x, (u32)(x >> 32)};
  merlin_frame(t, label, label_len, 8u);
  strobe_begin_op(t, EG_FLAG_AD);
#pragma unroll 1
  for (int i = 0; i < 8; ++i) strobe_absorb_byte(t, (w[i >> 2] >> (8 * (i & 3))) & 0xffu);
}
// Transcript::challenge_bytes(label, 64) -> 16 words
template <class S>
EG_HD void merlin_challenge64(Transcript<S>& t, const char* label, int label_len, u32 out[16]) {
  merlin_frame(t, label, label_len, 64u);
  strobe_begin_op(t, EG_FLAG_PRF);
#pragma unroll
  for (int w = 0; w < 16; ++w) out[w] = strobe_squeeze_word(t);
}

// Transcript::new(label): STROBE-128 init + "Merlin v1.0" + dom-sep
template <class S>
EG_HD void merlin_init(Transcript<S>& t, const char* label, int label_len) {
#pragma unroll 1
  for (int i = 0; i < 50; ++i) t.st.wr(i, 0u);
  const unsigned char head[18] = {1, EG_STROBE_R + 2, 1, 0, 1, 96, 'S', 'T', 'R', 'O', 'B', 'E', 'v', '1', '.', '0', '.', '2'};
#pragma unroll 1
  for (int i = 0; i < 18; ++i) strobe_xor_byte(t.st, (u32)i, head[i]);
  u64 a[25];
#pragma unroll
  for (int i = 0; i < 25; ++i) a[i] = (u64)t.st.rd(2 * i) | ((u64)t.st.rd(2 * i + 1) << 32);
  keccak_f1600(a);
#pragma unroll
  for (int i = 0; i < 25; ++i) { t.st.wr(2 * i, (u32)a[i]); t.st.wr(2 * i + 1, (u32)(a[i] >> 32)); }
  t.pos = 0; t.pos_begin = 0; t.cur_flags = 0;
  strobe_begin_op(t, EG_FLAG_META_AD);
  const char* proto = "Merlin v1.0";
#pragma unroll 1
  for (int i = 0; i < 11; ++i) strobe_absorb_byte(t, (u32)(unsigned char)proto[i]);
  merlin_append_bytes(t, "dom-sep", 7, label, label_len);
}

// state snapshot <-> 52 words (50 state + pos | pos_begin<<8 | cur_flags<<16, pad)
template <class S>
EG_HD void merlin_export(const Transcript<S>& t, u32 out[52]) {
#pragma unroll 1
  for (int i = 0; i < 50; ++i) out[i] = t.st.rd(i);
  out[50] = t.pos | (t.pos_begin << 8) | (t.cur_flags << 16);
  out[51] = 0;
}
template <class S, class W>
EG_HD void merlin_import(Transcript<S>& t, const W* in) {
#pragma unroll 1
  for (int i = 0; i < 50; ++i) t.st.wr(i, in[i]);
  const u32 m = in[50];
  t.pos = m & 0xffu; t.pos_begin = (m >> 8) & 0xffu; t.cur_flags = (m >> 16) & 0xffu;
}
template <class S>
EG_HD void merlin_clone(Transcript<S>& dst, const Transcript<S>& src) {
#pragma unroll 1
  for (int i = 0; i < 50; ++i) dst.st.wr(i, src.st.rd(i));
  dst.pos = src.pos; dst.pos_begin = src.pos_begin; dst.cur_flags = src.cur_flags;
}

}  // namespace eg

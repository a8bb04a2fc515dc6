// wire_json.hpp -- native wire ingest: the JSON that the reference's serde derives produce in human-readable formats
// (what examples/voting.rs:195-198 prints with serde_json) -> the packed binary layout of include/eg_hip.h.  Host-only C++.
//
// Mirrors, for the ballot types of the hot path:
//   deserialize_bytes / Base64UrlUnpadded (src/serde.rs:29-80): strings are base64url WITHOUT padding; the alphabet, padding and
//     non-zero trailing bits are errors (base64ct is strict), then the byte length must be 32 (ScalarHelper :191-206,
//     ElementHelper :254-269)
//   VecHelper<_, MIN> (:303-355): `ring_responses` and `ciphertext_responses` need at least 2 entries
//   the serde derives of EncryptedChoice (src/app/choice.rs:276-280: choices, range_proof, sum_proof), RingProof
//     (src/proofs/ring.rs:282-287: common_challenge, ring_responses), LogEqualityProof (src/proofs/log_equality.rs:96-101:
//     challenge, response), QuadraticVotingBallot (src/app/quadratic_voting.rs:205-217: votes, credit, credit_equivalence_proof),
//     CiphertextWithRangeProof (ciphertext, range_proof), RangeProof (src/proofs/range.rs:446-450: partial_ciphertexts + flattened
//     RingProof), SumOfSquaresProof (src/proofs/mul.rs:86-93: challenge, ciphertext_responses, sum_response), Ciphertext
//     (src/encryption.rs:96-101: random_element, blinded_element): unknown fields are skipped, missing and duplicate fields fail.
// Canonicity of scalars and validity of group elements are NOT decided here: they need group arithmetic, which only runs on the
// GPU (the verifier reports BadScalar / BadPoint with the item index, where serde would have failed).
//
// Verdict per object: EG_ST_OK (packed), EG_ST_MALFORMED (does not deserialise) or EG_PACK_RESHAPE (deserialises, but the number
// of choices / responses / partial ciphertexts differs from the election's: OptionsLenMismatch / LenMismatch territory, decided
// by the object path in the reference's order of checks).
#pragma once
#include <cstdint>
#include <cstring>
#include <functional>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <string>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

namespace egwire {

struct Cursor {
  const char* p;
  const char* end;
  bool fail = false;
  void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
  bool eat(char c) { ws(); if (p < end && *p == c) { ++p; return true; } return false; }
  bool peek(char c) { ws(); return p < end && *p == c; }
};

struct B64Table {
  uint8_t v[256];
  constexpr B64Table() : v() {
    for (int i = 0; i < 256; ++i) v[i] = 0x80;
    for (int i = 0; i < 26; ++i) { v['A' + i] = (uint8_t)i; v['a' + i] = (uint8_t)(26 + i); }
    for (int i = 0; i < 10; ++i) v['0' + i] = (uint8_t)(52 + i);
    v[(unsigned char)'-'] = 62; v[(unsigned char)'_'] = 63;
  }
};
static constexpr B64Table kB64{};

// A JSON string, unescaped into buf (serde_json unescapes every string and key before it is looked at, so a ballot whose emitter wrote
// "\u0041" for "A" or "\/" for "/" is as valid as the plain one): at most cap - 1 bytes; code points above 0x7f come out as 0xff
// (never part of a field name or of the base64url alphabet).  The slow path of parse_b64_32 / parse_key.
inline bool read_json_string(Cursor& c, char* buf, size_t cap, size_t& len) {
  len = 0;
  if (!c.eat('"')) return false;
  auto hex = [](char ch) { return ch >= '0' && ch <= '9' ? ch - '0' : ch >= 'a' && ch <= 'f' ? ch - 'a' + 10 : ch >= 'A' && ch <= 'F' ? ch - 'A' + 10 : -1; };
  while (c.p < c.end) {
    unsigned char ch = (unsigned char)*c.p++;
    if (ch == '"') { buf[len] = '\0'; return true; }
    if (ch < 0x20) return false;                         // control characters must be escaped in JSON
    if (ch == '\\') {
      if (c.p >= c.end) return false;
      const char e = *c.p++;
      switch (e) {
        case '"': ch = '"'; break;   case '\\': ch = '\\'; break;   case '/': ch = '/'; break;
        case 'b': ch = '\b'; break;  case 'f': ch = '\f'; break;    case 'n': ch = '\n'; break;
        case 'r': ch = '\r'; break;  case 't': ch = '\t'; break;
        case 'u': {
          if (c.end - c.p < 4) return false;
          int v = 0;
          for (int i = 0; i < 4; ++i) { const int h = hex(c.p[i]); if (h < 0) return false; v = v * 16 + h; }
          c.p += 4;
          ch = v < 0x80 ? (unsigned char)v : 0xff;
          break;
        }
        default: return false;
      }
    }
    if (len + 1 >= cap) return false;
    buf[len++] = (char)ch;
  }
  return false;
}

// a JSON string holding exactly 32 bytes of unpadded base64url (43 characters, the 2 trailing bits zero) -> out.
// Anything else in the string - another alphabet, padding, another length - fails.  Escaped characters take the slow path.
// the four characters of a group looked up already shifted into place (bit 31 = not in the alphabet): three ORs make the 24-bit word
struct B64Wide {
  uint32_t t[4][256];
  constexpr B64Wide() : t() {
    for (int k = 0; k < 4; ++k)
      for (int i = 0; i < 256; ++i) t[k][i] = kB64.v[i] == 0x80 ? 0x80000000u : (uint32_t)kB64.v[i] << (18 - 6 * k);
  }
};
static constexpr B64Wide kB64W{};
inline bool decode_b64_43(const unsigned char* s, uint8_t out[32]) {
  uint32_t bad = 0;
  for (int g = 0; g < 10; ++g) {                       // 40 characters -> 30 bytes
    const uint32_t w = kB64W.t[0][s[4 * g]] | kB64W.t[1][s[4 * g + 1]] | kB64W.t[2][s[4 * g + 2]] | kB64W.t[3][s[4 * g + 3]];
    bad |= w;
    out[3 * g] = (uint8_t)(w >> 16); out[3 * g + 1] = (uint8_t)(w >> 8); out[3 * g + 2] = (uint8_t)w;
  }
  // 3 characters -> 2 bytes + 2 bits that must be zero
  const uint32_t w = kB64W.t[1][s[40]] | kB64W.t[2][s[41]] | kB64W.t[3][s[42]];        // 18 bits: a << 12 | b << 6 | d
  bad |= w;
  out[30] = (uint8_t)(w >> 10); out[31] = (uint8_t)(w >> 2);
  return !(bad & 0x80000000u) && !(w & 3u);
}
inline bool parse_b64_32(Cursor& c, uint8_t out[32]) {
  c.ws();
  if (c.p >= c.end || *c.p != '"') return false;
  if (c.end - c.p >= 45 && c.p[44] == '"' && decode_b64_43(reinterpret_cast<const unsigned char*>(c.p + 1), out)) { c.p += 45; return true; }
  // not the plain form: either invalid, or a valid string with escapes - look only if a backslash comes before the closing quote
  const char* q = c.p + 1;
  bool escaped = false;
  while (q < c.end && *q != '"') { if (*q == '\\') { escaped = true; break; } ++q; }
  if (!escaped) return false;
  char buf[48];
  size_t len = 0;
  Cursor t = c;
  if (!read_json_string(t, buf, sizeof buf, len) || len != 43) return false;
  if (!decode_b64_43(reinterpret_cast<const unsigned char*>(buf), out)) return false;
  c.p = t.p;
  return true;
}

// skips any JSON value (for unknown fields); strings honour backslash escapes
inline bool skip_string(Cursor& c) {
  if (!c.eat('"')) return false;
  while (c.p < c.end) {
    const char ch = *c.p++;
    if (ch == '"') return true;
    if (ch == '\\') { if (c.p >= c.end) return false; ++c.p; }
  }
  return false;
}
inline bool skip_value(Cursor& c, int depth = 0) {
  if (depth > 32) return false;
  c.ws();
  if (c.p >= c.end) return false;
  if (*c.p == '"') return skip_string(c);
  if (*c.p == '{' || *c.p == '[') {
    const char close = *c.p == '{' ? '}' : ']';
    const bool obj = *c.p == '{';
    ++c.p;
    if (c.eat(close)) return true;
    for (;;) {
      if (obj) { if (!skip_string(c) || !c.eat(':')) return false; }
      if (!skip_value(c, depth + 1)) return false;
      if (c.eat(',')) continue;
      return c.eat(close);
    }
  }
  const char* s = c.p;
  while (c.p < c.end && *c.p != ',' && *c.p != '}' && *c.p != ']' && *c.p != ' ' && *c.p != '\n' && *c.p != '\t' && *c.p != '\r') ++c.p;
  return c.p > s;
}

// field name of an object member (every name of the wire format is plain ASCII; an escaped spelling of one is unescaped like
// serde_json does); longer names are unknown fields
inline bool parse_key(Cursor& c, char name[32]) {
  c.ws();
  if (c.p >= c.end || *c.p != '"') return false;
  const char* s = c.p + 1;
  const char* q = s;
  bool escaped = false;
  while (q < c.end && *q != '"') { if (*q == '\\') { escaped = true; ++q; if (q >= c.end) return false; } ++q; }
  if (q >= c.end) return false;
  const size_t len = (size_t)(q - s);
  if (escaped) {
    char buf[200];
    size_t n = 0;
    Cursor t = c;
    if (!read_json_string(t, buf, sizeof buf, n)) {       // a well-formed but very long escaped key is an unknown field
      if (!skip_string(c)) return false;
      name[0] = '\0';
      return c.eat(':');
    }
    if (n > 31) name[0] = '\0'; else { memcpy(name, buf, n); name[n] = '\0'; }
    c.p = t.p;
    return c.eat(':');
  }
  if (len > 31) name[0] = '\0';
  else { memcpy(name, s, len); name[len] = '\0'; }
  c.p = q + 1;
  return c.eat(':');
}

// generic object walker: fields[i] names; handler(i) parses the value of field i.  Every field is required exactly once.
template <class Handler>
inline bool parse_object(Cursor& c, const char* const* fields, int n_fields, Handler handler) {
  if (!c.eat('{')) return false;
  uint32_t seen = 0;
  if (!c.peek('}')) {
    for (;;) {
      char name[32];
      if (!parse_key(c, name)) return false;
      int idx = -1;
      for (int i = 0; i < n_fields; ++i) if (strcmp(name, fields[i]) == 0) { idx = i; break; }
      if (idx < 0) { if (!skip_value(c)) return false; }
      else {
        if (seen & (1u << idx)) return false;          // duplicate field
        seen |= 1u << idx;
        if (!handler(idx)) return false;
      }
      if (c.eat(',')) continue;
      break;
    }
  }
  if (!c.eat('}')) return false;
  return seen == (1u << n_fields) - 1u;                // missing field
}

using Bytes = std::vector<uint8_t>;

inline bool parse_item(Cursor& c, Bytes& out) {
  uint8_t b[32];
  if (!parse_b64_32(c, b)) return false;
  out.insert(out.end(), b, b + 32);
  return true;
}
// [ "..", ".." ] of 32-byte items; returns the count
inline bool parse_items(Cursor& c, Bytes& out, size_t& count, size_t minimum) {
  count = 0;
  if (!c.eat('[')) return false;
  if (!c.peek(']')) {
    for (;;) {
      if (!parse_item(c, out)) return false;
      ++count;
      if (c.eat(',')) continue;
      break;
    }
  }
  return c.eat(']') && count >= minimum;
}
inline bool parse_ciphertext(Cursor& c, Bytes& out) {
  static const char* const F[] = {"random_element", "blinded_element"};
  uint8_t r[32], b[32];
  if (!parse_object(c, F, 2, [&](int i) { return parse_b64_32(c, i == 0 ? r : b); })) return false;
  out.insert(out.end(), r, r + 32);
  out.insert(out.end(), b, b + 32);
  return true;
}
inline bool parse_ciphertexts(Cursor& c, Bytes& out, size_t& count) {
  count = 0;
  if (!c.eat('[')) return false;
  if (!c.peek(']')) {
    for (;;) {
      if (!parse_ciphertext(c, out)) return false;
      ++count;
      if (c.eat(',')) continue;
      break;
    }
  }
  return c.eat(']');
}

enum : uint32_t { ST_OK = 0, ST_MALFORMED = 13, PACK_RESHAPE = 0xfffffffeu };

// EncryptedChoice -> choices || e0 || responses [|| c || s]; returns the verdict
inline uint32_t pack_choice(Cursor& c, int n_options, bool single, uint8_t* dst) {
  static const char* const F[] = {"choices", "range_proof", "sum_proof"};
  static const char* const RING[] = {"common_challenge", "ring_responses"};
  static const char* const LOGEQ[] = {"challenge", "response"};
  static thread_local Bytes choices, ring_c, ring_r, sum;      // reused across objects: no allocation per ballot
  choices.clear(); ring_c.clear(); ring_r.clear(); sum.clear();
  size_t n_choices = 0, n_resp = 0;
  bool sum_null = false;
  const bool ok = parse_object(c, F, 3, [&](int i) {
    if (i == 0) return parse_ciphertexts(c, choices, n_choices);
    if (i == 1) return parse_object(c, RING, 2, [&](int k) { return k == 0 ? parse_item(c, ring_c) : parse_items(c, ring_r, n_resp, 2); });
    c.ws();
    if (c.end - c.p >= 4 && memcmp(c.p, "null", 4) == 0) { c.p += 4; sum_null = true; return true; }
    uint8_t ch[32], rs[32];
    if (!parse_object(c, LOGEQ, 2, [&](int k) { return parse_b64_32(c, k == 0 ? ch : rs); })) return false;
    sum.assign(ch, ch + 32); sum.insert(sum.end(), rs, rs + 32);
    return true;
  });
  if (!ok || sum_null == single) return ST_MALFORMED;       // S::Proof is `()` for MultiChoice, a LogEqualityProof for SingleChoice
  if (n_choices != (size_t)n_options || n_resp != 2 * (size_t)n_options) return PACK_RESHAPE;
  uint8_t* d = dst;
  memcpy(d, choices.data(), choices.size()); d += choices.size();
  memcpy(d, ring_c.data(), 32); d += 32;
  memcpy(d, ring_r.data(), ring_r.size()); d += ring_r.size();
  if (single) memcpy(d, sum.data(), 64);
  return ST_OK;
}

struct RangeShape { size_t rings, responses; };   // number of rings and total ring size of a RangeDecomposition

// { "ciphertext": .., "range_proof": { partial_ciphertexts, common_challenge, ring_responses } } -> ct || partials || e0 || responses
inline bool parse_ct_with_range(Cursor& c, Bytes& out, const RangeShape& shape, bool& shape_ok) {
  static const char* const F[] = {"ciphertext", "range_proof"};
  static const char* const RP[] = {"partial_ciphertexts", "common_challenge", "ring_responses"};
  static thread_local Bytes ct, partials, e0, resp;
  ct.clear(); partials.clear(); e0.clear(); resp.clear();
  size_t n_partials = 0, n_resp = 0;
  if (!parse_object(c, F, 2, [&](int i) {
        if (i == 0) return parse_ciphertext(c, ct);
        return parse_object(c, RP, 3, [&](int k) {
          if (k == 0) return parse_ciphertexts(c, partials, n_partials);
          if (k == 1) return parse_item(c, e0);
          return parse_items(c, resp, n_resp, 2);
        });
      }))
    return false;
  if (n_partials != shape.rings - 1 || n_resp != shape.responses) shape_ok = false;
  out.insert(out.end(), ct.begin(), ct.end());
  out.insert(out.end(), partials.begin(), partials.end());
  out.insert(out.end(), e0.begin(), e0.end());
  out.insert(out.end(), resp.begin(), resp.end());
  return true;
}

inline uint32_t pack_qv(Cursor& c, int n_options, const RangeShape& vote, const RangeShape& credit, size_t ballot_size, uint8_t* dst) {
  static const char* const F[] = {"votes", "credit", "credit_equivalence_proof"};
  static const char* const SQ[] = {"challenge", "ciphertext_responses", "sum_response"};
  static thread_local Bytes votes, cred, proof_c, proof_r, proof_s;
  votes.clear(); cred.clear(); proof_c.clear(); proof_r.clear(); proof_s.clear();
  size_t n_votes = 0, n_resp = 0;
  bool shape_ok = true;
  const bool ok = parse_object(c, F, 3, [&](int i) {
    if (i == 0) {
      if (!c.eat('[')) return false;
      if (!c.peek(']')) {
        for (;;) {
          if (!parse_ct_with_range(c, votes, vote, shape_ok)) return false;
          ++n_votes;
          if (c.eat(',')) continue;
          break;
        }
      }
      return c.eat(']');
    }
    if (i == 1) return parse_ct_with_range(c, cred, credit, shape_ok);
    return parse_object(c, SQ, 3, [&](int k) {
      if (k == 0) return parse_item(c, proof_c);
      if (k == 1) return parse_items(c, proof_r, n_resp, 2);
      return parse_item(c, proof_s);
    });
  });
  if (!ok) return ST_MALFORMED;
  if (!shape_ok || n_votes != (size_t)n_options || n_resp != 2 * (size_t)n_options) return PACK_RESHAPE;
  if (votes.size() + cred.size() + 32 + proof_r.size() + 32 != ballot_size) return PACK_RESHAPE;
  uint8_t* d = dst;
  memcpy(d, votes.data(), votes.size()); d += votes.size();
  memcpy(d, cred.data(), cred.size()); d += cred.size();
  memcpy(d, proof_c.data(), 32); d += 32;
  memcpy(d, proof_r.data(), proof_r.size()); d += proof_r.size();
  memcpy(d, proof_s.data(), 32);
  return ST_OK;
}

// ---- object path: ballots that deserialise but do not have the election's shape -----------------------------------------------------
// In the reference such a ballot reaches verify(), which reports OptionsLenMismatch (choice.rs:149-158, quadratic_voting.rs:295) or the
// LenMismatch of the first ill-shaped proof (ring.rs:310-315, range.rs:555-559, mul.rs:197-202) - unless an EARLIER proof fails first,
// or deserialisation already failed on a non-canonical scalar / invalid element (serde.rs:191-206,254-269).  The order is kept here:
//   1. every 32-byte item of the object, in the order of the struct's fields, is checked (GPU: `check_items`); the first bad one wins;
//   2. the options count; 3. the proofs in verify()'s order: the well-shaped ones before the first ill-shaped one ARE verified on the
//   GPU - the object is re-packed with an all-zero proof (which cannot verify) in place of every ill-shaped part (`verify_packed`) and
//   the verdict is read off the status word; if nothing earlier fails, the LenMismatch of the first ill-shaped proof is the verdict.
// Nothing here computes group or field arithmetic; the two callbacks run the product's GPU entry points.
struct AnyItems {                 // 32-byte items in struct-field order, tagged 'P' (group element) or 'S' (scalar)
  std::string kinds;
  Bytes data;
  size_t count() const { return kinds.size(); }
  void add(char kind, const uint8_t* b, size_t n_items) { kinds.append(n_items, kind); data.insert(data.end(), b, b + 32 * n_items); }
};
struct ChoiceAny { size_t n_choices = 0, n_resp = 0; bool sum_null = false; Bytes choices, e0, resp, sum; };
inline bool parse_choice_any(Cursor& c, ChoiceAny& o) {
  static const char* const F[] = {"choices", "range_proof", "sum_proof"};
  static const char* const RING[] = {"common_challenge", "ring_responses"};
  static const char* const LOGEQ[] = {"challenge", "response"};
  return parse_object(c, F, 3, [&](int i) {
    if (i == 0) return parse_ciphertexts(c, o.choices, o.n_choices);
    if (i == 1) return parse_object(c, RING, 2, [&](int k) { return k == 0 ? parse_item(c, o.e0) : parse_items(c, o.resp, o.n_resp, 2); });
    c.ws();
    if (c.end - c.p >= 4 && memcmp(c.p, "null", 4) == 0) { c.p += 4; o.sum_null = true; return true; }
    uint8_t ch[32], rs[32];
    if (!parse_object(c, LOGEQ, 2, [&](int k) { return parse_b64_32(c, k == 0 ? ch : rs); })) return false;
    o.sum.assign(ch, ch + 32); o.sum.insert(o.sum.end(), rs, rs + 32);
    return true;
  });
}
struct RangeAny { Bytes ct, partials, e0, resp; size_t n_partials = 0, n_resp = 0; };
inline bool parse_range_any(Cursor& c, RangeAny& r) {
  static const char* const F[] = {"ciphertext", "range_proof"};
  static const char* const RP[] = {"partial_ciphertexts", "common_challenge", "ring_responses"};
  return parse_object(c, F, 2, [&](int i) {
    if (i == 0) return parse_ciphertext(c, r.ct);
    return parse_object(c, RP, 3, [&](int k) {
      if (k == 0) return parse_ciphertexts(c, r.partials, r.n_partials);
      if (k == 1) return parse_item(c, r.e0);
      return parse_items(c, r.resp, r.n_resp, 2);
    });
  });
}
struct QvAny { std::vector<RangeAny> votes; RangeAny credit; Bytes c, resp, s; size_t n_resp = 0; };
inline bool parse_qv_any(Cursor& c, QvAny& o) {
  static const char* const F[] = {"votes", "credit", "credit_equivalence_proof"};
  static const char* const SQ[] = {"challenge", "ciphertext_responses", "sum_response"};
  return parse_object(c, F, 3, [&](int i) {
    if (i == 0) {
      if (!c.eat('[')) return false;
      if (!c.peek(']')) {
        for (;;) {
          o.votes.emplace_back();
          if (!parse_range_any(c, o.votes.back())) return false;
          if (c.eat(',')) continue;
          break;
        }
      }
      return c.eat(']');
    }
    if (i == 1) return parse_range_any(c, o.credit);
    return parse_object(c, SQ, 3, [&](int k) {
      if (k == 0) return parse_item(c, o.c);
      if (k == 1) return parse_items(c, o.resp, o.n_resp, 2);
      return parse_item(c, o.s);
    });
  });
}

enum : uint32_t { ST_BAD_SCALAR = 1, ST_BAD_POINT = 2, ST_OPTIONS_LEN = 3, ST_SUM_CHALLENGE = 4, ST_RANGE_LEN = 5, ST_QV_VARIANT_LEN = 7,
                  ST_QV_VARIANT_CHALLENGE = 8, ST_QV_CREDIT_RANGE_LEN = 9, ST_QV_CREDIT_RANGE_CHALLENGE = 10, ST_QV_CREDIT_EQUIV_LEN = 11 };
// check_items(kinds, data, ok): ok[i] = item i is a canonical scalar / valid element (one batched GPU call per kind);
// verify_packed(n, packed, status): the batch verifier on n packed ballots of the election's stride (substitutes never verify).
using CheckItemsFn = std::function<bool(const std::string& kinds, const Bytes& data, std::vector<uint8_t>& ok)>;
using VerifyPackedFn = std::function<bool(size_t n, const Bytes& packed, std::vector<uint32_t>& status)>;

// first bad item of every object (item counts in `counts`), or 0 when all are good: BAD_SCALAR / BAD_POINT | index << 8
inline bool first_invalid(const std::vector<AnyItems>& items, const CheckItemsFn& check, std::vector<uint32_t>& out) {
  std::string kinds; Bytes data;
  for (auto& it : items) { kinds += it.kinds; data.insert(data.end(), it.data.begin(), it.data.end()); }
  std::vector<uint8_t> ok;
  if (!kinds.empty() && !check(kinds, data, ok)) return false;
  out.assign(items.size(), 0);
  size_t pos = 0;
  for (size_t k = 0; k < items.size(); ++k) {
    for (size_t i = 0; i < items[k].count(); ++i)
      if (!ok[pos + i] && !out[k]) out[k] = (items[k].kinds[i] == 'P' ? ST_BAD_POINT : ST_BAD_SCALAR) | ((uint32_t)i << 8);
    pos += items[k].count();
  }
  return true;
}

// verdicts of EncryptedChoice objects whose shape is not the election's (text spans `odd`); status[k] for each of them
inline bool resolve_choice_objects(const char* json, const std::vector<std::pair<size_t, size_t>>& odd, int n_options, bool single,
                                   size_t stride, const CheckItemsFn& check, const VerifyPackedFn& verify, std::vector<uint32_t>& status) {
  const size_t n = (size_t)n_options;
  std::vector<ChoiceAny> objs(odd.size());
  std::vector<AnyItems> items(odd.size());
  status.assign(odd.size(), ST_MALFORMED);
  std::vector<char> parsed(odd.size(), 0);
  for (size_t k = 0; k < odd.size(); ++k) {
    Cursor c{json + odd[k].first, json + odd[k].first + odd[k].second};
    ChoiceAny& o = objs[k];
    if (!parse_choice_any(c, o) || o.sum_null == single) continue;
    c.ws();
    if (c.p != c.end) continue;
    parsed[k] = 1;
    AnyItems& it = items[k];
    it.add('P', o.choices.data(), 2 * o.n_choices);
    it.add('S', o.e0.data(), 1);
    it.add('S', o.resp.data(), o.n_resp);
    if (!o.sum_null) it.add('S', o.sum.data(), 2);
  }
  std::vector<uint32_t> bad;
  if (!first_invalid(items, check, bad)) return false;
  Bytes subs; std::vector<size_t> sub_for;
  for (size_t k = 0; k < odd.size(); ++k) {
    if (!parsed[k]) continue;
    const ChoiceAny& o = objs[k];
    if (bad[k]) status[k] = bad[k];
    else if (o.n_choices != n) status[k] = ST_OPTIONS_LEN;
    else if (!single || o.n_resp == 2 * n) status[k] = ST_RANGE_LEN;     // (a well-shaped object never comes here)
    else {                       // the sum proof is verified before the ring proof's length check (choice.rs:363-379)
      const size_t at = subs.size();
      subs.resize(at + stride, 0);
      memcpy(subs.data() + at, o.choices.data(), 64 * n);
      memcpy(subs.data() + at + 64 * n + 32 * (1 + 2 * n), o.sum.data(), 64);
      sub_for.push_back(k);
    }
  }
  if (!sub_for.empty()) {
    std::vector<uint32_t> st;
    if (!verify(sub_for.size(), subs, st)) return false;
    for (size_t i = 0; i < sub_for.size(); ++i) status[sub_for[i]] = (st[i] & 0xffu) == ST_SUM_CHALLENGE ? st[i] : (uint32_t)ST_RANGE_LEN;
  }
  return true;
}

inline bool resolve_qv_objects(const char* json, const std::vector<std::pair<size_t, size_t>>& odd, int n_options, const RangeShape& vote,
                               const RangeShape& credit, size_t stride, const CheckItemsFn& check, const VerifyPackedFn& verify,
                               std::vector<uint32_t>& status) {
  const size_t n = (size_t)n_options;
  std::vector<QvAny> objs(odd.size());
  std::vector<AnyItems> items(odd.size());
  status.assign(odd.size(), ST_MALFORMED);
  std::vector<char> parsed(odd.size(), 0);
  auto add_range = [](AnyItems& it, const RangeAny& r) {
    it.add('P', r.ct.data(), 2); it.add('P', r.partials.data(), 2 * r.n_partials); it.add('S', r.e0.data(), 1); it.add('S', r.resp.data(), r.n_resp);
  };
  for (size_t k = 0; k < odd.size(); ++k) {
    Cursor c{json + odd[k].first, json + odd[k].first + odd[k].second};
    QvAny& o = objs[k];
    if (!parse_qv_any(c, o)) continue;
    c.ws();
    if (c.p != c.end) continue;
    parsed[k] = 1;
    for (auto& v : o.votes) add_range(items[k], v);
    add_range(items[k], o.credit);
    items[k].add('S', o.c.data(), 1); items[k].add('S', o.resp.data(), o.n_resp); items[k].add('S', o.s.data(), 1);
  }
  std::vector<uint32_t> bad;
  if (!first_invalid(items, check, bad)) return false;
  auto shape_ok = [](const RangeAny& r, const RangeShape& sh) { return r.n_partials == sh.rings - 1 && r.n_resp == sh.responses; };
  auto put_range = [](Bytes& out, const RangeAny& r, const RangeShape& sh, bool good) {
    out.insert(out.end(), r.ct.begin(), r.ct.end());
    if (good) {
      out.insert(out.end(), r.partials.begin(), r.partials.end());
      out.insert(out.end(), r.e0.begin(), r.e0.end());
      out.insert(out.end(), r.resp.begin(), r.resp.end());
    } else {
      out.insert(out.end(), 64 * (sh.rings - 1) + 32 * (1 + sh.responses), 0);
    }
  };
  Bytes subs; std::vector<std::pair<size_t, size_t>> sub_for;      // (object, first ill-shaped proof: 0..n-1 votes, n credit, n+1 equivalence)
  for (size_t k = 0; k < odd.size(); ++k) {
    if (!parsed[k]) continue;
    const QvAny& o = objs[k];
    if (bad[k]) { status[k] = bad[k]; continue; }
    if (o.votes.size() != n) { status[k] = ST_OPTIONS_LEN; continue; }
    std::vector<char> good(n + 2);
    for (size_t v = 0; v < n; ++v) good[v] = shape_ok(o.votes[v], vote);
    good[n] = shape_ok(o.credit, credit);
    good[n + 1] = o.n_resp == 2 * n;
    size_t first_bad = 0;
    while (first_bad < n + 2 && good[first_bad]) ++first_bad;
    if (first_bad == n + 2) { status[k] = ST_MALFORMED; continue; }    // well-shaped after all: cannot happen for a RESHAPE verdict
    if (first_bad == 0) { status[k] = ST_QV_VARIANT_LEN; continue; }   // nothing is verified before the first vote's length check
    const size_t at = subs.size();
    for (size_t v = 0; v < n; ++v) put_range(subs, o.votes[v], vote, good[v]);
    put_range(subs, o.credit, credit, good[n]);
    if (good[n + 1]) {
      subs.insert(subs.end(), o.c.begin(), o.c.end()); subs.insert(subs.end(), o.resp.begin(), o.resp.end()); subs.insert(subs.end(), o.s.begin(), o.s.end());
    } else {
      subs.insert(subs.end(), 32 * (2 + 2 * n), 0);
    }
    if (subs.size() != at + stride) return false;
    sub_for.push_back({k, first_bad});
  }
  if (!sub_for.empty()) {
    std::vector<uint32_t> st;
    if (!verify(sub_for.size(), subs, st)) return false;
    for (size_t i = 0; i < sub_for.size(); ++i) {
      const size_t k = sub_for[i].first, first_bad = sub_for[i].second;
      const uint32_t kind = st[i] & 0xffu, detail = st[i] >> 8;
      // position of the reported failure in verify()'s order: votes 0..n-1, credit range, credit equivalence
      const size_t pos = (kind == ST_QV_VARIANT_LEN || kind == ST_QV_VARIANT_CHALLENGE) ? detail : kind == ST_QV_CREDIT_RANGE_CHALLENGE ? n : n + 1;
      if (kind != ST_OK && pos < first_bad) status[k] = st[i];            // an earlier, well-shaped proof fails first
      else if (first_bad < n) status[k] = ST_QV_VARIANT_LEN | ((uint32_t)first_bad << 8);
      else if (first_bad == n) status[k] = ST_QV_CREDIT_RANGE_LEN;
      else status[k] = ST_QV_CREDIT_EQUIV_LEN;
    }
  }
  return true;
}

// Start offsets of the top-level values of a stream: either one JSON array of objects, or objects back to back / one per line.
// Returns false for text that is neither.  Only brace depth and strings are tracked here; the per-object parser does the rest.
inline bool split_objects(const char* s, size_t len, std::vector<std::pair<size_t, size_t>>& spans) {
  size_t i = 0;
  auto ws = [&]() { while (i < len && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r')) ++i; };
  ws();
  const bool array = i < len && s[i] == '[';
  if (array) ++i;
  bool after_comma = false;
  for (;;) {
    ws();
    if (i >= len) return !array;
    if (array && s[i] == ']') { if (after_comma) return false; ++i; ws(); return i == len; }
    if (s[i] != '{') return false;
    const size_t start = i;
    int depth = 0;
    while (i < len) {
      const char ch = s[i];
      if (ch == '"') {                                  // skip the string at memchr speed; a quote after a backslash is escaped
        size_t q = i + 1;
        for (;;) {
          const void* hit = memchr(s + q, '"', len - q);
          if (!hit) return false;
          q = (size_t)((const char*)hit - s);
          size_t bs = 0;
          while (q - 1 - bs > i && s[q - 1 - bs] == '\\') ++bs;
          if ((bs & 1) == 0) break;
          ++q;
        }
        i = q + 1;
        continue;
      }
      ++i;
      if (ch == '{' || ch == '[') ++depth;
      else if (ch == '}' || ch == ']') { if (--depth == 0) break; }
      else if (ch == '\\') return false;                 // a backslash outside a string is not JSON
    }
    if (depth != 0) return false;
    spans.push_back({start, i - start});
    ws();
    after_comma = false;
    if (array) {
      if (i < len && s[i] == ',') { ++i; after_comma = true; continue; }
      if (i < len && s[i] == ']') continue;
      return false;
    }
  }
}

// ---- worker pool: eg_verify_*_json cuts and packs a text window by window; starting and joining `threads` std::threads twice per window
// cost several per cent of the call and left the join waiting for whichever thread shared its core with the GPU-driving thread.  The pool
// keeps threads - 1 workers for the length of a call; run() hands out [0, n) in grains from one atomic counter (the caller works too).
class WorkerPool {
 public:
  explicit WorkerPool(int threads) {
    for (int t = 1; t < threads; ++t) workers_.emplace_back([this]() { loop(); });
  }
  ~WorkerPool() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
    cv_job_.notify_all();
    for (auto& th : workers_) th.join();
  }
  WorkerPool(const WorkerPool&) = delete;
  WorkerPool& operator=(const WorkerPool&) = delete;
  int threads() const { return (int)workers_.size() + 1; }
  // fn(lo, hi) over [0, n) in pieces of `grain`; returns when every piece is done.  One run() at a time.
  void run(size_t n, size_t grain, const std::function<void(size_t, size_t)>& fn) {
    if (n == 0) return;
    if (grain < 1) grain = 1;
    if (workers_.empty() || n <= grain) { fn(0, n); return; }
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = &fn; n_ = n; grain_ = grain; next_.store(0); active_ = workers_.size(); ++generation_;
    }
    cv_job_.notify_all();
    drain();
    std::unique_lock<std::mutex> lk(mu_);
    cv_done_.wait(lk, [this]() { return active_ == 0; });
    fn_ = nullptr;
  }

 private:
  void drain() {
    for (;;) {
      const size_t lo = next_.fetch_add(grain_);
      if (lo >= n_) return;
      (*fn_)(lo, std::min(n_, lo + grain_));
    }
  }
  void loop() {
    size_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_job_.wait(lk, [&]() { return stop_ || generation_ != seen; });
        if (stop_) return;
        seen = generation_;
      }
      drain();
      std::lock_guard<std::mutex> lk(mu_);
      if (--active_ == 0) cv_done_.notify_all();
    }
  }
  std::vector<std::thread> workers_;
  std::mutex mu_;
  std::condition_variable cv_job_, cv_done_;
  const std::function<void(size_t, size_t)>* fn_ = nullptr;
  size_t n_ = 0, grain_ = 1, active_ = 0, generation_ = 0;
  std::atomic<size_t> next_{0};
  bool stop_ = false;
};

// ---- streaming splitter: the text is cut window by window, so that parsing and GPU verification of the first ballots start before
// the last bytes have been looked at (eg_verify_*_json).  Same verdicts as split_objects on the whole text.
struct SplitCursor {
  size_t pos = 0;        // next unread byte (always at the base depth, outside strings)
  size_t count = 0;      // values emitted so far
  bool started = false, array = false, closed = false;
};
// Emits the complete values that start in s[cur.pos, cur.pos + window) and advances the cursor past them; `done` = end of text
// reached (then the tail has been checked as well).  A window that holds no complete value is grown; at most `max_values` values are
// emitted per call (the caller's staging is finite, and a window of `{}` junk holds millions).  false = not a sequence of JSON objects.  Work inside a window: every chunk counts unescaped quotes (-> string state of the next chunks), then records its
// bracket events outside strings; a short sequential walk over the events (about 1 % of the bytes) finds the values.
// `partial` (the streaming entry points, StreamSplitter below): s[0, len) is a PIECE of a longer text.  Reaching its end inside a value,
// or between values, is then not an error: the call returns with done = true and cur.pos at the end of the last complete value, and
// s[cur.pos, len) - separators and / or the beginning of a value - is the caller's to carry over in front of the next piece.
inline bool split_next(const char* s, size_t len, size_t window, int threads, SplitCursor& cur,
                       std::vector<std::pair<size_t, size_t>>& spans, bool& done, size_t max_values = (size_t)-1,
                       WorkerPool* pool = nullptr, bool partial = false) {
  auto is_ws = [](char ch) { return ch == ' ' || ch == '\n' || ch == '\t' || ch == '\r'; };
  done = false;
  if (!cur.started) {
    while (cur.pos < len && is_ws(s[cur.pos])) ++cur.pos;
    if (cur.pos < len && s[cur.pos] == '[') { cur.array = true; ++cur.pos; }
    cur.started = true;
  }
  if (threads < 1) threads = 1;
  for (;;) {
    const size_t a = cur.pos, b = (window >= len - a) ? len : a + window;
    const size_t T = std::max<size_t>(1, std::min<size_t>((size_t)threads * (pool ? 4 : 1), (b - a) / 256 + 1));   // pool: pieces are handed out
    auto lo = [&](size_t t) { return a + (b - a) * t / T; };
    auto run = [&](auto fn) {
      if (pool) { pool->run(T, 1, [&](size_t lo_t, size_t hi_t) { for (size_t t = lo_t; t < hi_t; ++t) fn(t); }); return; }
      std::vector<std::thread> spawned;
      for (size_t t = 1; t < T; ++t) spawned.emplace_back(fn, t);
      fn(0);
      for (auto& th : spawned) th.join();
    };
    // ONE pass per chunk (round 3; two passes + a byte loop before: the splitter, not the parser, bounded eg_verify_*_json).  Whether a
    // chunk starts inside a string is known only after the chunks before it have been counted, so every chunk sorts its brackets by the
    // PARITY of the string delimiters seen so far: list 0 holds the brackets of the even regions (outside strings if the chunk starts
    // outside one), list 1 those of the odd regions; the prefix sum of the delimiter counts then picks one list per chunk.  The bytes
    // are searched 16 at a time (SSE2) for the six characters that matter: " \ { } [ ]
    // Only the brackets at the LOWEST levels of a chunk can be top-level ones (the absolute depth never goes below zero, so the top
    // level of a chunk that starts at depth D is its relative level -D <= the lowest level it reaches): a bracket is recorded only when
    // its level - before an opening one, after a closing one - is at or below the running minimum of its list.  ~2 events per ballot
    // instead of ~20; every list also keeps its total change of depth for the prefix sum.
    struct Ev { size_t pos; int step; int level; };      // +1 open, -1 close, 0 stray backslash; level relative to the chunk's start
    std::vector<size_t> quotes(T, 0);
    std::vector<std::vector<Ev>> ev2(2 * T);
    std::vector<long> delta2(2 * T, 0);
    run([&](size_t t) {
      size_t i = lo(t);
      const size_t e = lo(t + 1);
      size_t skip = (size_t)-1;                // position of a character escaped by the backslash before it
      { size_t bs = 0; while (i > a + bs && s[i - 1 - bs] == '\\') ++bs; if (bs & 1) skip = i; }
      size_t nq = 0;
      unsigned parity = 0;
      std::vector<Ev>* lists[2] = {&ev2[2 * t], &ev2[2 * t + 1]};
      int d[2] = {0, 0}, low[2] = {0, 0};
      auto special = [&](size_t p) {
        if (p == skip) return;
        const char ch = s[p];
        if (ch == '"') { ++nq; parity ^= 1u; }
        else if (ch == '\\') { skip = p + 1; lists[parity]->push_back({p, 0, d[parity]}); }
        else if (ch == '{' || ch == '[') {
          if (d[parity] <= low[parity]) lists[parity]->push_back({p, +1, d[parity]});
          ++d[parity];
        } else {
          --d[parity];
          if (d[parity] <= low[parity]) { lists[parity]->push_back({p, -1, d[parity]}); low[parity] = d[parity]; }
        }
      };
#if defined(__SSE2__)
      const __m128i q = _mm_set1_epi8('"'), b = _mm_set1_epi8('\\'), o1 = _mm_set1_epi8('{'), c1 = _mm_set1_epi8('}'),
                    o2 = _mm_set1_epi8('['), c2 = _mm_set1_epi8(']');
      for (; i + 16 <= e; i += 16) {
        const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s + i));
        const __m128i m = _mm_or_si128(_mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(v, q), _mm_cmpeq_epi8(v, b)),
                                                    _mm_or_si128(_mm_cmpeq_epi8(v, o1), _mm_cmpeq_epi8(v, c1))),
                                       _mm_or_si128(_mm_cmpeq_epi8(v, o2), _mm_cmpeq_epi8(v, c2)));
        unsigned bits = (unsigned)_mm_movemask_epi8(m);
        while (bits) { special(i + (size_t)__builtin_ctz(bits)); bits &= bits - 1; }
      }
#endif
      for (; i < e; ++i) {
        const char ch = s[i];
        if (ch == '"' || ch == '\\' || ch == '{' || ch == '}' || ch == '[' || ch == ']') special(i);
      }
      quotes[t] = nq;
      delta2[2 * t] = d[0]; delta2[2 * t + 1] = d[1];
    });
    std::vector<char> in_str(T, 0);
    for (size_t t = 1; t < T; ++t) in_str[t] = (char)((in_str[t - 1] + quotes[t - 1]) & 1);
    std::vector<const std::vector<Ev>*> events_of(T);
    std::vector<long> depth0(T + 1, 0);          // absolute depth at the start of every chunk (the window starts at the base depth)
    for (size_t t = 0; t < T; ++t) {
      events_of[t] = &ev2[2 * t + (size_t)in_str[t]];
      depth0[t + 1] = depth0[t] + delta2[2 * t + (size_t)in_str[t]];
    }
    // sequential walk over the recorded events
    long depth = depth0[T];                        // depth at the end of the window (used when no event ends the walk early)
    size_t start = 0, prev_end = a, emitted = 0, close_pos = 0;
    bool closed_here = false, bad = false, full = false;
    auto separators_ok = [&](size_t from, size_t to, bool before_value) {
      size_t commas = 0;
      for (size_t i = from; i < to; ++i) {
        if (s[i] == ',') ++commas;
        else if (!is_ws(s[i])) return false;
      }
      if (!before_value) return commas == 0;
      return cur.array ? commas == ((cur.count + emitted) ? 1u : 0u) : commas == 0;
    };
    for (size_t t = 0; t < T && !bad && !closed_here && !full; ++t)
      for (const Ev& e : *events_of[t]) {
        if (e.step == 0) { bad = true; break; }                    // a backslash outside a string is not JSON
        const long level = depth0[t] + e.level;                    // absolute: before an opening bracket, after a closing one
        if (e.step > 0) {
          if (level == 0) {
            if (s[e.pos] != '{' || !separators_ok(prev_end, e.pos, true)) { bad = true; break; }
            start = e.pos;
          }
        } else {
          if (level < 0) {                        // the closing bracket of a top-level array
            if (!cur.array || s[e.pos] != ']' || !separators_ok(prev_end, e.pos, false)) { bad = true; break; }
            closed_here = true; close_pos = e.pos;
            break;
          }
          if (level == 0) {
            spans.push_back({start, e.pos + 1 - start}); ++emitted; prev_end = e.pos + 1;
            if (emitted >= max_values) { full = true; break; }
          }
        }
      }
    if (bad) return false;
    if (full) { cur.count += emitted; cur.pos = prev_end; return true; }     // the rest of the window is looked at again by the next call
    if (closed_here) {
      for (size_t i = close_pos + 1; i < len; ++i) if (!is_ws(s[i])) return false;    // nothing but white space after the array
      cur.closed = true; cur.count += emitted; cur.pos = len; done = true;
      return true;
    }
    if (b == len) {                               // end of text without a closing bracket
      if (partial) { cur.count += emitted; cur.pos = prev_end; done = true; return true; }     // the rest belongs to the next piece
      if (depth != 0 || cur.array) return false;  // truncated value, or an array that never closes
      if (!separators_ok(prev_end, len, false)) return false;
      cur.count += emitted; cur.pos = len; done = true;
      return true;
    }
    if (emitted) { cur.count += emitted; cur.pos = prev_end; return true; }
    window *= 2;                                  // no complete value in the window: look further
  }
}

// The same result as split_objects, computed by `threads` workers: the streaming splitter with one window that spans the whole text
// (round 2 had a three-pass splitter of its own here; the fuzz tests hold both against the sequential split_objects).
inline bool split_objects_parallel(const char* s, size_t len, int threads, std::vector<std::pair<size_t, size_t>>& spans,
                                   size_t min_len = (size_t)1 << 20) {
  if (threads < 2 || len < min_len || len < 4 * (size_t)threads) return split_objects(s, len, spans);
  SplitCursor cur;
  bool done = false;
  return split_next(s, len, len, threads, cur, spans, done) && done;
}

// ---- the text in PIECES (eg_verify_json_begin / _feed / _end; examples/voting.rs:195-198 prints ballots one at a time) ----------------
// Pieces of any size; a value, a string, an escape sequence may straddle any number of piece boundaries.  Every piece is cut by the window
// splitter above in its `partial` mode; what follows the last complete value of a piece (separators, the beginning of a value) is
// carried over - only that, never the piece - and the lexical state at the end of the carry (bracket depth, inside a string, after a
// backslash) is kept, so that the end of the straddling value is found in the next piece by a walk over that value alone.  The buffers
// handed to the window splitter always begin right after a complete value (or at the start of the text), which is what its separator
// rules (one comma between the values of an array, none in a sequence, none before the closing bracket) need in order to hold across
// pieces exactly as they hold inside one text.  emit(base, spans, index of the first value): the values of one window; base is valid only
// during the call.
struct LexState { long depth = 0; bool in_str = false, esc = false; };
inline void lex_advance(const char* s, size_t n, LexState& st) {
  for (size_t i = 0; i < n; ++i) {
    const char ch = s[i];
    if (st.in_str) { if (st.esc) st.esc = false; else if (ch == '\\') st.esc = true; else if (ch == '"') st.in_str = false; continue; }
    if (ch == '"') st.in_str = true;
    else if (ch == '{' || ch == '[') ++st.depth;
    else if (ch == '}' || ch == ']') --st.depth;
  }
}
class StreamSplitter {
 public:
  typedef std::function<bool(const char*, const std::vector<std::pair<size_t, size_t>>&, size_t)> Emit;
  StreamSplitter(int threads, WorkerPool* pool, Emit emit, size_t window = (size_t)96 << 20, size_t max_values = (size_t)-1,
                 size_t max_carry = (size_t)256 << 20)
      : threads_(threads < 1 ? 1 : threads), pool_(pool), emit_(std::move(emit)), window_(window ? window : 1), max_values_(max_values ? max_values : 1),
        max_carry_(max_carry) {}
  bool feed(const char* s, size_t n) {
    if (failed_) return false;
    if (cur_.closed) {
      for (size_t i = 0; i < n; ++i) if (!is_ws(s[i])) return fail("text after the closing bracket of the array");
      return true;
    }
    if (!cur_.started && carry_.empty()) {         // the opening bracket, if any, must be seen by the window splitter: skip leading white space
      while (n && is_ws(*s)) { ++s; --n; }
      if (!n) return true;
    }
    if (!carry_.empty()) {
      // where does the value that straddles the boundary end (or, between values: where does the next value end, or the array close)?
      size_t h = 0;
      bool value_end = false, closer = false;
      for (; h < n; ++h) {
        const char ch = s[h];
        if (st_.in_str) { if (st_.esc) st_.esc = false; else if (ch == '\\') st_.esc = true; else if (ch == '"') st_.in_str = false; continue; }
        if (ch == '"') st_.in_str = true;
        else if (ch == '{' || ch == '[') ++st_.depth;
        else if (ch == '}' || ch == ']') {
          if (st_.depth == 0) { closer = true; ++h; break; }        // the array's closing bracket: taken along, the window splitter checks it
          if (--st_.depth == 0) { value_end = true; ++h; break; }
        }
      }
      if (carry_.size() + h > max_carry_) return fail("a value larger than the carry-over limit");
      carry_.append(s, h);
      if (!value_end && !closer) return true;        // the whole piece is inside the value: wait for more
      std::string x;
      x.swap(carry_);
      st_ = LexState();
      if (!run(x.data(), x.size())) return false;     // ends at a value's end or at the closing bracket: nothing is left over
      s += h; n -= h;
      if (cur_.closed) return feed(s, n);
    }
    return run(s, n);
  }
  // end of the text: nothing may be left but white space, and an array must have been closed
  bool finish() {
    if (failed_) return false;
    if (cur_.array && !cur_.closed) return fail("the array never closes");
    for (char ch : carry_) if (!is_ws(ch)) return fail(st_.depth || st_.in_str ? "the text ends inside a value" : "a separator at the end of the text");
    carry_.clear();
    return true;
  }
  size_t count() const { return cur_.count; }
  bool failed() const { return failed_; }
  const std::string& error() const { return err_; }
  size_t carried() const { return carry_.size(); }

 private:
  static bool is_ws(char ch) { return ch == ' ' || ch == '\n' || ch == '\t' || ch == '\r'; }
  bool fail(const char* why) { failed_ = true; if (err_.empty()) err_ = why; return false; }
  bool run(const char* s, size_t n) {
    cur_.pos = 0;
    for (;;) {
      spans_.clear();
      bool done = false;
      const size_t first = cur_.count;
      if (!split_next(s, n, window_, threads_, cur_, spans_, done, max_values_, pool_, true))
        return fail("the text is neither a JSON array of objects nor a sequence of JSON objects");
      if (!spans_.empty() && !emit_(s, spans_, first)) return fail("");
      if (done) break;
    }
    if (cur_.pos < n) {
      if (carry_.size() + (n - cur_.pos) > max_carry_) return fail("a value larger than the carry-over limit");
      carry_.append(s + cur_.pos, n - cur_.pos);
      lex_advance(s + cur_.pos, n - cur_.pos, st_);
    }
    return true;
  }
  int threads_;
  WorkerPool* pool_;
  Emit emit_;
  size_t window_, max_values_, max_carry_;
  SplitCursor cur_;
  std::string carry_;
  LexState st_;
  std::vector<std::pair<size_t, size_t>> spans_;
  bool failed_ = false;
  std::string err_;
};

template <class PackOne>
inline void pack_parallel(const char* json, const std::vector<std::pair<size_t, size_t>>& spans, size_t stride, int threads,
                          uint8_t* packed, uint32_t* status, PackOne pack_one, WorkerPool* pool = nullptr) {
  const size_t n = spans.size();
  if (threads < 1) threads = 1;
  if ((size_t)threads > n) threads = (int)std::max<size_t>(n, 1);
  auto work = [&](size_t lo, size_t hi) {
    for (size_t k = lo; k < hi; ++k) {
      Cursor c{json + spans[k].first, json + spans[k].first + spans[k].second};
      uint8_t* dst = packed + k * stride;
      uint32_t st = pack_one(c, dst);
      if (st == ST_OK) { c.ws(); if (c.p != c.end) st = ST_MALFORMED; }     // trailing characters after the object
      if (st != ST_OK) memset(dst, 0, stride);
      status[k] = st;
    }
  };
  if (pool) { pool->run(n, 128, work); return; }     // grains of 128 objects: the threads that share a core with others take fewer
  if (threads == 1) { work(0, n); return; }
  std::vector<std::thread> spawned;
  for (int t = 0; t < threads; ++t) spawned.emplace_back(work, n * t / threads, n * (t + 1) / threads);
  for (auto& th : spawned) th.join();
}

}  // namespace egwire

// elastic_elgamal_hip.hpp -- C++17 host-side mirror of the reference's interface for the ballot-verification
// path, header-only on top of the C ABI (eg_hip.h).  The reference is a compiled (Rust) library and no Rust
// toolchain exists in the build image, so this is the compiled-language face of the drop-in boundary; names,
// argument meaning and error variants follow the reference:
//   ChoiceParams::single / ::multi            src/app/choice.rs:160-196
//   EncryptedChoice::verify  (batched)        src/app/choice.rs:358-380      -> verify_batch
//   ChoiceVerificationError                   src/app/choice.rs:407-419
//   QuadraticVotingParams::new                src/app/quadratic_voting.rs:63-76
//   QuadraticVotingBallot::verify (batched)   src/app/quadratic_voting.rs:291-329
//   QuadraticVotingError                      src/app/quadratic_voting.rs:335-354
//   VerificationError                         src/proofs/mod.rs:63-80
//   Ristretto (Group backend, batched)        src/group/ristretto.rs:23-146
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "eg_hip.h"

namespace elastic_elgamal_hip {

using Bytes = std::vector<uint8_t>;
using Scalar = std::array<uint8_t, 32>;    // canonical little-endian, < l
using Element = std::array<uint8_t, 32>;   // canonical ristretto255 encoding

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
inline void check(int rc) { if (rc != EG_OK) throw Error(rc, eg_last_error()); }

class Context {
 public:
  explicit Context(int device = 0) { check(eg_init(device, &ctx_)); }
  ~Context() { eg_destroy(ctx_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  eg_ctx* raw() const { return ctx_; }
 private:
  eg_ctx* ctx_ = nullptr;
};

// VerificationError (proofs/mod.rs:63-80)
enum class VerificationError { ChallengeMismatch, LenMismatch };

// ChoiceVerificationError (choice.rs:407-419) + the deserialisation failures that serde reports earlier
struct ChoiceVerificationError {
  enum Kind { OptionsLenMismatch, Sum, Range, MalformedScalar, MalformedElement } kind;
  VerificationError inner = VerificationError::ChallengeMismatch;
  size_t item = 0;   // index of the malformed 32-byte item
  std::string to_string() const {
    switch (kind) {
      case OptionsLenMismatch: return "number of options in the ballot differs from expected";
      case Sum: return "cannot verify sum proof: restored challenge scalar does not match the one provided in the proof";
      case Range: return "cannot verify range proofs: restored challenge scalar does not match the one provided in the proof";
      case MalformedScalar: return "non-canonical scalar at item " + std::to_string(item);
      default: return "invalid group element at item " + std::to_string(item);
    }
  }
};
inline std::optional<ChoiceVerificationError> choice_error_from_status(uint32_t s) {
  switch (EG_STATUS_KIND(s)) {
    case EG_ST_OK: return std::nullopt;
    case EG_ST_BAD_SCALAR: return ChoiceVerificationError{ChoiceVerificationError::MalformedScalar, {}, EG_STATUS_DETAIL(s)};
    case EG_ST_BAD_POINT: return ChoiceVerificationError{ChoiceVerificationError::MalformedElement, {}, EG_STATUS_DETAIL(s)};
    case EG_ST_OPTIONS_LEN: return ChoiceVerificationError{ChoiceVerificationError::OptionsLenMismatch};
    case EG_ST_SUM_CHALLENGE: return ChoiceVerificationError{ChoiceVerificationError::Sum, VerificationError::ChallengeMismatch};
    case EG_ST_RANGE_LEN: return ChoiceVerificationError{ChoiceVerificationError::Range, VerificationError::LenMismatch};
    default: return ChoiceVerificationError{ChoiceVerificationError::Range, VerificationError::ChallengeMismatch};
  }
}

// QuadraticVotingError (quadratic_voting.rs:335-354)
struct QuadraticVotingError {
  enum Kind { Variant, CreditRange, CreditEquivalence, OptionsLenMismatch, MalformedScalar, MalformedElement } kind;
  size_t index = 0;   // Variant: zero-based option index; Malformed*: item index
  VerificationError inner = VerificationError::ChallengeMismatch;
};
inline std::optional<QuadraticVotingError> qv_error_from_status(uint32_t s) {
  const size_t d = EG_STATUS_DETAIL(s);
  switch (EG_STATUS_KIND(s)) {
    case EG_ST_OK: return std::nullopt;
    case EG_ST_BAD_SCALAR: return QuadraticVotingError{QuadraticVotingError::MalformedScalar, d};
    case EG_ST_BAD_POINT: return QuadraticVotingError{QuadraticVotingError::MalformedElement, d};
    case EG_ST_QV_VARIANT_LEN: return QuadraticVotingError{QuadraticVotingError::Variant, d, VerificationError::LenMismatch};
    case EG_ST_QV_VARIANT_CHALLENGE: return QuadraticVotingError{QuadraticVotingError::Variant, d};
    case EG_ST_QV_CREDIT_RANGE_LEN: return QuadraticVotingError{QuadraticVotingError::CreditRange, 0, VerificationError::LenMismatch};
    case EG_ST_QV_CREDIT_RANGE_CHALLENGE: return QuadraticVotingError{QuadraticVotingError::CreditRange};
    case EG_ST_QV_CREDIT_EQUIV_LEN: return QuadraticVotingError{QuadraticVotingError::CreditEquivalence, 0, VerificationError::LenMismatch};
    case EG_ST_QV_CREDIT_EQUIV_CHALLENGE: return QuadraticVotingError{QuadraticVotingError::CreditEquivalence};
    default: return QuadraticVotingError{QuadraticVotingError::OptionsLenMismatch};
  }
}

// Ciphertext (encryption.rs:96-101): random_element || blinded_element
struct Ciphertext { Element random_element{}, blinded_element{}; };

template <class E>
struct BatchVerdict {
  std::vector<std::optional<E>> results;   // per ballot: nullopt == Ok(..)
  std::vector<Ciphertext> totals;          // homomorphic sum of the ciphertexts of accepted ballots, per option
  size_t accepted() const { size_t n = 0; for (auto& r : results) n += !r.has_value(); return n; }
};

inline std::vector<Ciphertext> unpack_totals(const Bytes& t) {
  std::vector<Ciphertext> out(t.size() / 64);
  for (size_t k = 0; k < out.size(); ++k) {
    std::copy(t.begin() + 64 * k, t.begin() + 64 * k + 32, out[k].random_element.begin());
    std::copy(t.begin() + 64 * k + 32, t.begin() + 64 * k + 64, out[k].blinded_element.begin());
  }
  return out;
}

// ChoiceParams<G, S> (choice.rs:132-196); S is SingleChoice (sum proof) or MultiChoice
class ChoiceParams {
 public:
  static ChoiceParams single(const Context& ctx, const Element& receiver, size_t options_count) { return ChoiceParams(ctx, receiver, options_count, true); }
  static ChoiceParams multi(const Context& ctx, const Element& receiver, size_t options_count) { return ChoiceParams(ctx, receiver, options_count, false); }
  ~ChoiceParams() { eg_choice_params_destroy(p_); }
  ChoiceParams(ChoiceParams&& o) noexcept : p_(o.p_), n_(o.n_), single_(o.single_) { o.p_ = nullptr; }
  ChoiceParams(const ChoiceParams&) = delete;
  ChoiceParams& operator=(const ChoiceParams&) = delete;
  ChoiceParams& operator=(ChoiceParams&&) = delete;
  size_t options_count() const { return n_; }
  size_t ballot_size() const { return eg_choice_ballot_size((int)n_, single_); }
  // EncryptedChoice::verify for every packed ballot + totals[k] += vote[k] (examples/voting.rs:199-203)
  BatchVerdict<ChoiceVerificationError> verify_batch(const Bytes& packed) const {
    const size_t n = packed.size() / ballot_size();
    if (n * ballot_size() != packed.size())   // the analogue of check_options_count (choice.rs:149-158)
      throw Error(EG_ERR_BAD_ARG, "packed length is not a whole number of ballots for these parameters");
    std::vector<uint32_t> st(n);
    Bytes tally(64 * n_);
    check(eg_verify_choice_batch(p_, n, packed.data(), st.data(), tally.data()));
    BatchVerdict<ChoiceVerificationError> v;
    for (uint32_t s : st) v.results.push_back(choice_error_from_status(s));
    v.totals = unpack_totals(tally);
    return v;
  }
  // EncryptedChoice::new for synthetic voters base_seed + first + i (choice.rs:313-349), packed
  Bytes encrypt_batch(uint64_t base_seed, size_t first, size_t n, int n_selected = 0) const {
    Bytes out(n * ballot_size());
    check(eg_choice_encrypt_batch(p_, base_seed, first, n, n_selected, out.data()));
    return out;
  }
  // EncryptedChoice::single(&params, choice, rng) with the caller's choices (choice.rs:296-311); voter i draws from
  // ChaChaRng::seed_from_u64(base_seed + first + i).  VARIABLE TIME in the choice (see eg_hip.h): test / synthetic data only.
  Bytes encrypt_single_choices(uint64_t base_seed, size_t first, const std::vector<size_t>& choices) const {
    const size_t words = (n_ + 31) / 32;
    std::vector<uint32_t> sel(choices.size() * words, 0u);
    for (size_t i = 0; i < choices.size(); ++i) {
      if (choices[i] >= n_) throw Error(EG_ERR_BAD_ARG, "choice out of range");
      sel[i * words + choices[i] / 32] |= 1u << (choices[i] % 32);
    }
    Bytes out(choices.size() * ballot_size());
    check(eg_choice_encrypt_selected_batch(p_, base_seed, first, choices.size(), 0, sel.data(), out.data()));
    return out;
  }
  eg_choice_params* raw() const { return p_; }
 private:
  ChoiceParams(const Context& ctx, const Element& pk, size_t n, bool single) : n_(n), single_(single) {
    check(eg_choice_params_create(ctx.raw(), pk.data(), (int)n, single, &p_));
  }
  eg_choice_params* p_ = nullptr;
  size_t n_;
  int single_;
};

class QuadraticVotingParams {
 public:
  QuadraticVotingParams(const Context& ctx, const Element& receiver, size_t options, uint64_t credits) : n_(options), credits_(credits) {
    check(eg_qv_params_create(ctx.raw(), receiver.data(), (int)options, credits, &p_));
  }
  ~QuadraticVotingParams() { eg_qv_params_destroy(p_); }
  QuadraticVotingParams(const QuadraticVotingParams&) = delete;
  QuadraticVotingParams& operator=(const QuadraticVotingParams&) = delete;
  size_t options_count() const { return n_; }
  size_t ballot_size() const { return eg_qv_ballot_size(p_); }
  uint64_t credit_amount() const { return credits_; }
  // QuadraticVotingParams::max_votes (quadratic_voting.rs:84-87): isqrt(credit_amount)
  uint64_t max_votes() const { uint64_t r = 0; while ((r + 1) * (r + 1) <= credits_) ++r; return r; }
  BatchVerdict<QuadraticVotingError> verify_batch(const Bytes& packed) const {
    const size_t n = packed.size() / ballot_size();
    if (n * ballot_size() != packed.size()) throw Error(EG_ERR_BAD_ARG, "packed length is not a whole number of ballots");
    std::vector<uint32_t> st(n);
    Bytes tally(64 * n_);
    check(eg_verify_qv_batch(p_, n, packed.data(), st.data(), tally.data()));
    BatchVerdict<QuadraticVotingError> v;
    for (uint32_t s : st) v.results.push_back(qv_error_from_status(s));
    v.totals = unpack_totals(tally);
    return v;
  }
  // QuadraticVotingBallot::new(&params, votes, rng) for voters base_seed + first + i (quadratic_voting.rs:234-284); votes:
  // options_count() numbers per voter with sum(v^2) <= credits.  VARIABLE TIME in the votes (see eg_hip.h): test / synthetic data only.
  Bytes encrypt_votes_batch(uint64_t base_seed, size_t first, const std::vector<uint32_t>& votes) const {
    if (votes.size() % n_) throw Error(EG_ERR_BAD_ARG, "votes is not a whole number of ballots");
    const size_t n = votes.size() / n_;
    Bytes out(n * ballot_size());
    check(eg_qv_encrypt_votes_batch(p_, base_seed, first, n, 0, votes.data(), out.data()));
    return out;
  }
  eg_qv_params* raw() const { return p_; }
 private:
  eg_qv_params* p_ = nullptr;
  size_t n_;
  uint64_t credits_ = 0;
};

// One batch over several GPUs of this process (eg_verify_*_batch_multi): per_device[d] = the election's params created on the context
// of GPU d; contiguous slabs, one host thread per GPU inside the library, the slabs' tallies merged in the library.  What a
// single-process host like examples/voting.rs:179-213 calls when the node has more than one GPU.
inline BatchVerdict<ChoiceVerificationError> verify_batch_multi(const std::vector<const ChoiceParams*>& per_device, const Bytes& packed) {
  if (per_device.empty()) throw Error(EG_ERR_BAD_ARG, "no params objects");
  const size_t bs = per_device[0]->ballot_size(), n = packed.size() / bs;
  if (n * bs != packed.size()) throw Error(EG_ERR_BAD_ARG, "packed length is not a whole number of ballots for these parameters");
  std::vector<eg_choice_params*> raw;
  for (auto* p : per_device) raw.push_back(p->raw());
  std::vector<uint32_t> st(n);
  Bytes tally(64 * per_device[0]->options_count());
  check(eg_verify_choice_batch_multi(raw.data(), (int)raw.size(), n, packed.data(), st.data(), tally.data()));
  BatchVerdict<ChoiceVerificationError> v;
  for (uint32_t x : st) v.results.push_back(choice_error_from_status(x));
  v.totals = unpack_totals(tally);
  return v;
}
inline BatchVerdict<QuadraticVotingError> verify_batch_multi(const std::vector<const QuadraticVotingParams*>& per_device, const Bytes& packed) {
  if (per_device.empty()) throw Error(EG_ERR_BAD_ARG, "no params objects");
  const size_t bs = per_device[0]->ballot_size(), n = packed.size() / bs;
  if (n * bs != packed.size()) throw Error(EG_ERR_BAD_ARG, "packed length is not a whole number of ballots");
  std::vector<eg_qv_params*> raw;
  for (auto* p : per_device) raw.push_back(p->raw());
  std::vector<uint32_t> st(n);
  Bytes tally(64 * per_device[0]->options_count());
  check(eg_verify_qv_batch_multi(raw.data(), (int)raw.size(), n, packed.data(), st.data(), tally.data()));
  BatchVerdict<QuadraticVotingError> v;
  for (uint32_t x : st) v.results.push_back(qv_error_from_status(x));
  v.totals = unpack_totals(tally);
  return v;
}

// The same with every slab already resident on its GPU (eg_verify_*_batch_multi_device): slab d = counts[d] packed ballots at the device
// pointer d_ballots[d] on the device of per_device[d], verdicts (uint32 words) to d_status[d], streams[d] a hipStream_t of that device
// (empty vector: the null streams).  Returns the tally of this batch; every params object's running tally advances by its slab.  A
// failure in any slab throws and leaves every running tally as it was (eg_hip.h: "AFTER A FAILURE").
inline std::vector<Ciphertext> verify_batch_multi_device(const std::vector<const ChoiceParams*>& per_device, const std::vector<size_t>& counts,
                                                         const std::vector<const void*>& d_ballots, const std::vector<void*>& d_status,
                                                         const std::vector<void*>& streams = {}) {
  const size_t k = per_device.size();
  if (!k || counts.size() != k || d_ballots.size() != k || d_status.size() != k || (!streams.empty() && streams.size() != k))
    throw Error(EG_ERR_BAD_ARG, "one count, ballot pointer, status pointer (and stream) per params object");
  std::vector<eg_choice_params*> raw;
  for (auto* p : per_device) raw.push_back(p->raw());
  Bytes tally(64 * per_device[0]->options_count());
  check(eg_verify_choice_batch_multi_device(raw.data(), (int)k, counts.data(), d_ballots.data(), d_status.data(),
                                            streams.empty() ? nullptr : streams.data(), tally.data()));
  return unpack_totals(tally);
}
inline std::vector<Ciphertext> verify_batch_multi_device(const std::vector<const QuadraticVotingParams*>& per_device, const std::vector<size_t>& counts,
                                                         const std::vector<const void*>& d_ballots, const std::vector<void*>& d_status,
                                                         const std::vector<void*>& streams = {}) {
  const size_t k = per_device.size();
  if (!k || counts.size() != k || d_ballots.size() != k || d_status.size() != k || (!streams.empty() && streams.size() != k))
    throw Error(EG_ERR_BAD_ARG, "one count, ballot pointer, status pointer (and stream) per params object");
  std::vector<eg_qv_params*> raw;
  for (auto* p : per_device) raw.push_back(p->raw());
  Bytes tally(64 * per_device[0]->options_count());
  check(eg_verify_qv_batch_multi_device(raw.data(), (int)k, counts.data(), d_ballots.data(), d_status.data(),
                                        streams.empty() ? nullptr : streams.data(), tally.data()));
  return unpack_totals(tally);
}

// Wire ingest (src/serde.rs:19-80,179-355): ballots as JSON text in the reference's serde layout -> packed ballots, on the host.
// status[k]: EG_ST_OK (packed[k] valid), EG_ST_MALFORMED (does not deserialise) or EG_PACK_RESHAPE (wrong number of choices /
// responses / partial ciphertexts for this election: OptionsLenMismatch / LenMismatch territory).
struct PackedJson {
  Bytes packed;                  // n objects x ballot size; rejected slots are zero
  std::vector<uint32_t> status;
  size_t ballot_size = 0;
  Bytes accepted() const {       // the packed ballots with status OK, back to back (what verify_batch takes)
    Bytes out;
    for (size_t k = 0; k < status.size(); ++k)
      if (status[k] == EG_ST_OK) out.insert(out.end(), packed.begin() + k * ballot_size, packed.begin() + (k + 1) * ballot_size);
    return out;
  }
};
inline PackedJson pack_choice_json(size_t options, bool single, const std::string& text, int threads = 1) {
  PackedJson r;
  r.ballot_size = eg_choice_ballot_size((int)options, single);
  size_t n = 0;
  // first call sizes the output: with max_objects = 0 the library only counts (EG_ERR_BAD_ARG says "more objects than max_objects")
  const int rc = eg_choice_pack_json((int)options, single, text.data(), text.size(), threads, 0, nullptr, nullptr, &n);
  if (rc != EG_OK && n == 0) throw Error(rc, eg_last_error());
  r.packed.resize(n * r.ballot_size);
  r.status.resize(n);
  if (n) check(eg_choice_pack_json((int)options, single, text.data(), text.size(), threads, n, r.packed.data(), r.status.data(), &n));
  return r;
}
inline PackedJson pack_qv_json(size_t options, uint64_t credits, const std::string& text, int threads = 1) {
  PackedJson r;
  r.ballot_size = eg_qv_ballot_size_for((int)options, credits);
  size_t n = 0;
  const int rc = eg_qv_pack_json((int)options, credits, text.data(), text.size(), threads, 0, nullptr, nullptr, &n);
  if (rc != EG_OK && n == 0) throw Error(rc, eg_last_error());
  r.packed.resize(n * r.ballot_size);
  r.status.resize(n);
  if (n) check(eg_qv_pack_json((int)options, credits, text.data(), text.size(), threads, n, r.packed.data(), r.status.data(), &n));
  return r;
}

// The JSON text of a batch of ballots in PIECES (eg_verify_*_json_begin / eg_verify_json_feed / _take / _end): what a host does with
// ballots that arrive one at a time or in network-sized chunks (examples/voting.rs:195-198).  Pieces of any size, a ballot may straddle
// them.  RAII: a stream that is not finished is aborted (the params object's running tally is then what it was before).
class JsonStream {
 public:
  JsonStream(const ChoiceParams& p, int threads = 1) : options_(p.options_count()) { check(eg_verify_choice_json_begin(p.raw(), threads, &s_)); }
  JsonStream(const QuadraticVotingParams& p, int threads = 1) : options_(p.options_count()) { check(eg_verify_qv_json_begin(p.raw(), threads, &s_)); }
  // several params objects of ONE election, one per GPU (eg_verify_*_json_begin_multi): one parser, its packed windows dealt to the GPUs by
  // load, verdicts in text order; every params object tallies the windows it verified, finish() reports the text's tally
  JsonStream(const std::vector<const ChoiceParams*>& per_device, int threads = 1) : options_(per_device.at(0)->options_count()) {
    std::vector<eg_choice_params*> raw;
    for (auto* p : per_device) raw.push_back(p->raw());
    check(eg_verify_choice_json_begin_multi(raw.data(), (int)raw.size(), threads, &s_));
  }
  JsonStream(const std::vector<const QuadraticVotingParams*>& per_device, int threads = 1) : options_(per_device.at(0)->options_count()) {
    std::vector<eg_qv_params*> raw;
    for (auto* p : per_device) raw.push_back(p->raw());
    check(eg_verify_qv_json_begin_multi(raw.data(), (int)raw.size(), threads, &s_));
  }
  JsonStream(const JsonStream&) = delete;
  JsonStream& operator=(const JsonStream&) = delete;
  ~JsonStream() { if (s_) eg_verify_json_abort(s_); }
  // the next piece; returns the number of complete objects seen so far
  size_t feed(const char* text, size_t len) { size_t n = 0; check(eg_verify_json_feed(s_, text, len, &n)); objects_ = n; return n; }
  size_t feed(const std::string& piece) { return feed(piece.data(), piece.size()); }
  // a block handed over WITHOUT a copy (eg_verify_json_feed_owned): the stream keeps the shared buffer alive until the library gives it back
  size_t feed_owned(std::shared_ptr<const std::string> block) {
    auto* keep = new std::shared_ptr<const std::string>(std::move(block));
    size_t n = 0;
    const int rc = eg_verify_json_feed_owned(s_, (*keep)->data(), (*keep)->size(),
                                             [](void* user, const char*, size_t) { delete static_cast<std::shared_ptr<const std::string>*>(user); }, keep, &n);
    if (rc != EG_OK) { delete keep; check(rc); }          // a failed call leaves the block with the caller
    objects_ = n;
    return n;
  }
  // status words that are final so far, in order (never blocks)
  std::vector<uint32_t> take(size_t cap = 1 << 20) {
    std::vector<uint32_t> st(cap);
    size_t n = 0;
    check(eg_verify_json_take(s_, st.data(), cap, &n));
    st.resize(n);
    return st;
  }
  // flushes: the status words not yet taken; totals = the tally of the stream's ballots (the running tally of the params has them added)
  std::vector<uint32_t> finish(std::vector<Ciphertext>* totals = nullptr) {
    std::vector<uint32_t> st(objects_ + 1024);
    Bytes tally(64 * options_);
    size_t n = 0, total = 0;
    for (;;) {
      const int rc = eg_verify_json_end(s_, st.data(), st.size(), &n, &total, tally.data());
      if (rc == EG_OK) break;
      if (rc == EG_ERR_BAD_ARG && n > st.size()) { st.resize(n); continue; }     // more verdicts than feed() had reported yet: the stream is still open
      const std::string why = eg_last_error();
      s_ = nullptr;                // every other error: the library has destroyed the stream
      throw Error(rc, why);
    }
    s_ = nullptr;
    st.resize(n);
    objects_ = total;
    if (totals) *totals = unpack_totals(tally);
    return st;
  }
  size_t objects() const { return objects_; }
 private:
  eg_json_stream* s_ = nullptr;
  size_t options_, objects_ = 0;
};

// Ristretto: the Group backend (ristretto.rs), one problem per call shown here; *_batch in eg_hip.h for many.
// The reference's typed Elements cannot be invalid; here elements are byte strings, so the operations that take
// elements throw Error(EG_ERR_BAD_ARG) when an operand is not a valid ristretto255 encoding instead of silently
// treating it as the identity.
struct Ristretto {
  const Context& ctx;
  static void valid(uint8_t ok) { if (!ok) throw Error(EG_ERR_BAD_ARG, "operand is not a valid ristretto255 encoding"); }
  Scalar scalar_from_random_bytes(const std::array<uint8_t, 64>& wide) const { Scalar s; check(eg_scalar_from_wide_batch(ctx.raw(), 1, wide.data(), s.data())); return s; }
  std::optional<Scalar> deserialize_scalar(const Scalar& b) const { uint8_t ok = 0; check(eg_scalar_is_canonical_batch(ctx.raw(), 1, b.data(), &ok)); return ok ? std::optional<Scalar>(b) : std::nullopt; }
  std::optional<Element> deserialize_element(const Element& b) const { Element o; uint8_t ok = 0; check(eg_point_roundtrip_batch(ctx.raw(), 1, b.data(), o.data(), &ok)); return ok ? std::optional<Element>(o) : std::nullopt; }
  Element mul_generator(const Scalar& k) const { Element o; check(eg_mul_generator_batch(ctx.raw(), 1, k.data(), o.data())); return o; }
  Element vartime_double_mul_generator(const Scalar& k, const Element& p, const Scalar& r) const { Element o; uint8_t ok = 0; check(eg_vartime_double_mul_generator_batch(ctx.raw(), 1, k.data(), p.data(), r.data(), o.data(), &ok)); valid(ok); return o; }
  Element vartime_multi_mul(const std::vector<Scalar>& s, const std::vector<Element>& e) const {
    if (s.size() != e.size()) throw Error(EG_ERR_BAD_ARG, "scalars and elements differ in number");
    Bytes sb, eb; for (auto& x : s) sb.insert(sb.end(), x.begin(), x.end()); for (auto& x : e) eb.insert(eb.end(), x.begin(), x.end());
    Element o; uint8_t ok; check(eg_vartime_multi_mul_batch(ctx.raw(), 1, s.size(), sb.data(), eb.data(), o.data(), &ok)); valid(ok); return o;
  }
  // Element * &Scalar (ElementOps::Element: Mul<&Scalar>, group/mod.rs:136-143): one-term vartime_multi_mul.  VARIABLE TIME.
  Element mul(const Element& p, const Scalar& k) const { return vartime_multi_mul({k}, {p}); }
  // Scalar::from(u64) (ScalarOps::Scalar: From<u64>, group/mod.rs:70-80): the little-endian integer, canonical by construction
  static Scalar scalar_from_u64(uint64_t x) { Scalar s{}; for (int i = 0; i < 8; ++i) s[i] = (uint8_t)(x >> (8 * i)); return s; }
  Element add(const Element& a, const Element& b) const { Element o; uint8_t ok = 0; check(eg_point_add_batch(ctx.raw(), 1, a.data(), b.data(), 0, o.data(), &ok)); valid(ok); return o; }
  Element sub(const Element& a, const Element& b) const { Element o; uint8_t ok = 0; check(eg_point_add_batch(ctx.raw(), 1, a.data(), b.data(), 1, o.data(), &ok)); valid(ok); return o; }
  Element neg(const Element& a) const { return sub(identity(), a); }
  static Element identity() { return Element{}; }   // 32 zero bytes
  bool is_identity(const Element& a) const { uint8_t f = 0, ok = 0; check(eg_point_is_identity_batch(ctx.raw(), 1, a.data(), &f, &ok)); return f != 0; }
  Scalar invert_scalar(const Scalar& a) const { Scalar o; check(eg_scalar_invert_batch(ctx.raw(), 1, a.data(), o.data())); return o; }
  Scalar scalar_muladd(const Scalar& a, const Scalar& b, const Scalar& c) const { Scalar o; check(eg_scalar_muladd_batch(ctx.raw(), 1, a.data(), b.data(), c.data(), o.data())); return o; }
  Scalar scalar_neg(const Scalar& a) const { Scalar o; check(eg_scalar_neg_batch(ctx.raw(), 1, a.data(), o.data())); return o; }
};

// ---- tally stage (examples/voting.rs:122-177) ------------------------------------------------------------------------------------
// Params::combine_shares (sharing/mod.rs:302-325): the first `threshold` of the given (participant index, dh element) pairs ->
// the combined decryption [x]R, or nullopt when there are too few shares.  Verify the shares first (eg_share_params_create).
inline std::optional<Element> combine_shares(const Context& ctx, uint64_t shares, uint64_t threshold,
                                             const std::vector<std::pair<uint64_t, Element>>& given) {
  std::vector<uint64_t> idx; Bytes dh;
  for (auto& g : given) { idx.push_back(g.first); dh.insert(dh.end(), g.second.begin(), g.second.end()); }
  Element out; int combined = 0;
  check(eg_combine_shares(ctx.raw(), shares, threshold, idx.size(), idx.data(), dh.data(), out.data(), &combined));
  return combined ? std::optional<Element>(out) : std::nullopt;
}
// DiscreteLogTable (encryption.rs:260-298)
class DiscreteLogTable {
 public:
  DiscreteLogTable(const Context& ctx, const std::vector<uint64_t>& values) { check(eg_dlog_table_create(ctx.raw(), values.size(), values.data(), &t_)); }
  ~DiscreteLogTable() { eg_dlog_table_destroy(t_); }
  DiscreteLogTable(const DiscreteLogTable&) = delete;
  DiscreteLogTable& operator=(const DiscreteLogTable&) = delete;
  std::optional<uint64_t> get(const Element& decrypted_element) const {
    uint64_t v = 0; uint8_t found = 0;
    check(eg_dlog_table_get(t_, 1, decrypted_element.data(), &v, &found));
    return found ? std::optional<uint64_t>(v) : std::nullopt;
  }
 private:
  eg_dlog_table* t_ = nullptr;
};

}  // namespace elastic_elgamal_hip

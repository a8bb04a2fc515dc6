// plancheck.cpp -- TEST-ONLY build of the product's pure-host logic under AddressSanitizer + UBSan: the election parameter
// mirrors and plan builders (csrc/host_plan.hpp: RangeDecomposition::optimal, every build_*_plan, the flattening the engine
// uploads and its index check) and the native wire ingest (csrc/wire_json.hpp).  The product never loads this library.
#include <stdint.h>
#include <string.h>
#include <functional>
#include <string>
#include "../../elastic_elgamal_amd/csrc/host_plan.hpp"
#include "../../elastic_elgamal_amd/csrc/wire_json.hpp"

using namespace eghost;

static int check(Plan&& P, const uint8_t pk[32]) {
  if (P.pk_off >= 0) memcpy(P.blob.data() + P.pk_off, pk, 32);
  const FlatPlan F = flatten_plan(P);
  const std::string why = check_flat_plan(P, F);
  return why.empty() ? (int)F.jobs.size() : -1;
}

extern "C" {
// builds, flattens and checks a plan; returns its number of equations or -1 if the index check fails
int pc_plan(int kind, int n, unsigned long long v) {
  uint8_t pk[32]; memset(pk, 7, 32);
  size_t item = 0;
  switch (kind) {
    case 0: return check(build_choice_plan(n, true, v ? (int)v - 1 : -1), pk);      // v = rings per group of the ring-group walk + 1 (0: default)
    case 1: return check(build_choice_plan(n, false, v ? (int)v - 1 : -1), pk);
    case 2: return check(build_qv_plan(n, v), pk);
    case 3: return check(build_zero_plan(), pk);
    case 4: return check(build_bool_plan(), pk);
    case 5: return check(build_range_plan(v, &item), pk);
    case 6: return check(build_sumsq_plan(n, "test", &item), pk);
    case 7: return check(build_share_plan(10, 7, pk, (unsigned long long)n), pk);
    default: return -2;
  }
}
// A grouped choice plan with ONE table-backed term bent on purpose, so that the safety net of the ring-group walk can be seen to
// catch it: mutation 0 = untouched (must pass), 1 = a term's point slot replaced by the base of ANOTHER group that sits in
// the same table slot (passes every range check, would read a different ring's table), 2 = a term's table slot replaced by another slot
// of its own group.  Returns 0 if check_flat_plan accepts the plan, 1 if it refuses it, -2 if the plan offers no such term.
int pc_plan_mutated(int n, int single, int ring_group, int mutation, char* why_out, int cap) {
  uint8_t pk[32]; memset(pk, 7, 32);
  Plan P = build_choice_plan(n, single != 0, ring_group);
  if (P.pk_off >= 0) memcpy(P.blob.data() + P.pk_off, pk, 32);
  if (!P.grouped()) return -2;
  bool done = mutation == 0;
  for (auto& v : P.vterms) {
    if (done) break;
    if (v.base == 0xffff) continue;
    auto it = P.base_index_of.find(v.slot);
    if (it == P.base_index_of.end()) continue;
    const uint16_t i = it->second;
    for (size_t k = 0; k < P.base_slots.size() && !done; ++k) {
      if (mutation == 1 && P.base_group[k] != P.base_group[i] && P.base_local[k] == P.base_local[i]) { v.slot = P.base_slots[k]; done = true; }
      if (mutation == 2 && P.base_group[k] == P.base_group[i] && P.base_local[k] != P.base_local[i]) { v.base = P.base_local[k]; done = true; }
    }
  }
  if (!done) return -2;
  const FlatPlan F = flatten_plan(P);
  const std::string why = check_flat_plan(P, F);
  if (why_out && cap > 0) { strncpy(why_out, why.c_str(), cap - 1); why_out[cap - 1] = 0; }
  return why.empty() ? 0 : 1;
}
// The streaming splitter (egwire::StreamSplitter: eg_verify_json_begin / _feed / _end) against the one-shot splitter: the text is fed in
// pieces cut at the given offsets; the values it emits, in order, must be byte for byte the values split_objects finds in the whole text,
// and it must refuse a text (at any piece, or at finish) exactly when split_objects refuses it.  Returns the number of values, -1 if both
// refuse, -7 on any disagreement.  window / max_values small: many windows per piece.
int pc_stream_split(const char* json, size_t len, const size_t* cuts, size_t n_cuts, int threads, size_t window, size_t max_values) {
  std::vector<std::pair<size_t, size_t>> want;
  const bool want_ok = egwire::split_objects(json, len, want);
  std::vector<std::string> got;
  bool index_ok = true;
  egwire::WorkerPool pool(threads);
  egwire::StreamSplitter sp(threads, &pool, [&](const char* base, const std::vector<std::pair<size_t, size_t>>& spans, size_t first) {
    if (first != got.size()) index_ok = false;
    for (auto& sp : spans) got.emplace_back(base + sp.first, sp.second);
    return true;
  }, window, max_values);
  bool ok = true;
  size_t at = 0;
  for (size_t k = 0; k <= n_cuts && ok; ++k) {
    const size_t to = k < n_cuts ? cuts[k] : len;
    if (to < at || to > len) return -9;
    // every piece in a buffer of its own, of its exact size: a read past a piece's end is the sanitizer's to catch
    std::vector<char> piece(json + at, json + to);
    ok = sp.feed(piece.data(), piece.size());
    at = to;
  }
  if (ok) ok = sp.finish();
  if (ok != want_ok) return -7;
  if (!ok) return -1;
  if (!index_ok || got.size() != want.size() || sp.count() != want.size()) return -7;
  for (size_t i = 0; i < want.size(); ++i)
    if (got[i].size() != want[i].second || memcmp(got[i].data(), json + want[i].first, want[i].second)) return -7;
  return (int)want.size();
}
int pc_range(unsigned long long ub, char* buf, int cap) {
  const std::string s = optimal_range(ub).to_string();
  if ((int)s.size() + 1 > cap) return -1;
  memcpy(buf, s.c_str(), s.size() + 1);
  return (int)s.size();
}
// the native JSON packer on arbitrary bytes (fuzzed by the test): returns the number of objects or -1
int pc_pack_choice(int n_options, int single, const char* json, size_t len, int threads, uint8_t* packed, uint32_t* status, size_t max) {
  std::vector<std::pair<size_t, size_t>> spans;
  std::vector<std::pair<size_t, size_t>> seq;
  const bool ok = egwire::split_objects_parallel(json, len, threads, spans, 0), ok_seq = egwire::split_objects(json, len, seq);
  if (ok != ok_seq || (ok && spans != seq)) return -7;          // the two splitters must agree on every input
  if (!ok || spans.size() > max) return -1;
  egwire::pack_parallel(json, spans, choice_ballot_size(n_options, single != 0), threads, packed, status,
                        [&](egwire::Cursor& c, uint8_t* dst) { return egwire::pack_choice(c, n_options, single != 0, dst); });
  // the pooled form (what eg_verify_*_json runs) must pack the same bytes and verdicts
  const size_t stride = choice_ballot_size(n_options, single != 0);
  std::vector<uint8_t> packed2(spans.size() * stride + 1);
  std::vector<uint32_t> status2(spans.size() + 1);
  egwire::WorkerPool pool(threads);
  egwire::pack_parallel(json, spans, stride, threads, packed2.data(), status2.data(),
                        [&](egwire::Cursor& c, uint8_t* dst) { return egwire::pack_choice(c, n_options, single != 0, dst); }, &pool);
  if (memcmp(packed, packed2.data(), spans.size() * stride) || memcmp(status, status2.data(), spans.size() * sizeof(uint32_t))) return -8;
  return (int)spans.size();
}
int pc_pack_qv(int n_options, unsigned long long credits, const char* json, size_t len, int threads, uint8_t* packed, uint32_t* status, size_t max) {
  std::vector<std::pair<size_t, size_t>> spans, seq;
  const bool ok = egwire::split_objects_parallel(json, len, threads, spans, 0), ok_seq = egwire::split_objects(json, len, seq);
  if (ok != ok_seq || (ok && spans != seq)) return -7;
  if (!ok || spans.size() > max) return -1;
  const QvShape sh = qv_shape(n_options, credits);
  const egwire::RangeShape vote{sh.vote_range.rings.size(), (size_t)sh.vote_range.rings_size()};
  const egwire::RangeShape credit{sh.credit_range.rings.size(), (size_t)sh.credit_range.rings_size()};
  egwire::pack_parallel(json, spans, sh.ballot_size, threads, packed, status,
                        [&](egwire::Cursor& c, uint8_t* dst) { return egwire::pack_qv(c, n_options, vote, credit, sh.ballot_size, dst); });
  return (int)spans.size();
}
// the streaming splitter with windows of `window` bytes against the whole-text splitter: 1 = same verdict and spans, 0 = differ
int pc_split_windows(const char* json, size_t len, size_t window, int threads) {
  std::vector<std::pair<size_t, size_t>> whole, parts;
  const bool ok_whole = egwire::split_objects(json, len, whole);
  egwire::SplitCursor cur;
  bool done = false, ok = true;
  for (int guard = 0; ok && !done && guard < 1000000; ++guard) ok = egwire::split_next(json, len, window, threads, cur, parts, done);
  if (ok != ok_whole) return 0;
  if (ok && (parts != whole || cur.count != whole.size())) return 0;
  // and with a cap on the values per call (the ring of eg_verify_*_json is finite): same values, never more than the cap at once
  egwire::WorkerPool pool(threads);          // the pooled form of the splitter's passes must cut the same values
  for (size_t cap : {(size_t)1, (size_t)2, (size_t)5}) {
    egwire::SplitCursor c2;
    std::vector<std::pair<size_t, size_t>> capped;
    done = false; ok = true;
    for (int guard = 0; ok && !done && guard < 1000000; ++guard) {
      const size_t before = capped.size();
      ok = egwire::split_next(json, len, window, threads, c2, capped, done, cap, cap == 2 ? &pool : nullptr);
      if (ok && capped.size() - before > cap) return 0;
    }
    if (ok != ok_whole) return 0;
    if (ok && (capped != whole || c2.count != whole.size())) return 0;
  }
  return 1;
}
unsigned long long pc_qv_size(int n_options, unsigned long long credits) { return qv_shape(n_options, credits).ballot_size; }

// The whole JSON entry point of the library (eg_verify_*_json) with its two GPU services replaced by callbacks, so that the object
// path of wire_json.hpp (OptionsLenMismatch / LenMismatch in verify()'s order) can be checked against oracle/objects.c without a GPU:
// check(n, kinds, data, ok): validity of n 32-byte items; verify(n, packed, stride, status): the batch verifier.
typedef int (*pc_check_cb)(size_t n_items, const char* kinds, const uint8_t* data, uint8_t* ok);
typedef int (*pc_verify_cb)(size_t n, const uint8_t* packed, size_t stride, uint32_t* status);
static int resolve_common(const char* json, size_t len, size_t stride, uint32_t* status, size_t max, pc_check_cb check, pc_verify_cb verify,
                          const std::function<uint32_t(egwire::Cursor&, uint8_t*)>& pack_one,
                          const std::function<bool(const std::vector<std::pair<size_t, size_t>>&, const egwire::CheckItemsFn&,
                                                   const egwire::VerifyPackedFn&, std::vector<uint32_t>&)>& resolve) {
  std::vector<std::pair<size_t, size_t>> spans;
  if (!egwire::split_objects(json, len, spans) || spans.size() > max) return -1;
  std::vector<uint8_t> packed(spans.size() * stride + 1);
  std::vector<uint32_t> st(spans.size() + 1);
  egwire::pack_parallel(json, spans, stride, 2, packed.data(), st.data(), pack_one);
  std::vector<uint32_t> verdict(spans.size() + 1);
  if (!spans.empty() && verify(spans.size(), packed.data(), stride, verdict.data())) return -2;
  std::vector<std::pair<size_t, size_t>> odd; std::vector<size_t> at;
  for (size_t k = 0; k < spans.size(); ++k) {
    status[k] = st[k] == egwire::ST_OK ? verdict[k] : st[k];
    if (st[k] == egwire::PACK_RESHAPE) { odd.push_back(spans[k]); at.push_back(k); }
  }
  const egwire::CheckItemsFn ck = [&](const std::string& kinds, const egwire::Bytes& data, std::vector<uint8_t>& ok) {
    ok.assign(kinds.size(), 0);
    return check(kinds.size(), kinds.data(), data.data(), ok.data()) == 0;
  };
  const egwire::VerifyPackedFn vf = [&](size_t n, const egwire::Bytes& pk, std::vector<uint32_t>& out) {
    out.assign(n, 0);
    return verify(n, pk.data(), stride, out.data()) == 0;
  };
  std::vector<uint32_t> res;
  if (!odd.empty()) {
    if (!resolve(odd, ck, vf, res)) return -3;
    for (size_t i = 0; i < odd.size(); ++i) status[at[i]] = res[i];
  }
  return (int)spans.size();
}
int pc_resolve_choice(int n_options, int single, const char* json, size_t len, pc_check_cb check, pc_verify_cb verify, uint32_t* status, size_t max) {
  const size_t stride = choice_ballot_size(n_options, single != 0);
  return resolve_common(json, len, stride, status, max, check, verify,
                        [&](egwire::Cursor& c, uint8_t* dst) { return egwire::pack_choice(c, n_options, single != 0, dst); },
                        [&](const std::vector<std::pair<size_t, size_t>>& odd, const egwire::CheckItemsFn& ck, const egwire::VerifyPackedFn& vf,
                            std::vector<uint32_t>& out) { return egwire::resolve_choice_objects(json, odd, n_options, single != 0, stride, ck, vf, out); });
}
int pc_resolve_qv(int n_options, unsigned long long credits, const char* json, size_t len, pc_check_cb check, pc_verify_cb verify, uint32_t* status,
                  size_t max) {
  const QvShape sh = qv_shape(n_options, credits);
  const egwire::RangeShape vote{sh.vote_range.rings.size(), (size_t)sh.vote_range.rings_size()};
  const egwire::RangeShape credit{sh.credit_range.rings.size(), (size_t)sh.credit_range.rings_size()};
  return resolve_common(json, len, sh.ballot_size, status, max, check, verify,
                        [&](egwire::Cursor& c, uint8_t* dst) { return egwire::pack_qv(c, n_options, vote, credit, sh.ballot_size, dst); },
                        [&](const std::vector<std::pair<size_t, size_t>>& odd, const egwire::CheckItemsFn& ck, const egwire::VerifyPackedFn& vf,
                            std::vector<uint32_t>& out) { return egwire::resolve_qv_objects(json, odd, n_options, vote, credit, sh.ballot_size, ck, vf, out); });
}
}

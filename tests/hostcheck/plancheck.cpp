// plancheck.cpp -- TEST-ONLY build of the product's pure-host logic under AddressSanitizer + UBSan: the election parameter
// mirrors and plan builders (csrc/host_plan.hpp: RangeDecomposition::optimal, every build_*_plan, the flattening the engine
// uploads and its index check) and the native wire ingest (csrc/wire_json.hpp).  The product never loads this library.
#include <stdint.h>
#include <string.h>
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
    case 0: return check(build_choice_plan(n, true), pk);
    case 1: return check(build_choice_plan(n, false), pk);
    case 2: return check(build_qv_plan(n, v), pk);
    case 3: return check(build_zero_plan(), pk);
    case 4: return check(build_bool_plan(), pk);
    case 5: return check(build_range_plan(v, &item), pk);
    case 6: return check(build_sumsq_plan(n, "test", &item), pk);
    case 7: return check(build_share_plan(10, 7, pk, (unsigned long long)n), pk);
    default: return -2;
  }
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
  return 1;
}
unsigned long long pc_qv_size(int n_options, unsigned long long credits) { return qv_shape(n_options, credits).ballot_size; }
}

// voting.cpp -- the reference's examples/voting.rs on the GPU backend, in C++ on top of the C ABI:
//   `Args::vote`            (examples/voting.rs:179-213)  ./voting [votes] [options] [seed]
//   `Args::quadratic_vote`  (examples/voting.rs:219-269)  ./voting --qv [votes] [options] [credits] [seed]
// talliers' key -> voters create ballots from their own choices -> verify every ballot -> homomorphic totals -> decrypt and
// compare with the EXPECTED totals the voters' choices add up to.  Threshold sharing of the key (examples/voting.rs:105-120) is out
// of scope; a single key pair stands in for the shared key (tests/test_gpu_parity.py::test_threshold_tally_end_to_end
// runs the 7-of-10 tally stage).  With --devices N the batch goes through the in-process multi-GPU entry
// (eg_verify_*_batch_multi) over N contexts on the visible GPUs (all on GPU 0 when fewer are visible).  With --json the ballots travel
// the way examples/voting.rs:195-198 prints them - serde_json text, ONE BALLOT AT A TIME - into the streaming entry
// (JsonStream = eg_verify_choice_json_begin / eg_verify_json_feed / _end), which cuts, packs and verifies them as they arrive.
//
//   g++ -std=c++17 -Iinclude examples/voting.cpp -Lelastic_elgamal_amd -leg_hip -Wl,-rpath,$PWD/elastic_elgamal_amd -o voting
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>

#include "elastic_elgamal_hip.hpp"

using namespace elastic_elgamal_hip;

// the voters' randomness (rand::rng() in the reference): splitmix64, so that a run is reproducible from its seed
struct Rng {
  uint64_t s;
  uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
  uint64_t below(uint64_t n) { return next() % n; }
  bool chance(double p) { return (double)(next() >> 11) / 9007199254740992.0 < p; }
};

// Self::tally (examples/voting.rs:122-177) with one key: decrypt each total = blinded - [sk]random, look it up in
// DiscreteLogTable::new(0..=max), compare with the expected totals
static bool tally(const Context& ctx, const Ristretto& group, const Scalar& sk, const std::vector<Ciphertext>& totals,
                  const std::vector<uint64_t>& expected, uint64_t max_value) {
  std::vector<uint64_t> range(max_value + 1);
  for (uint64_t m = 0; m <= max_value; ++m) range[m] = m;
  const DiscreteLogTable lookup(ctx, range);
  bool ok = true;
  for (size_t k = 0; k < totals.size(); ++k) {
    const Element dh = group.mul(totals[k].random_element, sk);                 // Element * &Scalar
    const Element m_g = group.sub(totals[k].blinded_element, dh);
    const std::optional<uint64_t> found = lookup.get(m_g);
    if (!found) { printf("  variant #%zu: decryption failed\n", k + 1); return false; }
    printf("  variant #%zu decrypted tally: %llu (expected %llu)\n", k + 1, (unsigned long long)*found, (unsigned long long)expected[k]);
    ok = ok && *found == expected[k];
  }
  return ok;
}

// serde's human-readable form of an EncryptedChoice (src/serde.rs:19-80: every element and scalar as unpadded base64url; field names of
// choice.rs:276-280, ring.rs:282-287, log_equality.rs:96-101), from the packed ballot
static std::string b64url(const uint8_t* p, size_t n) {
  static const char* A = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789-_";
  std::string out;
  for (size_t i = 0; i < n; i += 3) {
    const uint32_t v = (uint32_t)p[i] << 16 | (i + 1 < n ? (uint32_t)p[i + 1] << 8 : 0u) | (i + 2 < n ? (uint32_t)p[i + 2] : 0u);
    out += A[v >> 18]; out += A[(v >> 12) & 63];
    if (i + 1 < n) out += A[(v >> 6) & 63];
    if (i + 2 < n) out += A[v & 63];
  }
  return out;
}
static std::string choice_to_json(const uint8_t* b, size_t options) {
  auto item = [&](size_t k) { return "\"" + b64url(b + 32 * k, 32) + "\""; };
  std::string s = "{\"choices\":[";
  for (size_t k = 0; k < options; ++k)
    s += std::string(k ? "," : "") + "{\"random_element\":" + item(2 * k) + ",\"blinded_element\":" + item(2 * k + 1) + "}";
  s += "],\"range_proof\":{\"common_challenge\":" + item(2 * options) + ",\"ring_responses\":[";
  for (size_t k = 0; k < 2 * options; ++k) s += std::string(k ? "," : "") + item(2 * options + 1 + k);
  s += "]},\"sum_proof\":{\"challenge\":" + item(4 * options + 1) + ",\"response\":" + item(4 * options + 2) + "}}";
  return s;
}

int main(int argc, char** argv) {
  bool qv = false, json = false;
  int devices = 1;
  std::vector<std::string> pos;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--qv")) qv = true;
    else if (!strcmp(argv[i], "--json")) json = true;
    else if (!strcmp(argv[i], "--devices") && i + 1 < argc) devices = atoi(argv[++i]);
    else pos.push_back(argv[i]);
  }
  auto arg = [&](size_t k, uint64_t dflt) { return k < pos.size() ? strtoull(pos[k].c_str(), nullptr, 10) : dflt; };
  const size_t votes = (size_t)arg(0, 1000), options = (size_t)arg(1, 5);
  const uint64_t credits = qv ? arg(2, 20) : 0, seed = arg(qv ? 3 : 2, 1);
  if (devices < 1 || devices > 16) { printf("--devices must be in 1..16\n"); return 2; }

  // one context per device slot; slot d uses GPU d when the process sees that many, else GPU 0
  std::vector<std::unique_ptr<Context>> ctxs;
  for (int d = 0; d < devices; ++d) {
    try { ctxs.push_back(std::make_unique<Context>(d)); }
    catch (const Error&) { if (d == 0) throw; ctxs.push_back(std::make_unique<Context>(0)); }
  }
  const Context& ctx = *ctxs[0];
  Ristretto group{ctx};

  // Keypair::generate: secret scalar from 64 seed-derived bytes, public key = [sk]G
  std::array<uint8_t, 64> wide{};
  for (size_t i = 0; i < 64; ++i) wide[i] = (uint8_t)((seed * 0x9E3779B97F4A7C15ull >> (i % 8 * 8)) + 31 * i);
  const Scalar sk = group.scalar_from_random_bytes(wide);
  const Element pk = group.mul_generator(sk);
  Rng rng{seed ^ 0xC0FFEEull};
  std::vector<uint64_t> expected(options, 0);
  bool ok = true;

  if (!qv) {
    std::vector<std::unique_ptr<ChoiceParams>> params;
    for (auto& c : ctxs) params.push_back(std::make_unique<ChoiceParams>(ChoiceParams::single(*c, pk, options)));
    std::vector<size_t> choices(votes);
    for (auto& c : choices) c = (size_t)rng.below(options);                            // rng.random_range(0..options_count)
    Bytes ballots = params[0]->encrypt_single_choices(seed, 0, choices);               // EncryptedChoice::single(&params, choice, rng)
    const size_t forged = votes > 3 ? 3 : votes;                                       // one forged ballot must be rejected
    if (forged < votes) ballots[forged * params[0]->ballot_size() + 64 * options + 40] ^= 1;
    for (size_t i = 0; i < votes; ++i) if (i != forged) expected[choices[i]] += 1;
    std::vector<const ChoiceParams*> per;
    for (auto& p : params) per.push_back(p.get());
    BatchVerdict<ChoiceVerificationError> verdict;
    if (json) {
      // println!("{}", serde_json::to_string_pretty(&encrypted)) per voter (examples/voting.rs:195-198): every ballot is fed as it is made
      // --devices N: ONE parser for all GPUs (eg_verify_choice_json_begin_multi: the packed windows are dealt to the params objects by load),
      // and every ballot's text is handed over without a copy (eg_verify_json_feed_owned: the stream gives the block back when it is through)
      std::unique_ptr<JsonStream> sp(devices > 1 ? new JsonStream(per, 4) : new JsonStream(*params[0], 4));
      JsonStream& stream = *sp;
      const size_t bs = params[0]->ballot_size();
      for (size_t i = 0; i < votes; ++i) {
        std::string text = choice_to_json(ballots.data() + i * bs, options) + "\n";
        if (devices > 1) stream.feed_owned(std::make_shared<const std::string>(std::move(text)));
        else stream.feed(text);
      }
      std::vector<Ciphertext> totals;
      for (uint32_t st : stream.finish(&totals)) verdict.results.push_back(choice_error_from_status(st));
      verdict.totals = totals;
      printf("(%zu ballots went through the JSON stream one at a time%s)\n", stream.objects(), devices > 1 ? ", one parser for all devices, blocks handed over" : "");
    } else
      verdict = devices > 1 ? verify_batch_multi(per, ballots) : params[0]->verify_batch(ballots);   // encrypted.verify(&params)
    printf("%zu of %zu ballots verified\n", verdict.accepted(), votes);
    for (size_t i = 0; i < verdict.results.size(); ++i)
      if (verdict.results[i]) printf("  voter #%zu rejected: %s\n", i + 1, verdict.results[i]->to_string().c_str());
    ok = verdict.accepted() == votes - (forged < votes ? 1 : 0) && tally(ctx, group, sk, verdict.totals, expected, votes);
  } else {
    std::vector<std::unique_ptr<QuadraticVotingParams>> params;
    for (auto& c : ctxs) params.push_back(std::make_unique<QuadraticVotingParams>(*c, pk, options, credits));
    // the reference's vote draw (examples/voting.rs:235-244): add one vote to a random option with probability 0.8 while the credit lasts
    std::vector<uint32_t> all(votes * options, 0u);
    for (size_t i = 0; i < votes; ++i) {
      uint32_t* v = &all[i * options];
      while (rng.chance(0.8)) {
        const size_t k = (size_t)rng.below(options);
        uint64_t credit = 0;
        for (size_t j = 0; j < options; ++j) { const uint64_t x = v[j] + (j == k ? 1u : 0u); credit += x * x; }
        if (credit > credits) break;
        v[k] += 1;
      }
    }
    Bytes ballots = params[0]->encrypt_votes_batch(seed, 0, all);                      // QuadraticVotingBallot::new(&vote_params, &votes, rng)
    const size_t forged = votes > 3 ? 3 : votes;
    if (forged < votes) ballots[forged * params[0]->ballot_size() + params[0]->ballot_size() - 40] ^= 1;
    for (size_t i = 0; i < votes; ++i) if (i != forged) for (size_t k = 0; k < options; ++k) expected[k] += all[i * options + k];
    std::vector<const QuadraticVotingParams*> per;
    for (auto& p : params) per.push_back(p.get());
    auto verdict = devices > 1 ? verify_batch_multi(per, ballots) : params[0]->verify_batch(ballots);   // encrypted.verify(&vote_params)
    printf("%zu of %zu quadratic-voting ballots verified\n", verdict.accepted(), votes);
    for (size_t i = 0; i < verdict.results.size(); ++i)
      if (verdict.results[i]) printf("  voter #%zu rejected (error kind %d)\n", i + 1, (int)verdict.results[i]->kind);
    const uint64_t max_votes = votes * params[0]->max_votes();                          // votes_count * vote_params.max_votes()
    ok = verdict.accepted() == votes - (forged < votes ? 1 : 0) && tally(ctx, group, sk, verdict.totals, expected, max_votes);
  }
  printf("%s: the decrypted totals %s the expected ones\n", ok ? "OK" : "MISMATCH", ok ? "equal" : "differ from");
  return ok ? 0 : 1;
}

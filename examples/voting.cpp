// voting.cpp -- the reference's examples/voting.rs:179-213 (`Args::vote`) on the GPU backend, in C++ on top of the
// C ABI: talliers' key -> voters create ballots -> verify every ballot -> homomorphic totals -> decrypt and
// compare with the expected counts.  Threshold sharing of the key (examples/voting.rs:105-120) is out of scope
// (SURVEY 2); a single key pair stands in for the shared key.
//
//   g++ -std=c++17 -Iinclude examples/voting.cpp -Lelastic_elgamal_amd -leg_hip -Wl,-rpath,$PWD/elastic_elgamal_amd -o voting
//   ./voting [votes=1000] [options=5] [seed=1]
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "elastic_elgamal_hip.hpp"

using namespace elastic_elgamal_hip;

int main(int argc, char** argv) {
  const size_t votes = argc > 1 ? strtoul(argv[1], nullptr, 10) : 1000;
  const size_t options = argc > 2 ? strtoul(argv[2], nullptr, 10) : 5;
  const uint64_t seed = argc > 3 ? strtoull(argv[3], nullptr, 10) : 1;
  Context ctx(0);
  Ristretto group{ctx};

  // Keypair::generate: secret scalar from 64 seed-derived bytes, public key = [sk]G
  std::array<uint8_t, 64> wide{};
  for (size_t i = 0; i < 64; ++i) wide[i] = (uint8_t)((seed * 0x9E3779B97F4A7C15ull >> (i % 8 * 8)) + 31 * i);
  const Scalar sk = group.scalar_from_random_bytes(wide);
  const Element pk = group.mul_generator(sk);

  ChoiceParams params = ChoiceParams::single(ctx, pk, options);
  Bytes ballots = params.encrypt_batch(seed, 0, votes);          // EncryptedChoice::single(&params, choice, rng) per voter
  if (votes > 3) ballots[3 * params.ballot_size() + 64 * options + 40] ^= 1;   // one forged ballot must be rejected

  auto verdict = params.verify_batch(ballots);                     // encrypted.verify(&params) for every voter
  printf("%zu of %zu ballots verified\n", verdict.accepted(), votes);
  for (size_t i = 0; i < verdict.results.size(); ++i)
    if (verdict.results[i]) printf("  voter #%zu rejected: %s\n", i + 1, verdict.results[i]->to_string().c_str());

  // tally(): decrypt each total = blinded - [sk]random, then look the element up in DiscreteLogTable::new(0..=votes)
  // (examples/voting.rs:130-131,166-172; the table's products come from the GPU in one batch)
  std::vector<uint64_t> range(votes + 1);
  for (uint64_t m = 0; m <= votes; ++m) range[m] = m;
  const DiscreteLogTable lookup(ctx, range);
  size_t sum = 0;
  for (size_t k = 0; k < options; ++k) {
    const Element dh = group.vartime_multi_mul({sk}, {verdict.totals[k].random_element});
    const Element m_g = group.sub(verdict.totals[k].blinded_element, dh);
    const std::optional<uint64_t> found = lookup.get(m_g);
    if (!found) { printf("decryption failed\n"); return 1; }
    printf("  option #%zu: %llu votes\n", k + 1, (unsigned long long)*found);
    sum += (size_t)*found;
  }
  const bool ok = sum == verdict.accepted();
  printf("%s: decrypted totals sum to %zu, %zu ballots were accepted\n", ok ? "OK" : "MISMATCH", sum, verdict.accepted());
  return ok ? 0 : 1;
}

// shim_calls.cpp -- drives every primitive-tier entry point with n = 1, the way a `Group` trait method does (INTEGRATION.md section 3:
// one problem per call), checks the results against each other through group identities, and prints the cost of one call.
// Test infrastructure (tests/test_gpu_parity.py::test_cpp_shim_calls builds and runs it); the numbers go to profiles/.
//   g++ -std=c++17 -O2 -Iinclude tests/cpp/shim_calls.cpp -Lelastic_elgamal_amd -leg_hip -Wl,-rpath,$PWD/elastic_elgamal_amd -o shim_calls
#include <chrono>
#include <cstdio>
#include <cstring>
#include <functional>

#include "elastic_elgamal_hip.hpp"

using namespace elastic_elgamal_hip;

static double us_per_call(int iters, const std::function<void()>& f) {
  f();                                                    // first call: scratch allocation
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < iters; ++i) f();
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
}

int main() {
  Context ctx(0);
  Ristretto g{ctx};
  int bad = 0;
  auto expect = [&](bool ok, const char* what) { if (!ok) { printf("FAIL: %s\n", what); ++bad; } };

  std::array<uint8_t, 64> wide{};
  for (int i = 0; i < 64; ++i) wide[i] = (uint8_t)(i * 37 + 11);
  const Scalar k = g.scalar_from_random_bytes(wide);
  for (int i = 0; i < 64; ++i) wide[i] = (uint8_t)(i * 91 + 5);
  const Scalar r = g.scalar_from_random_bytes(wide);
  const Scalar one = Ristretto::scalar_from_u64(1), zero = Ristretto::scalar_from_u64(0);
  const Element G = g.mul_generator(one), P = g.mul_generator(k);

  // identities between the trait methods (group/mod.rs:183-255)
  expect(g.mul(G, k) == P, "Element * &Scalar == mul_generator");
  expect(g.vartime_double_mul_generator(k, G, r) == g.mul_generator(g.scalar_muladd(k, one, r)), "[k]G + [r]G == [k + r]G");
  expect(g.vartime_multi_mul({k, r}, {P, G}) == g.add(g.mul(P, k), g.mul_generator(r)), "multi_mul == sum of products");
  expect(g.sub(g.add(P, G), G) == P && g.add(P, g.neg(P)) == Ristretto::identity(), "add / sub / neg");
  expect(g.is_identity(Ristretto::identity()) && !g.is_identity(P), "is_identity");
  expect(g.scalar_muladd(k, g.invert_scalar(k), zero) == one, "k * k^-1 == 1");
  expect(g.scalar_muladd(k, one, g.scalar_neg(k)) == zero, "k + (-k) == 0");
  expect(g.deserialize_element(P).has_value() && g.deserialize_scalar(k).has_value(), "deserialize of canonical encodings");
  Scalar big; big.fill(0xff);
  Element junk; junk.fill(0xff);
  expect(!g.deserialize_scalar(big).has_value() && !g.deserialize_element(junk).has_value(), "deserialize rejects non-canonical input");
  expect(Ristretto::scalar_from_u64(0x0102030405060708ull)[0] == 8 && Ristretto::scalar_from_u64(7)[31] == 0, "Scalar::from(u64)");

  const int it = 200;
  printf("primitive tier, one problem per call (n = 1), microseconds per call, %d calls each\n", it);
  printf("  scalar_from_random_bytes      %8.1f\n", us_per_call(it, [&] { (void)g.scalar_from_random_bytes(wide); }));
  printf("  deserialize_scalar            %8.1f\n", us_per_call(it, [&] { (void)g.deserialize_scalar(k); }));
  printf("  scalar a*b+c                  %8.1f\n", us_per_call(it, [&] { (void)g.scalar_muladd(k, r, one); }));
  printf("  invert_scalar                 %8.1f\n", us_per_call(it, [&] { (void)g.invert_scalar(k); }));
  printf("  deserialize_element           %8.1f\n", us_per_call(it, [&] { (void)g.deserialize_element(P); }));
  printf("  element add                   %8.1f\n", us_per_call(it, [&] { (void)g.add(P, G); }));
  printf("  mul_generator                 %8.1f\n", us_per_call(it, [&] { (void)g.mul_generator(k); }));
  printf("  Element * &Scalar             %8.1f\n", us_per_call(it, [&] { (void)g.mul(P, k); }));
  printf("  vartime_double_mul_generator  %8.1f\n", us_per_call(it, [&] { (void)g.vartime_double_mul_generator(k, P, r); }));
  printf("  vartime_multi_mul, 2 terms    %8.1f\n", us_per_call(it, [&] { (void)g.vartime_multi_mul({k, r}, {P, G}); }));
  printf("  vartime_multi_mul, 7 terms    %8.1f\n", us_per_call(it, [&] { (void)g.vartime_multi_mul({k, r, k, r, k, r, k}, {P, G, P, G, P, G, P}); }));
  printf("%s\n", bad ? "MISMATCH" : "OK: every identity holds");
  return bad ? 1 : 0;
}

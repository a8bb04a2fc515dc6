// Drives the pooled splitter and packer of csrc/wire_json.hpp (egwire::WorkerPool: what eg_verify_*_json runs its parser on) through
// plancheck.cpp's hooks in a -fsanitize=thread build: tests/test_plancheck.py::test_worker_pool_under_thread_sanitizer.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
extern "C" {
int pc_pack_choice(int n_options, int single, const char* json, size_t len, int threads, uint8_t* packed, uint32_t* status, size_t max);
int pc_split_windows(const char* json, size_t len, size_t window, int threads);
}
int main() {
  std::string one = "{\"choices\":[],\"range_proof\":{\"common_challenge\":\"AAAA\",\"ring_responses\":[]},\"sum_proof\":\"x\"}";
  std::string text = "[";
  for (int i = 0; i < 3000; ++i) { if (i) text += ","; text += one; }
  text += "]";
  std::vector<uint8_t> packed(3000 * 736 + 16);
  std::vector<uint32_t> st(3001);
  for (int rep = 0; rep < 5; ++rep) {
    int n = pc_pack_choice(5, 1, text.data(), text.size(), 8, packed.data(), st.data(), 3000);
    int ok = pc_split_windows(text.data(), text.size(), 5000, 8);
    printf("n=%d split=%d\n", n, ok);
  }
  return 0;
}

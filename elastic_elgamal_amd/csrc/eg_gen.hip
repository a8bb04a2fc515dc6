// eg_gen.hip -- translation unit of the synthetic-ballot generator kernels (prover_kernels.cuh).
#include "prover_kernels.cuh"

using namespace eg;

void eg_launch_choice_encrypt(int blocks, hipStream_t s, u64 seed0, size_t n, int n_options, int single, int n_selected,
                              const u32* selection, u64 rng_skip, const uint4* tabG, const uint4* tabK, const u32* prefixes, int pre_main, int pre_ring,
                              int pre_logeq, u32* out, u32 stride_words) {
  hipLaunchKernelGGL(k_choice_encrypt, dim3(blocks), dim3(NT), 0, s, seed0, n, n_options, single, n_selected, selection, rng_skip, tabG, tabK, prefixes,
                     pre_main, pre_ring, pre_logeq, out, stride_words);
}

void eg_launch_qv_encrypt(int blocks, hipStream_t s, u64 seed0, size_t n, int n_options, u64 credits, const u32* votes, u64 rng_skip,
                          const int* vote_range,
                          const int* credit_range, int pre_sumsq, const uint4* tabG, const uint4* tabK, const u32* prefixes,
                          u32* out, u32 stride_words, u32 vote_words, u32 credit_words) {
  // range arrays: [n_rings, pre_main, pre_ring, size0, step0, size1, step1, ...]
  GenRange v{}, c{};
  auto fill = [](GenRange& g, const int* a) {
    g.n_rings = a[0]; g.pre_main = a[1]; g.pre_ring = a[2];
    for (int i = 0; i < g.n_rings; ++i) { g.size[i] = (u32)a[3 + 2 * i]; g.step[i] = (u32)a[4 + 2 * i]; }
  };
  fill(v, vote_range); fill(c, credit_range);
  hipLaunchKernelGGL(k_qv_encrypt, dim3(blocks), dim3(NT), 0, s, seed0, n, n_options, credits, votes, rng_skip, v, c, pre_sumsq, tabG, tabK,
                     prefixes, out, stride_words, vote_words, credit_words);
}

// eg_gen.hip -- translation unit of the synthetic-ballot generator kernels (prover_kernels.cuh).
#include "prover_kernels.cuh"

using namespace eg;

void eg_launch_choice_encrypt(int blocks, hipStream_t s, u64 seed0, size_t n, int n_options, int single, int n_selected,
                              const u32* selection, u64 rng_skip, const uint4* tabG, const uint4* tabK, const u32* prefixes, int pre_main, int pre_ring,
                              int pre_logeq, u32* out, u32 stride_words, u32* gws) {
  hipLaunchKernelGGL(k_choice_encrypt, dim3(blocks), dim3(NT), 0, s, seed0, n, n_options, single, n_selected, selection, rng_skip, tabG, tabK, prefixes,
                     pre_main, pre_ring, pre_logeq, out, stride_words, gws);
}
unsigned eg_gen_choice_ws_words(int n_options) { return gen_choice_ws_words(n_options); }
unsigned eg_gen_qv_ws_words(int n_options, unsigned max_rings, unsigned max_responses) { return gen_qv_ws_words(n_options, max_rings, max_responses); }

void eg_launch_qv_encrypt(int blocks, hipStream_t s, u64 seed0, size_t n, int n_options, u64 credits, const u32* votes, u64 rng_skip,
                          int vote_rings, int vote_main, int vote_ring, const u32* d_vote_desc, int credit_rings, int credit_main,
                          int credit_ring, const u32* d_credit_desc, int pre_sumsq, const uint4* tabG, const uint4* tabK,
                          const u32* prefixes, u32* out, u32 stride_words, u32 vote_words, u32 credit_words, u32* gws) {
  const GenRange v{vote_rings, d_vote_desc, vote_main, vote_ring}, c{credit_rings, d_credit_desc, credit_main, credit_ring};
  hipLaunchKernelGGL(k_qv_encrypt, dim3(blocks), dim3(NT), 0, s, seed0, n, n_options, credits, votes, rng_skip, v, c, pre_sumsq, tabG, tabK,
                     prefixes, out, stride_words, vote_words, credit_words, gws);
}

// eg_gen.hip -- translation unit of the synthetic-ballot generator kernels (prover_kernels.cuh).
#include "prover_kernels.cuh"

using namespace eg;

void eg_launch_choice_encrypt(int blocks, hipStream_t s, u64 seed0, size_t n, int n_options, int single, int n_selected,
                              const uint4* tabG, const uint4* tabK, const u32* prefixes, int pre_main, int pre_ring,
                              int pre_logeq, u32* out, u32 stride_words) {
  hipLaunchKernelGGL(k_choice_encrypt, dim3(blocks), dim3(NT), 0, s, seed0, n, n_options, single, n_selected, tabG, tabK, prefixes,
                     pre_main, pre_ring, pre_logeq, out, stride_words);
}

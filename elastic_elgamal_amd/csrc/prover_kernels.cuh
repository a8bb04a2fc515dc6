// prover_kernels.cuh -- synthetic-ballot generation on the GPU (EncryptedChoice::new); filled in below.
#pragma once
#include "kernels.cuh"
namespace eg {
__global__ void __launch_bounds__(NT, 2) k_choice_encrypt(u64 seed0, size_t n, int n_options, int single, int n_selected,
                                                          const uint4* tabG, const uint4* tabK, const u32* key_words,
                                                          const u32* prefixes, uint4* ws, u32* out, u32 stride_words) {}
}  // namespace eg

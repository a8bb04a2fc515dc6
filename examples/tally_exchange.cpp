// tally_exchange.cpp -- the multi-GPU recipe for a host that is NOT Python: one process per GPU, ballots sharded contiguously, and the
// ONE exchange of the path done with librccl directly:
//     eg_verify_choice_batch_device  ->  eg_choice_tally_encode_device  ->  ncclAllGather  ->  eg_points_sum_device
// all on one HIP stream, no host round trip in between.  Point addition is not an RCCL reduction op, so the per-rank tallies travel as
// canonical encodings (64 bytes per option) and every rank adds the gathered ones itself.  This is what `totals[k] += vote[k]` of the
// reference's examples/voting.rs:199-203 becomes when the voters are spread over the GPUs of a node (BASELINE configs[4]).
//
//   hipcc -std=c++17 -Iinclude examples/tally_exchange.cpp -Lelastic_elgamal_amd -leg_hip -lrccl \
//         -Wl,-rpath,$PWD/elastic_elgamal_amd -o tally_exchange
//   one process per GPU:  RANK=r WORLD_SIZE=n LOCAL_RANK=r EG_NCCL_ID_FILE=/tmp/eg_nccl_id ./tally_exchange [total_ballots] [options] [seed]
//   (rank 0 writes the ncclUniqueId to EG_NCCL_ID_FILE, the others wait for it; with WORLD_SIZE unset it is a 1-rank run.)
// Every rank prints the merged tally's digest and the number of accepted ballots; with one rank the merged tally must equal the
// engine's own tally, which the program checks.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include "eg_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define NCCL_OK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_)); return 3; } } while (0)
#define EG_OK_(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s: %d %s\n", #x, r_, eg_last_error()); return 4; } } while (0)

static int env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }

int main(int argc, char** argv) {
  const size_t total = argc > 1 ? strtoull(argv[1], nullptr, 10) : 100000;
  const int options = argc > 2 ? atoi(argv[2]) : 5;
  const uint64_t seed = argc > 3 ? strtoull(argv[3], nullptr, 10) : 1;
  const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1), local = env_int("LOCAL_RANK", 0);
  HIP_OK(hipSetDevice(local));

  // communicator: rank 0 makes the id, the others read it from the rendezvous file
  ncclUniqueId id;
  const char* id_file = getenv("EG_NCCL_ID_FILE");
  if (rank == 0) {
    NCCL_OK(ncclGetUniqueId(&id));
    if (world > 1) {
      if (!id_file) { fprintf(stderr, "EG_NCCL_ID_FILE must be set when WORLD_SIZE > 1\n"); return 1; }
      std::ofstream f(std::string(id_file) + ".tmp", std::ios::binary);
      f.write(reinterpret_cast<const char*>(&id), sizeof id);
      f.close();
      rename((std::string(id_file) + ".tmp").c_str(), id_file);
    }
  } else {
    if (!id_file) { fprintf(stderr, "EG_NCCL_ID_FILE must be set when WORLD_SIZE > 1\n"); return 1; }
    for (int tries = 0;; ++tries) {
      std::ifstream f(id_file, std::ios::binary);
      if (f && f.read(reinterpret_cast<char*>(&id), sizeof id)) break;
      if (tries > 600) { fprintf(stderr, "no ncclUniqueId in %s after 60 s\n", id_file); return 1; }
      std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
  }
  ncclComm_t comm;
  NCCL_OK(ncclCommInitRank(&comm, world, id, rank));
  hipStream_t s;
  HIP_OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));

  // election: the key of the reference's snapshots (tests/snapshots.rs, seed 12345), single-choice ballots
  const uint8_t pk[32] = {0xa6, 0xad, 0xb6, 0xe9, 0xc0, 0xae, 0x8d, 0x54, 0xc2, 0x6e, 0x6e, 0x56, 0xb5, 0xcc, 0xd7, 0xa1,
                          0x6b, 0xb0, 0xe1, 0x95, 0x1a, 0xbe, 0x4d, 0x7e, 0xe7, 0x02, 0x8e, 0x3d, 0x4e, 0xca, 0x85, 0x31};
  eg_ctx* ctx = nullptr;
  EG_OK_(eg_init(local, &ctx));
  eg_choice_params* params = nullptr;
  EG_OK_(eg_choice_params_create(ctx, pk, options, 1, &params));
  const size_t stride = eg_choice_ballot_size(options, 1);

  // this rank's contiguous slab [lo, hi) of the batch: ballot i is made from seed + i whichever rank makes it
  const size_t lo = total * rank / world, hi = total * (rank + 1) / world, n = hi - lo;
  void *d_ballots = nullptr, *d_status = nullptr, *d_local = nullptr, *d_all = nullptr, *d_merged = nullptr, *d_bad = nullptr;
  const size_t tally_bytes = (size_t)options * 64;
  HIP_OK(hipMalloc(&d_ballots, n * stride + 16));
  HIP_OK(hipMalloc(&d_status, n * sizeof(uint32_t) + 16));
  HIP_OK(hipMalloc(&d_local, tally_bytes));
  HIP_OK(hipMalloc(&d_all, tally_bytes * world));
  HIP_OK(hipMalloc(&d_merged, tally_bytes));
  HIP_OK(hipMalloc(&d_bad, sizeof(uint32_t)));
  HIP_OK(hipMemsetAsync(d_bad, 0, sizeof(uint32_t), s));
  EG_OK_(eg_choice_encrypt_batch_device(params, seed, lo, n, 0, d_ballots, s));        // EncryptedChoice::single per voter, on the GPU

  // the step: verify + tally, then the one collective, then the local sum -- one stream, no host synchronisation in between
  EG_OK_(eg_choice_tally_reset_async(params, s));
  EG_OK_(eg_verify_choice_batch_device(params, n, d_ballots, d_status, s));
  EG_OK_(eg_choice_tally_encode_device(params, d_local, s));
  NCCL_OK(ncclAllGather(d_local, d_all, tally_bytes, ncclUint8, comm, s));
  EG_OK_(eg_points_sum_device(ctx, world, 2 * options, d_all, d_merged, d_bad, s));
  HIP_OK(hipStreamSynchronize(s));

  std::vector<uint8_t> merged(tally_bytes), own(tally_bytes);
  std::vector<uint32_t> status(n);
  uint32_t bad = 0;
  HIP_OK(hipMemcpy(merged.data(), d_merged, tally_bytes, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(status.data(), d_status, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost));
  size_t accepted = 0;
  for (uint32_t st : status) accepted += st == 0;
  uint64_t digest = 1469598103934665603ull;                                             // FNV-1a of the merged tally
  for (uint8_t b : merged) digest = (digest ^ b) * 1099511628211ull;
  printf("rank %d of %d: ballots [%zu, %zu), %zu accepted, merged tally digest %016llx, undecodable encodings %u\n", rank, world, lo,
         hi, accepted, (unsigned long long)digest, bad);
  int rc = bad ? 5 : 0;
  if (world == 1) {                                                                     // one rank: the exchange must be the identity
    EG_OK_(eg_choice_tally_encode(params, own.data()));
    const bool same = own == merged;
    printf("%s: the exchanged tally %s the engine's own tally, %zu of %zu ballots accepted\n", same && accepted == n ? "OK" : "MISMATCH",
           same ? "equals" : "differs from", accepted, n);
    if (!same || accepted != n) rc = 6;
  }
  eg_choice_params_destroy(params);
  eg_destroy(ctx);
  ncclCommDestroy(comm);
  for (void* p : {d_ballots, d_status, d_local, d_all, d_merged, d_bad}) (void)hipFree(p);
  (void)hipStreamDestroy(s);
  return rc;
}

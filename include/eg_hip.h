/*
 * eg_hip.h -- C ABI of libeg_hip.so: the MI355X (gfx950) Ristretto255 backend and batch ballot verifier
 * that stands in for the hot path of slowli/elastic-elgamal.
 *
 * Two tiers:
 *
 *  PRIMITIVE TIER -- batched equivalents of the reference's `Group` plugin trait for `Ristretto`
 *  (src/group/mod.rs:65-255, src/group/ristretto.rs:23-146).  A Rust shim
 *  `impl ScalarOps/ElementOps/Group for HipRistretto` binds these one-to-one.  All
 *  buffers are caller-owned HOST memory, 32-byte little-endian encodings exactly as
 *  serialize_scalar / serialize_element produce them; `n` independent problems per call run on the GPU.
 *
 *  BATCH TIER -- `for ballot in ballots { ballot.verify(&params) }` as one call: the reference has no
 *  equivalent entry point (it verifies one ballot at a time on one thread, examples/voting.rs:189-204);
 *  these are what a Rust host calls instead of the loop.  Ballots are packed back to back in the
 *  reference's own byte formats (see "wire layout" below).  `_device` variants take DEVICE pointers and a
 *  hipStream_t (as void*), so that torch / RCCL owned buffers can be passed without copies.
 *
 * Every function returns EG_OK (0) or a negative eg_error.  Nothing here falls back to the CPU: if no
 * gfx950 device is usable eg_init fails with EG_ERR_NO_DEVICE.
 *
 * TIMING (the `Group` contract, src/group/mod.rs:65-135,183-255): the reference requires scalar arithmetic, element addition,
 * `Element * &Scalar`, `mul_generator` and `multi_mul` to be CONSTANT-TIME and lets only the `vartime_*` methods depend on their
 * operands.  NOTHING in this library is constant-time: every multiplication indexes tables in device memory with digits of its scalar
 * (2^19-entry comb windows), the provers branch on the voter's choice, and a GPU shared with other processes is not a place for
 * secrets in the first place.  The library is the VERIFIER-side backend - `EncryptedChoice::verify`, `QuadraticVotingBallot::verify`
 * and everything under them call `vartime_double_mul_generator` / `vartime_multi_mul` on public data only (src/proofs/ring.rs:342-350,
 * log_equality.rs:160-164, mul.rs:213-247).  eg_mul_generator_batch, eg_scalar_*_batch, eg_point_add_batch and the
 * eg_*_encrypt_* provers exist for completeness of the trait, for tests and for making SYNTHETIC ballots (bench.py); a deployment that
 * encrypts real voters' choices, or multiplies by secret keys, keeps those operations on the CPU's constant-time backend
 * (curve25519-dalek), which is what the shim of INTEGRATION.md section 3 does.
 *
 * Threads: entry points may be called from any thread; calls on one context (and on the params objects created on
 * it) are serialised by a lock inside the context, and eg_last_error is per thread.  Host-pointer functions return
 * when the results are in the caller's buffers (they run on a stream owned by the context; the tally reset / encode host
 * forms first wait for the whole device, so they may follow `_device` calls directly).  `_device` / `_async` functions only
 * enqueue work on the hipStream_t they are given - NULL is HIP's null stream, as everywhere.  The context's per-lane workspace, which
 * every params object of a context and the primitive tier share, is ordered INSIDE the library (each user's stream waits for an event
 * of the previous user: no host synchronisation), so two params objects of one context may have work in flight at the same time - a JSON
 * stream on one, `_device` calls or eg_vartime_multi_mul_batch on the other - whatever the streams.  What a caller that uses several
 * streams with ONE params object must still order itself (events) are that object's own buffers: its chunk workspace and running tally.
 * LONG calls - eg_verify_*_json and eg_verify_*_batch_multi* - do not hold the context's lock from start to end, but they are one call all
 * the same: any other call on a params object they are using WAITS until they are over.
 *
 * RUN-TIME KNOBS.  Every environment variable the library looks at, complete (csrc/eg_hip.hip: struct Knobs, read_knobs - the only
 * place the environment is read).  They are read at TWO moments only - eg_init (context-wide knobs; the context keeps its copy) and the
 * creation of a params object (eg_*_params_create; the engine keeps its copy) - never per call: changing the environment afterwards
 * does nothing to live objects.  None of them changes a verdict or a tally (tests force each and compare with the oracle); the shipped
 * library has NO failure-injection switch (tests build their own copy with fault points: tests/faultlib).
 *
 *   name                  read at        default            effect
 *   EG_ALLOW_ANY_ARCH     eg_init        unset              accept a device that is not gfx950 (development only; kernels are built for gfx950)
 *   EG_COMB_BIG_BITS      eg_init        24                 window width of the WIDE fixed-base comb tables (11.8 GB per base); 0 = never build them
 *   EG_COMB_BIG_MIN       eg_init        524288             items an engine must have verified before the wide tables are built for it
 *   EG_MSM_BLOCKS_PER_CU  eg_init        32                 grid of the table / equation / multi-scalar kernels per CU (sets the 2.4 GB per-lane workspace)
 *   EG_MSM_LANES          eg_init        131072             lanes a multi-scalar call should cover before terms share doubling chains (1 = always chunks of 8)
 *   EG_MSM_BUCKET_MIN     eg_init        1048576            terms per product from which the bucket (Pippenger) method replaces Straus; eg_msm_scratch_bytes follows it
 *   EG_TEETH              params create  by plan (5 or 6)   comb shape of the per-ballot tables: 5 x 51 (2 KiB) or 6 x 43 (4 KiB)
 *   EG_STREAMS            params create  2                  work sets (each with its own stream) the chunks of a call alternate between: 1 or 2
 *   EG_CHUNK              params create  by plan / memory   ballots per chunk and work set (default 2^19, or 2^18 where two sets would pass 40 GB)
 *   EG_RING_GROUP         params create  by plan            ring-group walk of choice ballots: rings whose tables are resident at a time; 0 = all
 *                                                           (default: all below 256 options); a memory knob, 0.6-4 % slower
 *   EG_JSON_RING_KB       params create  1048576            pinned staging ring of eg_verify_*_json, KiB
 *   EG_JSON_WINDOW_KB     params create  98304              JSON text per parser window, KiB
 *   EG_JSON_GROWTH        params create  150                a second JSON submission is enqueued once it is this many per cent of the first
 *   EG_JSON_FIRST_MIN     params create  16384              packed ballots the JSON entry points wait for before their first GPU submission
 *   EG_JSON_ODD_MAX_MB    params create  256                text of ballots whose shape is not the election's that a JSON stream keeps for the
 *                                                           object path (they are resolved at the end); a text with more fails with EG_ERR_NOMEM
 *   EG_JSON_TRACE         params create  unset              timeline of the JSON submissions on stderr
 *
 * The Python mirror adds two of its own (elastic_elgamal_amd/__init__.py, read at import): EG_LIB = path of another build of this
 * library (A/B measurements, the fault-point build of the tests), EG_NO_TORCH_PRELOAD = do not import torch before loading the library.
 */
#ifndef EG_HIP_H
#define EG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  EG_OK = 0,
  EG_ERR_NO_DEVICE = -1,
  EG_ERR_HIP = -2,
  EG_ERR_BAD_ARG = -3,
  EG_ERR_BAD_PUBLIC_KEY = -4, /* PublicKey::from_bytes: invalid element or identity (keys/mod.rs:161-176) */
  EG_ERR_NOMEM = -5
} eg_error;

/* ---- per-ballot status words (uint32): low byte = kind, bits 8.. = detail ------------------------------
 * Kinds mirror the reference's error enums (the *_LEN kinds cannot arise in a packed batch, whose shape is fixed by the
 * params; they are produced by the object-ingest layer, elastic_elgamal_amd/ingest.py); precedence = order of checks in EncryptedChoice::verify
 * (choice.rs:358-380) and QuadraticVotingBallot::verify (quadratic_voting.rs:291-329), preceded by the
 * deserialisation-time rejections of serde (serde.rs:191-206,254-269) in wire order. */
enum {
  EG_ST_OK = 0,
  EG_ST_BAD_SCALAR = 1,           /* non-canonical scalar; detail = index of the 32-byte item in the ballot */
  EG_ST_BAD_POINT = 2,            /* invalid ristretto encoding; detail = item index */
  EG_ST_OPTIONS_LEN = 3,          /* ChoiceVerificationError::OptionsLenMismatch (choice.rs:409) */
  EG_ST_SUM_CHALLENGE = 4,        /* ChoiceVerificationError::Sum(ChallengeMismatch) (choice.rs:416) */
  EG_ST_RANGE_LEN = 5,            /* ...::Range(LenMismatch) (choice.rs:418, proofs/mod.rs:73-80) */
  EG_ST_RANGE_CHALLENGE = 6,      /* ...::Range(ChallengeMismatch) */
  EG_ST_QV_VARIANT_LEN = 7,       /* QuadraticVotingError::Variant{index, LenMismatch}; detail = index */
  EG_ST_QV_VARIANT_CHALLENGE = 8, /* QuadraticVotingError::Variant{index, ChallengeMismatch} */
  EG_ST_QV_CREDIT_RANGE_LEN = 9,
  EG_ST_QV_CREDIT_RANGE_CHALLENGE = 10, /* QuadraticVotingError::CreditRange (quadratic_voting.rs:341) */
  EG_ST_QV_CREDIT_EQUIV_LEN = 11,
  EG_ST_QV_CREDIT_EQUIV_CHALLENGE = 12, /* QuadraticVotingError::CreditEquivalence (:343) */
  EG_ST_MALFORMED = 13            /* the ballot object does not deserialise at all (bad base64url / byte length, fewer than 2
                                     responses, wrong proof kind: serde.rs:29-80,303-355); object-ingest layer only */
};
#define EG_STATUS_KIND(s) ((s) & 0xffu)
#define EG_STATUS_DETAIL(s) ((s) >> 8)

typedef struct eg_ctx eg_ctx;           /* one per process and GPU: device, stream, fixed-base table of G */
typedef struct eg_choice_params eg_choice_params; /* ChoiceParams<Ristretto, S> (choice.rs:132-196) */
typedef struct eg_qv_params eg_qv_params;         /* QuadraticVotingParams<Ristretto> (quadratic_voting.rs:47-76) */

/* ---- context ------------------------------------------------------------------------------------------------ */
/* Version of this ABI: bumped whenever an exported function changes its prototype or meaning (6 = round 6: eg_msm_scratch_bytes_ctx,
 * d_ok mandatory in eg_points_prepare_device, the multi-GPU JSON entries).  A binding checks it once after loading the library. */
#define EG_ABI_VERSION 6
int eg_abi_version(void);
int eg_init(int device, eg_ctx** out);   /* no reference analogue: the backend is a ZST */
void eg_destroy(eg_ctx* ctx);
const char* eg_last_error(void);         /* text of the last failure on this thread */
int eg_device_name(eg_ctx* ctx, char* buf, size_t cap);
int eg_synchronize(eg_ctx* ctx);         /* waits for all work on the context's device, whatever stream it was enqueued on */

/* ---- primitive tier (host buffers; n problems per call) ------------------------------------------------------- */
/* ScalarOps::scalar_from_random_bytes / Scalar::from_bytes_mod_order_wide (ristretto.rs:34-38) */
int eg_scalar_from_wide_batch(eg_ctx*, size_t n, const uint8_t* wide /*64n*/, uint8_t* out /*32n*/);
/* ScalarOps::deserialize_scalar -> is Some (ristretto.rs:59-62) */
int eg_scalar_is_canonical_batch(eg_ctx*, size_t n, const uint8_t* s /*32n*/, uint8_t* ok /*n*/);
/* a*b + c, -a  (Scalar Mul/Add/Neg used at ring.rs:192-193,339) */
int eg_scalar_muladd_batch(eg_ctx*, size_t n, const uint8_t* a, const uint8_t* b, const uint8_t* c, uint8_t* out);
int eg_scalar_neg_batch(eg_ctx*, size_t n, const uint8_t* a, uint8_t* out);
/* ScalarOps::invert_scalar / invert_scalars (group/mod.rs:104-118, ristretto.rs:40-52): 1/a mod l; 0 maps to 0 */
int eg_scalar_invert_batch(eg_ctx*, size_t n, const uint8_t* a, uint8_t* out);
/* ElementOps::deserialize_element then serialize_element (ristretto.rs:88-95): ok[i] = 1 for a valid
 * encoding, and out = re-encoding (equal to the input for every valid encoding) */
int eg_point_roundtrip_batch(eg_ctx*, size_t n, const uint8_t* in /*32n*/, uint8_t* out /*32n*/, uint8_t* ok /*n*/);
/* ElementOps::is_identity (ristretto.rs:80-82): is_identity[i] = 1 iff the encoding is valid and is the identity */
int eg_point_is_identity_batch(eg_ctx*, size_t n, const uint8_t* in /*32n*/, uint8_t* is_identity /*n*/, uint8_t* ok /*n*/);
/* Element Add / Sub (ristretto.rs:76-86 via RistrettoPoint ops); ok[i]=0 if an input fails to decode.
 * Element Neg is `subtract` with a = identity (32 zero bytes). */
int eg_point_add_batch(eg_ctx*, size_t n, const uint8_t* a, const uint8_t* b, int subtract, uint8_t* out, uint8_t* ok);
/* TranscriptForGroup's framing (proofs/mod.rs:39-57 over merlin 3.0.0): for each of n messages,
 * Transcript::new(proto); append_message(msg_label, msg_i); challenge_bytes(chal_label, out_len) -> out_i.  Lets known-answer
 * vectors of the transcript layer (the upstream merlin test vector) be checked on the device itself. */
int eg_merlin_challenge_batch(eg_ctx*, size_t n, const char* proto, size_t proto_len, const char* msg_label, size_t msg_label_len,
                              const uint8_t* msgs /* n x msg_len */, size_t msg_len, const char* chal_label, size_t chal_label_len,
                              uint8_t* out /* n x out_len */, size_t out_len);
/* Group::mul_generator / vartime_mul_generator (ristretto.rs:105-121); variable time like everything here (see TIMING): this IS
 * vartime_mul_generator, and stands in for mul_generator only where the scalar is public */
int eg_mul_generator_batch(eg_ctx*, size_t n, const uint8_t* k /*32n*/, uint8_t* out /*32n*/);
/* Group::vartime_double_mul_generator(k, P, r) = [k]P + [r]G (ristretto.rs:131-137) */
int eg_vartime_double_mul_generator_batch(eg_ctx*, size_t n, const uint8_t* k, const uint8_t* p, const uint8_t* r,
                                          uint8_t* out, uint8_t* ok);
/* Group::vartime_multi_mul(scalars, elements) (ristretto.rs:139-145): n problems of `terms` terms each,
 * scalars/points laid out [problem][term][32] */
int eg_vartime_multi_mul_batch(eg_ctx*, size_t n, size_t terms, const uint8_t* scalars, const uint8_t* points,
                               uint8_t* out, uint8_t* ok);
/* The same multi-scalar multiplication on DEVICE buffers, asynchronous on `stream`: out_i = enc( sum_t [k_it]P_it + [r_i]G ).
 * d_r may be NULL (no generator term) unless terms == 0; d_ok may be NULL.  Any number of terms up to 2^24: the terms of a problem
 * are evaluated in chunks of 8 on shared doubling chains (Straus, what dalek's vartime_multiscalar_mul does below 190 terms) and the
 * chunks' partial sums are added up with wavefront shuffles; from 2^20 terms per problem on, the bucket method (Pippenger, dalek's choice
 * above 190 terms; csrc/pippenger.cuh) takes over, one problem after the other: 13 ms instead of 21 for 2^22 terms.  A call that is cut
 * into several chunks per problem (more than 8 terms, or few problems of many terms) or that uses the bucket method needs d_scratch of
 * eg_msm_scratch_bytes_ctx(ctx, n, terms) bytes (0 when it needs none; ~250 bytes per term for the bucket method: 1 GB at 2^22 terms).  A caller
 * that keeps its operands in HBM pays no copy and no synchronisation. */
/* (named _ctx since ABI version 6: earlier builds exported this function under the name without the suffix, first WITHOUT the context
 * argument, then with it - a caller built against the old prototype must fail to link, not link and read a garbage size) */
size_t eg_msm_scratch_bytes_ctx(eg_ctx*, size_t n, size_t terms);   /* by the context: its switch to the bucket method is fixed at eg_init */
int eg_vartime_multi_mul_batch_device(eg_ctx*, size_t n, size_t terms, const void* d_scalars, const void* d_points, const void* d_r,
                                      void* d_scratch, void* d_out, void* d_ok, void* stream);

/* Prepared points.  In the reference an Element IS a decoded point, so Group::vartime_multi_mul (ristretto.rs:139-145) never pays a
 * decoding per product; the entry above, which takes 32-byte encodings, does - 285 field operations per term, a third of the bucket
 * method's time.  A caller that multiplies over one point set repeatedly decodes it once: eg_points_prepare_device writes
 * eg_prepared_point_size() (= 96) bytes per point - affine (x, y, xy), an opaque layout that is valid for this library build only, never
 * a wire format - and d_ok[i] = 1 if encoding i decodes.  An encoding that does not decode is prepared as the IDENTITY and contributes
 * nothing to a product; the reference cannot even construct such an Element (deserialize_element returns None), so d_ok is MANDATORY
 * (EG_ERR_BAD_ARG if NULL): the prepare call is the only place that says which inputs were not elements.  d_prepared must be 16-byte
 * aligned (hipMalloc'd memory is; EG_ERR_BAD_ARG otherwise - the kernels read it in 128-bit words), also in
 * eg_vartime_multi_mul_prepared_batch_device, which is eg_vartime_multi_mul_batch_device over such points (terms per problem x n problems,
 * 96 bytes each, same scratch, same paths) and has no ok output of its own. */
size_t eg_prepared_point_size(void);
int eg_points_prepare_device(eg_ctx*, size_t n, const void* d_encodings, void* d_prepared, void* d_ok, void* stream);
int eg_vartime_multi_mul_prepared_batch_device(eg_ctx*, size_t n, size_t terms, const void* d_scalars, const void* d_prepared, const void* d_r,
                                               void* d_scratch, void* d_out, void* stream);

/* ---- tally stage (examples/voting.rs:122-177) -------------------------------------------------------------------------------
 * Params::combine_shares (src/sharing/mod.rs:302-325, lagrange_coefficients :139-170): combines the FIRST `threshold` of the n given
 * decryption shares (zero-based participant indexes, dh elements of 32 bytes; verify them first: eg_share_params_create) into the
 * decryption [x]R of the shared key.  *combined = 0 (and EG_OK) when fewer than `threshold` shares are given (the reference
 * returns None); an index >= shares or a duplicate index is EG_ERR_BAD_ARG (the reference panics on the former).  All scalar and
 * group arithmetic runs on the GPU primitives. */
int eg_combine_shares(eg_ctx*, uint64_t shares, uint64_t threshold, size_t n, const uint64_t* indexes, const uint8_t* dh_elements,
                      uint8_t out[32], int* combined);
/* DiscreteLogTable (src/encryption.rs:260-298): new(values) computes [v]G for every non-zero value on the GPU; get() looks decrypted
 * elements up (found[i] = 0: not among the values; the identity is always 0).  The table itself lives in host memory. */
typedef struct eg_dlog_table eg_dlog_table;
int eg_dlog_table_create(eg_ctx*, size_t n, const uint64_t* values, eg_dlog_table** out);
void eg_dlog_table_destroy(eg_dlog_table*);
int eg_dlog_table_get(const eg_dlog_table*, size_t n, const uint8_t* elements /*32n*/, uint64_t* values, uint8_t* found);

/* ---- batch tier: EncryptedChoice --------------------------------------------------------------------------------
 * wire layout of one ballot (stride = eg_choice_ballot_size):
 *   n_options x Ciphertext::to_bytes (R || B, encryption.rs:155-160)
 *   RingProof::to_bytes            (e0 || s_0 .. s_{2n-1}, ring.rs:383-392)
 *   LogEqualityProof::to_bytes     (c || s, log_equality.rs:184-189)      -- SingleChoice only */
int eg_choice_params_create(eg_ctx*, const uint8_t pk[32], int n_options, int single, eg_choice_params** out);
void eg_choice_params_destroy(eg_choice_params*);
size_t eg_choice_ballot_size(int n_options, int single);
/* EncryptedChoice::verify over a batch + homomorphic tally of the accepted ballots
 * (examples/voting.rs:199-203).  status: n words.
 * Tally semantics (the same for the host and the `_device` forms, and for eg_verify_qv_batch): every call ADDS the
 * ciphertexts of the ballots it accepts to the running tally kept inside `params`; only eg_*_tally_reset clears the running
 * tally, eg_*_tally_add imports into it, eg_*_tally_encode reads it.  tally_out (may be NULL) additionally receives the tally
 * of THIS call's batch alone, n_options x 64 bytes (R || B); it never disturbs the running tally. */
int eg_verify_choice_batch(eg_choice_params*, size_t n, const uint8_t* ballots, uint32_t* status, uint8_t* tally_out);
/* device-resident variant: d_ballots / d_status are device pointers; the running tally stays on the device
 * inside `params` until eg_choice_tally_* is called.  Asynchronous on `stream` (internally the work forks onto two streams of the
 * params object and joins `stream` again before the call returns control of it), with TWO exceptions that block the host once:
 *  - the first call of a params object (and any later call with a larger batch) allocates the chunk workspace
 *    (hipDeviceSynchronize + hipMalloc; ~45 KB per ballot of a chunk for 5 options, two chunks of up to 2^19 ballots);
 *  - the call with which a params object has seen 2^19 ballots in total builds the WIDE fixed-base comb tables inline: about
 *    12 GB each for the generator (once per context) and the election key (once per params object), ~40 ms, with a
 *    hipStreamSynchronize on `stream`.  A host that needs strict asynchrony (or wants the memory accounted for up front) calls
 *    eg_choice_prepare_wide_tables / eg_qv_prepare_wide_tables right after creating the params; EG_COMB_BIG_BITS=0 switches the
 *    wide tables off (-1..2 % throughput).  If the wide tables do not fit the device memory the narrow ones stay in use. */
int eg_verify_choice_batch_device(eg_choice_params*, size_t n, const void* d_ballots, void* d_status, void* stream);
/* builds the wide comb tables now (synchronous); EG_ERR_NOMEM if they do not fit, EG_OK when they exist or are switched off */
int eg_choice_prepare_wide_tables(eg_choice_params*);
int eg_choice_tally_reset(eg_choice_params*);
int eg_choice_tally_encode(eg_choice_params*, uint8_t* out /* n_options*64 */);
/* running tally += the ciphertexts encoded in `in` (n_options*64 bytes, as eg_choice_tally_encode writes them): resume
 * from a checkpoint, or merge the tally of batches verified elsewhere.  EG_ERR_BAD_ARG (and slots with invalid encodings
 * untouched) if an encoding does not decode. */
int eg_choice_tally_add(eg_choice_params*, const uint8_t* in /* n_options*64 */);
/* asynchronous forms for the multi-GPU path: reset on a stream; write the canonical encodings (n_options*64 bytes)
 * to device memory, ready for an RCCL all-gather; then sum the gathered encodings with eg_points_sum_device. */
int eg_choice_tally_reset_async(eg_choice_params*, void* stream);
int eg_choice_tally_encode_device(eg_choice_params*, void* d_out, void* stream);
/* d_out[k] = encode( sum_r decode(d_in[r][k]) ), k < n_points, r < n_ranks; 32-byte encodings.  d_bad (device uint32, may be
 * NULL; the caller zeroes it) is incremented once per point slot that received an encoding which does not decode: such a
 * term contributes the identity, so a caller merging tallies from other ranks must treat *d_bad != 0 as a failed exchange. */
int eg_points_sum_device(eg_ctx*, int n_ranks, int n_points, const void* d_in, void* d_out, void* d_bad, void* stream);

/* ---- batch tier: QuadraticVotingBallot ------------------------------------------------------------------------------
 * wire layout (stride = eg_qv_ballot_size), serde field order of quadratic_voting.rs:205-217 / range.rs:446-450 /
 * mul.rs:86-93:
 *   n_options x [ ciphertext(64) || partial_ciphertexts(64 each) || common_challenge(32) || ring_responses ]
 *   credit:      [ ciphertext(64) || partial_ciphertexts          || common_challenge     || ring_responses ]
 *   SumOfSquaresProof: challenge(32) || ciphertext_responses(2n x 32) || sum_response(32) */
int eg_qv_params_create(eg_ctx*, const uint8_t pk[32], int n_options, uint64_t credits, eg_qv_params** out);
void eg_qv_params_destroy(eg_qv_params*);
size_t eg_qv_ballot_size(const eg_qv_params*);
int eg_verify_qv_batch(eg_qv_params*, size_t n, const uint8_t* ballots, uint32_t* status, uint8_t* tally_out);
int eg_verify_qv_batch_device(eg_qv_params*, size_t n, const void* d_ballots, void* d_status, void* stream);
int eg_qv_prepare_wide_tables(eg_qv_params*);      /* as eg_choice_prepare_wide_tables */
int eg_qv_tally_reset(eg_qv_params*);
int eg_qv_tally_encode(eg_qv_params*, uint8_t* out);
int eg_qv_tally_add(eg_qv_params*, const uint8_t* in);
int eg_qv_tally_reset_async(eg_qv_params*, void* stream);
int eg_qv_tally_encode_device(eg_qv_params*, void* d_out, void* stream);

/* ---- batch tier on several GPUs of ONE process (SURVEY.md 8b `device_mask`; the reference's host is one single-threaded process,
 * examples/voting.rs:179-213) -----------------------------------------------------------------------------------------------------
 * per_device[d], d < n_dev: params objects of the SAME election (same key, options, kind), each created on its own context
 * (eg_init(d, ..): one per GPU; two contexts on one GPU work too).
 *
 * eg_verify_*_batch_multi (HOST buffers): the batch is cut into contiguous slabs - slab d = ballots [n d / n_dev, n (d + 1) / n_dev) -
 * and one host thread per params object inside the library runs eg_verify_*_batch on its slab (its own device, streams, uploads; pin
 * the ballots with hipHostRegister / hipHostMalloc for full upload speed).  status: n words, in ballot order.
 *
 * eg_verify_*_batch_multi_device (DEVICE buffers): slab d is ALREADY RESIDENT on GPU d - a host that generated or received its ballots
 * per GPU pays no copy.  n_per_dev[d] ballots at d_ballots[d], verdicts to d_status[d] (n_per_dev[d] words), both device pointers on
 * per_device[d]'s device; streams[d] a hipStream_t of that device (streams == NULL: the null streams).  One host thread per slab
 * enqueues eg_verify_*_batch_device on its stream and waits for that stream: the call returns with every verdict written.
 *
 * Tallies.  Every params object ADDS the accepted ballots of its slab to its OWN running tally; tally_out (may be NULL, n_options*64
 * bytes) receives the tally of this call's whole batch alone, the slabs' tallies merged with the element addition of the primitive
 * tier - the in-process counterpart of the RCCL all-gather of the one-process-per-GPU path (examples/tally_exchange.cpp).
 * eg_*_tally_encode_multi = the sum of the running tallies (every encoding is checked: a running tally that does not decode, or params
 * objects of different elections, fail the call).
 *
 * A multi call HOLDS its params objects from its first check to its merge: it waits for a one-shot eg_verify_*_json that is running on
 * one of them, fails with EG_ERR_BAD_ARG - before it has touched anything - when an explicitly opened JSON stream owns one, and calls of
 * other threads on a held object wait until it is over.  Nothing in it drains a device: the running tallies are set aside behind an event
 * of the object's last asynchronous call, and slab d starts behind that copy on streams[d].  The calling thread's current HIP device is
 * put back to what it was when the call returns (the multi-GPU JSON entries do the same).
 *
 * AFTER A FAILURE.  An error in any slab fails the whole call; eg_last_error names the slab.  Verdicts of other slabs may have been
 * written, but NO running tally has advanced: every multi verify call sets the running tallies aside before it starts (a
 * device-to-device copy of 2 n_options points on each GPU) and puts them back when any slab, or the final merge, fails - the caller may simply retry the batch, or go on with
 * the next one, without resetting anything.  Only if putting them back fails as well (the device is gone) does the error text say
 * "could not be restored"; then reset every params object (eg_*_tally_reset) and re-import the last checkpoint (eg_*_tally_add).
 * Nothing thrown inside a slab's thread crosses the ABI: it becomes that slab's error. */
int eg_verify_choice_batch_multi(eg_choice_params* const* per_device, int n_dev, size_t n, const uint8_t* ballots, uint32_t* status,
                                 uint8_t* tally_out);
int eg_verify_qv_batch_multi(eg_qv_params* const* per_device, int n_dev, size_t n, const uint8_t* ballots, uint32_t* status,
                             uint8_t* tally_out);
int eg_verify_choice_batch_multi_device(eg_choice_params* const* per_device, int n_dev, const size_t* n_per_dev, const void* const* d_ballots,
                                        void* const* d_status, void* const* streams, uint8_t* tally_out);
int eg_verify_qv_batch_multi_device(eg_qv_params* const* per_device, int n_dev, const size_t* n_per_dev, const void* const* d_ballots,
                                    void* const* d_status, void* const* streams, uint8_t* tally_out);
int eg_choice_tally_encode_multi(eg_choice_params* const* per_device, int n_dev, uint8_t* out /* n_options*64 */);
int eg_qv_tally_encode_multi(eg_qv_params* const* per_device, int n_dev, uint8_t* out /* n_options*64 */);

/* ---- batch tier: single-ciphertext proofs --------------------------------------------------------------
 * PublicKey::verify_zero (keys/impls.rs:59-69)   item = ciphertext(64) || challenge || response              128 B
 * PublicKey::verify_bool (keys/impls.rs:100-112) item = ciphertext(64) || e0 || s0 || s1                     160 B
 * PublicKey::verify_range(keys/impls.rs:142-151) item = ciphertext(64) || partial_ciphertexts || e0 || responses
 *   with RangeDecomposition::optimal(upper_bound) (range.rs:148-153); eg_proof_item_size gives the stride.
 * status: EG_ST_OK, EG_ST_BAD_*, EG_ST_SUM_CHALLENGE (zero: ChallengeMismatch) or EG_ST_RANGE_CHALLENGE (bool/range). */
typedef struct eg_proof_params eg_proof_params;
enum { EG_PROOF_ZERO = 0, EG_PROOF_BOOL = 1, EG_PROOF_RANGE = 2, EG_PROOF_SHARE = 3, EG_PROOF_SUMSQ = 4 };
int eg_proof_params_create(eg_ctx*, const uint8_t pk[32], int kind, uint64_t upper_bound, eg_proof_params** out);
/* PublicKeySet::verify_share (sharing/key_set.rs:209-228) for one participant: item = the ciphertext's
 * random element R(32) || decryption share dh(32) || challenge || response; status EG_ST_SUM_CHALLENGE on ChallengeMismatch.
 * Verify with eg_verify_proof_batch. */
int eg_share_params_create(eg_ctx*, const uint8_t shared_key[32], uint64_t shares, uint64_t threshold, uint64_t index,
                           const uint8_t participant_key[32], eg_proof_params** out);
/* SumOfSquaresProof::verify(ciphertexts, sum_of_squares_ciphertext, receiver, transcript) (mul.rs:190-260) with
 * transcript = Transcript::new(label) (tests/snapshots.rs:133-153 uses b"test"): item = n_values value ciphertexts (64 B each) ||
 * sum-of-squares ciphertext (64) || challenge || ciphertext_responses (2 n_values x 32) || sum_response; status EG_ST_OK,
 * EG_ST_BAD_* or EG_ST_QV_CREDIT_EQUIV_CHALLENGE (ChallengeMismatch).  A wrong number of responses (LenMismatch, mul.rs:197-202)
 * cannot be expressed in a packed item.  Verify with eg_verify_proof_batch. */
int eg_sumsq_params_create(eg_ctx*, const uint8_t pk[32], int n_values, const char* label, size_t label_len, eg_proof_params** out);
void eg_proof_params_destroy(eg_proof_params*);
size_t eg_proof_item_size(const eg_proof_params*);
int eg_verify_proof_batch(eg_proof_params*, size_t n, const uint8_t* items, uint32_t* status);
int eg_verify_proof_batch_device(eg_proof_params*, size_t n, const void* d_items, void* d_status, void* stream);

/* ---- synthetic ballots on the GPU (EncryptedChoice::new / QuadraticVotingBallot::new) ---------
 * VARIABLE TIME in the choices / votes and in the RNG-drawn secrets (see TIMING at the top): these entry points make test and benchmark
 * inputs; they are not a voting client.
 * Ballot i of the call is produced from ChaChaRng::seed_from_u64(base_seed + first + i) with the reference's
 * RNG draw order (choice.rs:313-349, ring.rs:54-194, log_equality.rs:114-139, range.rs:462-534, mul.rs:107-181);
 * the voter's selection comes from a second stream seeded with the complemented seed.  n_selected is only
 * used for multi-choice params.  A ballot's secrets live in a per-lane workspace in device memory sized from the election's
 * shape, so the generators accept every election the verifiers accept (n_options up to 4000 / 256). */
int eg_choice_encrypt_batch_device(eg_choice_params*, uint64_t base_seed, size_t first, size_t n, int n_selected,
                                   void* d_out, void* stream);
int eg_choice_encrypt_batch(eg_choice_params*, uint64_t base_seed, size_t first, size_t n, int n_selected,
                            uint8_t* out /* host, n * eg_choice_ballot_size */);
int eg_qv_encrypt_batch_device(eg_qv_params*, uint64_t base_seed, size_t first, size_t n, void* d_out, void* stream);
/* The same provers with the CALLER's choices: EncryptedChoice::single(params, choice, rng) / ::new(params, &[bool], rng)
 * (choice.rs:296-349) and QuadraticVotingBallot::new(params, votes, rng) (quadratic_voting.rs:234-284).  selection:
 * ceil(n_options / 32) uint32 per ballot, bit k of the bitmask set <=> option k chosen (single-choice: exactly one bit); votes:
 * n_options uint32 per ballot with
 * sum(v^2) <= credits.  rng_skip = number of 64-byte draws the ballot's RNG, ChaChaRng::seed_from_u64(base_seed + first + i), has
 * already served: tests/snapshots.rs:107-161 draw the keypair first, so (12345, rng_skip = 1) reproduces the reference's
 * `encrypted-choice`, `encrypted-multi-choice` and `qv-ballot` snapshots byte for byte.  The host forms validate the choices
 * (EG_ERR_BAD_ARG); the device forms trust them, as the reference trusts its caller's assertions. */
int eg_choice_encrypt_selected_batch(eg_choice_params*, uint64_t base_seed, size_t first, size_t n, uint64_t rng_skip,
                                     const uint32_t* selection /* n x ceil(n_options / 32) */, uint8_t* out);
int eg_choice_encrypt_selected_batch_device(eg_choice_params*, uint64_t base_seed, size_t first, size_t n, uint64_t rng_skip,
                                            const void* d_selection, void* d_out, void* stream);
int eg_qv_encrypt_votes_batch(eg_qv_params*, uint64_t base_seed, size_t first, size_t n, uint64_t rng_skip,
                              const uint32_t* votes /* n x n_options */, uint8_t* out);
int eg_qv_encrypt_votes_batch_device(eg_qv_params*, uint64_t base_seed, size_t first, size_t n, uint64_t rng_skip,
                                     const void* d_votes, void* d_out, void* stream);

/* ---- wire ingest (host only, no GPU needed) ------------------------------------------------------------
 * The reference's serde layout in human-readable formats (src/serde.rs:19-80,179-355: every scalar / element an unpadded base64url
 * string; structures as derived on EncryptedChoice, RingProof, LogEqualityProof, QuadraticVotingBallot, RangeProof,
 * SumOfSquaresProof, Ciphertext) -> packed ballots.  `json` holds the ballots as ONE JSON array of objects or as objects back to
 * back / one per line (what examples/voting.rs:195-198 prints); object k is written to packed + k * ballot_size (zeroed unless
 * status[k] == EG_ST_OK) and gets status[k] =
 *   EG_ST_OK         packed;
 *   EG_ST_MALFORMED  does not deserialise: bad alphabet / padding / non-zero trailing bits / byte length != 32
 *                    (serde.rs:29-47,197,260), fewer than 2 ring_responses or ciphertext_responses (VecHelper<_, 2>, :303-355),
 *                    missing or duplicate field, sum_proof of the wrong kind;
 *   EG_PACK_RESHAPE  deserialises, but the number of choices / responses / partial ciphertexts is not the election's
 *                    (OptionsLenMismatch / LenMismatch, choice.rs:149-158, proofs/mod.rs:73-99): such objects go through the object
 *                    path (elastic_elgamal_amd/ingest.py), which applies the reference's order of checks.
 * Canonicity of scalars and validity of elements are decided later, by the GPU verifier (EG_ST_BAD_SCALAR / EG_ST_BAD_POINT with
 * the item index).  `threads` host threads parse in parallel.  *n_objects = number of objects found; EG_ERR_BAD_ARG if the text is
 * not a sequence of objects or holds more than max_objects of them. */
#define EG_PACK_RESHAPE 0xfffffffeu
int eg_choice_pack_json(int n_options, int single, const char* json, size_t json_len, int threads, size_t max_objects,
                        uint8_t* packed, uint32_t* status, size_t* n_objects);
int eg_qv_pack_json(int n_options, uint64_t credits, const char* json, size_t json_len, int threads, size_t max_objects,
                    uint8_t* packed, uint32_t* status, size_t* n_objects);
size_t eg_qv_ballot_size_for(int n_options, uint64_t credits);   /* eg_qv_ballot_size without a params object (host only) */

/* ---- the JSON text in PIECES (examples/voting.rs:195-198 prints ballots one at a time; src/serde.rs:19-80 is the layout) ----------------
 * eg_verify_{choice,qv}_json_begin opens a stream on a params object; eg_verify_json_feed takes the next piece of the text - ANY size, a
 * ballot (a string, an escape sequence) may straddle any number of pieces; the text as a whole is what eg_verify_*_json accepts: one JSON
 * array of objects, or objects back to back / one per line; eg_verify_json_end closes it.  Verdicts and tally are exactly those of the
 * one-shot entry on the concatenated text.
 *   feed   hands the piece to the stream's worker thread, which cuts it, packs its complete ballots on `threads` host threads into a
 *          pinned ring, and enqueues GPU work without waiting for it (the first submission once EG_JSON_FIRST_MIN ballots are packed, at
 *          most two in flight).  A piece below 8 MB is copied (one memcpy on the caller's thread; feed returns at once unless four 16 MB
 *          blocks - 64 MB of text - are already waiting); a larger piece is read in place and feed returns when the worker is through with it.  Either
 *          way nothing of `text` is referenced after feed returns.  *n_objects (may be NULL): complete objects the worker has cut so
 *          far - it may lag behind the pieces fed.  An error of the text (not a sequence of objects) or of the GPU found in a copied
 *          piece is reported by a LATER feed or take, and at the latest by end.
 *   take   (optional; waits at most for the worker to finish the window of the piece it is cutting - which, while the staging ring is
 *          full, includes the worker's own wait for the oldest submission - never for the GPU to drain) sends what has been copied so
 *          far to the worker, lets an idle GPU start on whatever is packed, and hands out, in order, the verdicts that are final so far: those of every ballot before the first one
 *          that is still on the GPU or whose shape is not the election's (such a ballot gets its OptionsLenMismatch / LenMismatch
 *          verdict from the object path, which runs at the end).
 *   end    waits for the GPU, resolves the ballots of another shape, writes the verdicts not yet taken (status: room for `cap`; if more
 *          are left the call fails with EG_ERR_BAD_ARG, *n_taken = the number that is left, and the stream stays open: call again with
 *          that much room, or abort), *n_objects = objects
 *          in the whole text, tally_out (may be NULL) = the tally of the STREAM's ballots; the params object's running tally has them
 *          added.  Destroys the stream - also when it reports an error of the text (not a sequence of objects, truncated) or of the GPU.
 *   abort  destroys the stream; the running tally is what it was before begin.  A failed feed leaves the stream dead: end returns the
 *          same error and cleans up, with the running tally as it was.
 * Host memory of a stream: the staging ring (EG_JSON_RING_KB), at most four 16 MB blocks of copied text, one status word per object, and the
 * TEXT of every ballot whose shape is not the election's (kept for the object path at the end): at most EG_JSON_ODD_MAX_MB (256 MB), beyond
 * which the stream fails with EG_ERR_NOMEM instead of buffering its input.
 * Between begin and end / abort the params object belongs to the stream: every other verify / tally call on it fails with EG_ERR_BAD_ARG
 * (the multi-GPU entries included: they refuse before they touch anything).
 * One stream per params object; several params objects (contexts, GPUs) may stream at the same time from different threads.
 * (The one-shot entries eg_verify_*_json run on the same pipeline, but a one-shot call is ONE call like any other: while it runs, calls of
 * other threads on the same params object - verify, tally, another one-shot, a begin - WAIT for it, they are not refused.) */
typedef struct eg_json_stream eg_json_stream;
int eg_verify_choice_json_begin(eg_choice_params*, int threads, eg_json_stream** out);
int eg_verify_qv_json_begin(eg_qv_params*, int threads, eg_json_stream** out);
int eg_verify_json_feed(eg_json_stream*, const char* text, size_t len, size_t* n_objects);
/* feed WITHOUT a copy and without waiting for the worker: the block stays the caller's memory but belongs to the library until it calls
 * release(user, text, len) - exactly once per successful call, from the stream's worker thread (or from the thread that ends / aborts the
 * stream, for blocks the worker never got to); release must not call back into the stream.  What a producer that receives the text in
 * its own buffers (a socket, a file mapping) uses: nothing is copied on its thread (eg_verify_json_feed copies a piece below 8 MB).
 * Returns at once unless ~64 MB of text are already waiting.  If the call FAILS, release is not called: the block is still the caller's. */
typedef void (*eg_json_release_fn)(void* user, const char* text, size_t len);
int eg_verify_json_feed_owned(eg_json_stream*, const char* text, size_t len, eg_json_release_fn release, void* user, size_t* n_objects);
/* a ready-made eg_json_release_fn: adds one to the size_t that `user` points to (atomically; NULL: does nothing).  For a caller that keeps its
 * blocks alive itself until the stream has ended and only wants to know that every block came back - and for bindings whose own callbacks
 * are expensive on a foreign thread (a Python callback takes the interpreter lock on the stream's worker thread, once per block). */
void eg_json_release_count(void* user, const char* text, size_t len);
int eg_verify_json_take(eg_json_stream*, uint32_t* status, size_t cap, size_t* n_taken);
int eg_verify_json_end(eg_json_stream*, uint32_t* status, size_t cap, size_t* n_taken, size_t* n_objects, uint8_t* tally_out);
void eg_verify_json_abort(eg_json_stream*);
/* JSON text -> verdicts and tally in one call (what a service that receives the output of examples/voting.rs:195-198 needs).  A pool of
 * `threads` host threads cuts the text into windows and packs them into a pinned ring (<= 1 GiB) while the calling thread uploads the
 * finished windows and enqueues their verification, two submissions in flight; memory: the ring, as much device staging, the chunk
 * workspace of the params object.  status[k]: the verify verdict of object k (as eg_verify_*_batch); EG_ST_MALFORMED for an object that
 * does not deserialise; for an object that deserialises with another shape than the election's, the reference's verdict from the object
 * path (EG_ST_OPTIONS_LEN, the LenMismatch variants, or whatever verify() says after them: src/app/choice.rs:358-380,
 * src/proofs/mod.rs:73-99).  Tally semantics as eg_verify_*_batch.  *n_objects = objects found; EG_ERR_BAD_ARG if the text is not a
 * sequence of JSON objects or holds more than max_objects (then no verdict of the call is valid; the call breaks off as soon as the parser
 * has cut one object too many - nothing further is parsed or verified). */
int eg_verify_choice_json(eg_choice_params*, const char* json, size_t json_len, int threads, size_t max_objects, uint32_t* status,
                          size_t* n_objects, uint8_t* tally_out);
int eg_verify_qv_json(eg_qv_params*, const char* json, size_t json_len, int threads, size_t max_objects, uint32_t* status,
                      size_t* n_objects, uint8_t* tally_out);

/* ---- the JSON entries over several GPUs of ONE process (the parser delivers 10 M ballots/s on 16 threads, one GPU verifies 6) ----
 * per_device[d]: params objects of the SAME election, each on its own context (as for eg_verify_*_batch_multi).  ONE parser - one splitter,
 * one pool of `threads` host threads - cuts and packs the text; every packed window (a few thousand ballots) goes to the params object with
 * the fewest ballots waiting or in flight, each of which has its own pinned ring, device staging and two submissions in flight (the pinned
 * budget of EG_JSON_RING_KB is divided among them, at least 128 MiB each).  status[k] is the verdict of object k of the TEXT, whichever GPU
 * verified it; every params object adds the accepted ballots of the windows it verified to its OWN running tally; tally_out = the tally of
 * the call's (the stream's) whole text, the lanes' tallies merged with the element addition of the primitive tier (eg_*_tally_encode_multi
 * sums the running tallies).  Ballots of another shape than the election's are resolved through the first object.  Everything else -
 * pieces of any size, feed / feed_owned / take / end / abort, errors, the params objects belonging to the stream - as for one object. */
int eg_verify_choice_json_multi(eg_choice_params* const* per_device, int n_dev, const char* json, size_t json_len, int threads, size_t max_objects,
                                uint32_t* status, size_t* n_objects, uint8_t* tally_out);
int eg_verify_qv_json_multi(eg_qv_params* const* per_device, int n_dev, const char* json, size_t json_len, int threads, size_t max_objects,
                            uint32_t* status, size_t* n_objects, uint8_t* tally_out);
int eg_verify_choice_json_begin_multi(eg_choice_params* const* per_device, int n_dev, int threads, eg_json_stream** out);
int eg_verify_qv_json_begin_multi(eg_qv_params* const* per_device, int n_dev, int threads, eg_json_stream** out);

/* ---- host-only introspection (no GPU needed; used by the CPU-side tests of the host logic) -------------------------------------
 * RangeDecomposition::optimal(upper_bound).to_string() (range.rs:110-124,148-305): the string hashed into the transcript */
int eg_range_decomposition(uint64_t upper_bound, char* buf, size_t cap);
/* JSON summary of the verification plan: kind 0 single-choice, 1 multi-choice, 2 quadratic voting (credits), 3 verify_zero,
 * 4 verify_bool, 5 verify_range (upper bound), 6 sum-of-squares proof over n_options values */
int eg_plan_describe(int kind, int n_options, uint64_t credits_or_bound, char* buf, size_t cap);

/* ---- measurement hooks (bench.py) ------------------------------------------------------------------------------------------
 * Average duration in milliseconds of the dominant kernel (k_eq_table<false, T>: the ring equations) over the launches since the last reset,
 * measured with HIP events on the stream the kernel was launched on; launches = number of launches averaged.
 * enable = 2: also run the chunks of a call one after the other on one work set instead of alternating between two streams, so that a
 * launch shares the chip with nothing and its duration is its own (a measurement mode: slower, same results). */
int eg_profile_enable(eg_ctx*, int enable);
int eg_profile_read(eg_ctx*, double* msm_ms_total, uint64_t* msm_launches, double* all_ms_total);
/* same for the second kernel (k_base_tables), covering the launches folded in by the last eg_profile_read */
int eg_profile_read_tables(eg_ctx*, double* tables_ms_total, uint64_t* tables_launches);

/* The VALU roof of the box the library runs on: the shipped field multiplication (csrc/fe25519.cuh: fe_mul, the function the table and
 * equation kernels inline) in a bare dependent chain on changing operands, three waves per SIMD, launched back to back for `seconds`
 * (0 < seconds <= 30); *fmul_g_per_s = 10^9 multiplications per second chip-wide and *sclk_mhz (may be NULL) = the shader clock the chip
 * held (s_memtime / s_memrealtime), both over the second half of the run.  bench.py divides its achieved field-multiplication rate by this
 * figure (`valu_roofline.box`) instead of by a constant measured on another box of the pool.  The call holds the context for its length
 * (other threads' calls on the context wait) and keeps the whole chip busy: a measurement hook, not something to call beside production
 * work.  No reference analogue (the reference's own helper-multiplication benches: benches/basics.rs:284-319). */
int eg_selfbench_fmul(eg_ctx*, double seconds, double* fmul_g_per_s, double* sclk_mhz);

/* ---- self-check of the fixed-base comb tables (election setup; no reference analogue: dalek's basepoint table is a compile-time
 * constant) ---------------------------------------------------------------------------------------------------------------------
 * The tables of G are built on the device run by run with one batched inversion per run.  This recomputes `samples` entries (the
 * corners of windows and runs first, the rest drawn from `seed`) one at a time from the generator and counts the entries that differ.
 * wide = 0: the table built by eg_init (EG_COMB_BITS windows); wide = 1: the wide table that large batches use (built now if absent;
 * EG_ERR_NOMEM if it does not fit).  *mismatches must come back 0. */
int eg_selfcheck_generator_table(eg_ctx*, int wide, size_t samples, uint64_t seed, uint64_t* mismatches);
/* window widths of the generator's comb tables: the one built by eg_init, and the wide one (0 until an engine of this context has
 * verified enough items to get it, or eg_selfcheck_generator_table(wide = 1) built it) */
int eg_comb_table_bits(eg_ctx*, int* narrow_bits, int* wide_bits);

#ifdef __cplusplus
}
#endif
#endif

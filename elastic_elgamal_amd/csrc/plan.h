// plan.h -- the verification PLAN: a flat, POD description of one ballot shape that the generic device
// engine (kernels in eg_hip.hip) executes for every ballot of a batch.
//
// The reference verifies a ballot by walking Rust objects (EncryptedChoice::verify, choice.rs:358-380;
// RangeProof::verify, range.rs:547-577; RingProof::verify, ring.rs:302-374; SumOfSquaresProof::verify,
// mul.rs:190-260; QuadraticVotingBallot::verify, quadratic_voting.rs:291-329).  All ballots of one election
// have the same shape, so the host flattens that walk ONCE per election into:
//   * wire items      which 32-byte items of the packed ballot are points / scalars
//   * derived points  sums / differences of decoded points and election constants
//                     (sum of ciphertexts choice.rs:363; B - x_j ring.rs:338; ct - sum(partials) range.rs:572)
//   * job classes     per stage: out = encode( sum_i [a_i]P_i + [g]G + [k]K )  -- every group-side equation
//   * hash programs   per stage: Merlin transcript ops producing the next challenges / final verdict flags
//   * status rules    flag -> error code, in the reference's order of checks
// Stages exist because equation j+1 of a ring needs the challenge hashed from equation j (ring.rs:354-360).
#pragma once
#include <stdint.h>

namespace egplan {

enum : uint8_t { SRC_NONE = 0, SRC_WIRE = 1, SRC_CHAL = 2 };

struct ScalarSrc {
  uint8_t kind;   // SRC_*
  uint8_t neg;    // use -scalar (ring.rs:339, log_equality.rs:160, mul.rs:205)
  uint16_t idx;   // wire item index or challenge slot
  uint32_t pad;
};

struct VarTerm {
  uint16_t slot;  // point slot (wire or derived)
  uint16_t base;  // 0xffff: multiply directly; else index of the base whose comb table was precomputed (k_base_tables)
  ScalarSrc s;
};

// A base that is the sum of other table-backed bases: its comb table is the entry-wise sum of theirs (k_sum_tables), tables
// being linear in the base.  Table index out_base >= the number of ordinary bases.
struct SumBase {
  uint16_t first, count;  // members: sum_members[first .. first + count), indices of ordinary bases
  uint16_t out_base;      // table index of the sum
  uint16_t pad;
};

struct JobClass {
  uint16_t term_first, term_count;  // variable-base terms
  ScalarSrc g, k;                   // fixed-base scalars for G and K
  uint16_t out_slot;                // compressed-output slot
  uint16_t enc_slot;                // if term_count == 0 and no g/k: just encode this point slot
  uint16_t defer;                   // 1: evaluate with halved scalars and leave the point for k_encode_batch (out = encode(2P))
  uint16_t pad;
};

struct DeriveTerm {
  uint16_t slot;
  uint8_t is_const;  // 0: ballot point slot, 1: election-constant point
  uint8_t neg;
};
struct DeriveClass {
  uint16_t term_first, term_count;
  uint16_t out_slot;
  uint16_t pad;
};

enum : uint32_t {
  OP_NEW = 1,          // a = label                 Transcript::new(label)
  OP_APPEND_BLOB,      // a = label, b = blob ref   append_message(label, constant bytes)
  OP_APPEND_WIRE,      // a = label, b = first item, c = item count
  OP_APPEND_CMP,       // a = label, b = slot, c = second slot or 0xffff
  OP_APPEND_U64,       // a = label, b = value
  OP_CHALLENGE,        // a = label, b = challenge slot, c = m  challenge_scalar -> slot; if m > 1 also m * challenge -> slot + 1
                       //                                       (the [e * m_j]G term of B - x_j, x_j = [m_j]G, ring.rs:338)
  OP_CHALLENGE_CHECK,  // a = label, b = wire item, c = flag    flag = (challenge_scalar == wire scalar)
  OP_LOAD_PREFIX,      // b = prefix index
  OP_SAVE_PREFIX,      // b = prefix index
  OP_LOAD_STATE,       // b = state slot
  OP_SAVE_STATE        // b = state slot
};
struct HashOp { uint32_t op, a, b, c; };
struct HashInst { uint32_t op_first, op_count; };

struct StatusRule {
  uint32_t flag_slot;
  uint32_t status;   // status word when the flag is false
};

struct WireItem {
  uint16_t item;     // 32-byte item index inside the ballot
  uint16_t slot;     // point slot for points; unused for scalars
};

inline uint32_t blob_ref(uint32_t off, uint32_t len) { return (off << 12) | len; }

}  // namespace egplan

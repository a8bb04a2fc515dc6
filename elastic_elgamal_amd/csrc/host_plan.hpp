// host_plan.hpp -- host-side mirror of the reference's election parameter objects and the flattening of
// their verify() walks into a device plan (plan.h).  Pure host logic, no arithmetic on secrets or points:
// all curve / hash work happens in the kernels.
//
// Mirrors: ChoiceParams::{single,multi} (choice.rs:132-196), RangeDecomposition::optimal + Display
// (range.rs:110-124,148-305), PreparedRange (range.rs:329-355), QuadraticVotingParams::new + isqrt
// (quadratic_voting.rs:63-76,127-143), and the transcript label schedules of ring.rs:290-293,317-368,
// log_equality.rs:167-173, range.rs:561-562, mul.rs:96-99,204-253.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include "plan.h"

namespace eghost {

using namespace egplan;

// ---- RangeDecomposition (range.rs) ---------------------------------------------------------------------
struct RingSpec { uint64_t size, step; };
struct RangeDecomposition {
  std::vector<RingSpec> rings;
  std::string to_string() const {  // Display, range.rs:110-124 (hashed into the transcript at :562)
    std::string s;
    for (size_t i = 0; i < rings.size(); ++i) {
      if (rings[i].step > 1) s += std::to_string(rings[i].step) + " * ";
      s += "0.." + std::to_string(rings[i].size);
      if (i + 1 < rings.size()) s += " + ";
    }
    return s;
  }
  uint64_t rings_size() const { uint64_t t = 0; for (auto& r : rings) t += r.size; return t; }
  uint64_t upper_bound() const { uint64_t t = 0; for (auto& r : rings) t += (r.size - 1) * r.step; return t + 1; }
};
struct OptimalDecomposition { RangeDecomposition d; uint64_t len; };

inline uint64_t lower_len_estimate(uint64_t ub) { return (uint64_t)std::ceil(std::log2((double)ub) * 3.0); }  // range.rs:302-305

inline OptimalDecomposition optimize_range(uint64_t ub, std::map<uint64_t, OptimalDecomposition>& memo) {  // range.rs:238-300
  auto it = memo.find(ub);
  if (it != memo.end()) return it->second;
  OptimalDecomposition opt;
  opt.len = ub + 2;
  opt.d.rings = {{ub, 1}};
  for (uint64_t first = 2;; ++first) {
    if (first + 2 > opt.len) break;
    const uint64_t remaining = ub - first;
    for (uint64_t mult = 2; mult <= first; ++mult) {
      if (remaining % mult) continue;
      const uint64_t inner_ub = remaining / mult + 1;
      if (inner_ub < 2) break;
      if (first + 2 + lower_len_estimate(inner_ub) > opt.len) continue;
      OptimalDecomposition inner = optimize_range(inner_ub, memo);
      const uint64_t cand = first + 2 + inner.len;
      if (cand < opt.len || (cand == opt.len && 1 + inner.d.rings.size() < opt.d.rings.size())) {
        opt.len = cand;
        opt.d = inner.d;
        for (auto& r : opt.d.rings) r.step *= mult;
        opt.d.rings.push_back({first, 1});
      }
    }
  }
  memo[ub] = opt;
  return opt;
}
inline RangeDecomposition optimal_range(uint64_t upper_bound) {
  std::map<uint64_t, OptimalDecomposition> memo;
  return optimize_range(upper_bound, memo).d;
}
inline uint64_t isqrt(uint64_t x) {  // quadratic_voting.rs:127-143
  uint64_t root = 0, p4 = 1ull << 62;
  while (p4 > x) p4 /= 4;
  while (p4 > 0) {
    if (x >= root + p4) { x -= root + p4; root = root / 2 + p4; } else root /= 2;
    p4 /= 4;
  }
  return root;
}

// ---- plan under construction ---------------------------------------------------------------------------------
struct Stage {
  std::vector<JobClass> jobs;
  std::vector<std::vector<HashOp>> insts;
  std::vector<uint16_t> deferred;   // output slots encoded by k_encode_batch (one batched inversion per ballot and stage)
};
// What happens to the per-ballot comb tables at the START of a stage (filled in by flatten_plan from the plan's base registry):
//   build       point slots whose tables are built now, into table slots 0, 1, .. (k_base_tables)
//   sums_direct every table of the ballot exists at once: the tables of the sums of bases follow at once (k_sum_tables)
//   acc         ring-group walk: this group's members are added into the sums' accumulators (k_sum_accumulate)
//   sum_finish  ring-group walk: the sums' tables are made from the accumulators (k_sum_finish); the groups' tables are dead by then

struct Plan {
  size_t stride = 0;  // bytes per ballot
  std::vector<WireItem> pt_items, sc_items;
  std::vector<std::vector<DeriveClass>> derive_levels;
  std::vector<DeriveTerm> dterms;
  std::vector<VarTerm> vterms;
  std::vector<Stage> stages;
  std::vector<StatusRule> rules;
  std::vector<uint32_t> tally_slots;
  std::vector<uint8_t> blob;
  std::vector<uint64_t> const_mults;          // election-constant points [m]G
  std::vector<std::vector<HashOp>> prefix_programs;
  int n_pt_slots = 0, n_cmp_slots = 0, n_chal_slots = 0, n_state_slots = 0, n_flag_slots = 0, n_prefixes = 0;
  std::map<std::string, uint32_t> blob_index;
  std::map<uint64_t, uint16_t> const_index;
  // bases with a precomputed comb table (ge_teeth_tables_build): point slots that several equations multiply
  std::vector<uint16_t> base_slots;
  std::map<uint16_t, uint16_t> base_index_of;
  // bases that are sums of ordinary bases: their tables are made from the members' tables (k_sum_tables) and are indexed after
  // all ordinary bases, so no ordinary base may be registered once the first sum exists (`late_base` records a violation,
  // check_flat_plan refuses the plan)
  std::vector<SumBase> sum_bases;
  std::vector<uint16_t> sum_members;
  bool late_base = false;
  // Ring-group walk: the rings of a choice ballot are independent chains until the final challenge, so a chunk
  // can hold the tables of ONE group of rings at a time - tables of group g, its stages, then the same table slots for group g + 1 -
  // while the tables of the sums of bases are accumulated group by group.  ring_group = rings per group (0: every table of a ballot
  // exists at once, the layout of rounds 1-3).  A base belongs to the group that was current when it was registered and sits in
  // table slot base_local[] of that group; group_stage[g] is the stage at whose start group g's tables are built; sum_stage the stage
  // at whose start the sums' tables must exist.
  int ring_group = 0, cur_group = 0, sum_stage = 0;
  std::vector<uint16_t> base_group, base_local;
  std::vector<int> group_stage{0};
  size_t table_uses = 0;              // products over ordinary bases' tables (plan_teeth)
  bool grouped() const { return ring_group > 0; }
  void begin_group(int g, int first_stage) {
    cur_group = g;
    if ((int)group_stage.size() <= g) group_stage.resize(g + 1, first_stage);
    group_stage[g] = first_stage;
  }
  size_t group_size(int g) const { size_t n = 0; for (uint16_t x : base_group) n += x == g; return n; }
  uint16_t base_index(uint16_t slot) {
    auto it = base_index_of.find(slot);
    if (it != base_index_of.end()) return it->second;
    if (!sum_bases.empty() && !grouped()) late_base = true;
    const uint16_t i = (uint16_t)base_slots.size();
    base_local.push_back((uint16_t)group_size(cur_group));
    base_group.push_back((uint16_t)cur_group);
    base_slots.push_back(slot);
    base_index_of[slot] = i;
    return i;
  }
  bool has_base(uint16_t slot) const { return base_index_of.count(slot) != 0; }
  // VarTerm::base is the TABLE SLOT the kernels index (the base's slot inside its group; without groups = its index)
  VarTerm term(uint16_t slot, ScalarSrc sc) { return VarTerm{slot, has_base(slot) ? base_local[base_index_of[slot]] : (uint16_t)0xffff, sc}; }
  VarTerm bterm(uint16_t slot, ScalarSrc sc) { ++table_uses; return VarTerm{slot, base_local[base_index(slot)], sc}; }
  // [sc](sum of the points in `member_slots`), every member a ring base: one table-backed term over the sum's own table.
  // `slot` is the derived point that holds the sum (not read by the equation kernel).  Members may be registered later (grouped plans
  // register a ring's bases when the ring is added): sum_members holds POINT SLOTS until flatten_plan resolves them.
  VarTerm sum_term(uint16_t slot, const std::vector<uint16_t>& member_slots, ScalarSrc sc) {
    if (!grouped()) for (uint16_t m : member_slots) (void)base_index(m);
    SumBase sb;
    sb.first = (uint16_t)sum_members.size();
    sb.count = (uint16_t)member_slots.size();
    // without groups the sums' tables follow the ordinary bases' tables; with groups they re-use the table slots from 0 on
    sb.out_base = (uint16_t)(grouped() ? sum_bases.size() : base_slots.size() + sum_bases.size());
    sb.pad = 0;
    sum_members.insert(sum_members.end(), member_slots.begin(), member_slots.end());
    sum_bases.push_back(sb);
    return VarTerm{slot, sb.out_base, sc};
  }
  // table slots a chunk needs per ballot
  size_t n_tables() const {
    if (!grouped()) return base_slots.size() + sum_bases.size();
    size_t m = sum_bases.size();
    for (size_t g = 0; g < group_stage.size(); ++g) m = std::max(m, group_size((int)g));
    return m;
  }

  uint32_t ref(const std::string& s) {  // label / constant message in the blob
    auto it = blob_index.find(s);
    if (it != blob_index.end()) return it->second;
    const uint32_t r = blob_ref((uint32_t)blob.size(), (uint32_t)s.size());
    blob.insert(blob.end(), s.begin(), s.end());
    blob_index[s] = r;
    return r;
  }
  int pk_off = -1;
  int gen_pre_main = -1, gen_pre_ring = -1, gen_pre_logeq = -1;  // prefix indices used by the ballot generator
  int gen_vote_main = -1, gen_vote_ring = -1, gen_credit_main = -1, gen_credit_ring = -1, gen_pre_sumsq = -1;
  uint32_t pk_ref() {  // 32 bytes of the election key; filled in when the params object is created
    if (pk_off < 0) { pk_off = (int)blob.size(); blob.insert(blob.end(), 32, 0); }
    return blob_ref((uint32_t)pk_off, 32);
  }
  uint16_t const_point(uint64_t m) {
    auto it = const_index.find(m);
    if (it != const_index.end()) return it->second;
    const uint16_t i = (uint16_t)const_mults.size();
    const_mults.push_back(m);
    const_index[m] = i;
    return i;
  }
  Stage& stage(size_t s) { if (stages.size() <= s) stages.resize(s + 1); return stages[s]; }
  uint16_t new_pt() { return (uint16_t)n_pt_slots++; }
  uint16_t new_cmp() { return (uint16_t)n_cmp_slots++; }
  uint16_t new_chal() { return (uint16_t)n_chal_slots++; }
  uint16_t new_state() { return (uint16_t)n_state_slots++; }
  uint16_t new_flag() { return (uint16_t)n_flag_slots++; }
  uint32_t new_prefix() { return (uint32_t)n_prefixes++; }

  uint16_t wire_point(uint16_t item) { const uint16_t s = new_pt(); pt_items.push_back({item, s}); return s; }
  void wire_scalar(uint16_t item) { sc_items.push_back({item, 0}); }

  uint16_t derive(int level, const std::vector<DeriveTerm>& terms) {
    if ((int)derive_levels.size() <= level) derive_levels.resize(level + 1);
    DeriveClass dc;
    dc.term_first = (uint16_t)dterms.size();
    dc.term_count = (uint16_t)terms.size();
    dc.out_slot = new_pt();
    dc.pad = 0;
    dterms.insert(dterms.end(), terms.begin(), terms.end());
    derive_levels[level].push_back(dc);
    return dc.out_slot;
  }
  // out = encode(sum terms + [g]G + [k]K); returns the compressed slot
  uint16_t job(size_t st, const std::vector<VarTerm>& terms, ScalarSrc g, ScalarSrc k) {
    JobClass jc;
    jc.term_first = (uint16_t)vterms.size();
    jc.term_count = (uint16_t)terms.size();
    jc.g = g; jc.k = k;
    jc.out_slot = new_cmp();
    jc.enc_slot = 0xffff;
    jc.defer = 1; jc.pad = 0;
    vterms.insert(vterms.end(), terms.begin(), terms.end());
    stage(st).jobs.push_back(jc);
    stage(st).deferred.push_back(jc.out_slot);
    return jc.out_slot;
  }
  uint16_t encode_job(size_t st, uint16_t pt_slot) {
    JobClass jc{};
    jc.term_first = 0; jc.term_count = 0;
    jc.g = ScalarSrc{SRC_NONE, 0, 0, 0}; jc.k = ScalarSrc{SRC_NONE, 0, 0, 0};
    jc.out_slot = new_cmp();
    jc.enc_slot = pt_slot;
    jc.defer = 0; jc.pad = 0;
    stage(st).jobs.push_back(jc);
    return jc.out_slot;
  }
};

// [a](P_1 + .. + P_m) as m table-backed terms with the same multiplier: the equation kernels evaluate all table-backed terms of
// an equation on ONE doubling chain (k_eq_table<true>), so a term costs one addition per column (43 with 6 teeth), against 252 doublings + 71 additions for a
// ladder over the sum itself.  Used where the reference multiplies a ciphertext that is, component-wise, the sum of ring
// ciphertexts (mul.rs:207-247 on the vote / credit ciphertexts of a range proof).
inline void push_sum_terms(Plan& P, std::vector<VarTerm>& out, const std::vector<uint16_t>& slots, ScalarSrc sc) {
  for (uint16_t s : slots) out.push_back(P.bterm(s, sc));
}

inline ScalarSrc wire_src(uint16_t item, bool neg = false) { return ScalarSrc{SRC_WIRE, (uint8_t)neg, item, 0}; }
inline ScalarSrc chal_src(uint16_t slot, bool neg = false) { return ScalarSrc{SRC_CHAL, (uint8_t)neg, slot, 0}; }
inline ScalarSrc no_src() { return ScalarSrc{SRC_NONE, 0, 0, 0}; }

// one ring of a RingProof
struct RingIn {
  uint16_t ptR, ptB;             // point slots of the ring's ciphertext
  bool enc_from_wire;            // "enc" bytes: wire items (canonical) or encodings of derived points
  uint16_t enc_item;             // first wire item (R then B) when enc_from_wire
  int derive_level;              // level at which ptB is available (0 = wire)
  std::vector<uint64_t> admissible;  // x_j = [admissible[j]] G  (admissible[0] == 0)
  uint16_t resp_item;            // wire item of s_0
};

// RingProof::verify (ring.rs:302-374) appended to the plan.  `setup` are the ops that bring a fresh transcript
// to the state the reference passes in (Transcript::new(label) [+ range header]).  Returns the verdict flag.
inline uint16_t add_ring_proof(Plan& P, const std::vector<HashOp>& setup, const std::vector<RingIn>& rings,
                               uint16_t challenge_item, size_t first_stage = 0, int* out_pre_main = nullptr,
                               int* out_pre_ring = nullptr) {
  // hoisted, election-constant prefixes (ring.rs:290-293, :329)
  const uint32_t pre_main = P.new_prefix(), pre_ring = P.new_prefix();
  if (out_pre_main) *out_pre_main = (int)pre_main;
  if (out_pre_ring) *out_pre_ring = (int)pre_ring;
  {
    std::vector<HashOp> prog = setup;
    prog.push_back({OP_APPEND_BLOB, P.ref("dom-sep"), P.ref("multi_ring_enc"), 0});
    prog.push_back({OP_APPEND_BLOB, P.ref("K"), P.pk_ref(), 0});   // "\x01pk" is patched to the key bytes
    prog.push_back({OP_SAVE_PREFIX, 0, pre_main, 0});
    prog.push_back({OP_APPEND_BLOB, P.ref("dom-sep"), P.ref("ring_enc"), 0});
    prog.push_back({OP_SAVE_PREFIX, 0, pre_ring, 0});
    P.prefix_programs.push_back(prog);
  }
  size_t max_size = 0;
  for (auto& r : rings) max_size = std::max(max_size, r.admissible.size());
  std::vector<std::pair<uint16_t, uint16_t>> terminal(rings.size());
  // ring-group walk: the rings of group g run their equations in stages first_stage + g * max_size + j, after their tables are built
  const size_t base_stage = first_stage;
  const size_t n_groups = P.grouped() ? (rings.size() + P.ring_group - 1) / P.ring_group : 1;
  for (size_t ri = 0; ri < rings.size(); ++ri) {
    const RingIn& r = rings[ri];
    const size_t s = r.admissible.size();
    if (P.grouped()) {
      const size_t g = ri / P.ring_group;
      first_stage = base_stage + g * max_size;
      if (ri % P.ring_group == 0) P.begin_group((int)g, (int)first_stage);
    }
    uint16_t enc_r = 0, enc_b = 0;
    if (!r.enc_from_wire) {
      enc_r = P.encode_job(first_stage, r.ptR);
      enc_b = P.encode_job(first_stage, r.ptB);
    }
    const uint16_t state = s > 2 ? P.new_state() : 0;
    uint16_t chal = 0;
    for (size_t j = 0; j < s; ++j) {
      // R_G = [s]G - [e]R ;  R_K = [s]K - [e](B - x_j) = [s]K + [e m_j]G - [e]B  with x_j = [m_j]G   (ring.rs:338-350).
      // Folding x_j into the generator term keeps B itself as the only variable base of the ring's K side, so the
      // comb tables of R and B are shared by all equations of the ring.  x_0 = O, so the folded multiplier e m_j is only ever
      // needed for a DERIVED challenge: the transcript kernel stores it next to the challenge (OP_CHALLENGE, slot + 1).
      const ScalarSrc e_pos = j == 0 ? wire_src(challenge_item) : chal_src(chal);
      ScalarSrc e = e_pos;
      e.neg = 1;
      const ScalarSrc resp = wire_src((uint16_t)(r.resp_item + j));
      const uint16_t cg = P.job(first_stage + j, {P.bterm(r.ptR, e)}, resp, no_src());
      ScalarSrc fold = no_src();
      if (r.admissible[j] == 1) fold = e_pos;
      else if (r.admissible[j] > 1) fold = chal_src((uint16_t)(chal + 1));      // j >= 1 here: admissible[0] == 0
      const uint16_t ck = P.job(first_stage + j, {P.bterm(r.ptB, e)}, fold, resp);
      if (j + 1 < s) {   // ring.rs:354-360
        std::vector<HashOp> ops;
        if (j == 0) {
          ops.push_back({OP_LOAD_PREFIX, 0, pre_ring, 0});
          if (r.enc_from_wire) ops.push_back({OP_APPEND_WIRE, P.ref("enc"), r.enc_item, 2});
          else ops.push_back({OP_APPEND_CMP, P.ref("enc"), enc_r, enc_b});
          ops.push_back({OP_APPEND_U64, P.ref("i"), (uint32_t)ri, 0});
          if (s > 2) ops.push_back({OP_SAVE_STATE, 0, state, 0});
        } else {
          ops.push_back({OP_LOAD_STATE, 0, state, 0});
        }
        ops.push_back({OP_APPEND_U64, P.ref("j"), (uint32_t)j, 0});
        ops.push_back({OP_APPEND_CMP, P.ref("R_G"), cg, 0xffff});
        ops.push_back({OP_APPEND_CMP, P.ref("R_K"), ck, 0xffff});
        chal = P.new_chal();
        const uint64_t m_next = r.admissible[j + 1];
        if (m_next > 1) (void)P.new_chal();                       // slot chal + 1 receives m_next * challenge
        ops.push_back({OP_CHALLENGE, P.ref("c"), chal, (uint32_t)(m_next > 1 ? m_next : 0)});
        P.stage(first_stage + j).insts.push_back(ops);
      } else {
        terminal[ri] = {cg, ck};
      }
    }
  }
  // common challenge (ring.rs:364-373)
  const uint16_t flag = P.new_flag();
  std::vector<HashOp> fin;
  fin.push_back({OP_LOAD_PREFIX, 0, pre_main, 0});
  for (auto& t : terminal) {
    fin.push_back({OP_APPEND_CMP, P.ref("R_G"), t.first, 0xffff});
    fin.push_back({OP_APPEND_CMP, P.ref("R_K"), t.second, 0xffff});
  }
  fin.push_back({OP_CHALLENGE_CHECK, P.ref("c"), challenge_item, flag});
  P.stage(base_stage + (n_groups - 1) * max_size + max_size - 1).insts.push_back(fin);
  return flag;
}

// ---- EncryptedChoice (choice.rs:358-380) ----------------------------------------------------------------------------
inline size_t choice_ballot_size(int n, bool single) { return (size_t)n * 64 + 32 * (size_t)(1 + 2 * n) + (single ? 64 : 0); }

// ring_group: rings per group of the ring-group walk, 0 = off (a single-choice ballot of 1 or 2 options evaluates its sum proof over the
// rings' tables on shared chains and needs them all at once: never grouped).  Rings have two members here, so a group takes two stages.
// The walk trades throughput for memory (profiles/r04_ab_experiments.txt, block 2: workspace per ballot 32.3 -> 16.7 KB for 5 options,
// 84 -> 26 KB for 16, at -4 % / -0.5 % throughput - a group's launches are a third the size and nothing comes back from the caches), so
// it is on by default only where all tables at once would squeeze the chunks to a few thousand ballots (2 KiB x 2 n tables per ballot:
// 1 MiB from 256 options on); EG_RING_GROUP chooses it for any election.
inline int choice_group_default(int n, bool single) {
  (void)single;
  return n >= 256 ? 8 : 0;
}
inline Plan build_choice_plan(int n, bool single, int ring_group = -1) {
  Plan P;
  if (ring_group < 0) ring_group = choice_group_default(n, single);
  if ((single && n <= 2) || ring_group >= n) ring_group = 0;
  P.ring_group = ring_group;
  const size_t tail = ring_group ? (size_t)((n + ring_group - 1) / ring_group) * 2 : 0;   // stage of the sum proof: after the last group
  P.stride = choice_ballot_size(n, single);
  std::vector<uint16_t> R(n), B(n);
  for (int k = 0; k < n; ++k) { R[k] = P.wire_point((uint16_t)(2 * k)); B[k] = P.wire_point((uint16_t)(2 * k + 1)); }
  const uint16_t ring_items = (uint16_t)(2 * n);   // e0, then 2n responses
  for (int i = 0; i < 1 + 2 * n; ++i) P.wire_scalar((uint16_t)(ring_items + i));
  const uint16_t sum_items = (uint16_t)(ring_items + 1 + 2 * n);
  if (single) { P.wire_scalar(sum_items); P.wire_scalar((uint16_t)(sum_items + 1)); }
  for (int k = 0; k < n; ++k) { P.tally_slots.push_back(R[k]); P.tally_slots.push_back(B[k]); }

  uint16_t sum_flag = 0;
  if (single) {
    // sum of ciphertexts (choice.rs:363), powers = (sum.R, sum.B - G) (choice.rs:83-86)
    std::vector<DeriveTerm> tr, tb;
    for (int k = 0; k < n; ++k) { tr.push_back({R[k], 0, 0}); tb.push_back({B[k], 0, 0}); }
    tb.push_back({P.const_point(1), 1, 1});
    const uint16_t p0 = P.derive(0, tr), p1 = P.derive(0, tb);
    const uint32_t pre = P.new_prefix();
    P.gen_pre_logeq = (int)pre;
    P.prefix_programs.push_back({{OP_NEW, P.ref("choice_encryption_sum"), 0, 0},
                                 {OP_APPEND_BLOB, P.ref("dom-sep"), P.ref("log_eq"), 0},
                                 {OP_APPEND_BLOB, P.ref("K"), P.pk_ref(), 0},
                                 {OP_SAVE_PREFIX, 0, pre, 0}});
    // LogEqualityProof::verify (log_equality.rs:153-180)
    const ScalarSrc c = wire_src(sum_items, true), s = wire_src((uint16_t)(sum_items + 1));
    uint16_t xg, xk;
    if (n <= 2) {
      // [-c](sum R_k) and [-c](sum B_k - G) = sum [-c]B_k + [c]G over the comb tables the rings need anyway, all terms on one
      // doubling chain (6 teeth: 42 doublings + 43 n additions, against 252 + 71 for a ladder over the sum)
      std::vector<VarTerm> tg, tk;
      push_sum_terms(P, tg, R, c);
      push_sum_terms(P, tk, B, c);
      xg = P.job(0, tg, s, no_src());
      xk = P.job(0, tk, wire_src(sum_items), s);
    } else {
      // more options: the comb tables of sum R_k and sum B_k are summed up from the ring bases' tables (6 (n - 1) + 31 additions,
      // no doublings: ge_teeth_tables_sum), and the two equations become ordinary one-table equations.  With the ring-group walk the
      // sums are accumulated group by group and the two equations run after the last group (stage `tail`).
      if (!P.grouped()) for (int k = 0; k < n; ++k) { (void)P.base_index(R[k]); (void)P.base_index(B[k]); }   // ring bases first, in ring order
      P.sum_stage = (int)tail;
      xg = P.job(tail, {P.sum_term(p0, R, c)}, s, no_src());
      xk = P.job(tail, {P.sum_term(p1, B, c)}, wire_src(sum_items), s);
    }
    const uint16_t e0 = P.encode_job(tail, p0), e1 = P.encode_job(tail, p1);
    sum_flag = P.new_flag();
    P.stage(tail).insts.push_back({{OP_LOAD_PREFIX, 0, pre, 0},
                                {OP_APPEND_CMP, P.ref("[r]G"), e0, 0xffff},
                                {OP_APPEND_CMP, P.ref("[r]K"), e1, 0xffff},
                                {OP_APPEND_CMP, P.ref("[x]G"), xg, 0xffff},
                                {OP_APPEND_CMP, P.ref("[x]K"), xk, 0xffff},
                                {OP_CHALLENGE_CHECK, P.ref("c"), sum_items, sum_flag}});
    P.rules.push_back({sum_flag, 4 /* EG_ST_SUM_CHALLENGE */});
  }
  std::vector<RingIn> rings;
  for (int k = 0; k < n; ++k) {
    RingIn r;
    r.ptR = R[k]; r.ptB = B[k]; r.enc_from_wire = true; r.enc_item = (uint16_t)(2 * k); r.derive_level = 0;
    r.admissible = {0, 1};   // [O, G] (choice.rs:370)
    r.resp_item = (uint16_t)(ring_items + 1 + 2 * k);
    rings.push_back(r);
  }
  const uint16_t flag = add_ring_proof(P, {{OP_NEW, P.ref("encrypted_choice_ranges"), 0, 0}}, rings, ring_items, 0,
                                       &P.gen_pre_main, &P.gen_pre_ring);
  P.rules.push_back({flag, 6 /* EG_ST_RANGE_CHALLENGE */});
  return P;
}

// ---- RangeProof::verify (range.rs:547-577) on items [first_item ...): ct(2) partials(2(r-1)) e0 responses ------------------
struct RangeOut {
  uint16_t flag; uint16_t ctR, ctB; uint16_t n_items; int pre_main, pre_ring;
  // the ring ciphertexts (partials, then last = ct - sum(partials), range.rs:564-572): ct = sum of them, component-wise, and
  // every one of them is a ring base with a comb table
  std::vector<uint16_t> ringR, ringB;
};
inline RangeOut add_range_proof(Plan& P, const RangeDecomposition& d, const std::string& label, uint16_t first_item) {
  const int nr = (int)d.rings.size();
  RangeOut out;
  uint16_t item = first_item;
  out.ctR = P.wire_point(item); out.ctB = P.wire_point((uint16_t)(item + 1));
  item += 2;
  std::vector<uint16_t> pR(nr), pB(nr);
  for (int i = 0; i < nr - 1; ++i) { pR[i] = P.wire_point(item); pB[i] = P.wire_point((uint16_t)(item + 1)); item += 2; }
  const uint16_t chal_item = item++;
  P.wire_scalar(chal_item);
  const uint16_t resp0 = item;
  for (uint64_t i = 0; i < d.rings_size(); ++i) P.wire_scalar(item++);
  out.n_items = (uint16_t)(item - first_item);
  // last = ct - sum(partials)  (range.rs:564-572)
  bool last_wire = nr == 1;
  if (nr == 1) { pR[0] = out.ctR; pB[0] = out.ctB; }
  else {
    std::vector<DeriveTerm> tr{{out.ctR, 0, 0}}, tb{{out.ctB, 0, 0}};
    for (int i = 0; i < nr - 1; ++i) { tr.push_back({pR[i], 0, 1}); tb.push_back({pB[i], 0, 1}); }
    pR[nr - 1] = P.derive(0, tr);
    pB[nr - 1] = P.derive(0, tb);
  }
  std::vector<RingIn> rings;
  uint16_t resp = resp0;
  for (int i = 0; i < nr; ++i) {
    RingIn r;
    r.ptR = pR[i]; r.ptB = pB[i];
    const bool derived = (i == nr - 1) && !last_wire;
    r.enc_from_wire = !derived;
    r.enc_item = (i == nr - 1) ? first_item : (uint16_t)(first_item + 2 + 2 * i);
    r.derive_level = derived ? 1 : 0;
    for (uint64_t j = 0; j < d.rings[i].size; ++j) r.admissible.push_back(j * d.rings[i].step);  // range.rs:343-349
    r.resp_item = resp;
    resp = (uint16_t)(resp + d.rings[i].size);
    rings.push_back(r);
  }
  const std::vector<HashOp> setup = {{OP_NEW, P.ref(label), 0, 0},
                                     {OP_APPEND_BLOB, P.ref("dom-sep"), P.ref("encryption_range_proof"), 0},
                                     {OP_APPEND_BLOB, P.ref("range"), P.ref(d.to_string()), 0}};   // range.rs:561-562
  out.flag = add_ring_proof(P, setup, rings, chal_item, 0, &out.pre_main, &out.pre_ring);
  out.ringR = pR; out.ringB = pB;
  return out;
}

// ---- QuadraticVotingBallot::verify (quadratic_voting.rs:291-329) ---------------------------------------------------------
struct QvShape { RangeDecomposition vote_range, credit_range; size_t vote_size, credit_size, ballot_size; };
inline QvShape qv_shape(int n, uint64_t credits) {
  QvShape s;
  s.vote_range = optimal_range(isqrt(credits) + 1);     // quadratic_voting.rs:67-69
  s.credit_range = optimal_range(credits + 1);
  auto sz = [](const RangeDecomposition& d) { return 64 + 64 * (d.rings.size() - 1) + 32 * (1 + d.rings_size()); };
  s.vote_size = sz(s.vote_range);
  s.credit_size = sz(s.credit_range);
  s.ballot_size = (size_t)n * s.vote_size + s.credit_size + 32 * (size_t)(2 * n + 2);
  return s;
}

inline Plan build_qv_plan(int n, uint64_t credits) {
  Plan P;
  const QvShape sh = qv_shape(n, credits);
  P.stride = sh.ballot_size;
  uint16_t item = 0;
  std::vector<RangeOut> votes;
  for (int i = 0; i < n; ++i) {   // :297-307
    RangeOut r = add_range_proof(P, sh.vote_range, "quadratic_voting_variant", item);
    item = (uint16_t)(item + r.n_items);
    votes.push_back(r);
    if (i == 0) { P.gen_vote_main = r.pre_main; P.gen_vote_ring = r.pre_ring; }
    P.rules.push_back({r.flag, 8u /* EG_ST_QV_VARIANT_CHALLENGE */ | ((uint32_t)i << 8)});
    P.tally_slots.push_back(r.ctR);
    P.tally_slots.push_back(r.ctB);
  }
  RangeOut credit = add_range_proof(P, sh.credit_range, "quadratic_voting_credit_range", item);   // :309-317
  item = (uint16_t)(item + credit.n_items);
  P.gen_credit_main = credit.pre_main; P.gen_credit_ring = credit.pre_ring;
  P.rules.push_back({credit.flag, 10 /* EG_ST_QV_CREDIT_RANGE_CHALLENGE */});
  // SumOfSquaresProof::verify (mul.rs:190-260)
  const uint16_t c_item = item;
  for (int i = 0; i < 2 * n + 2; ++i) P.wire_scalar((uint16_t)(c_item + i));
  const uint16_t sz_item = (uint16_t)(c_item + 1 + 2 * n);
  const uint32_t pre = P.new_prefix();
  P.gen_pre_sumsq = (int)pre;
  P.prefix_programs.push_back({{OP_NEW, P.ref("quadratic_voting_credit_equiv"), 0, 0},
                               {OP_APPEND_BLOB, P.ref("dom-sep"), P.ref("sum_of_squares"), 0},
                               {OP_APPEND_BLOB, P.ref("K"), P.pk_ref(), 0},
                               {OP_SAVE_PREFIX, 0, pre, 0}});
  std::vector<HashOp> ops{{OP_LOAD_PREFIX, 0, pre, 0}};
  const ScalarSrc neg_c = wire_src(c_item, true);
  std::vector<VarTerm> trz, tz;
  // wire items of the vote ciphertexts
  uint16_t vitem = 0;
  for (int i = 0; i < n; ++i) {
    const ScalarSrc s_r = wire_src((uint16_t)(c_item + 1 + 2 * i)), s_x = wire_src((uint16_t)(c_item + 2 + 2 * i));
    std::vector<VarTerm> tr, tx;
    push_sum_terms(P, tr, votes[i].ringR, neg_c);
    push_sum_terms(P, tx, votes[i].ringB, neg_c);
    const uint16_t er = P.job(0, tr, s_r, no_src());                                            // mul.rs:213-217
    const uint16_t ex = P.job(0, tx, s_x, s_r);                                                 // mul.rs:219-226
    ops.push_back({OP_APPEND_WIRE, P.ref("R_x"), vitem, 1});
    ops.push_back({OP_APPEND_WIRE, P.ref("X"), (uint32_t)(vitem + 1), 1});
    ops.push_back({OP_APPEND_CMP, P.ref("[e_r]G"), er, 0xffff});
    ops.push_back({OP_APPEND_CMP, P.ref("[e_x]G + [e_r]K"), ex, 0xffff});
    push_sum_terms(P, trz, votes[i].ringR, s_x);
    push_sum_terms(P, tz, votes[i].ringB, s_x);
    vitem = (uint16_t)(vitem + votes[i].n_items);
  }
  push_sum_terms(P, trz, credit.ringR, neg_c);
  push_sum_terms(P, tz, credit.ringB, neg_c);
  const uint16_t erz = P.job(0, trz, wire_src(sz_item), no_src());   // mul.rs:232-240
  const uint16_t ez = P.job(0, tz, no_src(), wire_src(sz_item));     // mul.rs:241-247
  ops.push_back({OP_APPEND_WIRE, P.ref("R_z"), vitem, 1});
  ops.push_back({OP_APPEND_WIRE, P.ref("Z"), (uint32_t)(vitem + 1), 1});
  ops.push_back({OP_APPEND_CMP, P.ref("[e_x]R_x + [e_z]G"), erz, 0xffff});
  ops.push_back({OP_APPEND_CMP, P.ref("[e_x]X + [e_z]K"), ez, 0xffff});
  const uint16_t flag = P.new_flag();
  ops.push_back({OP_CHALLENGE_CHECK, P.ref("c"), c_item, flag});
  P.stage(0).insts.push_back(ops);
  P.rules.push_back({flag, 12 /* EG_ST_QV_CREDIT_EQUIV_CHALLENGE */});
  return P;
}

// ---- PublicKey::verify_zero / verify_bool / verify_range (keys/impls.rs:59-69,100-112,142-151) ------------------------
inline Plan build_zero_plan() {   // item = ct(64) || challenge || response
  Plan P;
  P.stride = 128;
  const uint16_t R = P.wire_point(0), B = P.wire_point(1);
  P.wire_scalar(2); P.wire_scalar(3);
  const uint32_t pre = P.new_prefix();
  P.prefix_programs.push_back({{OP_NEW, P.ref("zero_encryption"), 0, 0},
                               {OP_APPEND_BLOB, P.ref("dom-sep"), P.ref("log_eq"), 0},
                               {OP_APPEND_BLOB, P.ref("K"), P.pk_ref(), 0},
                               {OP_SAVE_PREFIX, 0, pre, 0}});
  const ScalarSrc c = wire_src(2, true), s = wire_src(3);
  const uint16_t xg = P.job(0, {P.term(R, c)}, s, no_src());
  const uint16_t xk = P.job(0, {P.term(B, c)}, no_src(), s);
  const uint16_t flag = P.new_flag();
  P.stage(0).insts.push_back({{OP_LOAD_PREFIX, 0, pre, 0},
                              {OP_APPEND_WIRE, P.ref("[r]G"), 0, 1},
                              {OP_APPEND_WIRE, P.ref("[r]K"), 1, 1},
                              {OP_APPEND_CMP, P.ref("[x]G"), xg, 0xffff},
                              {OP_APPEND_CMP, P.ref("[x]K"), xk, 0xffff},
                              {OP_CHALLENGE_CHECK, P.ref("c"), 2, flag}});
  P.rules.push_back({flag, 4 /* ChallengeMismatch of the log-equality proof */});
  return P;
}
inline Plan build_bool_plan() {   // item = ct(64) || e0 || s0 || s1
  Plan P;
  P.stride = 160;
  RingIn r;
  r.ptR = P.wire_point(0); r.ptB = P.wire_point(1);
  for (int i = 2; i < 5; ++i) P.wire_scalar((uint16_t)i);
  r.enc_from_wire = true; r.enc_item = 0; r.derive_level = 0; r.admissible = {0, 1}; r.resp_item = 3;
  const uint16_t flag = add_ring_proof(P, {{OP_NEW, P.ref("bool_encryption"), 0, 0}}, {r}, 2);
  P.rules.push_back({flag, 6});
  return P;
}
inline Plan build_range_plan(uint64_t upper_bound, size_t* item_size) {   // ct || partials || e0 || responses
  Plan P;
  const RangeDecomposition d = optimal_range(upper_bound);
  RangeOut r = add_range_proof(P, d, "ciphertext_range", 0);
  P.stride = (size_t)r.n_items * 32;
  if (item_size) *item_size = P.stride;
  P.rules.push_back({r.flag, 6});
  return P;
}

// ---- SumOfSquaresProof::verify (mul.rs:190-260) on its own, with the caller's transcript label --------------------------------
// item = n value ciphertexts (64 each) || sum-of-squares ciphertext (64) || challenge || 2n ciphertext responses || sum response.
// Every ciphertext element is multiplied twice (once per-ciphertext, once in the (n+2)-term equations), so all of them get comb
// tables and the two long equations run on one doubling chain each.
inline Plan build_sumsq_plan(int n, const std::string& label, size_t* item_size) {
  Plan P;
  P.stride = (size_t)64 * (n + 1) + 32 * (size_t)(2 * n + 2);
  if (item_size) *item_size = P.stride;
  std::vector<uint16_t> R(n + 1), X(n + 1);
  for (int i = 0; i <= n; ++i) { R[i] = P.wire_point((uint16_t)(2 * i)); X[i] = P.wire_point((uint16_t)(2 * i + 1)); }
  const uint16_t c_item = (uint16_t)(2 * (n + 1));
  for (int i = 0; i < 2 * n + 2; ++i) P.wire_scalar((uint16_t)(c_item + i));
  const uint16_t sz_item = (uint16_t)(c_item + 1 + 2 * n);
  const uint32_t pre = P.new_prefix();
  P.prefix_programs.push_back({{OP_NEW, P.ref(label), 0, 0},
                               {OP_APPEND_BLOB, P.ref("dom-sep"), P.ref("sum_of_squares"), 0},
                               {OP_APPEND_BLOB, P.ref("K"), P.pk_ref(), 0},
                               {OP_SAVE_PREFIX, 0, pre, 0}});
  std::vector<HashOp> ops{{OP_LOAD_PREFIX, 0, pre, 0}};
  const ScalarSrc neg_c = wire_src(c_item, true);
  std::vector<VarTerm> trz, tz;
  for (int i = 0; i < n; ++i) {
    const ScalarSrc s_r = wire_src((uint16_t)(c_item + 1 + 2 * i)), s_x = wire_src((uint16_t)(c_item + 2 + 2 * i));
    const uint16_t er = P.job(0, {P.bterm(R[i], neg_c)}, s_r, no_src());      // mul.rs:213-217
    const uint16_t ex = P.job(0, {P.bterm(X[i], neg_c)}, s_x, s_r);           // mul.rs:219-226
    ops.push_back({OP_APPEND_WIRE, P.ref("R_x"), (uint32_t)(2 * i), 1});
    ops.push_back({OP_APPEND_WIRE, P.ref("X"), (uint32_t)(2 * i + 1), 1});
    ops.push_back({OP_APPEND_CMP, P.ref("[e_r]G"), er, 0xffff});
    ops.push_back({OP_APPEND_CMP, P.ref("[e_x]G + [e_r]K"), ex, 0xffff});
    trz.push_back(P.bterm(R[i], s_x));
    tz.push_back(P.bterm(X[i], s_x));
  }
  trz.push_back(P.bterm(R[n], neg_c));
  tz.push_back(P.bterm(X[n], neg_c));
  const uint16_t erz = P.job(0, trz, wire_src(sz_item), no_src());   // mul.rs:232-240
  const uint16_t ez = P.job(0, tz, no_src(), wire_src(sz_item));     // mul.rs:241-247
  ops.push_back({OP_APPEND_WIRE, P.ref("R_z"), (uint32_t)(2 * n), 1});
  ops.push_back({OP_APPEND_WIRE, P.ref("Z"), (uint32_t)(2 * n + 1), 1});
  ops.push_back({OP_APPEND_CMP, P.ref("[e_x]R_x + [e_z]G"), erz, 0xffff});
  ops.push_back({OP_APPEND_CMP, P.ref("[e_x]X + [e_z]K"), ez, 0xffff});
  const uint16_t flag = P.new_flag();
  ops.push_back({OP_CHALLENGE_CHECK, P.ref("c"), c_item, flag});
  P.stage(0).insts.push_back(ops);
  P.rules.push_back({flag, 12 /* EG_ST_QV_CREDIT_EQUIV_CHALLENGE: ChallengeMismatch of the sum-of-squares proof */});
  return P;
}

// ---- PublicKeySet::verify_share (sharing/key_set.rs:209-228): item = R(32) || dh(32) || challenge || response -------------
// LogEqualityProof with log_base = the ciphertext's random element R (per item) and powers = (participant key, dh):
//   X_G = [s]G - [c]KS   (KS = participant key: the engine's fixed base "K")      X_K = [s]R - [c]dh
inline Plan build_share_plan(uint64_t shares, uint64_t threshold, const uint8_t shared_key[32], uint64_t index) {
  Plan P;
  P.stride = 128;
  const uint16_t R = P.wire_point(0), dh = P.wire_point(1);
  P.wire_scalar(2); P.wire_scalar(3);
  const uint32_t pre = P.new_prefix();
  const std::string shared(reinterpret_cast<const char*>(shared_key), 32);
  P.prefix_programs.push_back({{OP_NEW, P.ref("elgamal_decryption_share"), 0, 0},
                               {OP_APPEND_U64, P.ref("n"), (uint32_t)shares, 0},          // key_set.rs:167-171
                               {OP_APPEND_U64, P.ref("t"), (uint32_t)threshold, 0},
                               {OP_APPEND_BLOB, P.ref("K"), P.ref(shared), 0},
                               {OP_APPEND_U64, P.ref("i"), (uint32_t)index, 0},
                               {OP_APPEND_BLOB, P.ref("dom-sep"), P.ref("log_eq"), 0},
                               {OP_SAVE_PREFIX, 0, pre, 0}});
  const ScalarSrc c = wire_src(2, true), s = wire_src(3);
  const uint16_t xg = P.job(0, {}, s, c);                               // [s]G + [-c]KS
  const uint16_t xk = P.job(0, {P.term(dh, c), P.term(R, s)}, no_src(), no_src());
  const uint16_t flag = P.new_flag();
  P.stage(0).insts.push_back({{OP_LOAD_PREFIX, 0, pre, 0},
                              {OP_APPEND_WIRE, P.ref("K"), 0, 1},        // log_base bytes = R as sent
                              {OP_APPEND_BLOB, P.ref("[r]G"), P.pk_ref(), 0},   // participant key bytes
                              {OP_APPEND_WIRE, P.ref("[r]K"), 1, 1},
                              {OP_APPEND_CMP, P.ref("[x]G"), xg, 0xffff},
                              {OP_APPEND_CMP, P.ref("[x]K"), xk, 0xffff},
                              {OP_CHALLENGE_CHECK, P.ref("c"), 2, flag}});
  P.rules.push_back({flag, 4});
  return P;
}

// ---- flattening of a plan into the arrays the kernels index (pure host logic, sanitizer-tested in tests/hostcheck) ------------
// Comb shape of the per-ballot tables (ge25519.cuh: Teeth<T>).  A table costs ~204 doublings + 21 additions with 5 teeth (16 entries,
// 51 columns) and 215 + 37 with 6 (32 entries, 43 columns); a product 50 + 51 against 42 + 43.  The rings of two of a choice ballot use a
// table twice: 5 teeth measured +2.4 %; the rings of 3 .. 7 of the range proofs use it 3 .. 7 times: 6 teeth, +5 % (A/B block 9).
inline int plan_teeth(const Plan& P) {
  if (P.base_slots.empty()) return 6;
  return P.table_uses <= 2 * P.base_slots.size() + P.base_slots.size() / 2 ? 5 : 6;          // <= 2.5 products per table
}

// the equations of a stage, sorted by kernel family (kernels.cuh): FAM_TABLE1 one table-backed base, FAM_TABLEN several
// table-backed bases on shared doubling chains, FAM_DIRECT1 one base without a table, FAM_GENERIC everything else, FAM_ENCODE plain
// encodings of point slots
enum { FAM_TABLE1 = 0, FAM_TABLEN = 1, FAM_DIRECT1 = 2, FAM_GENERIC = 3, FAM_ENCODE = 4, N_FAM = 5 };
constexpr int EG_MULTI_GROUP = 8;           // terms per shared doubling chain: 8 sign vectors = 72 KiB of LDS per block, two blocks per CU
inline int job_family(const JobClass& j, const std::vector<VarTerm>& vterms) {
  if (!j.defer) return FAM_ENCODE;
  if (j.term_count == 0) return FAM_GENERIC;
  for (unsigned t = 0; t < j.term_count; ++t)
    if (vterms[j.term_first + t].base == 0xffff) return j.term_count == 1 ? FAM_DIRECT1 : FAM_GENERIC;
  return j.term_count == 1 ? FAM_TABLE1 : FAM_TABLEN;
}
struct StageDev {
  int fam_first[N_FAM], fam_count[N_FAM], max_terms, inst_first, inst_count, defer_first, defer_count;
  // tables at the start of the stage (see Stage): build_slots[build_first ..) -> table slots 0 ..; sums_direct: k_sum_tables over
  // F.sums / F.sum_members; acc_count > 0: k_sum_accumulate over F.acc_sums[acc_first ..); sum_finish: k_sum_finish over F.sums
  int build_first, build_count, sums_direct, acc_first, acc_count, sum_finish;
};
struct LevelDev { int first, count; };
struct FlatPlan {
  std::vector<JobClass> jobs;
  std::vector<HashInst> insts;
  std::vector<HashOp> ops;
  std::vector<uint16_t> defer_slots;
  std::vector<DeriveClass> dclasses;
  std::vector<StageDev> stages;
  std::vector<LevelDev> levels;
  std::vector<uint16_t> build_slots;     // point slots, grouped by the stage that builds their tables
  std::vector<SumBase> sums;             // the sums of bases: members = sum_members[first ..) as TABLE SLOTS (ungrouped plans); out_base = table slot
  std::vector<uint16_t> sum_members;
  std::vector<SumBase> acc_sums;         // ring-group walk, per (group, sum) with members in the group: members as table slots of the group,
  std::vector<uint16_t> acc_members;     //   out_base = index of the sum (its accumulator), pad = 1 for the sum's first contribution
  int max_defer = 0, prefix_inst_first = 0, prefix_inst_count = 0;
};
inline FlatPlan flatten_plan(const Plan& P) {
  FlatPlan F;
  const size_t n_stages = std::max(P.stages.size(), (size_t)(P.sum_bases.empty() ? 0 : P.sum_stage + 1));
  std::vector<StageDev> tab(n_stages);
  for (auto& t : tab) t.build_first = t.build_count = t.sums_direct = t.acc_first = t.acc_count = t.sum_finish = 0;
  auto base_of = [&](uint16_t slot) -> int { auto it = P.base_index_of.find(slot); return it == P.base_index_of.end() ? -1 : (int)it->second; };
  for (size_t g = 0; g < P.group_stage.size(); ++g) {           // tables of group g: built at the start of its first stage, in slot order
    std::vector<uint16_t> slots(P.group_size((int)g));
    for (size_t i = 0; i < P.base_slots.size(); ++i)
      if (P.base_group[i] == g) slots[P.base_local[i]] = P.base_slots[i];
    if (slots.empty() || (size_t)P.group_stage[g] >= n_stages) continue;
    StageDev& t = tab[P.group_stage[g]];
    t.build_first = (int)F.build_slots.size(); t.build_count = (int)slots.size();
    F.build_slots.insert(F.build_slots.end(), slots.begin(), slots.end());
  }
  for (const SumBase& sb : P.sum_bases) {                        // members: point slots -> table slots
    SumBase r = sb;
    r.first = (uint16_t)F.sum_members.size();
    for (unsigned t = 0; t < sb.count; ++t) {
      const int b = base_of(P.sum_members[sb.first + t]);
      F.sum_members.push_back(b < 0 ? (uint16_t)0xffff : P.base_local[b]);
    }
    F.sums.push_back(r);
  }
  if (!P.sum_bases.empty()) {
    if (!P.grouped()) tab[0].sums_direct = 1;
    else {
      std::vector<char> seen(P.sum_bases.size(), 0);
      for (size_t g = 0; g < P.group_stage.size(); ++g) {
        if ((size_t)P.group_stage[g] >= n_stages) continue;
        StageDev& t = tab[P.group_stage[g]];
        t.acc_first = (int)F.acc_sums.size();
        for (size_t k = 0; k < P.sum_bases.size(); ++k) {
          SumBase r;
          r.first = (uint16_t)F.acc_members.size(); r.count = 0; r.out_base = (uint16_t)k; r.pad = 0;
          for (unsigned m = 0; m < P.sum_bases[k].count; ++m) {
            const int b = base_of(P.sum_members[P.sum_bases[k].first + m]);
            if (b >= 0 && P.base_group[b] == g) { F.acc_members.push_back(P.base_local[b]); ++r.count; }
          }
          if (!r.count) continue;
          r.pad = seen[k] ? 0 : 1;
          seen[k] = 1;
          F.acc_sums.push_back(r);
        }
        t.acc_count = (int)F.acc_sums.size() - t.acc_first;
      }
      tab[P.sum_stage].sum_finish = 1;
    }
  }
  for (size_t si = 0; si < n_stages; ++si) {
    static const Stage empty_stage;
    const Stage& st = si < P.stages.size() ? P.stages[si] : empty_stage;
    StageDev sd = tab[si];
    sd.defer_first = (int)F.defer_slots.size(); sd.defer_count = (int)st.deferred.size();
    F.defer_slots.insert(F.defer_slots.end(), st.deferred.begin(), st.deferred.end());
    F.max_defer = std::max(F.max_defer, std::min(sd.defer_count, 32));
    sd.max_terms = 0;
    for (int f = 0; f < N_FAM; ++f) {
      sd.fam_first[f] = (int)F.jobs.size();
      for (auto& j : st.jobs)
        if (job_family(j, P.vterms) == f) {
          F.jobs.push_back(j);
          if (f == FAM_TABLEN) sd.max_terms = std::max<int>(sd.max_terms, j.term_count);
        }
      sd.fam_count[f] = (int)F.jobs.size() - sd.fam_first[f];
    }
    sd.inst_first = (int)F.insts.size(); sd.inst_count = (int)st.insts.size();
    for (auto& prog : st.insts) {
      F.insts.push_back({(uint32_t)F.ops.size(), (uint32_t)prog.size()});
      F.ops.insert(F.ops.end(), prog.begin(), prog.end());
    }
    F.stages.push_back(sd);
  }
  F.prefix_inst_first = (int)F.insts.size();
  F.prefix_inst_count = (int)P.prefix_programs.size();
  for (auto& prog : P.prefix_programs) {
    F.insts.push_back({(uint32_t)F.ops.size(), (uint32_t)prog.size()});
    F.ops.insert(F.ops.end(), prog.begin(), prog.end());
  }
  for (auto& lvl : P.derive_levels) {
    F.levels.push_back({(int)F.dclasses.size(), (int)lvl.size()});
    F.dclasses.insert(F.dclasses.end(), lvl.begin(), lvl.end());
  }
  return F;
}

// Every index the kernels will dereference, checked against the sizes the engine allocates (a fault on the device can take
// the whole node down, so an inconsistent plan is refused on the host).  Returns "" when consistent.
inline std::string check_flat_plan(const Plan& P, const FlatPlan& F) {
  const size_t items = P.stride / 32;
  auto scalar_ok = [&](const ScalarSrc& s) {
    if (s.kind == SRC_NONE) return true;
    if (s.kind == SRC_WIRE) return (size_t)s.idx < items;
    return s.kind == SRC_CHAL && (int)s.idx < P.n_chal_slots;
  };
  if (P.stride % 32) return "stride is not a multiple of 32";
  for (auto& w : P.pt_items) if (w.item >= items || (int)w.slot >= P.n_pt_slots) return "wire point out of range";
  for (auto& w : P.sc_items) if (w.item >= items) return "wire scalar out of range";
  for (auto& d : F.dclasses) {
    if ((size_t)d.term_first + d.term_count > P.dterms.size() || (int)d.out_slot >= P.n_pt_slots) return "derive class out of range";
    for (unsigned t = 0; t < d.term_count; ++t) {
      const DeriveTerm& dt = P.dterms[d.term_first + t];
      if (dt.is_const ? dt.slot >= P.const_mults.size() : (int)dt.slot >= P.n_pt_slots) return "derive term out of range";
    }
  }
  for (uint16_t b : P.base_slots) if ((int)b >= P.n_pt_slots) return "base slot out of range";
  if (P.late_base) return "ordinary base registered after a sum base";
  if (P.base_group.size() != P.base_slots.size() || P.base_local.size() != P.base_slots.size()) return "base registry out of step";
  const size_t n_tab = P.n_tables();
  if (F.sums.size() != P.sum_bases.size()) return "sums lost in flattening";
  for (size_t i = 0; i < F.sums.size(); ++i) {
    const SumBase& sb = F.sums[i];
    if (sb.out_base != (P.grouped() ? i : P.base_slots.size() + i) || sb.out_base >= n_tab) return "sum base index out of order";
    if (sb.count == 0 || (size_t)sb.first + sb.count > F.sum_members.size()) return "sum base members out of range";
    for (unsigned t = 0; t < sb.count; ++t) {
      const uint16_t m = F.sum_members[sb.first + t];
      if (m == 0xffff) return "sum base member is not an ordinary base";
      if (!P.grouped() && m >= P.base_slots.size()) return "sum base member out of range";
    }
  }
  {
    // every ordinary base's table is built exactly once, into its own slot, by the stage its group starts with; accumulators cover
    // every member of every sum exactly once, the first contribution of a sum carries the `first` mark, and the sums are finished
    // at a stage after every group's tables
    size_t built = 0, acc_members = 0;
    std::vector<char> first_seen(F.sums.size(), 0);
    int last_acc_stage = -1, finish_stage = -1;
    for (size_t si = 0; si < F.stages.size(); ++si) {
      const StageDev& sd = F.stages[si];
      if (sd.build_count < 0 || (size_t)sd.build_first + sd.build_count > F.build_slots.size() || (size_t)sd.build_count > n_tab) return "table build range out of range";
      for (int k = 0; k < sd.build_count; ++k) if ((int)F.build_slots[sd.build_first + k] >= P.n_pt_slots) return "table build slot out of range";
      built += sd.build_count;
      if (sd.sums_direct && (P.grouped() || si != 0)) return "direct sum tables in a grouped plan";
      if (sd.acc_count < 0 || (size_t)sd.acc_first + sd.acc_count > F.acc_sums.size()) return "accumulate range out of range";
      for (int k = 0; k < sd.acc_count; ++k) {
        const SumBase& a = F.acc_sums[sd.acc_first + k];
        if (a.out_base >= F.sums.size() || a.count == 0 || (size_t)a.first + a.count > F.acc_members.size()) return "accumulate record out of range";
        for (unsigned m = 0; m < a.count; ++m) if ((int)F.acc_members[a.first + m] >= sd.build_count) return "accumulate member outside the group's tables";
        if ((a.pad != 0) != (first_seen[a.out_base] == 0)) return "accumulate first-mark out of order";
        first_seen[a.out_base] = 1;
        acc_members += a.count;
        last_acc_stage = (int)si;
      }
      if (sd.sum_finish) { if (finish_stage >= 0) return "sums finished twice"; finish_stage = (int)si; }
      if (sd.sum_finish && sd.build_count) return "sums finished in a stage that builds tables (they share the table slots)";
    }
    if (built != P.base_slots.size()) return "not every base table is built exactly once";
    if (P.grouped() && !F.sums.empty()) {
      size_t want = 0;
      for (auto& sb : F.sums) want += sb.count;
      if (acc_members != want || finish_stage <= last_acc_stage) return "sum accumulation incomplete or finished too early";
      for (char c : first_seen) if (!c) return "a sum has no members";
    } else if (finish_stage >= 0 || last_acc_stage >= 0) return "accumulators in an ungrouped plan";
  }
  size_t counted = 0, live = 0;      // live = table slots that hold a table while the stage's equations run
  // whose tables those are: ALL (ungrouped plan), the tables of ring group `resident` (>= 0), or the SUMS' tables (after sum_finish).  A
  // table-backed term must name a table that is resident when its equation runs AND the slot that its base's table was built into: a
  // term over a base of another group would pass the range check and silently read a different ring's table.
  enum { RES_ALL = -1, RES_SUMS = -2, RES_NONE = -3 };
  int resident = P.grouped() ? RES_NONE : RES_ALL;
  for (size_t si = 0; si < F.stages.size(); ++si) {
    const StageDev& sd = F.stages[si];
    if (sd.build_count) live = (size_t)sd.build_count + (sd.sums_direct ? F.sums.size() : 0);
    if (sd.sum_finish) live = F.sums.size();
    if (P.grouped()) {
      if (sd.build_count) {
        int g_here = -1;
        for (size_t g = 0; g < P.group_stage.size(); ++g)
          if (P.group_stage[g] == (int)si && P.group_size((int)g)) { if (g_here >= 0) return "two groups build their tables in one stage"; g_here = (int)g; }
        if (g_here < 0 || P.group_size(g_here) != (size_t)sd.build_count) return "a stage builds tables that are no group's";
        resident = g_here;
      }
      if (sd.sum_finish) resident = RES_SUMS;
    }
    for (int f = 0; f < N_FAM; ++f) {
      if (sd.fam_first[f] < 0 || (size_t)sd.fam_first[f] + sd.fam_count[f] > F.jobs.size()) return "family range out of range";
      counted += sd.fam_count[f];
      for (int k = 0; k < sd.fam_count[f]; ++k) {
        const JobClass& j = F.jobs[sd.fam_first[f] + k];
        if (job_family(j, P.vterms) != f) return "job in the wrong family";
        if ((size_t)j.term_first + j.term_count > P.vterms.size()) return "job terms out of range";
        if ((int)j.out_slot >= P.n_cmp_slots) return "job output out of range";
        if (!j.defer && (int)j.enc_slot >= P.n_pt_slots) return "encode job source out of range";
        if (!scalar_ok(j.g) || !scalar_ok(j.k)) return "job scalar out of range";
        for (unsigned t = 0; t < j.term_count; ++t) {
          const VarTerm& v = P.vterms[j.term_first + t];
          if (!scalar_ok(v.s) || v.s.kind == SRC_NONE) return "term scalar out of range";
          if (v.base == 0xffff ? (int)v.slot >= P.n_pt_slots : (v.base >= n_tab || v.base >= live)) return "term base out of range";
          if (v.base != 0xffff) {
            auto it = P.base_index_of.find(v.slot);
            if (it != P.base_index_of.end()) {            // an ordinary base: its own slot, in its own group's residency
              const uint16_t i = it->second;
              if (P.base_local[i] != v.base) return "term names another base's table slot";
              if (P.grouped() && resident != (int)P.base_group[i]) return "term over a base whose group's tables are not resident at this stage";
            } else {                                        // a sum of bases: the sums' tables
              const size_t first_sum = P.grouped() ? 0 : P.base_slots.size();
              if (v.base < first_sum || v.base - first_sum >= F.sums.size()) return "sum term names a table slot that is no sum's";
              if (P.grouped() && resident != RES_SUMS) return "sum term before the sums' tables are finished";
            }
          }
        }
      }
    }
    if ((size_t)sd.defer_first + sd.defer_count > F.defer_slots.size()) return "deferred range out of range";
    if ((size_t)sd.inst_first + sd.inst_count > F.insts.size()) return "program range out of range";
  }
  if (counted != F.jobs.size()) return "jobs lost in the family sort";
  for (uint16_t d : F.defer_slots) if ((int)d >= P.n_cmp_slots) return "deferred slot out of range";
  for (auto& in : F.insts) {
    if ((size_t)in.op_first + in.op_count > F.ops.size()) return "program ops out of range";
    for (uint32_t o = 0; o < in.op_count; ++o) {
      const HashOp& op = F.ops[in.op_first + o];
      auto blob_ok = [&](uint32_t ref) { return (size_t)(ref >> 12) + (ref & 0xfffu) <= P.blob.size(); };
      switch (op.op) {
        case OP_NEW: case OP_APPEND_U64: if (!blob_ok(op.a)) return "label out of range"; break;
        case OP_APPEND_BLOB: if (!blob_ok(op.a) || !blob_ok(op.b)) return "blob out of range"; break;
        case OP_APPEND_WIRE: if (!blob_ok(op.a) || (size_t)op.b + op.c > items) return "wire append out of range"; break;
        case OP_APPEND_CMP:
          if (!blob_ok(op.a) || (int)op.b >= P.n_cmp_slots || (op.c != 0xffffu && (int)op.c >= P.n_cmp_slots)) return "cmp append out of range";
          break;
        case OP_CHALLENGE:
          if (!blob_ok(op.a) || (int)op.b + (op.c > 1 ? 1 : 0) >= P.n_chal_slots) return "challenge slot out of range";
          break;
        case OP_CHALLENGE_CHECK: if (!blob_ok(op.a) || op.b >= items || (int)op.c >= P.n_flag_slots) return "challenge check out of range"; break;
        case OP_LOAD_PREFIX: case OP_SAVE_PREFIX: if ((int)op.b >= P.n_prefixes) return "prefix out of range"; break;
        case OP_LOAD_STATE: case OP_SAVE_STATE: if ((int)op.b >= P.n_state_slots) return "state slot out of range"; break;
        default: return "unknown transcript op";
      }
    }
  }
  for (auto& r : P.rules) if ((int)r.flag_slot >= P.n_flag_slots) return "rule flag out of range";
  for (uint32_t t : P.tally_slots) if ((int)t >= P.n_pt_slots) return "tally slot out of range";
  return "";
}

}  // namespace eghost

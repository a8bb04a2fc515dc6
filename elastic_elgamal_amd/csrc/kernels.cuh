// kernels.cuh -- the gfx950 kernels of the batch verifier.  Hand-written HIP for CDNA4; no MFMA (every
// product has two per-lane operands: this is modular-integer VALU work), wave64 throughout.
//
// Data layout in HBM (all per chunk of `cap` ballots, SoA so that lane == ballot is always coalesced):
//   pts   [slot][9][cap]  uint4   extended points, 36 limbs (X,Y,Z,T x 9 x 28.3 bit)    144 B / point
//   cmp   [slot][2][cap]  uint4   compressed outputs of the group equations               32 B
//   chal  [slot][2][cap]  uint4   derived challenges e_j                                   32 B
//   states[slot][52][cap] u32     saved ring transcripts (only when a ring has > 2 equations)
//   flags [slot][cap]     u32     proof verdict flags;  bad_item[cap] first malformed wire item
// The packed ballots themselves stay in wire (AoS) order and are read exactly where the reference reads
// them: decode, canonical checks, transcript appends and scalar operands.
//
// Kernels and what bounds them (DESIGN.md has the numbers):
//   k_decode_points / k_check_scalars / k_derive_points   VALU (1 inverse square root per point)
//   k_base_tables / k_eq_table / k_eq_generic   VALU-bound integer multiply-add: >90 % of the time
//   k_hash         LDS + VALU (Keccak-f[1600])
//   k_status, k_tally_*   trivial
#pragma once
#include <hip/hip_runtime.h>
#include "device_io.cuh"

// Waves per SIMD the register allocation of the two dominant kernels is held to (A/B knobs).  With 9 limbs the equation kernel fits
// 168 registers without scratch: three waves measured +0.6 % single-choice / +0.5..0.9 % quadratic voting in one call
// (profiles/r03_ab_experiments.txt); the table builder gains nothing from a third wave.
#ifndef EG_EQ_WAVES
#define EG_EQ_WAVES 3
#endif
#ifndef EG_TAB_WAVES
#define EG_TAB_WAVES 2
#endif

namespace eg {

// ---- scalar operand fetch ---------------------------------------------------------------------------------------
__device__ __forceinline__ void load_scalar(u32 s[8], const EngineBufs& B, u32 b, egplan::ScalarSrc src, bool halve = false) {
  if (src.kind == egplan::SRC_WIRE) load_wire_item(s, B, b, src.idx);
  else load32(s, B.chal, B.cap, src.idx, b);
  if (src.neg) { u32 t[8]; sc_neg(t, s);
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = t[i]; }
  if (halve) { u32 t[8]; sc_halve(t, s);
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = t[i]; }
}

// ---- k_decode_points: deserialize_element for every wire point (ristretto.rs:93-95) --------------------------------
// (two waves per SIMD: v_mad_u64_u32 reaches its full issue rate only with a second wave to alternate with, profiles/r01_ubench_valu_rates.txt)
__global__ void __launch_bounds__(NT, 2) k_decode_points(EngineBufs B, const egplan::WireItem* items, int n_items) {
  const size_t total = (size_t)n_items * B.n;
  for (size_t j = (size_t)blockIdx.x * NT + threadIdx.x; j < total; j += (size_t)gridDim.x * NT) {
    const u32 k = (u32)(j / B.n), b = (u32)(j % B.n);
    const egplan::WireItem it = items[k];
    u32 w[8];
    load_wire_item(w, B, b, it.item);
    ge p;
    const bool ok = ristretto_decode(p, w);
    store_pt(B.pts, B.cap, it.slot, b, p);
    if (!ok) atomicMin(&B.bad_item[b], (u32)it.item * 4u + 2u);
  }
}

// ---- k_check_scalars: deserialize_scalar canonical check (ristretto.rs:59-62) -------------------------------------------
__global__ void __launch_bounds__(NT) k_check_scalars(EngineBufs B, const egplan::WireItem* items, int n_items) {
  const size_t total = (size_t)n_items * B.n;
  for (size_t j = (size_t)blockIdx.x * NT + threadIdx.x; j < total; j += (size_t)gridDim.x * NT) {
    const u32 k = (u32)(j / B.n), b = (u32)(j % B.n);
    const egplan::WireItem it = items[k];
    u32 w[8];
    load_wire_item(w, B, b, it.item);
    if (!sc_is_canonical(w)) atomicMin(&B.bad_item[b], (u32)it.item * 4u + 1u);
  }
}

// ---- k_derive_points: sums / differences of points (choice.rs:363, ring.rs:338, range.rs:564-572) ----------------------------
__global__ void __launch_bounds__(NT) k_derive_points(EngineBufs B, const egplan::DeriveClass* classes,
                                                      const egplan::DeriveTerm* terms, int class_first, int n_classes) {
  const size_t total = (size_t)n_classes * B.n;
  for (size_t j = (size_t)blockIdx.x * NT + threadIdx.x; j < total; j += (size_t)gridDim.x * NT) {
    const u32 c = class_first + (u32)(j / B.n), b = (u32)(j % B.n);
    const egplan::DeriveClass dc = classes[c];
    ge acc;
    ge_identity(acc);
#pragma unroll 1
    for (u32 t = 0; t < dc.term_count; ++t) {
      const egplan::DeriveTerm dt = terms[dc.term_first + t];
      ge p;
      if (dt.is_const) load_const_pt(p, B.cpts, dt.slot);
      else load_pt(p, B.pts, B.cap, dt.slot, b);
      ge_cached pc;
      ge_to_cached(pc, p);
      ge_cached_cneg(pc, dt.neg != 0);
      ge_p1p1 r;
      ge_add(r, acc, pc);
      ge_add_to_p3(acc, r);
    }
    store_pt(B.pts, B.cap, dc.out_slot, b, acc);
  }
}

// ---- k_base_tables: comb tables of every ring base (once per base and ballot; shared by all equations of the ring) -----------
template <int T>
__global__ void __launch_bounds__(NT, EG_TAB_WAVES) k_base_tables(EngineBufs B, const unsigned short* base_slots, int n_bases) {
  const size_t total = (size_t)n_bases * B.n;
  WsRows tmp;
  tmp.init(B.ws);
  for (size_t j = (size_t)blockIdx.x * NT + threadIdx.x; j < total; j += (size_t)gridDim.x * NT) {
    const u32 k = (u32)(j / B.n), b = (u32)(j % B.n);
    ge p;
    load_pt(p, B.pts, B.cap, base_slots[k], b);
    BaseTable bt{B.btab + ((size_t)k * B.cap + b) * btab_quads<T>()};
    ge_teeth_tables_build<T>(bt, tmp, p);
  }
}

// ---- k_sum_tables: comb table of a base that is the sum of ring bases, from their tables (ge_teeth_tables_sum; no doublings) ----
#ifndef EG_SUM_WAVES
#define EG_SUM_WAVES 2
#endif
template <int T>
__global__ void __launch_bounds__(NT, EG_SUM_WAVES) k_sum_tables(EngineBufs B, const egplan::SumBase* sums, const unsigned short* members, int n_sums) {
  const size_t total = (size_t)n_sums * B.n;
  WsRows tmp;
  tmp.init(B.ws);
  for (size_t j = (size_t)blockIdx.x * NT + threadIdx.x; j < total; j += (size_t)gridDim.x * NT) {
    const u32 k = (u32)(j / B.n), b = (u32)(j % B.n);
    const egplan::SumBase sb = sums[k];
    BaseTable out{B.btab + ((size_t)sb.out_base * B.cap + b) * btab_quads<T>()};
    ge_teeth_tables_sum<T>(out, tmp, (int)sb.count, [&](int t, int g, ge_cached& e) {
      const BaseTable bt{B.btab + ((size_t)members[sb.first + t] * B.cap + b) * btab_quads<T>()};
      bt.load(e, g);
    });
  }
}

// ---- ring-group walk: the sums' tables when only one group of ring tables exists at a time (ge_teeth_sum_accumulate) ---------------------
// k_sum_accumulate: one lane = one accumulator entry (sum, first-flip step t) of one ballot: adds the entries of this group's members.
// recs: per (group, sum) {first, count: members as table slots of the group; out_base: index of the sum; pad: 1 = first contribution}.
template <int T>
__global__ void __launch_bounds__(NT, EG_SUM_WAVES) k_sum_accumulate(EngineBufs B, const egplan::SumBase* recs, const unsigned short* members, int n_recs) {
  const size_t total = (size_t)n_recs * T * B.n;
  for (size_t j = (size_t)blockIdx.x * NT + threadIdx.x; j < total; j += (size_t)gridDim.x * NT) {
    const u32 q = (u32)(j / B.n), b = (u32)(j % B.n);
    const egplan::SumBase rec = recs[q / T];
    BaseTable acc{B.sacc + ((size_t)rec.out_base * B.cap + b) * (T * BTAB_ENTRY_QUADS)};
    ge_teeth_sum_accumulate<T>(acc, (int)(q % T), rec.pad != 0, (int)rec.count, [&](int t, int g, ge_cached& e) {
      const BaseTable bt{B.btab + ((size_t)members[rec.first + t] * B.cap + b) * btab_quads<T>()};
      bt.load(e, g);
    });
  }
}
// k_sum_finish: one lane = one sum of one ballot: its table (table slot sums[k].out_base) from its accumulator
template <int T>
__global__ void __launch_bounds__(NT, EG_SUM_WAVES) k_sum_finish(EngineBufs B, const egplan::SumBase* sums, int n_sums) {
  const size_t total = (size_t)n_sums * B.n;
  WsRows tmp;
  tmp.init(B.ws);
  for (size_t j = (size_t)blockIdx.x * NT + threadIdx.x; j < total; j += (size_t)gridDim.x * NT) {
    const u32 k = (u32)(j / B.n), b = (u32)(j % B.n);
    BaseTable out{B.btab + ((size_t)sums[k].out_base * B.cap + b) * btab_quads<T>()};
    const BaseTable acc{B.sacc + ((size_t)k * B.cap + b) * (T * BTAB_ENTRY_QUADS)};
    ge_teeth_tables_sum<T>(out, tmp, 1, [&](int, int g, ge_cached& e) { acc.load(e, teeth_first_flip_index<T>(g)); });
  }
}

// ---- group equations: P = sum_i [a_i]P_i + [g]G + [k]K (halved scalars; k_encode_batch then emits encode(2P)) ------------------
// One lane = one group equation of one ballot: vartime_double_mul_generator / vartime_multi_mul (ring.rs:342-350,
// log_equality.rs:160-164, mul.rs:213-247).  Persistent blocks stride over (class, ballot); lanes of a wave share the class, so
// control flow is uniform.  The equations are split by shape into kernels whose live registers fit the file without scratch
// (one kernel carrying every shape spilled 159 VGPRs):
//   k_eq_table<false>  one variable base with a comb table (every ring equation) + the fixed-base combs   -- the dominant kernel
//   k_eq_table<true>   several table-backed bases evaluated on ONE doubling chain (Straus); sign vectors of the scalars in LDS
//   k_eq_direct        one base without a table (radix-16 ladder over a per-lane workspace table): the log-equality equations
//   k_eq_generic       anything else (any mix of terms, evaluated term by term)
//   k_encode_plain     serialize_element of points that are not produced by an equation (derived ciphertexts)
__device__ __forceinline__ void eq_fixed_terms(ge& acc, const EngineBufs& B, u32 b, const egplan::JobClass& jc) {
  const FixedTable tg(B.tabG), tk(B.tabK);
  if (jc.g.kind != egplan::SRC_NONE) {
    u32 s[8], dg[EG_COMB_WORDS];
    load_scalar(s, B, b, jc.g, true);
    sc_recode_comb(dg, s);
    ge_fixed_mul_add(acc, tg, dg);
  }
  if (jc.k.kind != egplan::SRC_NONE) {
    u32 s[8], dg[EG_COMB_WORDS];
    load_scalar(s, B, b, jc.k, true);
    sc_recode_comb(dg, s);
    ge_fixed_mul_add(acc, tk, dg);
  }
}

template <bool MULTI, int T>
__global__ void __launch_bounds__(NT, EG_EQ_WAVES) k_eq_table(EngineBufs B, const egplan::JobClass* classes, const egplan::VarTerm* terms,
                                                    int class_first, int n_classes, int group) {
  extern __shared__ u32 eq_signs[];          // MULTI: [group][9][NT] sign vectors of the multipliers (sc_teeth_signs)
  const size_t total = (size_t)n_classes * B.n;
  for (size_t j = (size_t)blockIdx.x * NT + threadIdx.x; j < total; j += (size_t)gridDim.x * NT) {
    const u32 c = class_first + (u32)(j / B.n), b = (u32)(j % B.n);
    const egplan::JobClass jc = classes[c];
    ge acc;
    if (!MULTI) {
      const egplan::VarTerm vt = terms[jc.term_first];
      u32 s[8];
      load_scalar(s, B, b, vt.s, true);
      u64 rows[T];
      sc_recode_teeth<T>(rows, s);
      BaseTable bt{B.btab + ((size_t)vt.base * B.cap + b) * btab_quads<T>()};
      ge_teeth_mul<T>(acc, bt, rows);
    } else {
      // groups of up to `group` terms share a doubling chain (the group size is what fits LDS at two blocks per CU)
      const int nt = (int)jc.term_count;
#pragma unroll 1
      for (int t0 = 0; t0 < nt; t0 += group) {
        const int m = min(group, nt - t0);
#pragma unroll 1
        for (int t = 0; t < m; ++t) {
          u32 s[8], sg[9];
          load_scalar(s, B, b, terms[jc.term_first + t0 + t].s, true);
          sc_teeth_signs<T>(sg, s);
#pragma unroll
          for (int w = 0; w < 9; ++w) eq_signs[(t * 9 + w) * NT + threadIdx.x] = sg[w];
        }
        ge part;
        ge_teeth_mul_multi<T>(part, m,
            [&](int t, int col, int& idx, bool& neg) {
              sc_teeth_column<T>([&](int w) { return eq_signs[(t * 9 + w) * NT + threadIdx.x]; }, col, idx, neg);
            },
            [&](int t, int idx, ge_cached& e) {
              const BaseTable bt{B.btab + ((size_t)terms[jc.term_first + t0 + t].base * B.cap + b) * btab_quads<T>()};
              bt.load(e, idx);
            });
        if (t0 == 0) acc = part;
        else { ge sum; ge_add_full(sum, acc, part); acc = sum; }
      }
    }
    eq_fixed_terms(acc, B, b, jc);
    store_pt(B.dpt, B.cap, jc.out_slot, b, acc);       // encoded (as 2 * acc) by k_encode_batch
  }
}

// one base WITHOUT a table (used once: the sum of the ciphertexts in the log-equality proof, log_equality.rs:160-164): uniform signed
// radix-16 ladder over a per-lane table {1..8}P kept in a workspace slice
__global__ void __launch_bounds__(NT, 2) k_eq_direct(EngineBufs B, const egplan::JobClass* classes, const egplan::VarTerm* terms,
                                                     int class_first, int n_classes) {
  const size_t total = (size_t)n_classes * B.n;
  WsTable tab;
  tab.init(B.ws);
  for (size_t j = (size_t)blockIdx.x * NT + threadIdx.x; j < total; j += (size_t)gridDim.x * NT) {
    const u32 c = class_first + (u32)(j / B.n), b = (u32)(j % B.n);
    const egplan::JobClass jc = classes[c];
    const egplan::VarTerm vt = terms[jc.term_first];
    u32 s[8], dg[8];
    load_scalar(s, B, b, vt.s, true);
    sc_recode_radix16(dg, s);
    ge p, acc;
    load_pt(p, B.pts, B.cap, vt.slot, b);
    ge_var_table_build(tab, p);
    ge_var_mul(acc, tab, dg);
    eq_fixed_terms(acc, B, b, jc);
    store_pt(B.dpt, B.cap, jc.out_slot, b, acc);
  }
}

template <int T>
__global__ void __launch_bounds__(NT, 2) k_eq_generic(EngineBufs B, const egplan::JobClass* classes,
                                                      const egplan::VarTerm* terms, int class_first, int n_classes) {
  const size_t total = (size_t)n_classes * B.n;
  WsTable tab;
  tab.init(B.ws);
  for (size_t j = (size_t)blockIdx.x * NT + threadIdx.x; j < total; j += (size_t)gridDim.x * NT) {
    const u32 c = class_first + (u32)(j / B.n), b = (u32)(j % B.n);
    const egplan::JobClass jc = classes[c];
    ge acc;
    if (jc.term_count == 0) ge_identity(acc);
#pragma unroll 1
    for (u32 t = 0; t < jc.term_count; ++t) {
      const egplan::VarTerm vt = terms[jc.term_first + t];
      u32 s[8];
      load_scalar(s, B, b, vt.s, true);
      ge part;
      if (vt.base != 0xffffu) {
        BaseTable bt{B.btab + ((size_t)vt.base * B.cap + b) * btab_quads<T>()};
        u64 rows[T];
        sc_recode_teeth<T>(rows, s);
        ge_teeth_mul<T>(part, bt, rows);
      } else {
        ge p;
        load_pt(p, B.pts, B.cap, vt.slot, b);
        u32 dg[8];
        sc_recode_radix16(dg, s);
        ge_var_table_build(tab, p);
        ge_var_mul(part, tab, dg);
      }
      if (t == 0) acc = part;
      else { ge sum; ge_add_full(sum, acc, part); acc = sum; }
    }
    eq_fixed_terms(acc, B, b, jc);
    store_pt(B.dpt, B.cap, jc.out_slot, b, acc);
  }
}

// serialize_element (ristretto.rs:88-90) of point slots: the "enc" bytes of derived ciphertexts (range.rs:572), the sum of the
// choices (choice.rs:83-86).  class: enc_slot -> out_slot.
__global__ void __launch_bounds__(NT, 2) k_encode_plain(EngineBufs B, const egplan::JobClass* classes, int class_first, int n_classes) {
  const size_t total = (size_t)n_classes * B.n;
  for (size_t j = (size_t)blockIdx.x * NT + threadIdx.x; j < total; j += (size_t)gridDim.x * NT) {
    const u32 c = class_first + (u32)(j / B.n), b = (u32)(j % B.n);
    const egplan::JobClass jc = classes[c];
    ge p;
    load_pt(p, B.pts, B.cap, jc.enc_slot, b);
    u32 out[8];
    ristretto_encode(out, p);
    store32(B.cmp, B.cap, jc.out_slot, b, out);
  }
}

// ---- k_encode_batch: serialize_element for all deferred commitments of a ballot with ONE field inversion ------------------------
// (ge_double_encode_prepare / _finish: the commitments were evaluated with halved scalars, out = encode(2P).)
__device__ __forceinline__ void encw_store(u32* encw, u32 cap, u32 slot, u32 b, const fe& f) {
#pragma unroll
  for (int i = 0; i < EG_NL; ++i) encw[((size_t)slot * EG_NL + i) * cap + b] = f.v[i];
}
__device__ __forceinline__ void encw_load(fe& f, const u32* encw, u32 cap, u32 slot, u32 b) {
#pragma unroll
  for (int i = 0; i < EG_NL; ++i) f.v[i] = encw[((size_t)slot * EG_NL + i) * cap + b];
}
__global__ void __launch_bounds__(NT, 2) k_encode_batch(EngineBufs B, const unsigned short* slots, int n_slots) {
  for (u32 b = blockIdx.x * NT + threadIdx.x; b < B.n; b += gridDim.x * NT) {
    fe prod; fe_1(prod);
    u32 zero_mask = 0;
#pragma unroll 1
    for (int k = 0; k < n_slots; ++k) {
      ge p;
      load_pt(p, B.dpt, B.cap, slots[k], b);
      fe n; bool zero;
      ge_double_encode_prepare(n, zero, p);
      zero_mask |= (zero ? 1u : 0u) << (k & 31);
      encw_store(B.encw, B.cap, 2 * k, b, prod);       // prefix product before k
      encw_store(B.encw, B.cap, 2 * k + 1, b, n);
      fe t; fe_mul(t, prod, n); prod = t;
    }
    fe inv;
    fe_invert(inv, prod);
#pragma unroll 1
    for (int k = n_slots - 1; k >= 0; --k) {
      fe pre, n, inv_n, t;
      encw_load(pre, B.encw, B.cap, 2 * k, b);
      encw_load(n, B.encw, B.cap, 2 * k + 1, b);
      fe_mul(inv_n, inv, pre);                          // 1 / N_k
      fe_mul(t, inv, n); inv = t;                       // inverse of the prefix product before k
      ge p;
      load_pt(p, B.dpt, B.cap, slots[k], b);
      u32 out[8];
      ge_double_encode_finish(out, p, inv_n, ((zero_mask >> (k & 31)) & 1u) != 0);
      store32(B.cmp, B.cap, slots[k], b, out);
    }
  }
}

// ---- k_hash: Merlin transcript programs (proofs/mod.rs:39-57 + the per-proof label schedules) --------------------------------------
__global__ void __launch_bounds__(NT) k_hash(EngineBufs B, const egplan::HashInst* insts, const egplan::HashOp* ops,
                                             int inst_first, int n_insts) {
  __shared__ u32 lds[50 * NT];
  const size_t j = (size_t)blockIdx.x * NT + threadIdx.x;
  if (j >= (size_t)n_insts * B.n) return;
  const u32 b = (u32)(j % B.n);
  const egplan::HashInst hi = insts[inst_first + (u32)(j / B.n)];
  Transcript<LdsState> t;
  t.st.base = lds + threadIdx.x;
  t.pos = 0; t.pos_begin = 0; t.cur_flags = 0;
#pragma unroll 1
  for (u32 o = 0; o < hi.op_count; ++o) {
    const egplan::HashOp op = ops[hi.op_first + o];
    const char* label = reinterpret_cast<const char*>(B.blob) + (op.a >> 12);
    const int label_len = (int)(op.a & 0xfffu);
    switch (op.op) {
      case egplan::OP_NEW:
        merlin_init(t, label, label_len);
        break;
      case egplan::OP_APPEND_BLOB:
        merlin_append_bytes(t, label, label_len, reinterpret_cast<const char*>(B.blob) + (op.b >> 12), (int)(op.b & 0xfffu));
        break;
      case egplan::OP_APPEND_WIRE: {
        merlin_frame(t, label, label_len, op.c * 32u);
        strobe_begin_op(t, EG_FLAG_AD);
#pragma unroll 1
        for (u32 it = 0; it < op.c; ++it) {
          u32 w[8];
          load_wire_item(w, B, b, op.b + it);
          strobe_absorb_words(t, w, 32);
        }
        break;
      }
      case egplan::OP_APPEND_CMP: {
        const u32 n = (op.c == 0xffffu) ? 1u : 2u;
        merlin_frame(t, label, label_len, n * 32u);
        strobe_begin_op(t, EG_FLAG_AD);
        u32 w[8];
        load32(w, B.cmp, B.cap, op.b, b);
        strobe_absorb_words(t, w, 32);
        if (n == 2) { load32(w, B.cmp, B.cap, op.c, b); strobe_absorb_words(t, w, 32); }
        break;
      }
      case egplan::OP_APPEND_U64:
        merlin_append_u64(t, label, label_len, (u64)op.b);
        break;
      case egplan::OP_CHALLENGE: {
        u32 wide[16], e[8];
        merlin_challenge64(t, label, label_len, wide);
        sc_from_wide(e, wide);
        store32(B.chal, B.cap, op.b, b, e);
        if (op.c > 1u) {            // m * e for the folded admissible value of the next equation (ring.rs:338)
          u32 m[8], t[8];
          const u32 z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
          sc_from_u64(m, (u64)op.c);
          sc_muladd(t, e, m, z);
          store32(B.chal, B.cap, op.b + 1u, b, t);
        }
        break;
      }
      case egplan::OP_CHALLENGE_CHECK: {
        u32 wide[16], e[8], want[8];
        merlin_challenge64(t, label, label_len, wide);
        sc_from_wide(e, wide);
        load_wire_item(want, B, b, op.b);
        B.flags[(size_t)op.c * B.cap + b] = sc_eq(e, want) ? 1u : 0u;
        break;
      }
      case egplan::OP_LOAD_PREFIX:
        merlin_import(t, B.prefixes + (size_t)op.b * 52);
        break;
      case egplan::OP_SAVE_PREFIX: {
        u32 w[52];
        merlin_export(t, w);
#pragma unroll 1
        for (int i = 0; i < 52; ++i) B.prefixes[(size_t)op.b * 52 + i] = w[i];
        break;
      }
      case egplan::OP_LOAD_STATE: {
#pragma unroll 1
        for (int i = 0; i < 50; ++i) t.st.wr(i, B.states[((size_t)op.b * 52 + i) * B.cap + b]);
        const u32 m = B.states[((size_t)op.b * 52 + 50) * B.cap + b];
        t.pos = m & 0xffu; t.pos_begin = (m >> 8) & 0xffu; t.cur_flags = (m >> 16) & 0xffu;
        break;
      }
      case egplan::OP_SAVE_STATE: {
#pragma unroll 1
        for (int i = 0; i < 50; ++i) B.states[((size_t)op.b * 52 + i) * B.cap + b] = t.st.rd(i);
        B.states[((size_t)op.b * 52 + 50) * B.cap + b] = t.pos | (t.pos_begin << 8) | (t.cur_flags << 16);
        break;
      }
      default: break;
    }
  }
}

// ---- k_status: first failing check wins, in the reference's order (choice.rs:358-380, quadratic_voting.rs:291-329) ---------------------
__global__ void __launch_bounds__(NT) k_status(EngineBufs B, const egplan::StatusRule* rules, int n_rules) {
  const u32 b = blockIdx.x * NT + threadIdx.x;
  if (b >= B.n) return;
  const u32 bad = B.bad_item[b];
  u32 st = 0;
  if (bad != 0xffffffffu) {
    st = (bad & 3u) | ((bad >> 2) << 8);
  } else {
#pragma unroll 1
    for (int r = 0; r < n_rules; ++r) {
      if (B.flags[(size_t)rules[r].flag_slot * B.cap + b] == 0u) { st = rules[r].status; break; }
    }
  }
  B.status[b] = st;
}

// ---- tally: totals[k] += vote[k] over accepted ballots (examples/voting.rs:199-203) ---------------------------------------------------------
// acc = sum over the 64 lanes of a wavefront, in lane 0: wavefront-shuffle point accumulation (six rounds of 36 shuffles + one addition)
__device__ __forceinline__ void wave_reduce_points(ge& acc) {
#pragma unroll 1
  for (int off = 32; off >= 1; off >>= 1) {
    u32 w[PT_WORDS];
    ge_to_words(w, acc);
#pragma unroll
    for (int k = 0; k < PT_WORDS; ++k) w[k] = (u32)__shfl_down((int)w[k], off, 64);
    ge other, sum;
    words_to_ge(other, w);
    ge_add_full(sum, acc, other);
    acc = sum;
  }
}
// acc = sum over the block, in thread 0: every wavefront folds its 64 lanes with shuffles, the four wavefront sums meet in LDS
__device__ __forceinline__ void block_reduce_points(ge& acc, u32* lds /* [NT / 64][PT_WORDS] */) {
  wave_reduce_points(acc);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
    u32 w[PT_WORDS];
    ge_to_words(w, acc);
#pragma unroll
    for (int i = 0; i < PT_WORDS; ++i) lds[wave * PT_WORDS + i] = w[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll 1
    for (int k = 1; k < NT / 64; ++k) {
      u32 w[PT_WORDS];
#pragma unroll
      for (int i = 0; i < PT_WORDS; ++i) w[i] = lds[k * PT_WORDS + i];
      ge other, sum;
      words_to_ge(other, w);
      ge_add_full(sum, acc, other);
      acc = sum;
    }
  }
}

// grid (G, n_slots): block (x, k) sums point slot tally_slots[k] over its share of accepted ballots
__global__ void __launch_bounds__(NT) k_tally_partial(EngineBufs B, const u32* tally_slots, u32* partial /* [n_slots][G][PT_WORDS] */) {
  __shared__ u32 lds[PT_WORDS * (NT / 64)];
  const u32 slot = tally_slots[blockIdx.y];
  ge acc;
  ge_identity(acc);
  for (u32 b = blockIdx.x * NT + threadIdx.x; b < B.n; b += gridDim.x * NT) {
    if (B.status[b] != 0u) continue;
    ge p, sum;
    load_pt(p, B.pts, B.cap, slot, b);
    ge_add_full(sum, acc, p);
    acc = sum;
  }
  block_reduce_points(acc, lds);
  if (threadIdx.x == 0) {
    u32 w[PT_WORDS];
    ge_to_words(w, acc);
    for (int i = 0; i < PT_WORDS; ++i) partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * PT_WORDS + i] = w[i];
  }
}
// one block per slot: tally[k] += sum_x partial[k][x]
__global__ void __launch_bounds__(NT) k_tally_final(const u32* partial, int G, u32* tally /* [n_slots][PT_WORDS] */) {
  __shared__ u32 lds[PT_WORDS * (NT / 64)];
  ge acc;
  ge_identity(acc);
  for (int x = threadIdx.x; x < G; x += NT) {
    u32 w[PT_WORDS];
    for (int i = 0; i < PT_WORDS; ++i) w[i] = partial[((size_t)blockIdx.x * G + x) * PT_WORDS + i];
    ge p, sum;
    words_to_ge(p, w);
    ge_add_full(sum, acc, p);
    acc = sum;
  }
  block_reduce_points(acc, lds);
  if (threadIdx.x == 0) {
    u32 w[PT_WORDS];
    for (int i = 0; i < PT_WORDS; ++i) w[i] = tally[(size_t)blockIdx.x * PT_WORDS + i];
    ge cur, sum;
    words_to_ge(cur, w);
    ge_add_full(sum, cur, acc);
    ge_to_words(w, sum);
    for (int i = 0; i < PT_WORDS; ++i) tally[(size_t)blockIdx.x * PT_WORDS + i] = w[i];
  }
}
__global__ void k_tally_init(u32* tally, int n_slots) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_slots) return;
  ge id; ge_identity(id);
  u32 w[PT_WORDS]; ge_to_words(w, id);
  for (int i = 0; i < PT_WORDS; ++i) tally[(size_t)k * PT_WORDS + i] = w[i];
}
__global__ void k_tally_encode(const u32* tally, int n_slots, u32* out /* [n_slots][8] */) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_slots) return;
  u32 w[PT_WORDS];
  for (int i = 0; i < PT_WORDS; ++i) w[i] = tally[(size_t)k * PT_WORDS + i];
  ge p; words_to_ge(p, w);
  u32 o[8]; ristretto_encode(o, p);
  for (int i = 0; i < 8; ++i) out[(size_t)k * 8 + i] = o[i];
}

// tally[k] += decode(in[k]): resumes a running tally from its canonical encodings (checkpoint / merge of an earlier batch);
// bad counts the encodings that fail to decode (the tally is then left untouched for that slot)
__global__ void k_tally_add_encoded(const u32* in, int n_slots, u32* tally, u32* bad) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_slots) return;
  u32 w[8];
  for (int i = 0; i < 8; ++i) w[i] = in[(size_t)k * 8 + i];
  ge p;
  if (!ristretto_decode(p, w)) { atomicAdd(bad, 1u); return; }
  u32 t[PT_WORDS];
  for (int i = 0; i < PT_WORDS; ++i) t[i] = tally[(size_t)k * PT_WORDS + i];
  ge cur, sum; words_to_ge(cur, t);
  ge_add_full(sum, cur, p);
  ge_to_words(t, sum);
  for (int i = 0; i < PT_WORDS; ++i) tally[(size_t)k * PT_WORDS + i] = t[i];
}

// tally[k] += src[k] (extended points): puts a set-aside running tally back after a host call tallied its own batch
__global__ void k_tally_add_points(const u32* src, int n_slots, u32* tally) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_slots) return;
  u32 a[PT_WORDS], b[PT_WORDS];
  for (int i = 0; i < PT_WORDS; ++i) { a[i] = src[(size_t)k * PT_WORDS + i]; b[i] = tally[(size_t)k * PT_WORDS + i]; }
  ge p, q, sum; words_to_ge(p, a); words_to_ge(q, b);
  ge_add_full(sum, q, p);
  ge_to_words(b, sum);
  for (int i = 0; i < PT_WORDS; ++i) tally[(size_t)k * PT_WORDS + i] = b[i];
}

// out[k] = encode( sum_r decode(in[r][k]) ): merges the per-GPU tallies after the all-gather.  Encodings are
// canonical, so the result does not depend on the order of ranks or on how ballots were sharded.  An encoding that does
// not decode counts in *bad (if given) and contributes the identity: the caller must treat bad != 0 as a failed exchange.
__global__ void k_points_sum(const u32* in, int n_ranks, int n_points, u32* out, u32* bad) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_points) return;
  ge acc; ge_identity(acc);
  bool ok = true;
  for (int r = 0; r < n_ranks; ++r) {
    u32 w[8];
    for (int i = 0; i < 8; ++i) w[i] = in[((size_t)r * n_points + k) * 8 + i];
    ge p, sum;
    ok = ok & ristretto_decode(p, w);
    ge_add_full(sum, acc, p); acc = sum;
  }
  u32 o[8]; ristretto_encode(o, acc);
  for (int i = 0; i < 8; ++i) out[(size_t)k * 8 + i] = o[i];
  if (!ok && bad) atomicAdd(bad, 1u);
}

// ---- election setup --------------------------------------------------------------------------------------------------------------------------------
// Fixed-base comb tables: tab[w*E + k-1] = niels([k * 2^(B w)] Base), w < ceil(254 / B), 1 <= k <= E = 2^(B-1).
// k_comb_window_bases: bases[w] = [2^(B w)] Base (one lane, B * windows doublings).
__global__ void k_comb_window_bases(const u32* base_words /* PT_WORDS */, int bits, u32* bases /* [windows][PT_WORDS] */) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  u32 bw[PT_WORDS];
  for (int i = 0; i < PT_WORDS; ++i) bw[i] = base_words[i];
  ge p; words_to_ge(p, bw);
  const int windows = comb_windows(bits);
#pragma unroll 1
  for (int w = 0; w < windows; ++w) {
    ge_to_words(bw, p);
    for (int i = 0; i < PT_WORDS; ++i) bases[w * PT_WORDS + i] = bw[i];
#pragma unroll 1
    for (int i = 0; i < bits; ++i) { ge d; ge_dbl_full(d, p); p = d; }
  }
}
// k_build_fixed_table: one lane = a run of COMB_RUN consecutive entries of one window.  The run starts from [k0]P_w (double-and-add)
// and proceeds by additions of P_w; the entries are made affine with ONE field inversion per run (Montgomery's trick: the prefix
// products of the Z coordinates sit in `scratch`, the projective points wait in their own table slots).  ~16 field multiplications
// per entry, against ~440 + an inversion when every entry was computed on its own (round 1): a 20-bit table takes ~1 ms instead of
// 70 ms, which is what makes the 24-bit tables (92 M entries per base) affordable.
constexpr int COMB_RUN = 64;
constexpr int COMB_HEADER_QUADS = 8;      // the allocation starts one cache line before the first entry; tab[-1] is the header
__global__ void __launch_bounds__(NT) k_build_fixed_table(const u32* bases, int bits, uint4* tab, u32* scratch /* [COMB_RUN][EG_NL][lanes] */) {
  const int windows = comb_windows(bits), entries = comb_entries(bits), runs = entries / COMB_RUN;
  const size_t lanes = (size_t)windows * runs;
  const size_t lane = (size_t)blockIdx.x * NT + threadIdx.x;
  if (lane == 0) tab[-1] = make_uint4((u32)bits, (u32)windows, (u32)entries, 0u);
  if (lane >= lanes) return;
  const int w = (int)(lane / runs), k0 = (int)(lane % runs) * COMB_RUN + 1;
  u32 bw[PT_WORDS];
#pragma unroll 1
  for (int i = 0; i < PT_WORDS; ++i) bw[i] = bases[w * PT_WORDS + i];
  ge p; words_to_ge(p, bw);
  ge_cached pc; ge_to_cached(pc, p);
  ge q; ge_identity(q);
#pragma unroll 1
  for (int bit = bits - 1; bit >= 0; --bit) {        // k0 <= 2^(B-1)
    ge d; ge_dbl_full(d, q); q = d;
    if ((k0 >> bit) & 1) { ge_p1p1 t; ge_add(t, q, pc); ge_add_to_p3(q, t); }
  }
  uint4* slot = tab + ((size_t)w * entries + (k0 - 1)) * 8;
  fe prod; fe_1(prod);
#pragma unroll 1
  for (int i = 0; i < COMB_RUN; ++i) {
    u32 o[32];
#pragma unroll
    for (int j = 0; j < EG_NL; ++j) { o[j] = q.X.v[j]; o[EG_NL + j] = q.Y.v[j]; o[2 * EG_NL + j] = q.Z.v[j]; }
#pragma unroll
    for (int j = 3 * EG_NL; j < 32; ++j) o[j] = 0;
#pragma unroll
    for (int qd = 0; qd < 8; ++qd) slot[(size_t)i * 8 + qd] = make_uint4(o[4 * qd], o[4 * qd + 1], o[4 * qd + 2], o[4 * qd + 3]);
#pragma unroll
    for (int j = 0; j < EG_NL; ++j) scratch[((size_t)i * EG_NL + j) * lanes + lane] = prod.v[j];     // product of the Z's before entry i
    fe t; fe_mul(t, prod, q.Z); prod = t;
    if (i + 1 < COMB_RUN) { ge_p1p1 s; ge_add(s, q, pc); ge_add_to_p3(q, s); }
  }
  fe inv;
  fe_invert(inv, prod);
  const fe d2 = EG_FE_2D;
#pragma unroll 1
  for (int i = COMB_RUN - 1; i >= 0; --i) {
    u32 o[32];
#pragma unroll
    for (int qd = 0; qd < 8; ++qd) {
      const uint4 v = slot[(size_t)i * 8 + qd];
      o[4 * qd] = v.x; o[4 * qd + 1] = v.y; o[4 * qd + 2] = v.z; o[4 * qd + 3] = v.w;
    }
    fe X, Y, Z, pre;
    fe_0(X); fe_0(Y); fe_0(Z); fe_0(pre);
#pragma unroll
    for (int j = 0; j < EG_NL; ++j) { X.v[j] = o[j]; Y.v[j] = o[EG_NL + j]; Z.v[j] = o[2 * EG_NL + j]; pre.v[j] = scratch[((size_t)i * EG_NL + j) * lanes + lane]; }
    fe zi, t, x, y;
    fe_mul(zi, inv, pre);                 // 1 / Z_i
    fe_mul(t, inv, Z); inv = t;           // inverse of the product before entry i
    fe_mul(x, X, zi);
    fe_mul(y, Y, zi);
    ge_niels n;
    fe_add(n.ypx, y, x); fe_carry(n.ypx);
    fe_sub(n.ymx, y, x); fe_carry(n.ymx);
    fe_mul(n.xy2d, x, y);
    fe_mul(n.xy2d, n.xy2d, d2);
#pragma unroll
    for (int j = 0; j < EG_NL; ++j) { o[j] = n.ypx.v[j]; o[EG_NL + j] = n.ymx.v[j]; o[2 * EG_NL + j] = n.xy2d.v[j]; }
#pragma unroll
    for (int j = 3 * EG_NL; j < 32; ++j) o[j] = 0;
#pragma unroll
    for (int qd = 0; qd < 8; ++qd) slot[(size_t)i * 8 + qd] = make_uint4(o[4 * qd], o[4 * qd + 1], o[4 * qd + 2], o[4 * qd + 3]);
  }
}
// self-check of a comb table: sampled entries (and the corners of windows and runs) recomputed one by one, the way round 1 built every
// entry ([k 2^(B w)]Base by doublings and a double-and-add, an inversion per entry), compared as canonical field elements
__global__ void __launch_bounds__(NT) k_check_fixed_table(const u32* base_words /* PT_WORDS */, const uint4* tab, size_t samples, u64 seed,
                                                          unsigned long long* mismatches) {
  const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  if (i >= samples) return;
  const int bits = (int)reinterpret_cast<const u32*>(tab)[-4];
  const int windows = comb_windows(bits), entries = comb_entries(bits);
  const size_t total = (size_t)windows * entries;
  u64 x = seed + 0x9e3779b97f4a7c15ull * (u64)(i + 1);                 // splitmix64
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull; x = (x ^ (x >> 27)) * 0x94d049bb133111ebull; x ^= x >> 31;
  size_t idx = (size_t)(x % total);
  const size_t corners[8] = {0, 1, COMB_RUN - 1, COMB_RUN, (size_t)entries - 1, (size_t)entries, total - COMB_RUN, total - 1};
  if (i < 8) idx = corners[i];
  const int w = (int)(idx / entries), k = (int)(idx % entries) + 1;
  u32 bw[PT_WORDS];
  for (int j = 0; j < PT_WORDS; ++j) bw[j] = base_words[j];
  ge p; words_to_ge(p, bw);
#pragma unroll 1
  for (int j = 0; j < bits * w; ++j) { ge d; ge_dbl_full(d, p); p = d; }
  ge q; ge_identity(q);
#pragma unroll 1
  for (int bit = bits - 1; bit >= 0; --bit) {
    ge d; ge_dbl_full(d, q); q = d;
    if ((k >> bit) & 1) { ge s; ge_add_full(s, q, p); q = s; }
  }
  ge_niels want; ge_to_niels(want, q);
  ge_niels got;
  const FixedTable ft(tab);
  ft.load(got, (int)idx);
  if (!(fe_eq(want.ypx, got.ypx) & fe_eq(want.ymx, got.ymx) & fe_eq(want.xy2d, got.xy2d))) atomicAdd(mismatches, 1ull);
}
// out[0] = generator words (always); if pk != null: out[1] = decoded key, flags[0] = valid, flags[1] = identity
__global__ void k_setup_points(const u32* pk_words, u32* out_words, u32* flags) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  ge g; ge_generator(g);
  u32 w[PT_WORDS]; ge_to_words(w, g);
  for (int i = 0; i < PT_WORDS; ++i) out_words[i] = w[i];
  if (pk_words) {
    u32 pw[8];
    for (int i = 0; i < 8; ++i) pw[i] = pk_words[i];
    ge k;
    const bool ok = ristretto_decode(k, pw);
    ge_to_words(w, k);
    for (int i = 0; i < PT_WORDS; ++i) out_words[PT_WORDS + i] = w[i];
    flags[0] = ok ? 1u : 0u;
    flags[1] = (fe_iszero(k.X) | fe_iszero(k.Y)) ? 1u : 0u;
  }
}
// cpts[i] = [m_i] G for small multipliers (admissible values, range.rs:341-355)
__global__ void __launch_bounds__(NT) k_const_points(const u64* mults, int n, const uint4* tabG, uint4* cpts) {
  const int i = blockIdx.x * NT + threadIdx.x;
  if (i >= n) return;
  u32 s[8], dg[EG_COMB_WORDS];
  sc_from_u64(s, mults[i]);
  sc_recode_comb(dg, s);
  ge acc; ge_identity(acc);
  const FixedTable tg(tabG);
  ge_fixed_mul_add(acc, tg, dg);
  u32 w[PT_WORDS]; ge_to_words(w, acc);
  for (int q = 0; q < PT_QUADS; ++q) cpts[(size_t)i * PT_QUADS + q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
}

// ---- primitive tier kernels (AoS 32-byte items, one lane per problem) ---------------------------------------------------------------------------------------
__device__ __forceinline__ void ld8(u32 w[8], const u32* p) {
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i] = p[i];
}
__device__ __forceinline__ void st8(u32* p, const u32 w[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = w[i];
}
__global__ void __launch_bounds__(NT) k_prim_scalar_from_wide(size_t n, const u32* wide, u32* out) {
  const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  if (i >= n) return;
  u32 w[16], o[8];
#pragma unroll
  for (int k = 0; k < 16; ++k) w[k] = wide[i * 16 + k];
  sc_from_wide(o, w);
  st8(out + i * 8, o);
}
__global__ void __launch_bounds__(NT) k_prim_scalar_canonical(size_t n, const u32* s, unsigned char* ok) {
  const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  if (i >= n) return;
  u32 w[8]; ld8(w, s + i * 8);
  ok[i] = sc_is_canonical(w) ? 1 : 0;
}
__global__ void __launch_bounds__(NT) k_prim_scalar_muladd(size_t n, const u32* a, const u32* b, const u32* c, u32* out) {
  const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  if (i >= n) return;
  u32 aw[8], bw[8], cw[8], o[8];
  ld8(aw, a + i * 8); ld8(bw, b + i * 8); ld8(cw, c + i * 8);
  sc_muladd(o, aw, bw, cw);
  st8(out + i * 8, o);
}
__global__ void __launch_bounds__(NT) k_prim_scalar_neg(size_t n, const u32* a, u32* out) {
  const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  if (i >= n) return;
  u32 aw[8], o[8]; ld8(aw, a + i * 8);
  sc_neg(o, aw);
  st8(out + i * 8, o);
}
__global__ void __launch_bounds__(NT) k_prim_scalar_invert(size_t n, const u32* a, u32* out) {
  const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  if (i >= n) return;
  u32 aw[8], o[8]; ld8(aw, a + i * 8);
  sc_invert(o, aw);
  st8(out + i * 8, o);
}
// ElementOps::is_identity on encodings: the identity's only valid encoding is 32 zero bytes
__global__ void __launch_bounds__(NT, 2) k_prim_point_is_identity(size_t n, const u32* in, unsigned char* is_id, unsigned char* ok) {
  const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  if (i >= n) return;
  u32 w[8]; ld8(w, in + i * 8);
  ge p;
  const bool valid = ristretto_decode(p, w);
  u32 any = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) any |= w[k];
  ok[i] = valid ? 1 : 0;
  is_id[i] = (valid && any == 0) ? 1 : 0;
}
__global__ void __launch_bounds__(NT, 2) k_prim_point_roundtrip(size_t n, const u32* in, u32* out, unsigned char* ok) {
  const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  if (i >= n) return;
  u32 w[8], o[8]; ld8(w, in + i * 8);
  ge p;
  ok[i] = ristretto_decode(p, w) ? 1 : 0;
  ristretto_encode(o, p);
  st8(out + i * 8, o);
}
__global__ void __launch_bounds__(NT, 2) k_prim_point_add(size_t n, const u32* a, const u32* b, int subtract, u32* out, unsigned char* ok) {
  const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  if (i >= n) return;
  u32 aw[8], bw[8], o[8]; ld8(aw, a + i * 8); ld8(bw, b + i * 8);
  ge p, q, r;
  const bool okk = ristretto_decode(p, aw) & ristretto_decode(q, bw);
  if (subtract) ge_sub_full(r, p, q); else ge_add_full(r, p, q);
  ristretto_encode(o, r);
  st8(out + i * 8, o);
  ok[i] = okk ? 1 : 0;
}
// Transcript::new(proto); append_message(msg_label, msg_i); challenge_bytes(chal_label, out_len) for n independent messages:
// the Merlin framing of proofs/mod.rs:39-57 as a primitive, so that known-answer vectors (the upstream merlin KAT) run on the GPU
__global__ void __launch_bounds__(NT) k_prim_merlin(size_t n, const unsigned char* labels, int proto_len, int msg_label_len,
                                                    int chal_label_len, const unsigned char* msgs, int msg_len, unsigned char* out,
                                                    int out_len) {
  __shared__ u32 lds[50 * NT];
  const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  if (i >= n) return;
  Transcript<LdsState> t;
  t.st.base = lds + threadIdx.x;
  const char* l = reinterpret_cast<const char*>(labels);
  merlin_init(t, l, proto_len);
  merlin_append_bytes(t, l + proto_len, msg_label_len, reinterpret_cast<const char*>(msgs) + i * (size_t)msg_len, msg_len);
  merlin_frame(t, l + proto_len + msg_label_len, chal_label_len, (u32)out_len);
  strobe_begin_op(t, EG_FLAG_PRF);
#pragma unroll 1
  for (int k = 0; k < out_len; ++k) out[i * (size_t)out_len + k] = (unsigned char)strobe_squeeze_byte(t);
}

// ---- Group::mul_generator / vartime_double_mul_generator / vartime_multi_mul (ristretto.rs:105-145) as ONE multi-scalar multiplication --------
// out = enc( sum_t [k_t]P_t + [r]G ) for n problems of `terms` terms (terms may be 0, then r must be given).
// Straus' interleaved method, what dalek's vartime_multiscalar_mul does below 190 terms: the terms of a problem are cut into chunks of
// up to `chunk` (<= MSM_CHUNK) terms; one lane evaluates one chunk on ONE chain of 252 doublings shared by its terms (per-lane radix-16
// tables {1..8}P_t in the workspace, signed digits in LDS): 252 doublings + 71 additions per TERM became 252 / chunk + 71.  Problems
// with more than one chunk leave their partial sums in `partial`; k_prim_msm_reduce adds them up with wave shuffles (one wave per
// problem) and finishes (generator term, encoding).  A bucket method does not pay at the sizes a GPU sees them: the 252 sequential
// doublings of a single product are ~0.4 ms for one lane whatever the algorithm, and 2^16 terms in chunks of 8 already keep 8192 lanes
// busy for about that long (bench.py --workload msm).
constexpr int MSM_CHUNK = 8;
__device__ __forceinline__ int msm_digit(const u32* dig, int t, int i) {
  const u32 w = dig[(t * 8 + (i >> 3)) * NT + threadIdx.x];
  const int nib = (int)((w >> (4 * (i & 7))) & 15u);
  return nib >= 8 ? nib - 16 : nib;
}
__device__ __forceinline__ void msm_finish(ge& acc, size_t i, const u32* r, const uint4* tabG, u32* out) {
  if (r) {
    u32 s[8], dg[EG_COMB_WORDS]; ld8(s, r + i * 8);
    sc_recode_comb(dg, s);
    const FixedTable tg(tabG);
    ge_fixed_mul_add(acc, tg, dg);
  }
  u32 o[8]; ristretto_encode(o, acc);
  st8(out + i * 8, o);
}
// A PREPARED point: a ristretto255 element decoded once (k_prim_points_prepare), kept in device memory as affine (x, y, t = xy), 3 x 256
// bits packed = 96 bytes, Z = 1 implied.  Products over one point set (eg_vartime_multi_mul_prepared_batch_device) then skip the 285
// field operations of a decoding per term - a third of the bucket method's time (profiles/r04_msm_by_size.txt).  An encoding that does
// not decode is prepared as the identity (0, 1, 0) and flagged by the prepare call.
constexpr int PREP_WORDS = 24;
EG_D void prepared_load(ge& p, const u32* src) {
  u32 w[PREP_WORDS];
#pragma unroll
  for (int q = 0; q < 6; ++q) { const uint4 v = reinterpret_cast<const uint4*>(src)[q]; w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w; }
  fe_unpack8(p.X, w); fe_unpack8(p.Y, w + 8); fe_unpack8(p.T, w + 16); fe_1(p.Z);
}
__global__ void __launch_bounds__(NT, 2) k_prim_points_prepare(size_t n, const u32* in, u32* out, unsigned char* ok) {
  const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  if (i >= n) return;
  u32 pw[8];
  ld8(pw, in + i * 8);
  ge p;
  const bool good = ristretto_decode(p, pw);          // p = the identity when the encoding does not decode
  fe_carry(p.X); fe_carry(p.Y); fe_carry(p.T);
  u32 w[PREP_WORDS];
  fe_pack8(w, p.X); fe_pack8(w + 8, p.Y); fe_pack8(w + 16, p.T);
  uint4* dst = reinterpret_cast<uint4*>(out + i * PREP_WORDS);
#pragma unroll
  for (int q = 0; q < 6; ++q) dst[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
  if (ok) ok[i] = good ? 1 : 0;
}
template <bool PREPARED>
__global__ void __launch_bounds__(NT, 2) k_prim_msm(size_t n, int terms, int chunk, int n_chunks, const u32* scalars, const u32* points,
                                                    const u32* r, const uint4* tabG, uint4* ws, u32* partial, unsigned char* ok_partial,
                                                    u32* out, unsigned char* ok) {
  extern __shared__ u32 msm_dig[];                       // [chunk][8][NT] radix-16 digits of the chunk's scalars
  uint4* my_ws = ws + ((size_t)blockIdx.x * NT + threadIdx.x) * (size_t)chunk * WS_QUADS;
  const size_t total = n * (size_t)n_chunks;
  for (size_t j = (size_t)blockIdx.x * NT + threadIdx.x; j < total; j += (size_t)gridDim.x * NT) {
    const size_t i = j / n_chunks;
    const int t0 = (int)(j % n_chunks) * chunk, m = min(chunk, terms - t0);
    bool okk = true;
#pragma unroll 1
    for (int t = 0; t < m; ++t) {
      u32 s[8], dg[8];
      ld8(s, scalars + (i * terms + t0 + t) * 8);
      ge p;
      if constexpr (PREPARED) prepared_load(p, points + (i * terms + t0 + t) * (size_t)PREP_WORDS);
      else {
        u32 pw[8];
        ld8(pw, points + (i * terms + t0 + t) * 8);
        okk = okk & ristretto_decode(p, pw);
      }
      sc_recode_radix16(dg, s);
#pragma unroll
      for (int w = 0; w < 8; ++w) msm_dig[(t * 8 + w) * NT + threadIdx.x] = dg[w];
      WsTable tab; tab.base = my_ws + (size_t)t * WS_QUADS;
      ge_var_table_build(tab, p);
    }
    ge acc; ge_identity(acc);
    ge_cached ident; ge_cached_identity(ident);
#pragma unroll 1
    for (int d = 63; d >= 0 && m > 0; --d) {
      ge_p1p1 tt;
      if (d != 63) {
        ge_p2 q; q.X = acc.X; q.Y = acc.Y; q.Z = acc.Z;
#pragma unroll 1
        for (int k = 0; k < 3; ++k) { ge_dbl(tt, q.X, q.Y, q.Z); ge_dbl_to_p2(q, tt); }
        ge_dbl(tt, q.X, q.Y, q.Z);
        ge_dbl_to_p3(acc, tt);
      }
#pragma unroll 1
      for (int t = 0; t < m; ++t) {
        const int dv = msm_digit(msm_dig, t, d), ad = dv < 0 ? -dv : dv;
        WsTable tab; tab.base = my_ws + (size_t)t * WS_QUADS;
        ge_cached c; tab.load(c, ad == 0 ? 0 : ad - 1);
        fe_cmov(c.YpX, ident.YpX, ad == 0); fe_cmov(c.YmX, ident.YmX, ad == 0);
        fe_cmov(c.Z2, ident.Z2, ad == 0); fe_cmov(c.T2d, ident.T2d, ad == 0);
        ge_cached_cneg(c, dv < 0);
        ge_add(tt, acc, c);
        if (t + 1 == m && d > 0) {              // a doubling follows: T is not needed
          ge_p2 q; ge_add_to_p2(q, tt);
          acc.X = q.X; acc.Y = q.Y; acc.Z = q.Z;
        } else {
          ge_add_to_p3(acc, tt);
        }
      }
    }
    if (n_chunks == 1) {
      msm_finish(acc, i, r, tabG, out);
      if (ok) ok[i] = okk ? 1 : 0;
    } else {
      u32 w[PT_WORDS];
      ge_to_words(w, acc);
#pragma unroll
      for (int q = 0; q < PT_QUADS; ++q)
        reinterpret_cast<uint4*>(partial)[j * PT_QUADS + q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
      ok_partial[j] = okk ? 1 : 0;
    }
  }
}
// one wavefront per 64 partial sums of a problem: folds them into one (lane = partial, shuffle reduction).  Long products (2^16 single-term
// chunks) are folded 64-fold per pass until at most 64 partial sums per problem are left for k_prim_msm_reduce.
__global__ void __launch_bounds__(NT, 2) k_prim_msm_fold(size_t n, int n_in, int n_out, const u32* in, const unsigned char* ok_in, u32* out,
                                                         unsigned char* ok_out) {
  const size_t wave = ((size_t)blockIdx.x * NT + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= n * (size_t)n_out) return;                  // whole wavefronts leave together
  const size_t i = wave / n_out;
  const int c = (int)(wave % n_out) * 64 + lane;
  ge acc; ge_identity(acc);
  bool okk = true;
  if (c < n_in) {
    const size_t j = i * (size_t)n_in + c;
    u32 w[PT_WORDS];
#pragma unroll
    for (int q = 0; q < PT_QUADS; ++q) {
      const uint4 v = reinterpret_cast<const uint4*>(in)[j * PT_QUADS + q];
      w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
    }
    words_to_ge(acc, w);
    okk = ok_in[j] != 0;
  }
  wave_reduce_points(acc);
  const bool all_ok = __all(okk ? 1 : 0) != 0;
  if (lane == 0) {
    u32 w[PT_WORDS];
    ge_to_words(w, acc);
#pragma unroll
    for (int q = 0; q < PT_QUADS; ++q) reinterpret_cast<uint4*>(out)[wave * PT_QUADS + q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
    ok_out[wave] = all_ok ? 1 : 0;
  }
}
// one wavefront per problem: sums the problem's partial sums, adds the generator term, encodes
__global__ void __launch_bounds__(NT, 2) k_prim_msm_reduce(size_t n, int n_chunks, const u32* partial, const unsigned char* ok_partial,
                                                           const u32* r, const uint4* tabG, u32* out, unsigned char* ok) {
  const size_t wave = ((size_t)blockIdx.x * NT + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= n) return;                                  // whole wavefronts leave together
  ge acc; ge_identity(acc);
  bool okk = true;
#pragma unroll 1
  for (int c = lane; c < n_chunks; c += 64) {
    const size_t j = wave * (size_t)n_chunks + c;
    u32 w[PT_WORDS];
#pragma unroll
    for (int q = 0; q < PT_QUADS; ++q) {
      const uint4 v = reinterpret_cast<const uint4*>(partial)[j * PT_QUADS + q];
      w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
    }
    ge p, sum; words_to_ge(p, w);
    ge_add_full(sum, acc, p); acc = sum;
    okk = okk & (ok_partial[j] != 0);
  }
  wave_reduce_points(acc);
  const bool all_ok = __all(okk ? 1 : 0) != 0;
  if (lane == 0) {
    msm_finish(acc, wave, r, tabG, out);
    if (ok) ok[wave] = all_ok ? 1 : 0;
  }
}

// ---- measurement hook (eg_selfbench_fmul; bench.py's valu_roofline.box) -------------------------------------------------------------------
// The field multiplication of fe25519.cuh - the very function the table and equation kernels inline - in a bare dependent chain on
// changing 255-bit operands: what tools/ubench/field_bench.hip measures, inside the shipped library so that the bench can calibrate the
// VALU roof on the box it runs on.  A block asks for a third of a CU's LDS, so exactly three blocks = three waves per SIMD are resident
// (the occupancy of k_eq_table).  stamps[wave] = (shader clock cycles, 100 MHz real-time ticks) of the chain: the clock the chip held.
__global__ void __launch_bounds__(NT) k_selfbench_fmul(u32* out, uint2* stamps, u32 seed, int iters) {
  u32 wa[8], wb[8];
  u32 x0 = seed ^ (((u32)blockIdx.x * NT + threadIdx.x) * 0x85ebca6bu);
#pragma unroll
  for (int i = 0; i < 8; ++i) { x0 ^= x0 << 13; x0 ^= x0 >> 17; x0 ^= x0 << 5; wa[i] = x0; }
#pragma unroll
  for (int i = 0; i < 8; ++i) { x0 ^= x0 << 13; x0 ^= x0 >> 17; x0 ^= x0 << 5; wb[i] = x0; }
  wa[7] &= 0x3fffffffu; wb[7] &= 0x3fffffffu;
  fe x, y;
  fe_from_words(x, wa); fe_from_words(y, wb);
  const u64 c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) { fe_mul(x, x, y); fe_mul(y, y, x); }
  const u64 c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  const size_t g = (size_t)blockIdx.x * NT + threadIdx.x;
  if ((threadIdx.x & 63) == 0) stamps[g >> 6] = make_uint2((u32)(c1 - c0), (u32)(r1 - r0));
  u32 o[8];
  fe_to_words(o, x);
#pragma unroll
  for (int i = 0; i < 8; ++i) out[g * 16 + i] = o[i];
  fe_to_words(o, y);
#pragma unroll
  for (int i = 0; i < 8; ++i) out[g * 16 + 8 + i] = o[i];
}

}  // namespace eg

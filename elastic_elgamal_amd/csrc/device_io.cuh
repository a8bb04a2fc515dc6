// device_io.cuh -- HBM / LDS layouts shared by the verifier and generator kernels (see kernels.cuh header).
#pragma once
#include <hip/hip_runtime.h>
#include "ge25519.cuh"
#include "sc25519.cuh"
#include "merlin.cuh"
#include "plan.h"

namespace eg {

constexpr int NT = 256;               // threads per block everywhere (4 wavefronts, one per SIMD)
constexpr int PT_WORDS = 4 * EG_NL;   // a point (four field elements) as limbs: 36 words
constexpr int PT_QUADS = PT_WORDS / 4;   // = 9 uint4: the unit of every SoA point row
static_assert(PT_WORDS % 4 == 0, "a point must be a whole number of uint4");
constexpr int WS_QUADS = 8 * PT_QUADS;   // var-base table: 8 entries x 9 uint4 per lane

struct EngineBufs {
  const u32* wire;      // packed ballots of this chunk
  u32 stride_words;     // ballot stride / 4
  u32 n;                // ballots in this chunk
  u32 cap;              // SoA pitch (chunk capacity)
  uint4* pts;
  uint4* cmp;
  uint4* chal;
  u32* states;
  u32* flags;
  u32* bad_item;
  u32* status;          // [n] output
  const uint4* tabG;    // fixed-base comb table of the generator (FixedTable: header + [window][entry] x 8 uint4)
  const uint4* tabK;    // fixed-base table of the election key
  const uint4* cpts;    // election-constant points [idx][PT_QUADS]
  u32* prefixes;        // hoisted transcript prefixes [idx][52]
  const unsigned char* blob;  // labels and constant messages
  uint4* ws;            // per-lane variable-base tables (direct multiplications)
  uint4* dpt;           // deferred commitments P with out = encode(2P)   [cmp slot][PT_QUADS][cap]
  u32* encw;            // k_encode_batch scratch: prefix products and N   [2 * slot][EG_NL][cap]
  uint4* btab;          // comb tables of the ring bases [table slot][cap] x btab_quads<T>() uint4 (16 or 32 packed entries of 128 B: 2 or 4 KiB)
  uint4* sacc;          // ring-group walk: accumulators of the sums of bases [sum][cap] x T packed entries (k_sum_accumulate / k_sum_finish)
};
constexpr int BTAB_ENTRY_QUADS = 8;   // packed entries: 4 field elements x 256 bits = 128 bytes = ONE cache line per lookup
template <int T> constexpr int btab_quads() { return Teeth<T>::ENTRIES * BTAB_ENTRY_QUADS; }    // T = the plan's comb shape (ge25519.cuh)
constexpr int btab_quads_of(int teeth) { return (1 << (teeth - 1)) * BTAB_ENTRY_QUADS; }

// ---- SoA accessors ------------------------------------------------------------------------------------
__device__ __forceinline__ void words_to_fe4(fe& a, fe& b, fe& c, fe& d, const u32 w[PT_WORDS]) {
#pragma unroll
  for (int i = 0; i < EG_NL; ++i) { a.v[i] = w[i]; b.v[i] = w[EG_NL + i]; c.v[i] = w[2 * EG_NL + i]; d.v[i] = w[3 * EG_NL + i]; }
}
__device__ __forceinline__ void fe4_to_words(u32 w[PT_WORDS], const fe& a, const fe& b, const fe& c, const fe& d) {
#pragma unroll
  for (int i = 0; i < EG_NL; ++i) { w[i] = a.v[i]; w[EG_NL + i] = b.v[i]; w[2 * EG_NL + i] = c.v[i]; w[3 * EG_NL + i] = d.v[i]; }
}
__device__ __forceinline__ void words_to_ge(ge& p, const u32 w[PT_WORDS]) { words_to_fe4(p.X, p.Y, p.Z, p.T, w); }
__device__ __forceinline__ void ge_to_words(u32 w[PT_WORDS], const ge& p) { fe4_to_words(w, p.X, p.Y, p.Z, p.T); }
__device__ __forceinline__ void load_pt(ge& p, const uint4* pts, u32 cap, u32 slot, u32 b) {
  u32 w[PT_WORDS];
#pragma unroll
  for (int q = 0; q < PT_QUADS; ++q) {
    const uint4 v = pts[((size_t)slot * PT_QUADS + q) * cap + b];
    w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
  }
  words_to_ge(p, w);
}
__device__ __forceinline__ void store_pt(uint4* pts, u32 cap, u32 slot, u32 b, const ge& p) {
  u32 w[PT_WORDS];
  ge_to_words(w, p);
#pragma unroll
  for (int q = 0; q < PT_QUADS; ++q) pts[((size_t)slot * PT_QUADS + q) * cap + b] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
}
__device__ __forceinline__ void load_const_pt(ge& p, const uint4* cpts, u32 idx) {
  u32 w[PT_WORDS];
#pragma unroll
  for (int q = 0; q < PT_QUADS; ++q) {
    const uint4 v = cpts[(size_t)idx * PT_QUADS + q];
    w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
  }
  words_to_ge(p, w);
}
__device__ __forceinline__ void load32(u32 w[8], const uint4* arr, u32 cap, u32 slot, u32 b) {
  const uint4 a = arr[((size_t)slot * 2) * cap + b], c = arr[((size_t)slot * 2 + 1) * cap + b];
  w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = c.x; w[5] = c.y; w[6] = c.z; w[7] = c.w;
}
__device__ __forceinline__ void store32(uint4* arr, u32 cap, u32 slot, u32 b, const u32 w[8]) {
  arr[((size_t)slot * 2) * cap + b] = make_uint4(w[0], w[1], w[2], w[3]);
  arr[((size_t)slot * 2 + 1) * cap + b] = make_uint4(w[4], w[5], w[6], w[7]);
}
__device__ __forceinline__ void load_wire_item(u32 w[8], const EngineBufs& B, u32 b, u32 item) {
  const uint4* p = reinterpret_cast<const uint4*>(B.wire + (size_t)b * B.stride_words + (size_t)item * 8);
  const uint4 a = p[0], c = p[1];
  w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = c.x; w[5] = c.y; w[6] = c.z; w[7] = c.w;
}

// ---- table I/O policies --------------------------------------------------------------------------------------
// per-lane variable-base table {1..8}P in a global workspace (1.1 KiB per lane: too big for registers or LDS)
// Each lane owns 1152 contiguous bytes (8 entries x 144 B): a lookup touches only the lane's own 2 cache lines.
// (The first layout, [entry][quad][lane], fetched ~5x more lines than it used because lanes with different digits
// shared 128-B lines: profiles/r01_bench_pmc_counters.txt of the first measurement.)
struct WsTable {
  uint4* base;   // ws + global_lane * 80
  __device__ __forceinline__ void init(uint4* ws) { base = ws + ((size_t)blockIdx.x * NT + threadIdx.x) * WS_QUADS; }
  __device__ __forceinline__ void store(int e, const ge_cached& c) {
    u32 w[PT_WORDS];
    fe4_to_words(w, c.YpX, c.YmX, c.Z2, c.T2d);
#pragma unroll
    for (int q = 0; q < PT_QUADS; ++q) base[e * PT_QUADS + q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
  }
  __device__ __forceinline__ void load(ge_cached& c, int e) const {
    u32 w[PT_WORDS];
#pragma unroll
    for (int q = 0; q < PT_QUADS; ++q) {
      const uint4 v = base[e * PT_QUADS + q];
      w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
    }
    words_to_fe4(c.YpX, c.YmX, c.Z2, c.T2d, w);
  }
};
// The same workspace, word-interleaved across the lanes of the grid ([entry][quad][lane]): for scratch whose entry index is the SAME
// in every lane (the steps of a table's Gray-code walk), so that a wave's access is one coalesced 1-KiB row per quad.
struct WsRows {
  uint4* base;     // ws + global lane
  size_t stride;   // lanes of the grid
  __device__ __forceinline__ void init(uint4* ws) { base = ws + ((size_t)blockIdx.x * NT + threadIdx.x); stride = (size_t)gridDim.x * NT; }
  __device__ __forceinline__ void store(int e, const ge_cached& c) {
    u32 w[PT_WORDS];
    fe4_to_words(w, c.YpX, c.YmX, c.Z2, c.T2d);
#pragma unroll
    for (int q = 0; q < PT_QUADS; ++q) base[(size_t)(e * PT_QUADS + q) * stride] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
  }
  __device__ __forceinline__ void load(ge_cached& c, int e) const {
    u32 w[PT_WORDS];
#pragma unroll
    for (int q = 0; q < PT_QUADS; ++q) {
      const uint4 v = base[(size_t)(e * PT_QUADS + q) * stride];
      w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
    }
    words_to_fe4(c.YpX, c.YmX, c.Z2, c.T2d, w);
  }
};
// ---- packed table entries ------------------------------------------------------------------------------------------------------------
// A field element with limbs in class 1 (every limb within its 29 / 28 bits, limb 1 a hair above) is < 2^256 as an integer: eight
// 32-bit words.  Four of them are 128 bytes - ONE cache line per table lookup instead of the two that a 160-byte entry straddles
// (the lookups' traffic is ~11 % of the equation kernel).  Packing costs a carry sweep and ~15 instructions per
// element when an entry is stored, unpacking ~16 per element at every lookup (fe_pack8 / fe_unpack8, fe25519.cuh).  (The measurement-only
// layouts of round 2 - unpacked 160-byte entries, one-line and shared-entry bounds - are logged in profiles/r02_ab_experiments.txt
// and no longer live in this header.)
// comb table of one (base, ballot): 2^(T-1) cached entries, contiguous (ge_teeth_tables_build / ge_teeth_mul)
struct BaseTable {
  uint4* base;
  __device__ __forceinline__ void store(int e, const ge_cached& c) {
    fe a = c.YpX, b = c.YmX, z = c.Z2;
    fe_carry(a); fe_carry(b); fe_carry(z);          // stored entries come lazily: classes 2, 3, 2 (T2d is a product: class 1)
    u32 w[32];
    fe_pack8(w, a); fe_pack8(w + 8, b); fe_pack8(w + 16, z); fe_pack8(w + 24, c.T2d);
#pragma unroll
    for (int q = 0; q < 8; ++q) base[e * BTAB_ENTRY_QUADS + q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
  }
  __device__ __forceinline__ void load(ge_cached& c, int e) const {
    u32 w[32];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const uint4 v = base[e * BTAB_ENTRY_QUADS + q];
      w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
    }
    fe_unpack8(c.YpX, w); fe_unpack8(c.YmX, w + 8); fe_unpack8(c.Z2, w + 16); fe_unpack8(c.T2d, w + 24);
  }
};
// fixed-base comb table shared by every lane: entry = 8 uint4 (3 x EG_NL = 27 limbs used).  The uint4 in front of the first entry is the
// table's header {window bits, windows, entries per window, 0} (k_build_fixed_table), so a table pointer says how it is cut.
struct FixedTable {
  const uint4* tab;
  int bits;
  __device__ __forceinline__ explicit FixedTable(const uint4* t) : tab(t), bits((int)reinterpret_cast<const u32*>(t)[-4]) {}
  __device__ __forceinline__ void load(ge_niels& c, int idx) const {
    u32 w[32];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const uint4 v = tab[(size_t)idx * 8 + q];
      w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < EG_NL; ++i) { c.ypx.v[i] = w[i]; c.ymx.v[i] = w[EG_NL + i]; c.xy2d.v[i] = w[2 * EG_NL + i]; }
  }
};
// transcript state: word-interleaved LDS column per lane (positions are wave-uniform => conflict free)
struct LdsState {
  u32* base;
  __device__ __forceinline__ u32 rd(int i) const { return base[i * NT]; }
  __device__ __forceinline__ void wr(int i, u32 v) { base[i * NT] = v; }
};


}  // namespace eg

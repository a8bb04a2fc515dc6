// pippenger.cuh -- the bucket method for ONE very large multi-scalar multiplication (Group::vartime_multi_mul with >= 2^17 terms;
// src/group/ristretto.rs:139-145 -> dalek switches from Straus to Pippenger above 190 terms).  The reference never multiplies more than
// 7 terms; this exists because the primitive tier takes any number of terms and Straus pays 64 additions per term (kernels.cuh:
// k_prim_msm, chunks of 8 terms on one doubling chain), the bucket method W = ceil(257 / c) of them:
//   sum_t [k_t]P_t = sum_w 2^(c w) sum_b b * S_(w,b),   S_(w,b) = sum of +-P_t over the terms whose signed digit w is +-b.
// Pipeline (one problem at a time, everything on the caller's stream, no host round trip):
//   k_pip_prepare     one lane = one term: decode the point (affine: Z = 1) into a packed Niels entry, cut the scalar into W signed c-bit
//                     digits, count the terms per bucket (atomicAdd)
//   k_pip_scan        prefix sums over the W 2^(c-1) buckets (one block): list offsets, and the number of PIECES of <= S entries
//                     each bucket is cut into
//   k_pip_fill        one lane = one (term, window): append the term to its bucket's index list (atomic cursor; the ORDER inside a
//                     bucket depends on timing, the sum does not - and the result is a canonical encoding)
//   k_pip_sum_terms   one lane = one piece of one bucket: mixed additions (7 multiplications) of up to 128 terms
//   k_pip_sum_points  one lane = up to 64 consecutive partial sums of one bucket: repeated until every bucket has ONE sum.  A bucket is
//                     as long as the scalars make it - all terms of a window share a bucket when the scalars are equal, and the top window
//                     of 253-bit scalars has two buckets for everything - so no lane ever sums a whole bucket: the depth is fixed by the
//                     number of terms (4 levels cover 2^24), not by the data.
//   k_pip_window      one lane = 16 consecutive buckets of one window: running sums give sum (b - b0) S_b and sum S_b, the weight of the
//                     segment's first bucket is a 14-bit double-and-add; wavefront-shuffle sum over 64 segments (wave_reduce_points), and
//                     lane 0 multiplies the wave's sum by 2^(c w).  The W 2^(c-11) results are partial sums of the problem and go through
//                     k_prim_msm_fold / k_prim_msm_reduce (generator term, encoding) like the partial sums of the Straus path.
// Floor: the c w doublings of the top window are the same ~0.4 ms of one lane that bound every other schedule.
#pragma once
#include "kernels.cuh"

namespace eg {

constexpr int PIP_SEG = 16;            // buckets per lane of k_pip_window
constexpr int PIP_NIELS_WORDS = 24;    // (y + x, y - x, 2dxy), 256 bits each
constexpr int PIP_MIN_C = 12, PIP_MAX_C = 15;    // 2^(c-1) buckets per window: whole wavefronts of segments from c = 12 on; |digit| <= 2^14 fits 15 bits
__host__ __device__ inline int pip_windows(int c) { return (256 + c) / c; }      // any 256-bit scalar plus the carry of the signed recoding

constexpr int PIP_S_TERMS = 128;       // entries per piece of a bucket at the first level (terms), ...
constexpr int PIP_S_POINTS = 64;       // ... and at the following levels (partial sums)
struct PipBufs {
  u32* niels;              // [terms][24]
  unsigned short* digits;  // [W][terms]: |digit| in bits 0..14, sign in bit 15
  u32* counts;             // [W * B] terms per bucket
  u32* offsets;            // [W * B] exclusive prefix sums of counts
  u32* cursors;            // [W * B] fill cursors
  u32* idx;                // [W * terms] term | sign << 31, grouped by bucket
  u32* all_ok;             // [1] 1 while every point decoded
};
// one level of the bucket sums: bucket q owns entries [off[q], off[q] + cnt[q]) of the level's input and is cut into pieces[q] =
// ceil(cnt[q] / S) pieces; piece u (global numbering by the prefix sums piece0[]) becomes entry u of the next level
struct PipLevel { const u32* cnt; const u32* off; const u32* pieces; const u32* piece0; const u32* total; };

// atomicAdd(&ctr[key], 1) for the active lanes of a wavefront, returning each lane's old value.  When many lanes of the wavefront hold
// the same key - the top window of 253-bit scalars has two or three digits for every term, equal scalars one per window - the lanes of
// a key are served by ONE atomic of their leader (a counter hammered by 2^18 single increments costs milliseconds); with keys that differ
// the plain atomic of every lane is faster, so the aggregation runs only if at least a quarter of the wavefront shares the first key.
__device__ __forceinline__ u32 pip_wave_atomic_inc(u32* ctr, size_t key, bool active) {
  u32 old = 0;
  const unsigned long long act = __ballot(active);
  if (act == 0ull) return 0u;
  const int lane = threadIdx.x & 63;
  const int first = __ffsll((long long)act) - 1;
  const size_t key0 = (size_t)__shfl((long long)key, first, 64);
  if (__popcll(__ballot(active && key == key0)) < 16) {
    if (active) old = atomicAdd(&ctr[key], 1u);
    return old;
  }
  unsigned long long rem = act;
  while (rem) {                                    // wave-uniform loop: one round per distinct key
    const int leader = __ffsll((long long)rem) - 1;
    const size_t k = (size_t)__shfl((long long)key, leader, 64);
    const unsigned long long same = __ballot(active && key == k) & rem;
    u32 base = 0;
    if (lane == leader) base = atomicAdd(&ctr[k], (u32)__popcll(same));
    base = (u32)__shfl((int)base, leader, 64);
    if (active && key == k) old = base + (u32)__popcll(same & ((1ull << lane) - 1ull));
    rem &= ~same;
  }
  return old;
}

template <bool PREPARED>      // PREPARED: points are prepared points (kernels.cuh: 96 bytes, already decoded)
__global__ void __launch_bounds__(NT, 2) k_pip_prepare(size_t terms, int c, const u32* scalars, const u32* points, PipBufs P) {
  const size_t t0 = (size_t)blockIdx.x * NT + threadIdx.x;
  const bool live = t0 < terms;                               // dead lanes stay for the wave-level atomics
  const size_t t = live ? t0 : terms - 1;
  const int W = pip_windows(c), B = 1 << (c - 1);
  u32 s[8];
  ld8(s, scalars + t * 8);
  ge p;
  if constexpr (PREPARED) prepared_load(p, points + t * (size_t)PREP_WORDS);
  else {
    u32 pw[8];
    ld8(pw, points + t * 8);
    if (!ristretto_decode(p, pw) && live) atomicAnd(P.all_ok, 0u);     // p is then the identity: the term contributes nothing
  }
  if (live) {
    const fe d2 = EG_FE_2D;
    fe ypx, ymx, xy2d;
    fe_add(ypx, p.Y, p.X); fe_carry(ypx);
    fe_sub(ymx, p.Y, p.X); fe_carry(ymx);
    fe_mul(xy2d, p.T, d2);                                     // Z = 1 after decoding: T = xy
    u32 w[PIP_NIELS_WORDS];
    fe_pack8(w, ypx); fe_pack8(w + 8, ymx); fe_pack8(w + 16, xy2d);
    uint4* dst = reinterpret_cast<uint4*>(P.niels + t * PIP_NIELS_WORDS);
#pragma unroll
    for (int q = 0; q < 6; ++q) dst[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
  }
  u32 carry = 0;
#pragma unroll 1
  for (int w = 0; w < W; ++w) {
    const int off = w * c, wi = off >> 5, sh = off & 31;
    u32 lo = 0, hi = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { lo = (wi == j) ? s[j] : lo; hi = (wi + 1 == j) ? s[j] : hi; }
    const u64 v = (u64)lo | ((u64)hi << 32);
    u32 raw = ((u32)(v >> sh) & ((1u << c) - 1u)) + carry;     // 0 .. 2^c
    carry = raw > (u32)B ? 1u : 0u;
    u32 mag = carry ? (1u << c) - raw : raw;                    // |digit| <= 2^(c-1)
    if (live) P.digits[(size_t)w * terms + t] = (unsigned short)(mag | (carry << 15));
    (void)pip_wave_atomic_inc(P.counts, (size_t)w * B + (mag ? mag - 1 : 0u), live && mag != 0u);
  }
}

// ---- prefix sums over the buckets, for the term lists and for every level of pieces at once -------------------------------------------------------
// Sequence 0 = counts (-> list offsets); sequence l + 1 = pieces of level l: ceil(counts / 128), then ceil(. / 64) of the level before.
// Three launches: sums of tiles of 1024 buckets, an exclusive scan of the tile sums by one block, the scan inside every tile.
constexpr int PIP_SEQ = 6;             // offsets + up to 5 levels
struct PipScan { u32* offsets; u32* pieces[PIP_SEQ - 1]; u32* piece0[PIP_SEQ - 1]; u32* totals; u32* cursors; u32* tile_sums; };
__device__ __forceinline__ void pip_tuple(u32 v[PIP_SEQ], u32 cnt, int levels) {
  v[0] = cnt;
  u32 m = (cnt + PIP_S_TERMS - 1u) / PIP_S_TERMS;
#pragma unroll
  for (int l = 0; l < PIP_SEQ - 1; ++l) { v[l + 1] = l < levels ? m : 0u; m = (m + PIP_S_POINTS - 1u) / PIP_S_POINTS; }
}
__global__ void __launch_bounds__(NT) k_pip_scan_tiles(const u32* counts, u32 n, int levels, PipScan S) {
  __shared__ u32 red[PIP_SEQ][NT / 64];
  u32 sum[PIP_SEQ] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const u32 i = blockIdx.x * 1024u + threadIdx.x * 4u + e;
    u32 v[PIP_SEQ];
    pip_tuple(v, i < n ? counts[i] : 0u, levels);
#pragma unroll
    for (int k = 0; k < PIP_SEQ; ++k) sum[k] += v[k];
  }
#pragma unroll
  for (int k = 0; k < PIP_SEQ; ++k) {
    u32 x = sum[k];
    for (int off = 32; off >= 1; off >>= 1) x += (u32)__shfl_down((int)x, off, 64);
    if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = x;
  }
  __syncthreads();
  if (threadIdx.x < PIP_SEQ) S.tile_sums[blockIdx.x * PIP_SEQ + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}
__global__ void k_pip_scan_tops(u32 n_tiles, PipScan S) {          // one wavefront per sequence: tile sums -> exclusive prefix sums, totals
  const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;           // launched with 64 * PIP_SEQ threads
  u32 run = 0;
  for (u32 b0 = 0; b0 < n_tiles; b0 += 64) {
    const u32 b = b0 + lane;
    const u32 v = b < n_tiles ? S.tile_sums[b * PIP_SEQ + k] : 0u;
    u32 x = v;
    for (int off = 1; off < 64; off <<= 1) { const u32 y = (u32)__shfl_up((int)x, off, 64); if (lane >= off) x += y; }
    if (b < n_tiles) S.tile_sums[b * PIP_SEQ + k] = run + x - v;
    run += (u32)__shfl((int)x, 63, 64);
  }
  if (lane == 0 && k >= 1) S.totals[k - 1] = run;
}
__global__ void __launch_bounds__(NT) k_pip_scan_apply(const u32* counts, u32 n, int levels, PipScan S) {
  __shared__ u32 part[PIP_SEQ][NT];
  u32 v[4][PIP_SEQ], sum[PIP_SEQ] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const u32 i = blockIdx.x * 1024u + threadIdx.x * 4u + e;
    pip_tuple(v[e], i < n ? counts[i] : 0u, levels);
#pragma unroll
    for (int k = 0; k < PIP_SEQ; ++k) sum[k] += v[e][k];
  }
#pragma unroll
  for (int k = 0; k < PIP_SEQ; ++k) part[k][threadIdx.x] = sum[k];
  __syncthreads();
  for (int d = 1; d < NT; d <<= 1) {              // Hillis-Steele inclusive scan over the 256 lanes, all sequences together
    u32 add[PIP_SEQ];
#pragma unroll
    for (int k = 0; k < PIP_SEQ; ++k) add[k] = (int)threadIdx.x >= d ? part[k][threadIdx.x - d] : 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PIP_SEQ; ++k) part[k][threadIdx.x] += add[k];
    __syncthreads();
  }
  u32 run[PIP_SEQ];
#pragma unroll
  for (int k = 0; k < PIP_SEQ; ++k) run[k] = S.tile_sums[blockIdx.x * PIP_SEQ + k] + (threadIdx.x ? part[k][threadIdx.x - 1] : 0u);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const u32 i = blockIdx.x * 1024u + threadIdx.x * 4u + e;
    if (i < n) {
      S.offsets[i] = run[0];
      S.cursors[i] = 0u;
#pragma unroll
      for (int l = 0; l < PIP_SEQ - 1; ++l)
        if (l < levels) { S.pieces[l][i] = v[e][l + 1]; S.piece0[l][i] = run[l + 1]; }
    }
#pragma unroll
    for (int k = 0; k < PIP_SEQ; ++k) run[k] += v[e][k];
  }
}

__global__ void __launch_bounds__(NT) k_pip_fill(size_t terms, int c, PipBufs P) {
  const int W = pip_windows(c), B = 1 << (c - 1);
  const size_t total = terms * (size_t)W, rounds = (total + (size_t)gridDim.x * NT - 1) / ((size_t)gridDim.x * NT);
  for (size_t r = 0; r < rounds; ++r) {                       // every lane runs every round: the atomics are wave-level
    const size_t j = r * (size_t)gridDim.x * NT + (size_t)blockIdx.x * NT + threadIdx.x;
    const bool live = j < total;
    const size_t jj = live ? j : total - 1;
    const size_t w = jj / terms, t = jj % terms;
    const u32 d = P.digits[jj], mag = d & 0x7fffu;
    const size_t q = w * B + (mag ? mag - 1 : 0u);
    const u32 pos = pip_wave_atomic_inc(P.cursors, q, live && mag != 0u);
    if (live && mag) P.idx[P.offsets[q] + pos] = (u32)t | ((d >> 15) << 31);
  }
}

__device__ __forceinline__ void pip_load_niels(ge_niels& e, const u32* niels, u32 t) {
  const uint4* src = reinterpret_cast<const uint4*>(niels + (size_t)t * PIP_NIELS_WORDS);
  u32 w[PIP_NIELS_WORDS];
#pragma unroll
  for (int q = 0; q < 6; ++q) { const uint4 v = src[q]; w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w; }
  fe_unpack8(e.ypx, w); fe_unpack8(e.ymx, w + 8); fe_unpack8(e.xy2d, w + 16);
}

// the bucket that piece u belongs to: the last q with piece0[q] <= u (empty buckets share their successor's piece0 and are skipped)
__device__ __forceinline__ u32 pip_bucket_of(const u32* piece0, u32 n_buckets, u32 u) {
  u32 lo = 0, hi = n_buckets;                    // invariant: piece0[lo] <= u, (hi == n or piece0[hi] > u)
  while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (piece0[mid] <= u) lo = mid; else hi = mid; }
  return lo;
}
__device__ __forceinline__ void pip_store_point(uint4* dst, const ge& p) {
  u32 w[PT_WORDS];
  ge_to_words(w, p);
#pragma unroll
  for (int k = 0; k < PT_QUADS; ++k) dst[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
}
__device__ __forceinline__ void pip_load_point(ge& p, const uint4* src) {
  u32 w[PT_WORDS];
#pragma unroll
  for (int k = 0; k < PT_QUADS; ++k) { const uint4 v = src[k]; w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w; }
  words_to_ge(p, w);
}
// first level: piece u of bucket q = terms idx[off[q] + k S ..) -> out[u]
__global__ void __launch_bounds__(NT, 2) k_pip_sum_terms(PipBufs P, PipLevel L, u32 n_buckets, uint4* out) {
  const u32 total = *L.total;
  for (u32 u = blockIdx.x * NT + threadIdx.x; u < total; u += gridDim.x * NT) {
    const u32 q = pip_bucket_of(L.piece0, n_buckets, u), k = u - L.piece0[q];
    const u32 beg = L.off[q] + k * PIP_S_TERMS, end = min(L.off[q] + L.cnt[q], beg + PIP_S_TERMS);
    ge acc; ge_identity(acc);
#pragma unroll 1
    for (u32 i = beg; i < end; ++i) {
      const u32 e = P.idx[i];
      ge_niels n;
      pip_load_niels(n, P.niels, e & 0x7fffffffu);
      ge_niels_cneg(n, (e >> 31) != 0);
      ge_p1p1 r; ge_madd(r, acc, n);
      ge_add_to_p3(acc, r);
    }
    pip_store_point(out + (size_t)u * PT_QUADS, acc);
  }
}
// following levels: piece u of bucket q = partial sums in[off[q] + k S ..) -> out[u]
__global__ void __launch_bounds__(NT, 2) k_pip_sum_points(const uint4* in, PipLevel L, u32 n_buckets, uint4* out) {
  const u32 total = *L.total;
  for (u32 u = blockIdx.x * NT + threadIdx.x; u < total; u += gridDim.x * NT) {
    const u32 q = pip_bucket_of(L.piece0, n_buckets, u), k = u - L.piece0[q];
    const u32 beg = L.off[q] + k * PIP_S_POINTS, end = min(L.off[q] + L.cnt[q], beg + PIP_S_POINTS);
    ge acc;
    pip_load_point(acc, in + (size_t)beg * PT_QUADS);
#pragma unroll 1
    for (u32 i = beg + 1; i < end; ++i) {
      ge p, t1;
      pip_load_point(p, in + (size_t)i * PT_QUADS);
      ge_add_full(t1, acc, p); acc = t1;
    }
    pip_store_point(out + (size_t)u * PT_QUADS, acc);
  }
}

// grid: W * (B / PIP_SEG) lanes, whole wavefronts per window; out: one partial sum (PT_WORDS) per wavefront, already times 2^(c w)
// sums / cnt / off: the last level (cnt[q] is 0 or 1; the bucket's sum is sums[off[q]])
__global__ void __launch_bounds__(NT, 2) k_pip_window(int c, PipBufs P, const uint4* sums, const u32* cnt, const u32* off, u32* partial,
                                                      unsigned char* ok_partial) {
  const int W = pip_windows(c), B = 1 << (c - 1), segs = B / PIP_SEG;
  const size_t lane = (size_t)blockIdx.x * NT + threadIdx.x;
  if (lane >= (size_t)W * segs) return;                                              // whole wavefronts leave together (segs is a multiple of 64)
  const int w = (int)(lane / segs), s = (int)(lane % segs), b0 = s * PIP_SEG;
  ge run, tot;
  ge_identity(run); ge_identity(tot);
#pragma unroll 1
  for (int j = b0 + PIP_SEG - 1; j >= b0; --j) {
    const size_t q = (size_t)w * B + j;
    ge t1;
    if (cnt[q]) {                                   // (an empty bucket leaves the running sum as it is)
      ge sb;
      pip_load_point(sb, sums + (size_t)off[q] * PT_QUADS);
      ge_add_full(t1, run, sb); run = t1;
    }
    ge_add_full(t1, tot, run); tot = t1;
  }
  // tot = sum (j - b0 + 1) S_j; the weights are j + 1: add [b0] run (b0 < 2^14, a multiple of PIP_SEG).  tot is not needed while
  // [b0] run is made: it waits in LDS (word-interleaved, 36 KB per block), which keeps the double-and-add below 256 registers
  // without scratch (round 4: 256 VGPR + 20 B)
  __shared__ u32 tot_park[PT_WORDS][NT];
  {
    u32 w[PT_WORDS];
    ge_to_words(w, tot);
#pragma unroll
    for (int k = 0; k < PT_WORDS; ++k) tot_park[k][threadIdx.x] = w[k];
  }
  ge m; ge_identity(m);
#pragma unroll 1
  for (int bit = 13; bit >= 0; --bit) {
    ge d; ge_dbl_full(d, m); m = d;
    if ((b0 >> bit) & 1) { ge t1; ge_add_full(t1, m, run); m = t1; }
  }
  {
    u32 w[PT_WORDS];
#pragma unroll
    for (int k = 0; k < PT_WORDS; ++k) w[k] = tot_park[k][threadIdx.x];
    words_to_ge(tot, w);
    ge t1; ge_add_full(t1, tot, m); tot = t1;
  }
  wave_reduce_points(tot);                        // the 64 segments of a wavefront belong to one window (segs is a multiple of 64)
  if ((threadIdx.x & 63) == 0) {
    if (c * w > 0) {                              // times 2^(c w): doublings without T until the last one
      ge_p2 q; q.X = tot.X; q.Y = tot.Y; q.Z = tot.Z;
      ge_p1p1 t;
#pragma unroll 1
      for (int k = 0; k < c * w - 1; ++k) { ge_dbl(t, q.X, q.Y, q.Z); ge_dbl_to_p2(q, t); }
      ge_dbl(t, q.X, q.Y, q.Z);
      ge_dbl_to_p3(tot, t);
    }
    const size_t wave = lane >> 6;
    u32 ow[PT_WORDS];
    ge_to_words(ow, tot);
#pragma unroll
    for (int k = 0; k < PT_QUADS; ++k)
      reinterpret_cast<uint4*>(partial)[wave * PT_QUADS + k] = make_uint4(ow[4 * k], ow[4 * k + 1], ow[4 * k + 2], ow[4 * k + 3]);
    ok_partial[wave] = (wave == 0) ? (unsigned char)(*P.all_ok != 0u) : (unsigned char)1;
  }
}

}  // namespace eg

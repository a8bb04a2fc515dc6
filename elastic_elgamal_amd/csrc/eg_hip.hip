// eg_hip.hip -- host side of libeg_hip.so: context, the plan-driven verification engine and the C ABI
// declared in include/eg_hip.h.  One process drives one GPU (one rank per GPU under torch.distributed);
// everything is launched on a single HIP stream, per chunk of ballots:
//   decode -> canonical checks -> derived points -> [ group equations -> transcript hashing ] x stages
//   -> status -> tally
// There is no CPU fallback: every numeric result comes out of the kernels in kernels.cuh.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <condition_variable>
#include <deque>
#include <list>
#include <unordered_map>
#include <vector>

#include "../../include/eg_hip.h"
#include "host_plan.hpp"
#include "wire_json.hpp"
#include "kernels.cuh"
#include "pippenger.cuh"

using namespace eg;

// defined in eg_gen.hip (separate translation unit so the two compile in parallel)
void eg_launch_qv_encrypt(int blocks, hipStream_t s, u64 seed0, size_t n, int n_options, u64 credits, const u32* votes, u64 rng_skip,
                          int vote_rings, int vote_main, int vote_ring, const u32* d_vote_desc, int credit_rings, int credit_main,
                          int credit_ring, const u32* d_credit_desc, int pre_sumsq, const uint4* tabG, const uint4* tabK,
                          const u32* prefixes, u32* out, u32 stride_words, u32 vote_words, u32 credit_words, u32* gws);
void eg_launch_choice_encrypt(int blocks, hipStream_t s, u64 seed0, size_t n, int n_options, int single, int n_selected,
                              const u32* selection, u64 rng_skip, const uint4* tabG, const uint4* tabK, const u32* prefixes, int pre_main, int pre_ring,
                              int pre_logeq, u32* out, u32 stride_words, u32* gws);
unsigned eg_gen_choice_ws_words(int n_options);
unsigned eg_gen_qv_ws_words(int n_options, unsigned max_rings, unsigned max_responses);

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(x)                                                                                   \
  do {                                                                                              \
    hipError_t e_ = (x);                                                                            \
    if (e_ != hipSuccess)                                                                           \
      return fail(EG_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_) + " (" + __FILE__ + ":" + std::to_string(__LINE__) + ")"); \
  } while (0)

#define TRY_(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)
// An entry point that works on SEVERAL devices (the multi-GPU calls, a JSON stream with one lane per GPU) leaves the calling thread's current
// device as it found it: a host that drives its own HIP work (or torch) on that thread must not find another device selected afterwards.
struct KeepDevice {
  int dev = -1;
  KeepDevice() { if (hipGetDevice(&dev) != hipSuccess) { dev = -1; (void)hipGetLastError(); } }
  ~KeepDevice() { if (dev >= 0) (void)hipSetDevice(dev); }
};
// runs at scope exit, on every path (early returns of HIPCHK included)
struct ScopeExit {
  std::function<void()> fn;
  ~ScopeExit() { if (fn) fn(); }
};
// Every run-time knob of the library, read from the environment in ONE place and at two moments only: eg_init (the context keeps a copy)
// and the creation of a params object (the engine keeps its own copy, so a host may change the environment between two elections).  No
// entry point reads the environment per call.  The table in include/eg_hip.h ("Run-time knobs") is the documentation; keep it in step.
struct Knobs {
  int teeth = 0;                    // EG_TEETH            0 = by plan (5 or 6)
  int streams = 2;                  // EG_STREAMS          work sets (1 or 2)
  size_t chunk = 0;                 // EG_CHUNK            0 = by plan and memory
  int ring_group = -1;              // EG_RING_GROUP       -1 = by plan (off below 256 options), 0 = off, g = rings per group
  int comb_big_bits = -1;           // EG_COMB_BIG_BITS    -1 = EG_COMB_BITS_BIG
  long long comb_big_min = -1;      // EG_COMB_BIG_MIN     -1 = 2^19 items
  int msm_blocks_per_cu = 32;       // EG_MSM_BLOCKS_PER_CU
  size_t msm_lanes = (size_t)1 << 17;       // EG_MSM_LANES
  size_t msm_bucket_min = (size_t)1 << 20;  // EG_MSM_BUCKET_MIN
  size_t json_ring_kb = 0, json_window_kb = 0;   // EG_JSON_RING_KB, EG_JSON_WINDOW_KB   0 = 1 GiB ring, 96 MiB windows
  size_t json_growth = 150;         // EG_JSON_GROWTH      per cent
  size_t json_first_min = 0;        // EG_JSON_FIRST_MIN   0 = the stream's default (stream_begin)
  size_t json_odd_max_mb = 256;     // EG_JSON_ODD_MAX_MB  text of wrong-shape ballots a JSON stream keeps for the object path
  bool json_trace = false;          // EG_JSON_TRACE
  bool allow_any_arch = false;      // EG_ALLOW_ANY_ARCH
};
static Knobs read_knobs() {
  Knobs k;
  auto num = [](const char* name, long long dflt) { const char* v = getenv(name); return v && *v ? strtoll(v, nullptr, 10) : dflt; };
  { const long long t = num("EG_TEETH", 0); if (t == 5 || t == 6) k.teeth = (int)t;
#ifdef EG_TEETH7
    if (t == 7) k.teeth = 7;
#endif
  }
  if (getenv("EG_STREAMS")) k.streams = num("EG_STREAMS", 2) >= 2 ? 2 : 1;
  k.chunk = (size_t)std::max<long long>(0, num("EG_CHUNK", 0));
  if (getenv("EG_RING_GROUP")) k.ring_group = (int)std::max<long long>(0, num("EG_RING_GROUP", 0));
  if (getenv("EG_COMB_BIG_BITS")) k.comb_big_bits = (int)num("EG_COMB_BIG_BITS", -1);
  if (getenv("EG_COMB_BIG_MIN")) k.comb_big_min = std::max<long long>(0, num("EG_COMB_BIG_MIN", -1));
  k.msm_blocks_per_cu = (int)std::max<long long>(1, num("EG_MSM_BLOCKS_PER_CU", 32));
  k.msm_lanes = (size_t)std::max<long long>(1, num("EG_MSM_LANES", 1 << 17));
  k.msm_bucket_min = (size_t)std::max<long long>(0, num("EG_MSM_BUCKET_MIN", 1 << 20));
  k.json_ring_kb = (size_t)std::max<long long>(0, num("EG_JSON_RING_KB", 0));
  k.json_window_kb = (size_t)std::max<long long>(0, num("EG_JSON_WINDOW_KB", 0));
  k.json_growth = (size_t)std::max<long long>(0, num("EG_JSON_GROWTH", 150));
  k.json_first_min = (size_t)std::max<long long>(0, num("EG_JSON_FIRST_MIN", 0));
  k.json_odd_max_mb = (size_t)std::max<long long>(1, num("EG_JSON_ODD_MAX_MB", 256));
  k.json_trace = getenv("EG_JSON_TRACE") != nullptr;
  k.allow_any_arch = getenv("EG_ALLOW_ANY_ARCH") != nullptr;
  return k;
}
// Fault points: places where a TEST build can make a call fail on purpose.  The shipped library has none - the macro is the constant
// false - and holds no name of a switch; tests/faultlib builds a second library with -DEG_FAULT_POINTS_H=<its header>, which defines the macro.
#ifdef EG_FAULT_POINTS_H
#include EG_FAULT_POINTS_H
#else
#define EG_FAULT_POINT(name) false
#endif
enum { PROF_CALL = 0, PROF_MSM = 1, PROF_TABLES = 2 };
struct ProfSpan { hipEvent_t a, b; int kind; };

struct eg_ctx {
  std::recursive_mutex mu;   // serialises the C entry points of one context (and of the params created on it)
  Knobs knobs;               // the environment as eg_init found it (read_knobs)
  std::atomic<int> refs{1};  // the caller's reference + one per live params object
  int device = 0;
  hipStream_t stream = nullptr;
  int cus = 0;
  std::string name;
  uint4* tabG = nullptr;     // fixed-base comb table of the generator, EG_COMB_BITS wide (points at the first entry; comb_table_build)
  uint4* tabG_big = nullptr; // the same, big_bits wide: built when the first large batch arrives (ensure_big_tables)
  int big_bits = EG_COMB_BITS_BIG;      // 0: never build wide tables (EG_COMB_BIG_BITS)
  size_t big_min = (size_t)1 << 19;     // an engine that has verified this many items gets the wide tables (EG_COMB_BIG_MIN)
  bool big_failed = false;              // the wide tables did not fit the device memory: do not try again
  u32* gen_words = nullptr;  // generator as PT_WORDS limbs
  int resident_blocks = 0;   // blocks of the equation kernel that the chip holds at once (two per CU)
  int msm_blocks = 0;        // grid of the table / equation kernels: EG_GRID_OVERSUBSCRIBE x resident_blocks, each block striding over its share
  uint4* ws = nullptr;       // per-lane workspace of those kernels (msm_blocks * WS_QUADS * NT uint4)
  // The workspace is shared by EVERY engine of the context and by the primitive tier, in two halves (work set k of an engine uses half k;
  // a one-set call and k_prim_msm use both).  Calls are serialised on the host by `mu`, but their kernels are not: a JSON stream's
  // submissions, `_device` calls on caller streams and the work sets' own streams all run asynchronously.  Every user therefore makes its
  // stream wait for the last user of the halves it is about to write (ws_acquire) and leaves an event behind (ws_release): the GPU side
  // is ordered without any host synchronisation, whatever stream, engine or thread the previous user was.
  hipEvent_t ws_done[2] = {nullptr, nullptr};
  hipStream_t ws_last[2] = {nullptr, nullptr};
  bool ws_used[2] = {false, false};
  void* prim_scratch = nullptr;   // device scratch of the primitive tier, kept between calls and grown on demand (prim_bufs)
  size_t prim_scratch_bytes = 0;
  bool prof = false;
  bool prof_serial = false;    // eg_profile_enable(ctx, 2): chunks one after the other on one work set, so that a launch's duration is its own
  std::vector<ProfSpan> spans;
  std::vector<hipEvent_t> event_pool;
  double msm_ms = 0, all_ms = 0, tables_ms = 0;
  uint64_t msm_launches = 0, tables_launches = 0;
};

// see eg_ctx::ws_done.  halves: bit k = half k of the workspace.  Called under the context's lock.
static int ws_acquire(eg_ctx* c, unsigned halves, hipStream_t s) {
  for (int h = 0; h < 2; ++h)
    if (((halves >> h) & 1u) && c->ws_used[h] && c->ws_last[h] != s) HIPCHK(hipStreamWaitEvent(s, c->ws_done[h], 0));
  return EG_OK;
}
static int ws_release(eg_ctx* c, unsigned halves, hipStream_t s) {
  for (int h = 0; h < 2; ++h)
    if ((halves >> h) & 1u) { HIPCHK(hipEventRecord(c->ws_done[h], s)); c->ws_used[h] = true; c->ws_last[h] = s; }
  return EG_OK;
}

static int prof_begin(eg_ctx* c, hipStream_t s, int kind, size_t* idx) {
  if (!c->prof) return EG_OK;
  hipEvent_t a, b;
  for (hipEvent_t* e : {&a, &b}) {
    if (!c->event_pool.empty()) { *e = c->event_pool.back(); c->event_pool.pop_back(); }
    else HIPCHK(hipEventCreate(e));
  }
  HIPCHK(hipEventRecord(a, s));
  *idx = c->spans.size();
  c->spans.push_back({a, b, kind});
  return EG_OK;
}
static int prof_end(eg_ctx* c, hipStream_t s, size_t idx) {
  if (!c->prof) return EG_OK;
  HIPCHK(hipEventRecord(c->spans[idx].b, s));
  return EG_OK;
}

template <class T>
static int upload(T** dptr, const std::vector<T>& v, hipStream_t s) {
  *dptr = nullptr;
  const size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
  HIPCHK(hipMalloc((void**)dptr, bytes));
  if (!v.empty()) HIPCHK(hipMemcpyAsync(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
  return EG_OK;
}

// Builds the comb table of one fixed base (kernels.cuh: k_comb_window_bases + k_build_fixed_table) on stream s and waits for it
// (the scratch of the batched inversion is released before returning).  *out points at the first entry; the allocation starts
// COMB_HEADER_QUADS uint4 earlier (a whole cache line, so that the 128-byte entries stay line-aligned; the table's header is the
// last uint4 of it): release with comb_table_free.
static void comb_table_free(uint4* tab) { if (tab) (void)hipFree(tab - COMB_HEADER_QUADS); }
static int comb_table_build(const u32* d_base_words, int bits, hipStream_t s, uint4** out, bool soft_nomem = false) {
  *out = nullptr;
  if (bits < 8 || bits > 26) return fail(EG_ERR_BAD_ARG, "comb width out of range");
  const int windows = comb_windows(bits), entries = comb_entries(bits);
  const size_t lanes = (size_t)windows * (entries / COMB_RUN);
  uint4* base = nullptr;
  u32 *d_bases = nullptr, *scratch = nullptr;
  hipError_t he = hipMalloc((void**)&base, ((size_t)windows * entries * 8 + COMB_HEADER_QUADS) * sizeof(uint4));
  if (he == hipSuccess) he = hipMalloc((void**)&d_bases, (size_t)windows * PT_WORDS * sizeof(u32));
  if (he == hipSuccess) he = hipMalloc((void**)&scratch, (size_t)COMB_RUN * EG_NL * lanes * sizeof(u32));
  if (he != hipSuccess) {
    if (base) (void)hipFree(base);
    if (d_bases) (void)hipFree(d_bases);
    if (scratch) (void)hipFree(scratch);
    if (he == hipErrorOutOfMemory && soft_nomem) { (void)hipGetLastError(); return EG_ERR_NOMEM; }
    HIPCHK(he);
  }
  hipLaunchKernelGGL(k_comb_window_bases, dim3(1), dim3(64), 0, s, d_base_words, bits, d_bases);
  hipLaunchKernelGGL(k_build_fixed_table, dim3((unsigned)((lanes + NT - 1) / NT)), dim3(NT), 0, s, d_bases, bits, base + COMB_HEADER_QUADS, scratch);
  const hipError_t se = hipStreamSynchronize(s);
  (void)hipFree(d_bases); (void)hipFree(scratch);
  if (se != hipSuccess) { (void)hipFree(base); HIPCHK(se); }
  *out = base + COMB_HEADER_QUADS;
  return EG_OK;
}

// ---------------------------------------------------------------------------------------------------------------
using eghost::StageDev; using eghost::LevelDev;
using eghost::FAM_TABLE1; using eghost::FAM_TABLEN; using eghost::FAM_DIRECT1; using eghost::FAM_GENERIC; using eghost::FAM_ENCODE; using eghost::N_FAM;
using eghost::EG_MULTI_GROUP; using eghost::job_family;

struct Engine {
  eg_ctx* ctx = nullptr;
  eghost::Plan plan;
  int n_options = 0;
  // device plan
  egplan::WireItem *d_pt_items = nullptr, *d_sc_items = nullptr;
  egplan::DeriveClass* d_dclasses = nullptr;
  egplan::DeriveTerm* d_dterms = nullptr;
  egplan::JobClass* d_jobs = nullptr;
  egplan::VarTerm* d_vterms = nullptr;
  egplan::HashInst* d_insts = nullptr;
  egplan::HashOp* d_ops = nullptr;
  egplan::StatusRule* d_rules = nullptr;
  u32* d_tally_slots = nullptr;
  unsigned short* d_base_slots = nullptr;      // FlatPlan::build_slots: point slots by the stage that builds their tables
  egplan::SumBase* d_sum_bases = nullptr;      // FlatPlan::sums / sum_members (members as table slots)
  unsigned short* d_sum_members = nullptr;
  egplan::SumBase* d_acc_sums = nullptr;       // ring-group walk: FlatPlan::acc_sums / acc_members
  unsigned short* d_acc_members = nullptr;
  int n_sums = 0;
  unsigned short* d_defer_slots = nullptr;
  int max_defer = 0;
  unsigned char* d_blob = nullptr;
  uint4 *d_tabK = nullptr, *d_cpts = nullptr;   // d_tabK: comb table of the election key (first entry; see comb_table_build)
  uint4* d_tabK_big = nullptr;                 // its wide form, built with the context's (ensure_big_tables)
  bool use_big = false;                        // this call reads the wide tables
  size_t items_seen = 0;                       // items verified by this engine so far (the wide tables are built once it passes ctx->big_min)
  u32* d_prefixes = nullptr;
  u32* d_key_words = nullptr;   // [0..PT_WORDS) generator, [PT_WORDS..2 PT_WORDS) key
  std::vector<StageDev> stages;
  std::vector<LevelDev> levels;
  int prefix_inst_first = 0, prefix_inst_count = 0;
  // Chunk workspace: TWO sets, each with its own stream.  Consecutive chunks of a call alternate between the sets, so that the launch
  // tail of one chunk's kernels (the last blocks of a grid leave most CUs idle) is filled by the other chunk's kernels; the sets join
  // the caller's stream at the end of the call (fork / join with events).  Set 1 tallies into its own accumulator, which the join
  // adds to the running tally: the order of the additions is fixed by the code, not by timing.  (EG_STREAMS=1: one set, one stream.)
  struct WorkSet {
    uint4 *pts = nullptr, *cmp = nullptr, *chal = nullptr;
    u32 *states = nullptr, *flags = nullptr, *bad_item = nullptr;
    uint4 *btab = nullptr, *dpt = nullptr, *sacc = nullptr;
    u32* encw = nullptr;
    u32* partial = nullptr;
    u32* tally = nullptr;      // set 0: the engine's running tally; set 1: this call's share, added to set 0's at the join
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
  } set[2];
  int n_sets = 2;
  Knobs knobs;                 // the environment as the creation of this params object found it (read_knobs)
  uint8_t key_bytes[32] = {0}; // the election key as given (canonical encoding): two params objects belong to one election if these agree
  int teeth = 6;               // comb shape of the per-ballot tables (plan_teeth: 5 x 51 when a table serves two products, else 6 x 43)
  hipEvent_t fork = nullptr;
  u32 cap = 0, max_cap = 0;
  int tally_blocks = 64;
  u32* tally = nullptr;        // [2n][PT_WORDS] running tally (extended points) = set[0].tally
  u32* tally_saved = nullptr;  // [2n][PT_WORDS] the running tally set aside while a host call computes its per-batch tally
  u32* tally_saved2 = nullptr; // the same for the JSON entry points, which call the host form piece by piece
  u32* tally_saved3 = nullptr; // the running tally as a multi-GPU call found it (TallyRollback: put back if any slab fails)
  u32* d_tally_enc = nullptr;  // [2n][8] device staging of engine_tally_encode (no allocation per call)
  uint8_t* json_ring = nullptr;        // pinned staging of the JSON entry points: a ring of packed ballots that the parser threads fill
  size_t json_ring_bytes = 0;          // window by window while earlier windows are uploaded and verified
  u32* json_status_ring = nullptr;     // pinned verdicts of the ballots in the ring
  size_t json_ring_ballots = 0;
  hipStream_t json_ctl[2] = {nullptr, nullptr};   // control streams of consecutive windows (fork / join of a window's chunks)
  struct eg_json_stream* stream_open = nullptr;   // the JSON stream (eg_verify_json_begin ... _end) that owns this engine's work sets, ring and tally just now
  // LONG calls - the ones that do not hold the context's lock from start to end: a one-shot JSON call (its worker thread takes the lock piece
  // by piece) and a multi-GPU call (one host thread per slab) - hold this mutex for their whole length; every other entry point on the
  // params object WAITS for them (EG_WAIT_LONG).  `reserved` marks the engine as held by a multi-GPU call (set and cleared under the
  // context's lock): between the moment such a call sets the running tallies aside and its merge or roll-back, nobody else may verify, tally
  // or open a stream on the object.
  std::mutex long_call_mu;
  bool reserved = false;
  // the tail of the last asynchronous call that touched the running tally (engine_touch): what a later call on ANOTHER stream has to wait
  // for instead of draining the whole device (TallyRollback)
  hipEvent_t last_done = nullptr, saved_ev = nullptr;
  bool last_used = false;
  // staging for the host-pointer API
  hipStream_t copy_stream = nullptr;
  unsigned char* d_wire = nullptr;
  u32* d_status = nullptr;
  size_t staging_ballots = 0;
  // generators: per-lane workspace (grown on demand) and the ring shapes of the two range proofs of a QV ballot
  u32* gen_ws = nullptr;
  size_t gen_ws_bytes = 0;
  u32* d_gen_desc = nullptr;
};

static int grid_for(size_t lanes, int cap_blocks);
// blocks of a generator launch and its per-lane workspace: at most ~2 GiB of scratch, whatever the election's size
static int gen_workspace(Engine* e, size_t n, unsigned words, int* blocks_out) {
  const size_t budget = (size_t)2 << 30;
  int blocks = grid_for(n, e->ctx->cus * 4);
  const size_t per_block = (size_t)NT * words * sizeof(u32);
  if ((size_t)blocks * per_block > budget) blocks = (int)std::max<size_t>(1, budget / per_block);
  const size_t need = (size_t)blocks * per_block;
  if (need > e->gen_ws_bytes) {
    HIPCHK(hipDeviceSynchronize());     // an earlier generator launch may still use the old slice
    if (e->gen_ws) (void)hipFree(e->gen_ws);
    e->gen_ws = nullptr; e->gen_ws_bytes = 0;
    HIPCHK(hipMalloc((void**)&e->gen_ws, need));
    e->gen_ws_bytes = need;
  }
  *blocks_out = blocks;
  return EG_OK;
}

static void engine_free(Engine* e) {
  if (!e) return;
  void* ptrs[] = {e->d_pt_items, e->d_sc_items, e->d_dclasses, e->d_dterms, e->d_jobs, e->d_vterms, e->d_insts, e->d_ops,
                  e->d_rules, e->d_tally_slots, e->d_base_slots, e->d_sum_bases, e->d_sum_members, e->d_acc_sums, e->d_acc_members, e->d_defer_slots, e->d_blob, e->d_cpts, e->d_prefixes, e->d_key_words,
                  e->tally_saved, e->tally_saved2, e->tally_saved3, e->d_tally_enc, e->d_wire, e->d_status, e->gen_ws, e->d_gen_desc};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  for (auto& w : e->set) {
    void* sp[] = {w.pts, w.cmp, w.chal, w.states, w.flags, w.bad_item, w.btab, w.dpt, w.sacc, w.encw, w.partial, w.tally};
    for (void* p : sp) if (p) (void)hipFree(p);
    if (w.stream) (void)hipStreamDestroy(w.stream);
    if (w.done) (void)hipEventDestroy(w.done);
  }
  if (e->fork) (void)hipEventDestroy(e->fork);
  if (e->last_done) (void)hipEventDestroy(e->last_done);
  if (e->saved_ev) (void)hipEventDestroy(e->saved_ev);
  comb_table_free(e->d_tabK); comb_table_free(e->d_tabK_big);
  if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
  if (e->json_ring) (void)hipHostFree(e->json_ring);
  if (e->json_status_ring) (void)hipHostFree(e->json_status_ring);
  for (hipStream_t st : e->json_ctl) if (st) (void)hipStreamDestroy(st);
  delete e;
}

static EngineBufs make_bufs(const Engine* e, int set, const void* d_ballots, u32 n, void* d_status) {
  const Engine::WorkSet& w = e->set[set];
  EngineBufs B;
  B.wire = reinterpret_cast<const u32*>(d_ballots);
  B.stride_words = (u32)(e->plan.stride / 4);
  B.n = n;
  B.cap = e->cap;
  B.pts = w.pts; B.cmp = w.cmp; B.chal = w.chal; B.states = w.states; B.flags = w.flags; B.bad_item = w.bad_item;
  B.status = reinterpret_cast<u32*>(d_status);
  B.tabG = e->use_big ? e->ctx->tabG_big : e->ctx->tabG; B.tabK = e->use_big ? e->d_tabK_big : e->d_tabK; B.cpts = e->d_cpts; B.prefixes = e->d_prefixes; B.blob = e->d_blob;
  // each set works in its own part of the context's per-lane workspace (the grids of its kernels are 1 / n_sets of msm_blocks)
  B.ws = e->ctx->ws + (size_t)set * (e->ctx->msm_blocks / e->n_sets) * WS_QUADS * NT;
  B.btab = w.btab;
  B.sacc = w.sacc;
  B.dpt = w.dpt;
  B.encw = w.encw;
  return B;
}

// bytes of chunk workspace per ballot of ONE work set (the buffers engine_reserve allocates for it)
static size_t engine_bytes_per_ballot(const Engine* e) {
  const eghost::Plan& P = e->plan;
  const size_t pt = (size_t)PT_WORDS * sizeof(u32);
  return (size_t)std::max(P.n_pt_slots, 1) * pt + (size_t)std::max(P.n_cmp_slots, 1) * (32 + pt) + (size_t)std::max(P.n_chal_slots, 1) * 32 +
         (size_t)std::max(P.n_state_slots, 1) * 208 + (size_t)std::max(P.n_flag_slots, 1) * 4 + 4 +
         std::max<size_t>(P.n_tables(), 1) * btab_quads_of(e->teeth) * 16 + (size_t)std::max(e->max_defer, 1) * 2 * EG_NL * sizeof(u32) +
         (P.grouped() ? P.sum_bases.size() * (size_t)e->teeth * BTAB_ENTRY_QUADS * 16 : 0);
}

// (re)allocate the per-chunk SoA buffers of every work set for chunks of up to `want` ballots
static void engine_release_sets(Engine* e) {
  for (auto& w : e->set) {
    void** bufs[] = {(void**)&w.pts, (void**)&w.cmp, (void**)&w.chal, (void**)&w.states, (void**)&w.flags, (void**)&w.bad_item,
                     (void**)&w.btab, (void**)&w.dpt, (void**)&w.encw, (void**)&w.sacc};
    for (void** b : bufs) { if (*b) (void)hipFree(*b); *b = nullptr; }
  }
  e->cap = 0;
}
static int engine_reserve(Engine* e, u32 want) {
  if (want <= e->cap) return EG_OK;
  const eghost::Plan& P = e->plan;
  HIPCHK(hipDeviceSynchronize());   // earlier batches may still be running on a caller's stream
  engine_release_sets(e);
  const size_t cap = (want + NT - 1) / NT * NT;
  const size_t sizes[] = {(size_t)std::max(P.n_pt_slots, 1) * PT_QUADS * cap * sizeof(uint4),
                          (size_t)std::max(P.n_cmp_slots, 1) * 2 * cap * sizeof(uint4),
                          (size_t)std::max(P.n_chal_slots, 1) * 2 * cap * sizeof(uint4),
                          (size_t)std::max(P.n_state_slots, 1) * 52 * cap * sizeof(u32),
                          (size_t)std::max(P.n_flag_slots, 1) * cap * sizeof(u32),
                          cap * sizeof(u32),
                          std::max<size_t>(P.n_tables(), 1) * cap * btab_quads_of(e->teeth) * sizeof(uint4),
                          (size_t)std::max(P.n_cmp_slots, 1) * PT_QUADS * cap * sizeof(uint4),
                          (size_t)std::max(e->max_defer, 1) * 2 * EG_NL * cap * sizeof(u32),
                          P.grouped() ? std::max<size_t>(P.sum_bases.size(), 1) * cap * (size_t)e->teeth * BTAB_ENTRY_QUADS * sizeof(uint4) : (size_t)256};
  for (int k = 0; k < e->n_sets; ++k) {
    Engine::WorkSet& w = e->set[k];
    void** bufs[] = {(void**)&w.pts, (void**)&w.cmp, (void**)&w.chal, (void**)&w.states, (void**)&w.flags, (void**)&w.bad_item,
                     (void**)&w.btab, (void**)&w.dpt, (void**)&w.encw, (void**)&w.sacc};
    for (size_t i = 0; i < sizeof(sizes) / sizeof(sizes[0]); ++i) {
      const hipError_t he = hipMalloc(bufs[i], sizes[i]);
      if (he == hipErrorOutOfMemory) {        // the caller retries with smaller chunks
        (void)hipGetLastError();
        engine_release_sets(e);
        return fail(EG_ERR_NOMEM, "device memory exhausted while reserving the chunk workspace");
      }
      HIPCHK(he);
    }
  }
  e->cap = (u32)cap;
  return EG_OK;
}

static unsigned blocks_of(size_t n) { return (unsigned)std::max<size_t>(1, (n + NT - 1) / NT); }
static int grid_for(size_t lanes, int cap_blocks) {
  size_t blocks = (lanes + NT - 1) / NT;
  if (blocks < 1) blocks = 1;
  if ((size_t)cap_blocks < blocks) blocks = (size_t)cap_blocks;
  return (int)blocks;
}

// build device state for a plan + election key
static int engine_create(eg_ctx* ctx, eghost::Plan&& plan, const uint8_t pk[32], int n_options, Engine** out) {
  std::unique_ptr<Engine, void (*)(Engine*)> e(new Engine(), engine_free);
  e->ctx = ctx;
  e->plan = std::move(plan);
  e->n_options = n_options;
  eghost::Plan& P = e->plan;
  hipStream_t s = ctx->stream;
  HIPCHK(hipSetDevice(ctx->device));
  if (P.pk_off >= 0) memcpy(P.blob.data() + P.pk_off, pk, 32);

  // flatten stages / levels / programs (pure host logic: host_plan.hpp)
  const eghost::FlatPlan F = eghost::flatten_plan(P);
  {
    const std::string why = eghost::check_flat_plan(P, F);
    if (!why.empty()) return fail(EG_ERR_BAD_ARG, "internal: inconsistent verification plan: " + why);
  }
  e->stages = F.stages;
  e->levels = F.levels;
  e->max_defer = F.max_defer;
  e->prefix_inst_first = F.prefix_inst_first;
  e->prefix_inst_count = F.prefix_inst_count;
  const std::vector<egplan::JobClass>& jobs = F.jobs;
  const std::vector<egplan::HashInst>& insts = F.insts;
  const std::vector<egplan::HashOp>& ops = F.ops;
  const std::vector<uint16_t>& defer_slots = F.defer_slots;
  const std::vector<egplan::DeriveClass>& dclasses = F.dclasses;
  int rc;
  if ((rc = upload(&e->d_pt_items, P.pt_items, s))) return rc;
  if ((rc = upload(&e->d_sc_items, P.sc_items, s))) return rc;
  if ((rc = upload(&e->d_dclasses, dclasses, s))) return rc;
  if ((rc = upload(&e->d_dterms, P.dterms, s))) return rc;
  if ((rc = upload(&e->d_jobs, jobs, s))) return rc;
  if ((rc = upload(&e->d_vterms, P.vterms, s))) return rc;
  if ((rc = upload(&e->d_insts, insts, s))) return rc;
  if ((rc = upload(&e->d_ops, ops, s))) return rc;
  if ((rc = upload(&e->d_rules, P.rules, s))) return rc;
  if ((rc = upload(&e->d_tally_slots, P.tally_slots, s))) return rc;
  if ((rc = upload(&e->d_base_slots, F.build_slots, s))) return rc;
  if ((rc = upload(&e->d_sum_bases, F.sums, s))) return rc;
  if ((rc = upload(&e->d_sum_members, F.sum_members, s))) return rc;
  if ((rc = upload(&e->d_acc_sums, F.acc_sums, s))) return rc;
  if ((rc = upload(&e->d_acc_members, F.acc_members, s))) return rc;
  e->n_sums = (int)F.sums.size();
  if ((rc = upload(&e->d_defer_slots, defer_slots, s))) return rc;
  if ((rc = upload(&e->d_blob, P.blob, s))) return rc;

  // election key: decode, reject invalid / identity (keys/mod.rs:161-176), fixed-base table
  u32* d_pk = nullptr;
  u32* d_flags = nullptr;
  HIPCHK(hipMalloc((void**)&d_pk, 32));
  HIPCHK(hipMalloc((void**)&d_flags, 8));
  HIPCHK(hipMalloc((void**)&e->d_key_words, 2 * PT_WORDS * sizeof(u32)));
  HIPCHK(hipMemcpyAsync(d_pk, pk, 32, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_setup_points, dim3(1), dim3(64), 0, s, d_pk, e->d_key_words, d_flags);
  u32 hflags[2] = {0, 0};
  HIPCHK(hipMemcpyAsync(hflags, d_flags, 8, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  (void)hipFree(d_pk); (void)hipFree(d_flags);
  if (!hflags[0]) return fail(EG_ERR_BAD_PUBLIC_KEY, "public key is not a valid ristretto255 encoding");
  if (hflags[1]) return fail(EG_ERR_BAD_PUBLIC_KEY, "public key is the identity");
  if ((rc = comb_table_build(e->d_key_words + PT_WORDS, EG_COMB_BITS, s, &e->d_tabK))) return rc;

  // election-constant points [m]G
  {
    u64* d_m = nullptr;
    std::vector<uint64_t> mults = P.const_mults;
    if (mults.empty()) mults.push_back(0);
    HIPCHK(hipMalloc((void**)&d_m, mults.size() * 8));
    HIPCHK(hipMemcpyAsync(d_m, mults.data(), mults.size() * 8, hipMemcpyHostToDevice, s));
    HIPCHK(hipMalloc((void**)&e->d_cpts, mults.size() * PT_QUADS * sizeof(uint4)));
    hipLaunchKernelGGL(k_const_points, dim3((unsigned)((mults.size() + NT - 1) / NT)), dim3(NT), 0, s, d_m, (int)mults.size(),
                       ctx->tabG, e->d_cpts);
    HIPCHK(hipStreamSynchronize(s));
    (void)hipFree(d_m);
  }

  // chunk workspace: sized lazily by engine_reserve() for the batches actually seen (up to EG_CHUNK ballots per chunk and work set)
  e->teeth = eghost::plan_teeth(e->plan);
  e->knobs = read_knobs();
  memcpy(e->key_bytes, pk, 32);
  if (e->knobs.teeth) e->teeth = e->knobs.teeth;     // measurement knob
  e->n_sets = e->knobs.streams;
  // Two work sets whose kernels fill each other's launch tails (profiles/r03_ab_experiments.txt, blocks 3 and 9; M single-choice ballots/s):
  // with 6-tooth tables (57.6 KB per ballot and set) one set of 2^20 ballots 6.06, one set of 2^18 5.81 (-4 %), two sets of 2^18 6.05,
  // two sets of 2^19 6.13 (+1 %), two sets of 2^17 5.90; with the 5-tooth tables of the choice ballots (32.3 KB) two sets of 2^18 6.28,
  // of 2^19 6.35.  Default: two sets of 2^19 ballots where that stays within 40 GB (the choice ballots of up to ~6 options: 34 GB),
  // else of 2^18 (16 options: 44 GB; quadratic voting 5 / 20: 42 GB); EG_CHUNK overrides.
  {
    const size_t per = engine_bytes_per_ballot(e.get()) * (size_t)e->n_sets;
    const u32 two_sets = per * 524288u <= ((size_t)40 << 30) ? 524288u : 262144u;
    e->max_cap = e->knobs.chunk ? (u32)std::min<size_t>(e->knobs.chunk, 1u << 30) : (e->n_sets == 2 ? two_sets : 1048576u);
  }
  {
    // Large elections keep the workspace within half of the free device memory (~58 KB per ballot and set for 5 options, 150 KB for 16).
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    const size_t per = engine_bytes_per_ballot(e.get()) * (size_t)e->n_sets;
    const size_t limit = per ? free_b / 2 / per : e->max_cap;
    if (limit < e->max_cap) e->max_cap = (u32)(limit / NT * NT);
  }
  if (e->max_cap < NT) e->max_cap = NT;
  e->max_cap = (e->max_cap + NT - 1) / NT * NT;
  const size_t tally_bytes = std::max<size_t>(P.tally_slots.size(), 1) * PT_WORDS * sizeof(u32);
  for (int k = 0; k < e->n_sets; ++k) {
    Engine::WorkSet& w = e->set[k];
    HIPCHK(hipMalloc((void**)&w.partial, tally_bytes * e->tally_blocks));
    HIPCHK(hipMalloc((void**)&w.tally, tally_bytes));
    hipLaunchKernelGGL(k_tally_init, dim3(blocks_of(P.tally_slots.size())), dim3(NT), 0, s, w.tally, (int)P.tally_slots.size());
    if (e->n_sets > 1) {
      HIPCHK(hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
      HIPCHK(hipEventCreateWithFlags(&w.done, hipEventDisableTiming));
    }
  }
  if (e->n_sets > 1) HIPCHK(hipEventCreateWithFlags(&e->fork, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&e->last_done, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&e->saved_ev, hipEventDisableTiming));
  e->tally = e->set[0].tally;
  HIPCHK(hipMalloc((void**)&e->tally_saved, tally_bytes));
  HIPCHK(hipMalloc((void**)&e->tally_saved2, tally_bytes));
  HIPCHK(hipMalloc((void**)&e->tally_saved3, tally_bytes));
  HIPCHK(hipMalloc((void**)&e->d_tally_enc, std::max<size_t>(P.tally_slots.size(), 1) * 32));
  HIPCHK(hipMalloc((void**)&e->d_prefixes, (size_t)std::max(P.n_prefixes, 1) * 52 * sizeof(u32)));

  // hoisted transcript prefixes: run the prefix programs once (a single lane each)
  if (e->prefix_inst_count) {
    EngineBufs B = make_bufs(e.get(), 0, nullptr, 1, nullptr);
    hipLaunchKernelGGL(k_hash, dim3(grid_for(e->prefix_inst_count, 1 << 20)), dim3(NT), 0, s, B, e->d_insts, e->d_ops,
                       e->prefix_inst_first, e->prefix_inst_count);
  }
  HIPCHK(hipStreamSynchronize(s));
  HIPCHK(hipGetLastError());
  *out = e.release();
  ctx->refs.fetch_add(1);   // dropped by params_destroy
  return EG_OK;
}

// Wide comb tables for G (per context) and K (per engine): built when an engine has seen ctx->big_min items (in one batch or over
// many calls: a host batch arrives in pieces, JSON text window by window), kept for the life of the objects.  A table that does not fit the device memory is not an error: the narrow tables stay in use.
static int ensure_big_tables(Engine* e, hipStream_t s) {
  eg_ctx* ctx = e->ctx;
  if (ctx->big_bits == 0 || ctx->big_failed) return EG_OK;
  if (!ctx->tabG_big) {
    const int rc = comb_table_build(ctx->gen_words, ctx->big_bits, s, &ctx->tabG_big, true);
    if (rc == EG_ERR_NOMEM) { ctx->big_failed = true; return EG_OK; }
    if (rc) return rc;
  }
  if (!e->d_tabK_big) {
    const int rc = comb_table_build(e->d_key_words + PT_WORDS, ctx->big_bits, s, &e->d_tabK_big, true);
    if (rc == EG_ERR_NOMEM) { ctx->big_failed = true; return EG_OK; }
    if (rc) return rc;
  }
  return EG_OK;
}

// the kernels that touch the per-ballot comb tables exist once per comb shape (template parameter T, ge25519.cuh: Teeth<T>)
#ifdef EG_TEETH7      // measurement builds only (tools/build_variant.sh teeth7 -DEG_TEETH7): the 7 x 37 comb (64 entries, 8 KiB per table) as a third shape
#define EG_WITH_TEETH(teeth, ...) do { if ((teeth) == 5) { constexpr int T = 5; __VA_ARGS__; } else if ((teeth) == 7) { constexpr int T = 7; __VA_ARGS__; } else { constexpr int T = 6; __VA_ARGS__; } } while (0)
#else
#define EG_WITH_TEETH(teeth, ...) do { if ((teeth) == 5) { constexpr int T = 5; __VA_ARGS__; } else { constexpr int T = 6; __VA_ARGS__; } } while (0)
#endif

// marks the end of an asynchronous call that touched the engine's running tally on stream s (Engine::last_done)
static int engine_touch(Engine* e, hipStream_t s) {
  HIPCHK(hipEventRecord(e->last_done, s));
  e->last_used = true;
  return EG_OK;
}
// verify n ballots (device pointers), accumulating accepted ciphertexts into the running tally
// flags (the streaming JSON entry points overlap consecutive calls on different control streams `s`):
//   VD_FORCE_SETS     always run on the work sets' own streams (never on `s` with set 0's buffers), whatever the size of the batch;
//   VD_KEEP_SET_TALLY leave set 1's share of the tally in its accumulator at the join (the caller merges once, when everything has landed).
enum { VD_FORCE_SETS = 1, VD_KEEP_SET_TALLY = 2 };
static int engine_verify_device(Engine* e, size_t n, const void* d_ballots, void* d_status, hipStream_t s, int flags = 0) {
  eg_ctx* ctx = e->ctx;     // s may be the null stream: a NULL hipStream_t means what it means everywhere in HIP
  const eghost::Plan& P = e->plan;
  size_t all_idx = 0;
  int rc;
  e->items_seen += n;
  if (e->items_seen >= ctx->big_min && (rc = ensure_big_tables(e, s))) return rc;
  e->use_big = e->d_tabK_big != nullptr && ctx->tabG_big != nullptr;
  if ((rc = prof_begin(ctx, s, PROF_CALL, &all_idx))) return rc;
  // equal-sized chunks (each a multiple of the block size) so that the persistent grids stay balanced on the last chunk; with two
  // work sets a batch that is worth splitting gets an even number of chunks, so that both streams carry the same load
  const bool two = e->n_sets == 2 && ((flags & VD_FORCE_SETS) || (n >= (size_t)ctx->resident_blocks * NT / 2 && !ctx->prof_serial));
  auto chunks_for = [&](size_t cap) {
    size_t k = (n + cap - 1) / cap;
    if (two) k = std::max<size_t>(2, (k + 1) / 2 * 2);
    return k;
  };
  size_t n_chunks = chunks_for(e->max_cap);
  while (n) {
    const int rr = engine_reserve(e, (u32)(((n + n_chunks - 1) / n_chunks + NT - 1) / NT * NT));
    if (rr == EG_OK) break;
    if (rr != EG_ERR_NOMEM || e->max_cap <= 16 * NT) return rr;
    e->max_cap = (e->max_cap / 2 + NT - 1) / NT * NT;    // other allocations took the memory this engine counted on: halve the chunks
    n_chunks = chunks_for(e->max_cap);
  }
  const size_t even = n_chunks ? ((n + n_chunks - 1) / n_chunks + NT - 1) / NT * NT : 0;
  // join: the caller's stream continues after both sets; set 1's share of the tally moves into the running tally.  It runs on EVERY
  // way out once the sets have been forked (the scope guard below): an early return between fork and join (a failed launch, event or
  // allocation) used to leave the caller's stream ahead of work still queued on the sets' streams - the caller would then free or
  // re-use buffers those kernels still write, and set 1's accumulator would carry a stale share into the next call.
  bool forked = false, joined = false;
  auto join = [&]() -> hipError_t {
    joined = true;
    hipError_t first = hipSuccess;
    for (int k = 0; k < 2; ++k) {
      if (ws_release(ctx, 1u << k, e->set[k].stream) != EG_OK && first == hipSuccess) first = hipErrorUnknown;   // the next user of this half waits for it
      hipError_t he = hipEventRecord(e->set[k].done, e->set[k].stream);
      if (he == hipSuccess) he = hipStreamWaitEvent(s, e->set[k].done, 0);
      if (he != hipSuccess && first == hipSuccess) first = he;
    }
    const int ns = (int)P.tally_slots.size();
    if (ns && !(flags & VD_KEEP_SET_TALLY)) {
      hipLaunchKernelGGL(k_tally_add_points, dim3(blocks_of((size_t)ns)), dim3(NT), 0, s, e->set[1].tally, ns, e->set[0].tally);
      hipLaunchKernelGGL(k_tally_init, dim3(blocks_of((size_t)ns)), dim3(NT), 0, s, e->set[1].tally, ns);
    }
    return first;
  };
  ScopeExit rejoin{[&]() {
    if (!forked || joined) return;
    if (join() != hipSuccess) (void)hipDeviceSynchronize();     // the streams could not even be tied together: drain the device instead
  }};
  bool whole_ws = false;       // a one-set call holds both halves of the context's per-lane workspace on `s` until it returns
  ScopeExit release_ws{[&]() { if (whole_ws) (void)ws_release(ctx, 3u, s); }};
  if (two) {                 // fork: both work sets start after whatever the caller's stream holds so far
    HIPCHK(hipEventRecord(e->fork, s));
    forked = true;
    for (int k = 0; k < 2; ++k) {
      HIPCHK(hipStreamWaitEvent(e->set[k].stream, e->fork, 0));
      TRY_(ws_acquire(ctx, 1u << k, e->set[k].stream));      // ... and after the last user of its half of the workspace, whoever that was
    }
  } else if (n) {
    TRY_(ws_acquire(ctx, 3u, s));
    whole_ws = true;
  }
  const bool inject_failure = two && EG_FAULT_POINT(after_fork);     // constant false in the shipped library (see EG_FAULT_POINT above)
  const int msm_blocks = ctx->msm_blocks / (two ? 2 : 1);
  size_t chunk = 0;
  for (size_t off = 0; off < n; off += even, ++chunk) {
    const int set = two ? (int)(chunk & 1) : 0;
    hipStream_t cs = two ? e->set[set].stream : s;
    const Engine::WorkSet& w = e->set[set];
    const u32 cn = (u32)std::min<size_t>(even, n - off);
    EngineBufs B = make_bufs(e, set, reinterpret_cast<const unsigned char*>(d_ballots) + off * P.stride, cn,
                             reinterpret_cast<u32*>(d_status) + off);
    if (!two) B.ws = ctx->ws;
    const int wide = ctx->cus * 32;     // grid-stride kernels that need no per-lane workspace: well oversubscribed (see eg_init)
    HIPCHK(hipMemsetAsync(w.bad_item, 0xff, (size_t)cn * sizeof(u32), cs));
    hipLaunchKernelGGL(k_decode_points, dim3(grid_for((size_t)P.pt_items.size() * cn, wide)), dim3(NT), 0, cs, B, e->d_pt_items,
                       (int)P.pt_items.size());
    hipLaunchKernelGGL(k_check_scalars, dim3(grid_for((size_t)P.sc_items.size() * cn, wide)), dim3(NT), 0, cs, B, e->d_sc_items,
                       (int)P.sc_items.size());
    for (auto& lv : e->levels)
      if (lv.count)
        hipLaunchKernelGGL(k_derive_points, dim3(grid_for((size_t)lv.count * cn, wide)), dim3(NT), 0, cs, B, e->d_dclasses,
                           e->d_dterms, lv.first, lv.count);
    for (auto& st : e->stages) {
      // the comb tables this stage starts with (host_plan.hpp: Stage): every table of the ballot before stage 0, or - ring-group walk -
      // the tables of one group of rings in the table slots the previous group has finished with
      if (st.build_count) {
        size_t pi = 0;
        if ((rc = prof_begin(ctx, cs, PROF_TABLES, &pi))) return rc;
        EG_WITH_TEETH(e->teeth, hipLaunchKernelGGL(k_base_tables<T>, dim3(grid_for((size_t)st.build_count * cn, msm_blocks)), dim3(NT), 0, cs, B,
                                                   e->d_base_slots + st.build_first, st.build_count));
        if ((rc = prof_end(ctx, cs, pi))) return rc;
      }
      if (st.sums_direct)
        EG_WITH_TEETH(e->teeth, hipLaunchKernelGGL(k_sum_tables<T>, dim3(grid_for((size_t)e->n_sums * cn, msm_blocks)), dim3(NT), 0, cs, B,
                                                   e->d_sum_bases, e->d_sum_members, e->n_sums));
      if (st.acc_count)
        EG_WITH_TEETH(e->teeth, hipLaunchKernelGGL(k_sum_accumulate<T>, dim3(grid_for((size_t)st.acc_count * T * cn, msm_blocks)), dim3(NT), 0, cs, B,
                                                   e->d_acc_sums + st.acc_first, e->d_acc_members, st.acc_count));
      if (st.sum_finish)
        EG_WITH_TEETH(e->teeth, hipLaunchKernelGGL(k_sum_finish<T>, dim3(grid_for((size_t)e->n_sums * cn, msm_blocks)), dim3(NT), 0, cs, B,
                                                   e->d_sum_bases, e->n_sums));
      if (st.fam_count[FAM_TABLE1]) {
        size_t pi = 0;
        if ((rc = prof_begin(ctx, cs, PROF_MSM, &pi))) return rc;
        EG_WITH_TEETH(e->teeth, hipLaunchKernelGGL((k_eq_table<false, T>), dim3(grid_for((size_t)st.fam_count[FAM_TABLE1] * cn, msm_blocks)), dim3(NT), 0, cs,
                                                   B, e->d_jobs, e->d_vterms, st.fam_first[FAM_TABLE1], st.fam_count[FAM_TABLE1], 1));
        if ((rc = prof_end(ctx, cs, pi))) return rc;
      }
      if (st.fam_count[FAM_TABLEN]) {
        const int group = std::min(st.max_terms, EG_MULTI_GROUP);
        EG_WITH_TEETH(e->teeth, hipLaunchKernelGGL((k_eq_table<true, T>), dim3(grid_for((size_t)st.fam_count[FAM_TABLEN] * cn, msm_blocks)), dim3(NT),
                                                   (size_t)group * 9 * NT * sizeof(u32), cs, B, e->d_jobs, e->d_vterms, st.fam_first[FAM_TABLEN],
                                                   st.fam_count[FAM_TABLEN], group));
      }
      if (st.fam_count[FAM_DIRECT1])
        hipLaunchKernelGGL(k_eq_direct, dim3(grid_for((size_t)st.fam_count[FAM_DIRECT1] * cn, msm_blocks)), dim3(NT), 0, cs, B,
                           e->d_jobs, e->d_vterms, st.fam_first[FAM_DIRECT1], st.fam_count[FAM_DIRECT1]);
      if (st.fam_count[FAM_GENERIC])
        EG_WITH_TEETH(e->teeth, hipLaunchKernelGGL(k_eq_generic<T>, dim3(grid_for((size_t)st.fam_count[FAM_GENERIC] * cn, msm_blocks)), dim3(NT), 0, cs, B,
                           e->d_jobs, e->d_vterms, st.fam_first[FAM_GENERIC], st.fam_count[FAM_GENERIC]));
      if (st.fam_count[FAM_ENCODE])
        hipLaunchKernelGGL(k_encode_plain, dim3(grid_for((size_t)st.fam_count[FAM_ENCODE] * cn, wide)), dim3(NT), 0, cs, B, e->d_jobs,
                           st.fam_first[FAM_ENCODE], st.fam_count[FAM_ENCODE]);
      for (int d0 = 0; d0 < st.defer_count; d0 += 32)   // one batched inversion per ballot and group of <= 32 commitments
        hipLaunchKernelGGL(k_encode_batch, dim3(grid_for(cn, wide)), dim3(NT), 0, cs, B, e->d_defer_slots + st.defer_first + d0,
                           std::min(32, st.defer_count - d0));
      if (st.inst_count)
        hipLaunchKernelGGL(k_hash, dim3(grid_for((size_t)st.inst_count * cn, 1 << 30)), dim3(NT), 0, cs, B, e->d_insts, e->d_ops,
                           st.inst_first, st.inst_count);
    }
    hipLaunchKernelGGL(k_status, dim3((cn + NT - 1) / NT), dim3(NT), 0, cs, B, e->d_rules, (int)P.rules.size());
    if (inject_failure && chunk == 0) return fail(EG_ERR_HIP, "injected failure between fork and join (fault point of a test build)");
    if (P.tally_slots.empty()) continue;
    const int G = std::min<int>(e->tally_blocks, (int)((cn + NT - 1) / NT));
    hipLaunchKernelGGL(k_tally_partial, dim3(G, (unsigned)P.tally_slots.size()), dim3(NT), 0, cs, B, e->d_tally_slots, w.partial);
    hipLaunchKernelGGL(k_tally_final, dim3((unsigned)P.tally_slots.size()), dim3(NT), 0, cs, w.partial, G, w.tally);
  }
  if (two) HIPCHK(join());
  if ((rc = prof_end(ctx, s, all_idx))) return rc;
  if ((rc = engine_touch(e, s))) return rc;
  HIPCHK(hipGetLastError());
  return EG_OK;
}

// encodes a tally held in device memory (the running tally, or a snapshot of it) into host bytes
static int engine_tally_encode_from(Engine* e, const u32* d_tally, uint8_t* out, bool drain_device = true) {
  hipStream_t s = e->ctx->stream;
  if (drain_device) HIPCHK(hipDeviceSynchronize());   // batches enqueued on caller streams by the _device entry points must have landed
  else if (e->last_used) HIPCHK(hipStreamWaitEvent(s, e->last_done, 0));      // (a caller that knows its streams have been waited for: only the engine's last call)
  const int ns = (int)e->plan.tally_slots.size();
  if (!ns) return EG_OK;
  hipLaunchKernelGGL(k_tally_encode, dim3(blocks_of((size_t)ns)), dim3(NT), 0, s, d_tally, ns, e->d_tally_enc);
  HIPCHK(hipMemcpyAsync(out, e->d_tally_enc, (size_t)ns * 32, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return EG_OK;
}
static int engine_tally_encode(Engine* e, uint8_t* out) { return engine_tally_encode_from(e, e->tally, out); }

// running tally += the points encoded in `in` (n_slots x 32 bytes): checkpoint / resume and merging of earlier batches
static int engine_tally_add(Engine* e, const uint8_t* in) {
  hipStream_t s = e->ctx->stream;
  HIPCHK(hipDeviceSynchronize());
  const int ns = (int)e->plan.tally_slots.size();
  u32* d_in = nullptr;
  HIPCHK(hipMalloc((void**)&d_in, (size_t)ns * 32 + 4));
  u32* d_bad = d_in + (size_t)ns * 8;
  HIPCHK(hipMemsetAsync(d_bad, 0, 4, s));
  HIPCHK(hipMemcpyAsync(d_in, in, (size_t)ns * 32, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_tally_add_encoded, dim3(blocks_of((size_t)ns)), dim3(NT), 0, s, d_in, ns, e->tally, d_bad);
  u32 bad = 0;
  HIPCHK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  (void)hipFree(d_in);
  return bad ? fail(EG_ERR_BAD_ARG, "tally contains an invalid ristretto255 encoding") : EG_OK;
}

// device-side staging of the host forms: room for n packed ballots and their verdicts
static int engine_stage_reserve(Engine* e, size_t n) {
  if (n <= e->staging_ballots) return EG_OK;
  if (e->d_wire) (void)hipFree(e->d_wire);
  if (e->d_status) (void)hipFree(e->d_status);
  e->d_wire = nullptr; e->d_status = nullptr; e->staging_ballots = 0;
  HIPCHK(hipMalloc((void**)&e->d_wire, std::max<size_t>(n, 1) * e->plan.stride));
  HIPCHK(hipMalloc((void**)&e->d_status, std::max<size_t>(n, 1) * sizeof(u32)));
  e->staging_ballots = n;
  return EG_OK;
}

static int engine_verify_host(Engine* e, size_t n, const uint8_t* ballots, uint32_t* status, uint8_t* tally_out) {
  hipStream_t s = e->ctx->stream;
  HIPCHK(hipSetDevice(e->ctx->device));
  // The host form runs on the context's own stream but shares the engine's workspaces and running tally with whatever
  // `_device` calls enqueued on caller streams: wait for them, like the other host forms do.
  HIPCHK(hipDeviceSynchronize());
  const int ns = (int)e->plan.tally_slots.size();
  { const int rc = engine_stage_reserve(e, n); if (rc) return rc; }
  // fallible set-up comes BEFORE the running tally is set aside, and a scope guard puts it back on EVERY exit path (round 2 merged it
  // only on the explicit error paths: an early HIPCHK return in between lost the running tally)
  if (n && !e->copy_stream) HIPCHK(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
  // tally_out is the tally of THIS batch; the running tally keeps accumulating across calls (only eg_*_tally_reset clears
  // it).  The running tally is set aside, the batch is tallied from the identity, and the two are merged afterwards.
  bool set_aside = false;
  ScopeExit restore{[&]() {   // running tally = saved + this batch
    if (!set_aside) return;
    hipLaunchKernelGGL(k_tally_add_points, dim3(blocks_of((size_t)ns)), dim3(NT), 0, s, e->tally_saved, ns, e->tally);
    (void)hipStreamSynchronize(s);
  }};
  if (tally_out && ns) {
    HIPCHK(hipMemcpyAsync(e->tally_saved, e->tally, (size_t)ns * PT_WORDS * sizeof(u32), hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(k_tally_init, dim3(blocks_of((size_t)ns)), dim3(NT), 0, s, e->tally, ns);
    set_aside = true;
  }
  if (n && e->items_seen + n >= e->ctx->big_min) {     // the whole batch counts: its first piece already reads the wide comb tables
    const int rc = ensure_big_tables(e, s);
    if (rc) return rc;
  }
  if (n) {
    // Pipeline: the copy stream uploads piece k+1 while piece k is verified (host staging, not the kernels, is
    // the scaling risk when ballots arrive in host memory).  The first piece is small so that the exposed upload is short;
    // the rest are as large as the engine's chunks (large chunks waste less of each kernel's last round).
    std::vector<std::pair<size_t, size_t>> pieces;   // (offset, count)
    {
      const size_t lanes = (size_t)e->ctx->resident_blocks * NT;
      size_t off = 0;
      if (n >= lanes) {       // (a quarter of a mid-sized batch: 262 144 ballots +5 %, 131 072 +2.5 % over uploading them whole; below that the split costs more)
        const size_t f = std::min(lanes, (n / 4 + NT - 1) / NT * NT);
        pieces.push_back({0, f}); off = f;
      }
      const size_t piece_cap = (size_t)e->max_cap * e->n_sets;      // one piece = one chunk per work set
      const size_t rem = n - off, k = (rem + piece_cap - 1) / piece_cap;
      const size_t even = ((rem + k - 1) / k + NT - 1) / NT * NT;
      for (; off < n; off += even) pieces.push_back({off, std::min(even, n - off)});
    }
    std::vector<hipEvent_t> uploaded(pieces.size(), nullptr);
    size_t largest = 0;
    for (auto& pc : pieces) largest = std::max(largest, pc.second);
    int rc = engine_reserve(e, (u32)std::min<size_t>((largest + e->n_sets - 1) / e->n_sets + NT, e->max_cap));   // no regrowth of the workspace mid-pipeline
    if (rc == EG_ERR_NOMEM) rc = EG_OK;                // engine_verify_device falls back to smaller chunks
    for (size_t k = 0; k < pieces.size() && rc == EG_OK; ++k) {
      const size_t off = pieces[k].first, m = pieces[k].second;
      hipError_t he = hipEventCreateWithFlags(&uploaded[k], hipEventDisableTiming);
      if (he == hipSuccess) he = hipMemcpyAsync(e->d_wire + off * e->plan.stride, ballots + off * e->plan.stride, m * e->plan.stride,
                                                hipMemcpyHostToDevice, e->copy_stream);
      if (he == hipSuccess) he = hipEventRecord(uploaded[k], e->copy_stream);
      if (he == hipSuccess) he = hipStreamWaitEvent(s, uploaded[k], 0);
      if (he != hipSuccess) { rc = fail(EG_ERR_HIP, std::string("host upload: ") + hipGetErrorString(he)); break; }
      rc = engine_verify_device(e, m, e->d_wire + off * e->plan.stride, e->d_status + off, s);
    }
    // one download at the end: a device-to-pageable copy would stall the host (and the next upload) behind chunk k
    if (rc == EG_OK && hipMemcpyAsync(status, e->d_status, n * sizeof(u32), hipMemcpyDeviceToHost, s) != hipSuccess)
      rc = fail(EG_ERR_HIP, "status download failed");
    (void)hipStreamSynchronize(e->copy_stream);
    (void)hipStreamSynchronize(s);
    for (hipEvent_t ev : uploaded) if (ev) (void)hipEventDestroy(ev);
    if (rc) return rc;
  }
  HIPCHK(hipStreamSynchronize(s));
  if (tally_out) return engine_tally_encode(e, tally_out);      // (the guard merges the running tally back after the encoding)
  return EG_OK;
}

// ---------------------------------------------------------------------------------------------------------------
struct eg_choice_params { Engine* eng; int n_options; int single; };
struct eg_qv_params { Engine* eng; int n_options; uint64_t credits; eghost::QvShape shape; };
struct eg_proof_params { Engine* eng; int kind; size_t item_size; };
// between eg_verify_*_json_begin and eg_verify_json_end / _abort the params object belongs to the stream: its work sets, staging ring and
// running tally are in use, so every other verify / tally call on it is refused
static int refuse_if_streaming(const Engine* e) {
  return e->stream_open ? fail(EG_ERR_BAD_ARG, "a JSON stream is open on this params object (eg_verify_json_end or _abort it first)") : EG_OK;
}
void eg_verify_json_abort(struct eg_json_stream* S);
static bool stream_is_one_shot(const struct eg_json_stream* S);
// A LONG call (Engine::long_call_mu) - a one-shot JSON call (eg_verify_*_json: its worker thread takes the context's lock piece by piece) or a
// multi-GPU call (eg_verify_*_batch_multi*: one host thread per slab) - does not hold the context's lock for its length, but it is still
// ONE call on the params object as far as other threads are concerned: their calls on that object wait for it, as they would for any other
// entry point (the long call holds the engine's long_call_mu from start to end; the waiter drops the context's lock, queues on that mutex,
// takes the lock again and looks again).  An explicitly opened stream is the caller's own doing: calls are refused.
// (lk_ is the unique_lock of EG_LOCK / EG_LOCK_P in the calling entry point.)
#define EG_WAIT_LONG_ONLY(e)                                                                     \
  do {                                                                                           \
    while (((e)->stream_open && stream_is_one_shot((e)->stream_open)) || (e)->reserved) {        \
      lk_.unlock();                                                                              \
      { std::lock_guard<std::mutex> wait_((e)->long_call_mu); }                                  \
      lk_.lock();                                                                                \
    }                                                                                            \
  } while (0)
#define EG_WAIT_JSON(e)                                                                          \
  do {                                                                                           \
    EG_WAIT_LONG_ONLY(e);                                                                        \
    TRY(refuse_if_streaming(e));                                                                 \
  } while (0)

// The context is reference counted: the caller holds one reference (dropped by eg_destroy) and every params object
// created on it holds another, so params may be destroyed after the context they were created on.
static void ctx_release(eg_ctx* c) {
  if (!c || c->refs.fetch_sub(1) != 1) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (auto& sp : c->spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
  for (auto& ev : c->event_pool) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : c->ws_done) if (ev) (void)hipEventDestroy(ev);
  comb_table_free(c->tabG); comb_table_free(c->tabG_big);
  if (c->gen_words) (void)hipFree(c->gen_words);
  if (c->ws) (void)hipFree(c->ws);
  if (c->prim_scratch) (void)hipFree(c->prim_scratch);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}
// destroys a params object: its engine under the context lock, then its reference on the context
template <class Params>
static void params_destroy(Params* p) {
  if (!p) return;
  eg_ctx* c = p->eng->ctx;
  if (p->eng->stream_open) eg_verify_json_abort(p->eng->stream_open);      // a stream left open dies with its params object (its worker thread
                                                                             // takes the context's lock: joined before we take it)
  {
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    (void)hipSetDevice(c->device);
    engine_free(p->eng);
    delete p;
  }
  ctx_release(c);
}


// every entry point runs under its context's lock AND on its context's device (hipSetDevice is per thread and cheap): a process that holds
// contexts on several GPUs may call any function of any of them from any thread
#define EG_LOCK(c) std::unique_lock<std::recursive_mutex> lk_; if (c) { lk_ = std::unique_lock<std::recursive_mutex>((c)->mu); (void)hipSetDevice((c)->device); }
#define EG_LOCK_P(p) EG_LOCK((p) ? (p)->eng->ctx : (eg_ctx*)nullptr)

extern "C" {

const char* eg_last_error(void) { return g_err.c_str(); }
int eg_abi_version(void) { return EG_ABI_VERSION; }

int eg_init(int device, eg_ctx** out) {
  if (!out) return fail(EG_ERR_BAD_ARG, "out is null");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(EG_ERR_NO_DEVICE, "no HIP device visible");
  if (device < 0 || device >= count) return fail(EG_ERR_BAD_ARG, "device index out of range");
  HIPCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  std::unique_ptr<eg_ctx> c(new eg_ctx());
  c->device = device;
  c->cus = prop.multiProcessorCount;
  c->name = std::string(prop.name) + " (" + prop.gcnArchName + ")";
  c->knobs = read_knobs();
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos && !c->knobs.allow_any_arch)
    return fail(EG_ERR_NO_DEVICE, "device is " + c->name + ", this library is built for gfx950 (MI355X) only");
  HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HIPCHK(hipMalloc((void**)&c->gen_words, 2 * PT_WORDS * sizeof(u32)));
  hipLaunchKernelGGL(k_setup_points, dim3(1), dim3(64), 0, c->stream, (const u32*)nullptr, c->gen_words, (u32*)nullptr);
  {
    const int rc = comb_table_build(c->gen_words, EG_COMB_BITS, c->stream, &c->tabG);
    if (rc) return rc;
  }
  if (c->knobs.comb_big_bits >= 0) c->big_bits = c->knobs.comb_big_bits;
  if (c->knobs.comb_big_min >= 0) c->big_min = (size_t)c->knobs.comb_big_min;
  if (c->big_bits != 0 && (c->big_bits <= EG_COMB_BITS || c->big_bits > 26)) return fail(EG_ERR_BAD_ARG, "EG_COMB_BIG_BITS must be 0 or in (EG_COMB_BITS, 26]");
  int per_cu = 0;
  HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_eq_table<false, 6>, NT, 0));
  // the shared-chain kernel keeps up to EG_MULTI_GROUP sign vectors per lane in dynamic LDS (9 KiB per term and block)
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_eq_table<true, 5>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             EG_MULTI_GROUP * 9 * NT * (int)sizeof(u32)));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_eq_table<true, 6>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             EG_MULTI_GROUP * 9 * NT * (int)sizeof(u32)));
#ifdef EG_TEETH7
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_eq_table<true, 7>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             EG_MULTI_GROUP * 9 * NT * (int)sizeof(u32)));
#endif
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_prim_msm<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             MSM_CHUNK * 8 * NT * (int)sizeof(u32)));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_prim_msm<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             MSM_CHUNK * 8 * NT * (int)sizeof(u32)));
  if (per_cu < 1) per_cu = 1;
  c->resident_blocks = per_cu * c->cus;
  // A grid of exactly the resident blocks makes every block do the same number of rounds, so the slowest CU sets the time and the
  // last round runs part-empty; 32 blocks per CU, handed out as others finish, measured +1.5 % (single) / +1.8 % (QV) in one
  // call in round 2 (2 -> 4 -> 8 -> 16 -> 32 -> 64 blocks per CU: 5.33 / 5.38 / 5.41 / 5.39 / 5.40 / 5.44 M ballots/s).  The price is the
  // per-lane workspace: 2.4 GB.
  per_cu = c->knobs.msm_blocks_per_cu;      // 32
  c->msm_blocks = per_cu * c->cus;
  HIPCHK(hipMalloc((void**)&c->ws, (size_t)c->msm_blocks * WS_QUADS * NT * sizeof(uint4)));
  for (hipEvent_t& ev : c->ws_done) HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipGetLastError());
  *out = c.release();
  return EG_OK;
}

void eg_destroy(eg_ctx* c) { ctx_release(c); }

int eg_device_name(eg_ctx* c, char* buf, size_t cap) { EG_LOCK(c);
  if (!c || !buf || !cap) return fail(EG_ERR_BAD_ARG, "bad argument");
  snprintf(buf, cap, "%s, %d CUs, msm grid %d blocks", c->name.c_str(), c->cus, c->msm_blocks);
  return EG_OK;
}

int eg_synchronize(eg_ctx* c) { EG_LOCK(c);
  if (!c) return fail(EG_ERR_BAD_ARG, "ctx is null");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipDeviceSynchronize());      // _device / _async work may sit on caller streams, not only on the context's own
  return EG_OK;
}

int eg_profile_enable(eg_ctx* c, int enable) { EG_LOCK(c);
  if (!c) return fail(EG_ERR_BAD_ARG, "ctx is null");
  c->prof = enable != 0;
  c->prof_serial = enable == 2;
  return EG_OK;
}

int eg_profile_read(eg_ctx* c, double* msm_ms_total, uint64_t* msm_launches, double* all_ms_total) { EG_LOCK(c);
  if (!c) return fail(EG_ERR_BAD_ARG, "ctx is null");
  HIPCHK(hipDeviceSynchronize());
  for (auto& sp : c->spans) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, sp.a, sp.b));
    if (sp.kind == PROF_MSM) { c->msm_ms += ms; c->msm_launches++; }
    else if (sp.kind == PROF_TABLES) { c->tables_ms += ms; c->tables_launches++; }
    else c->all_ms += ms;
    c->event_pool.push_back(sp.a);
    c->event_pool.push_back(sp.b);
  }
  c->spans.clear();
  if (msm_ms_total) *msm_ms_total = c->msm_ms;
  if (msm_launches) *msm_launches = c->msm_launches;
  if (all_ms_total) *all_ms_total = c->all_ms;
  c->msm_ms = 0; c->all_ms = 0; c->msm_launches = 0;
  return EG_OK;
}
int eg_profile_read_tables(eg_ctx* c, double* tables_ms_total, uint64_t* tables_launches) { EG_LOCK(c);
  if (!c) return fail(EG_ERR_BAD_ARG, "ctx is null");
  if (tables_ms_total) *tables_ms_total = c->tables_ms;
  if (tables_launches) *tables_launches = c->tables_launches;
  c->tables_ms = 0; c->tables_launches = 0;
  return EG_OK;
}

int eg_selfcheck_generator_table(eg_ctx* c, int wide, size_t samples, uint64_t seed, uint64_t* mismatches) { EG_LOCK(c);
  if (!c || !mismatches) return fail(EG_ERR_BAD_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  if (wide && !c->tabG_big) {
    if (c->big_bits == 0) return fail(EG_ERR_BAD_ARG, "wide comb tables are switched off (EG_COMB_BIG_BITS=0)");
    const int rc = comb_table_build(c->gen_words, c->big_bits, s, &c->tabG_big, true);
    if (rc == EG_ERR_NOMEM) return fail(EG_ERR_NOMEM, "the wide comb table does not fit the device memory");
    if (rc) return rc;
  }
  unsigned long long* d_bad = nullptr;
  HIPCHK(hipMalloc((void**)&d_bad, sizeof(unsigned long long)));
  HIPCHK(hipMemsetAsync(d_bad, 0, sizeof(unsigned long long), s));
  if (samples)
    hipLaunchKernelGGL(k_check_fixed_table, dim3(blocks_of(samples)), dim3(NT), 0, s, c->gen_words, wide ? c->tabG_big : c->tabG, samples,
                       (u64)seed, d_bad);
  unsigned long long bad = 0;
  const hipError_t he = hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, s);
  const hipError_t se = hipStreamSynchronize(s);
  (void)hipFree(d_bad);
  HIPCHK(he); HIPCHK(se);
  *mismatches = (uint64_t)bad;
  return EG_OK;
}

int eg_comb_table_bits(eg_ctx* c, int* narrow_bits, int* wide_bits) { EG_LOCK(c);
  if (!c) return fail(EG_ERR_BAD_ARG, "ctx is null");
  if (narrow_bits) *narrow_bits = EG_COMB_BITS;
  if (wide_bits) *wide_bits = c->tabG_big ? c->big_bits : 0;
  return EG_OK;
}

// ---- primitive tier ---------------------------------------------------------------------------------------------
struct DevBuf {      // a device buffer owned for the length of one call (batch-tier helpers)
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int alloc(size_t bytes) { HIPCHK(hipMalloc(&p, std::max<size_t>(bytes, 16))); return EG_OK; }
  int put(const void* src, size_t bytes, hipStream_t s) { if (bytes) HIPCHK(hipMemcpyAsync(p, src, bytes, hipMemcpyHostToDevice, s)); return EG_OK; }
  int get(void* dst, size_t bytes, hipStream_t s) { if (bytes) HIPCHK(hipMemcpyAsync(dst, p, bytes, hipMemcpyDeviceToHost, s)); return EG_OK; }
};
#define TRY(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)
// The primitive tier works out of ONE device scratch area per context, kept between calls (round 2 paid three to five hipMalloc /
// hipFree per call).  Every primitive call runs under the context lock and ends with a stream synchronisation, so the area is free
// again when the next call starts; it only ever grows.
static int prim_bufs(eg_ctx* c, std::initializer_list<size_t> sizes, std::initializer_list<void**> ptrs) {
  size_t total = 0;
  for (size_t b : sizes) total += (b + 255) / 256 * 256;
  if (total > c->prim_scratch_bytes) {
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->prim_scratch) (void)hipFree(c->prim_scratch);
    c->prim_scratch = nullptr; c->prim_scratch_bytes = 0;
    const size_t want = std::max<size_t>(total + total / 4, (size_t)1 << 20);
    HIPCHK(hipMalloc(&c->prim_scratch, want));
    c->prim_scratch_bytes = want;
  }
  size_t off = 0;
  auto it = ptrs.begin();
  for (size_t b : sizes) { **it = static_cast<char*>(c->prim_scratch) + off; off += (b + 255) / 256 * 256; ++it; }
  return EG_OK;
}
static int h2d(void* dst, const void* src, size_t bytes, hipStream_t s) { if (bytes) HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s)); return EG_OK; }
static int d2h(void* dst, const void* src, size_t bytes, hipStream_t s) { if (bytes) HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s)); return EG_OK; }

int eg_scalar_from_wide_batch(eg_ctx* c, size_t n, const uint8_t* wide, uint8_t* out) { EG_LOCK(c);
  if (!c || (n && (!wide || !out))) return fail(EG_ERR_BAD_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  void *a, *o;
  TRY(prim_bufs(c, {n * 64, n * 32}, {&a, &o})); TRY(h2d(a, wide, n * 64, c->stream));
  hipLaunchKernelGGL(k_prim_scalar_from_wide, dim3(blocks_of(n)), dim3(NT), 0, c->stream, n, (const u32*)a, (u32*)o);
  TRY(d2h(out, o, n * 32, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return EG_OK;
}
int eg_scalar_is_canonical_batch(eg_ctx* c, size_t n, const uint8_t* s_, uint8_t* ok) { EG_LOCK(c);
  if (!c || (n && (!s_ || !ok))) return fail(EG_ERR_BAD_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  void *a, *o;
  TRY(prim_bufs(c, {n * 32, n}, {&a, &o})); TRY(h2d(a, s_, n * 32, c->stream));
  hipLaunchKernelGGL(k_prim_scalar_canonical, dim3(blocks_of(n)), dim3(NT), 0, c->stream, n, (const u32*)a, (unsigned char*)o);
  TRY(d2h(ok, o, n, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return EG_OK;
}
int eg_scalar_muladd_batch(eg_ctx* c, size_t n, const uint8_t* a_, const uint8_t* b_, const uint8_t* c_, uint8_t* out) { EG_LOCK(c);
  if (!c || (n && (!a_ || !b_ || !c_ || !out))) return fail(EG_ERR_BAD_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  void *a, *b, *cc, *o;
  TRY(prim_bufs(c, {n * 32, n * 32, n * 32, n * 32}, {&a, &b, &cc, &o}));
  TRY(h2d(a, a_, n * 32, c->stream)); TRY(h2d(b, b_, n * 32, c->stream)); TRY(h2d(cc, c_, n * 32, c->stream));
  hipLaunchKernelGGL(k_prim_scalar_muladd, dim3(blocks_of(n)), dim3(NT), 0, c->stream, n, (const u32*)a, (const u32*)b,
                     (const u32*)cc, (u32*)o);
  TRY(d2h(out, o, n * 32, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return EG_OK;
}
int eg_scalar_neg_batch(eg_ctx* c, size_t n, const uint8_t* a_, uint8_t* out) { EG_LOCK(c);
  if (!c || (n && (!a_ || !out))) return fail(EG_ERR_BAD_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  void *a, *o;
  TRY(prim_bufs(c, {n * 32, n * 32}, {&a, &o})); TRY(h2d(a, a_, n * 32, c->stream));
  hipLaunchKernelGGL(k_prim_scalar_neg, dim3(blocks_of(n)), dim3(NT), 0, c->stream, n, (const u32*)a, (u32*)o);
  TRY(d2h(out, o, n * 32, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return EG_OK;
}
int eg_scalar_invert_batch(eg_ctx* c, size_t n, const uint8_t* a_, uint8_t* out) { EG_LOCK(c);
  if (!c || (n && (!a_ || !out))) return fail(EG_ERR_BAD_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  void *a, *o;
  TRY(prim_bufs(c, {n * 32, n * 32}, {&a, &o})); TRY(h2d(a, a_, n * 32, c->stream));
  hipLaunchKernelGGL(k_prim_scalar_invert, dim3(blocks_of(n)), dim3(NT), 0, c->stream, n, (const u32*)a, (u32*)o);
  TRY(d2h(out, o, n * 32, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return EG_OK;
}
int eg_point_is_identity_batch(eg_ctx* c, size_t n, const uint8_t* in, uint8_t* is_identity, uint8_t* ok) { EG_LOCK(c);
  if (!c || (n && (!in || !is_identity || !ok))) return fail(EG_ERR_BAD_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  void *a, *f, *o;
  TRY(prim_bufs(c, {n * 32, n, n}, {&a, &f, &o})); TRY(h2d(a, in, n * 32, c->stream));
  hipLaunchKernelGGL(k_prim_point_is_identity, dim3(blocks_of(n)), dim3(NT), 0, c->stream, n, (const u32*)a,
                     (unsigned char*)f, (unsigned char*)o);
  TRY(d2h(is_identity, f, n, c->stream)); TRY(d2h(ok, o, n, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return EG_OK;
}
int eg_point_roundtrip_batch(eg_ctx* c, size_t n, const uint8_t* in, uint8_t* out, uint8_t* ok) { EG_LOCK(c);
  if (!c || (n && (!in || !out || !ok))) return fail(EG_ERR_BAD_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  void *a, *o, *k;
  TRY(prim_bufs(c, {n * 32, n * 32, n}, {&a, &o, &k})); TRY(h2d(a, in, n * 32, c->stream));
  hipLaunchKernelGGL(k_prim_point_roundtrip, dim3(blocks_of(n)), dim3(NT), 0, c->stream, n, (const u32*)a, (u32*)o,
                     (unsigned char*)k);
  TRY(d2h(out, o, n * 32, c->stream)); TRY(d2h(ok, k, n, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return EG_OK;
}
int eg_point_add_batch(eg_ctx* c, size_t n, const uint8_t* a_, const uint8_t* b_, int subtract, uint8_t* out, uint8_t* ok) { EG_LOCK(c);
  if (!c || (n && (!a_ || !b_ || !out || !ok))) return fail(EG_ERR_BAD_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  void *a, *b, *o, *k;
  TRY(prim_bufs(c, {n * 32, n * 32, n * 32, n}, {&a, &b, &o, &k}));
  TRY(h2d(a, a_, n * 32, c->stream)); TRY(h2d(b, b_, n * 32, c->stream));
  hipLaunchKernelGGL(k_prim_point_add, dim3(blocks_of(n)), dim3(NT), 0, c->stream, n, (const u32*)a, (const u32*)b, subtract,
                     (u32*)o, (unsigned char*)k);
  TRY(d2h(out, o, n * 32, c->stream)); TRY(d2h(ok, k, n, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return EG_OK;
}
// How the terms of a problem are cut: chunks of up to MSM_CHUNK terms share a doubling chain (less work per term), but a call with few
// terms in all should still cover the chip (a lone 2^16-term product in chunks of 8 keeps only 8192 lanes busy: 3.1 ms; in chunks of 1
// it is 65 536 independent ladders, ~1 ms): the chunk shrinks until the call has ~2 waves per SIMD worth of lanes.
static void msm_plan(const eg_ctx* ctx, size_t n, size_t terms, int* chunk, int* n_chunks) {
  const size_t lanes_wanted = ctx->knobs.msm_lanes;     // 2^17 (EG_MSM_LANES; tests: 1 = always full chunks)
  size_t c = std::min<size_t>(std::max<size_t>(terms, 1), MSM_CHUNK);
  c = std::min(c, std::max<size_t>(1, n * terms / lanes_wanted));
  *chunk = (int)c;
  *n_chunks = (int)std::max<size_t>(1, (terms + c - 1) / c);
}
// scratch of a call that is cut into several chunks per problem: two areas (the folds ping-pong between them), each holding the partial
// sums (PT_WORDS words) and the decode flags (one byte) of up to `count` chunks
struct MsmScratch { u32* part[2]; unsigned char* ok[2]; };
static size_t msm_area_bytes(size_t count) { return (count * PT_WORDS * sizeof(u32) + 255) / 256 * 256 + (count + 255) / 256 * 256; }
static size_t msm_scratch_total(size_t n, size_t nc) { return nc > 1 ? msm_area_bytes(n * nc) + msm_area_bytes(n * ((nc + 63) / 64)) : 0; }
static MsmScratch msm_scratch_at(void* base, size_t n, size_t nc) {
  MsmScratch m{{nullptr, nullptr}, {nullptr, nullptr}};
  if (nc <= 1 || !base) return m;
  char* p = static_cast<char*>(base);
  const size_t c0 = n * nc, c1 = n * ((nc + 63) / 64);
  m.part[0] = reinterpret_cast<u32*>(p); m.ok[0] = reinterpret_cast<unsigned char*>(p + (c0 * PT_WORDS * sizeof(u32) + 255) / 256 * 256);
  p += msm_area_bytes(c0);
  m.part[1] = reinterpret_cast<u32*>(p); m.ok[1] = reinterpret_cast<unsigned char*>(p + (c1 * PT_WORDS * sizeof(u32) + 255) / 256 * 256);
  return m;
}
// `count` partial sums per problem in m.part[0] -> out: 64-fold wavefront-shuffle passes, then one wavefront per problem adds them up, adds the
// generator term and encodes
static void msm_fold_reduce(eg_ctx* c, size_t n, int count, const MsmScratch& m, const u32* d_r, u32* d_out, unsigned char* d_ok, hipStream_t s) {
  int cur = 0;
  while (count > 64) {                 // 64-fold per pass, one wavefront per 64 partial sums
    const int next = (count + 63) / 64;
    hipLaunchKernelGGL(k_prim_msm_fold, dim3((unsigned)((n * (size_t)next * 64 + NT - 1) / NT)), dim3(NT), 0, s, n, count, next,
                       m.part[cur], m.ok[cur], m.part[cur ^ 1], m.ok[cur ^ 1]);
    cur ^= 1; count = next;
  }
  hipLaunchKernelGGL(k_prim_msm_reduce, dim3((unsigned)((n * 64 + NT - 1) / NT)), dim3(NT), 0, s, n, count, m.part[cur], m.ok[cur],
                     d_r, c->tabG, d_out, d_ok);
}

// ---- the bucket method for one very large product (pippenger.cuh) ----------------------------------------------------------------------------
// Window width by size: 2^(c-1) buckets per window with ~64 terms each.  Measured against the Straus path (tools/msm_probe.py,
// profiles/r04_msm_by_size.txt): 2^19 terms 3.6 ms against 3.3, 2^20 4.8 against 5.7, 2^21 7.3 against 10.8, 2^22 13.0 against 21.5 - the
// bucket method takes over from 2^20 terms; EG_MSM_BUCKET_MIN moves the switch (tests force either path at sizes the oracle can follow).
static int pip_window_bits(size_t terms) {
  int c = PIP_MIN_C;
  while (c < PIP_MAX_C && ((size_t)64 << (c - 1)) < terms) ++c;      // 2^17 -> 12, 2^18 -> 13, 2^19 -> 14, >= 2^20 -> 15
  return c;
}
static bool msm_uses_buckets(const eg_ctx* ctx, size_t terms) { return terms >= std::max<size_t>(ctx->knobs.msm_bucket_min, (size_t)1 << 12); }
constexpr int PIP_MAX_LEVELS = PIP_SEQ - 1;
struct PipLayout {
  size_t niels, digits, counts, offsets, cursors, idx, pieces[PIP_MAX_LEVELS], piece0[PIP_MAX_LEVELS], totals, tiles, psum[2], flag, part, total;
  size_t psum_points;       // capacity of each of the two partial-sum buffers
  int c, W, B, waves, levels;
};
static PipLayout pip_layout(size_t terms) {
  PipLayout L;
  L.c = pip_window_bits(terms); L.W = pip_windows(L.c); L.B = 1 << (L.c - 1); L.waves = L.W * (L.B / PIP_SEG) / 64;
  const size_t nb = (size_t)L.W * L.B;
  // levels: pieces of 128 terms, then of 64 partial sums, until the longest possible bucket (all terms) is down to one sum
  L.levels = 1;
  for (size_t n = (terms + PIP_S_TERMS - 1) / PIP_S_TERMS; n > 1; n = (n + PIP_S_POINTS - 1) / PIP_S_POINTS) ++L.levels;
  L.psum_points = (size_t)L.W * terms / PIP_S_TERMS + nb + 64;      // every bucket adds at most one ragged piece per level
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) / 256 * 256; return at; };
  L.niels = take(terms * PIP_NIELS_WORDS * sizeof(u32));
  L.digits = take((size_t)L.W * terms * sizeof(unsigned short));
  L.counts = take(nb * sizeof(u32));
  L.offsets = take(nb * sizeof(u32));
  L.cursors = take(nb * sizeof(u32));
  L.idx = take((size_t)L.W * terms * sizeof(u32));
  for (int l = 0; l < PIP_MAX_LEVELS; ++l) { L.pieces[l] = take(nb * sizeof(u32)); L.piece0[l] = take(nb * sizeof(u32)); }
  L.totals = take(256);
  L.tiles = take(((nb + 1023) / 1024) * PIP_SEQ * sizeof(u32));
  L.psum[0] = take(L.psum_points * PT_QUADS * sizeof(uint4));
  L.psum[1] = take(L.psum_points * PT_QUADS * sizeof(uint4));
  L.flag = take(256);
  L.part = take(msm_scratch_total(1, (size_t)L.waves) + 256);
  L.total = off;
  return L;
}
static int pip_launch(eg_ctx* c, size_t terms, const u32* d_scalars, const u32* d_points, const u32* d_r, void* d_scratch, u32* d_out,
                      unsigned char* d_ok, hipStream_t s, bool prepared = false) {
  const PipLayout L = pip_layout(terms);
  if (L.levels > PIP_MAX_LEVELS) return fail(EG_ERR_BAD_ARG, "too many terms for the bucket method");
  char* base = static_cast<char*>(d_scratch);
  auto at = [&](size_t o) { return reinterpret_cast<u32*>(base + o); };
  PipBufs P;
  P.niels = at(L.niels); P.digits = reinterpret_cast<unsigned short*>(base + L.digits);
  P.counts = at(L.counts); P.offsets = at(L.offsets); P.cursors = at(L.cursors); P.idx = at(L.idx); P.all_ok = at(L.flag);
  const u32 nb = (u32)((size_t)L.W * L.B);
  u32* totals = at(L.totals);
  uint4* psum[2] = {reinterpret_cast<uint4*>(base + L.psum[0]), reinterpret_cast<uint4*>(base + L.psum[1])};
  HIPCHK(hipMemsetAsync(P.counts, 0, (size_t)nb * sizeof(u32), s));
  HIPCHK(hipMemsetAsync(P.all_ok, 0x01, sizeof(u32), s));
  if (prepared) hipLaunchKernelGGL(k_pip_prepare<true>, dim3((unsigned)((terms + NT - 1) / NT)), dim3(NT), 0, s, terms, L.c, d_scalars, d_points, P);
  else hipLaunchKernelGGL(k_pip_prepare<false>, dim3((unsigned)((terms + NT - 1) / NT)), dim3(NT), 0, s, terms, L.c, d_scalars, d_points, P);
  {
    PipScan S;
    S.offsets = P.offsets; S.cursors = P.cursors; S.totals = totals; S.tile_sums = at(L.tiles);
    for (int l = 0; l < PIP_SEQ - 1; ++l) { S.pieces[l] = at(L.pieces[l]); S.piece0[l] = at(L.piece0[l]); }
    const unsigned tiles = (nb + 1023u) / 1024u;
    hipLaunchKernelGGL(k_pip_scan_tiles, dim3(tiles), dim3(NT), 0, s, (const u32*)P.counts, nb, L.levels, S);
    hipLaunchKernelGGL(k_pip_scan_tops, dim3(1), dim3(64 * PIP_SEQ), 0, s, tiles, S);
    hipLaunchKernelGGL(k_pip_scan_apply, dim3(tiles), dim3(NT), 0, s, (const u32*)P.counts, nb, L.levels, S);
  }
  hipLaunchKernelGGL(k_pip_fill, dim3(grid_for(terms * (size_t)L.W, c->cus * 64)), dim3(NT), 0, s, terms, L.c, P);
  // level 0: pieces of the term lists; levels 1 ..: pieces of the partial sums of the level before
  const u32* cnt = P.counts;
  const u32* off = P.offsets;
  size_t bound = L.psum_points;
  for (int l = 0; l < L.levels; ++l) {
    u32 *pieces = at(L.pieces[l]), *piece0 = at(L.piece0[l]);
    const PipLevel lv{cnt, off, pieces, piece0, totals + l};
    const int grid = grid_for(bound, c->cus * 64);
    if (l == 0) hipLaunchKernelGGL(k_pip_sum_terms, dim3(grid), dim3(NT), 0, s, P, lv, nb, psum[0]);
    else hipLaunchKernelGGL(k_pip_sum_points, dim3(grid), dim3(NT), 0, s, (const uint4*)psum[(l - 1) & 1], lv, nb, psum[l & 1]);
    cnt = pieces; off = piece0;
    bound = bound / PIP_S_POINTS + nb + 64;
  }
  const MsmScratch m = msm_scratch_at(base + L.part, 1, (size_t)L.waves);
  hipLaunchKernelGGL(k_pip_window, dim3((unsigned)(((size_t)L.waves * 64 + NT - 1) / NT)), dim3(NT), 0, s, L.c, P,
                     (const uint4*)psum[(L.levels - 1) & 1], cnt, off, m.part[0], m.ok[0]);
  msm_fold_reduce(c, 1, L.waves, m, d_r, d_out, d_ok, s);
  HIPCHK(hipGetLastError());
  return EG_OK;
}

static size_t msm_chunks(const eg_ctx* ctx, size_t n, size_t terms) { int c, k; msm_plan(ctx, n, terms, &c, &k); return (size_t)k; }
// scratch a call needs on the device (0: none)
static size_t msm_scratch_for(const eg_ctx* ctx, size_t n, size_t terms) {
  return msm_uses_buckets(ctx, terms) ? pip_layout(terms).total : msm_scratch_total(n, msm_chunks(ctx, n, terms));
}
// out[i] = enc( sum_t [k_it]P_it + [r_i]G ) on device pointers (kernels.cuh: k_prim_msm, k_prim_msm_fold, k_prim_msm_reduce; very large
// products: pippenger.cuh, one problem after the other); asynchronous on s
static int prim_msm_launch(eg_ctx* c, size_t n, size_t terms, const u32* d_scalars, const u32* d_points, const u32* d_r, void* d_scratch,
                           u32* d_out, unsigned char* d_ok, hipStream_t s, bool prepared = false) {
  const size_t pw = prepared ? PREP_WORDS : 8;          // words per point: a prepared point (96 bytes) or an encoding (32)
  if (msm_uses_buckets(c, terms)) {
    for (size_t i = 0; i < n; ++i)
      TRY(pip_launch(c, terms, d_scalars + i * terms * 8, d_points + i * terms * pw, d_r ? d_r + i * 8 : nullptr, d_scratch, d_out + i * 8,
                     d_ok ? d_ok + i : nullptr, s, prepared));
    return EG_OK;
  }
  int chunk, n_chunks;
  msm_plan(c, n, terms, &chunk, &n_chunks);
  const MsmScratch m = msm_scratch_at(d_scratch, n, (size_t)n_chunks);
  // every lane owns `chunk` tables in the per-lane workspace: the grid shrinks accordingly (the workspace is msm_blocks x one table)
  const int grid = grid_for(n * (size_t)n_chunks, std::max(1, c->msm_blocks / chunk));
  TRY(ws_acquire(c, 3u, s));       // the lanes' tables live in the context's per-lane workspace, which engines with work in flight share
  if (prepared) hipLaunchKernelGGL(k_prim_msm<true>, dim3(grid), dim3(NT), (size_t)chunk * 8 * NT * sizeof(u32), s, n, (int)terms, chunk, n_chunks,
                                   d_scalars, d_points, d_r, c->tabG, c->ws, m.part[0], m.ok[0], d_out, d_ok);
  else hipLaunchKernelGGL(k_prim_msm<false>, dim3(grid), dim3(NT), (size_t)chunk * 8 * NT * sizeof(u32), s, n, (int)terms, chunk, n_chunks, d_scalars,
                          d_points, d_r, c->tabG, c->ws, m.part[0], m.ok[0], d_out, d_ok);
  TRY(ws_release(c, 3u, s));
  if (n_chunks > 1) msm_fold_reduce(c, n, n_chunks, m, d_r, d_out, d_ok, s);
  HIPCHK(hipGetLastError());
  return EG_OK;
}
static int prim_msm(eg_ctx* c, size_t n, size_t terms, const uint8_t* scalars, const uint8_t* points, const uint8_t* r,
                    uint8_t* out, uint8_t* ok) {
  HIPCHK(hipSetDevice(c->device));
  if (n == 0) return EG_OK;
  if (terms > ((size_t)1 << 24) || n > ((size_t)1 << 32)) return fail(EG_ERR_BAD_ARG, "at most 2^24 terms per problem");
  const size_t need = msm_scratch_for(c, n, terms);
  void *sc, *pt, *rr, *o, *k, *scratch;
  TRY(prim_bufs(c, {n * terms * 32, n * terms * 32, n * 32, n * 32, n, need}, {&sc, &pt, &rr, &o, &k, &scratch}));
  TRY(h2d(sc, scalars, n * terms * 32, c->stream)); TRY(h2d(pt, points, n * terms * 32, c->stream));
  if (r) TRY(h2d(rr, r, n * 32, c->stream));
  TRY(prim_msm_launch(c, n, terms, (const u32*)sc, (const u32*)pt, r ? (const u32*)rr : (const u32*)nullptr, need ? scratch : nullptr,
                      (u32*)o, (unsigned char*)k, c->stream));
  TRY(d2h(out, o, n * 32, c->stream));
  if (ok) TRY(d2h(ok, k, n, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipGetLastError());
  return EG_OK;
}
int eg_mul_generator_batch(eg_ctx* c, size_t n, const uint8_t* k, uint8_t* out) { EG_LOCK(c);
  if (!c || (n && (!k || !out))) return fail(EG_ERR_BAD_ARG, "bad argument");
  return prim_msm(c, n, 0, nullptr, nullptr, k, out, nullptr);
}
int eg_vartime_double_mul_generator_batch(eg_ctx* c, size_t n, const uint8_t* k, const uint8_t* p, const uint8_t* r, uint8_t* out,
                                          uint8_t* ok) { EG_LOCK(c);
  if (!c || (n && (!k || !p || !r || !out || !ok))) return fail(EG_ERR_BAD_ARG, "bad argument");
  return prim_msm(c, n, 1, k, p, r, out, ok);
}
int eg_vartime_multi_mul_batch(eg_ctx* c, size_t n, size_t terms, const uint8_t* scalars, const uint8_t* points, uint8_t* out,
                               uint8_t* ok) { EG_LOCK(c);
  if (!c || (n && !out) || (n && terms && (!scalars || !points))) return fail(EG_ERR_BAD_ARG, "bad argument");
  return prim_msm(c, n, terms, scalars, points, nullptr, out, ok);
}
// the same on DEVICE buffers, asynchronous on `stream` (a caller that keeps its operands in HBM pays no copies and no synchronisation;
// what bench.py --workload msm times).  d_scratch must hold eg_msm_scratch_bytes_ctx(ctx, n, terms) bytes (0 for <= 8 terms).
size_t eg_msm_scratch_bytes_ctx(eg_ctx* c, size_t n, size_t terms) { return c ? msm_scratch_for(c, n, terms) : 0; }
int eg_vartime_multi_mul_batch_device(eg_ctx* c, size_t n, size_t terms, const void* d_scalars, const void* d_points, const void* d_r,
                                      void* d_scratch, void* d_out, void* d_ok, void* stream) { EG_LOCK(c);
  if (!c || (n && !d_out) || (n && terms && (!d_scalars || !d_points)) || (n && !terms && !d_r)) return fail(EG_ERR_BAD_ARG, "bad argument");
  if (terms > ((size_t)1 << 24)) return fail(EG_ERR_BAD_ARG, "at most 2^24 terms per problem");
  const size_t need = msm_scratch_for(c, n, terms);
  if (n && need && !d_scratch) return fail(EG_ERR_BAD_ARG, "this call is cut into several chunks per problem (or uses the bucket method) and needs d_scratch (eg_msm_scratch_bytes_ctx)");
  if (n == 0) return EG_OK;
  HIPCHK(hipSetDevice(c->device));
  return prim_msm_launch(c, n, terms, (const u32*)d_scalars, (const u32*)d_points, (const u32*)d_r, need ? d_scratch : nullptr, (u32*)d_out,
                         (unsigned char*)d_ok, (hipStream_t)stream);
}

// Prepared points: decode a point set ONCE, multiply over it many times (ristretto.rs:139-145: vartime_multi_mul takes Elements, which in
// the reference ARE decoded points - its callers never pay a decoding per product; eg_vartime_multi_mul_batch_device, which takes
// encodings, does).
size_t eg_prepared_point_size(void) { return PREP_WORDS * sizeof(u32); }
int eg_points_prepare_device(eg_ctx* c, size_t n, const void* d_encodings, void* d_prepared, void* d_ok, void* stream) { EG_LOCK(c);
  if (!c || (n && (!d_encodings || !d_prepared))) return fail(EG_ERR_BAD_ARG, "bad argument");
  if (n && !d_ok) return fail(EG_ERR_BAD_ARG, "d_ok is mandatory: an encoding that does not decode is prepared as the identity, and this is the only place that says so");
  if (reinterpret_cast<uintptr_t>(d_prepared) & 15u) return fail(EG_ERR_BAD_ARG, "d_prepared must be 16-byte aligned");
  if (n == 0) return EG_OK;
  hipLaunchKernelGGL(k_prim_points_prepare, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, (hipStream_t)stream, n, (const u32*)d_encodings,
                     (u32*)d_prepared, (unsigned char*)d_ok);
  HIPCHK(hipGetLastError());
  return EG_OK;
}
int eg_vartime_multi_mul_prepared_batch_device(eg_ctx* c, size_t n, size_t terms, const void* d_scalars, const void* d_prepared, const void* d_r,
                                               void* d_scratch, void* d_out, void* stream) { EG_LOCK(c);
  if (!c || (n && !d_out) || (n && terms && (!d_scalars || !d_prepared)) || (n && !terms && !d_r)) return fail(EG_ERR_BAD_ARG, "bad argument");
  if (terms > ((size_t)1 << 24)) return fail(EG_ERR_BAD_ARG, "at most 2^24 terms per problem");
  if (reinterpret_cast<uintptr_t>(d_prepared) & 15u) return fail(EG_ERR_BAD_ARG, "d_prepared must be 16-byte aligned");
  const size_t need = msm_scratch_for(c, n, terms);
  if (n && need && !d_scratch) return fail(EG_ERR_BAD_ARG, "this call is cut into several chunks per problem (or uses the bucket method) and needs d_scratch (eg_msm_scratch_bytes_ctx)");
  if (n == 0) return EG_OK;
  return prim_msm_launch(c, n, terms, (const u32*)d_scalars, (const u32*)d_prepared, (const u32*)d_r, need ? d_scratch : nullptr, (u32*)d_out,
                         nullptr, (hipStream_t)stream, true);
}

// The VALU roof of THIS box: the shipped fe_mul in a bare chain (k_selfbench_fmul), launched back to back for `seconds`; the rate and the
// shader clock are taken over the second half, when the power management has settled (a burst of tens of milliseconds runs 10 % faster:
// profiles/r03_ubench_field_sustained.txt).  Boxes of one pool differ by +-3 %, so a fraction of a constant measured elsewhere cannot tell
// a slow box from a regression.
int eg_selfbench_fmul(eg_ctx* c, double seconds, double* fmul_g_per_s, double* sclk_mhz) { EG_LOCK(c);
  if (!c || !fmul_g_per_s || !(seconds > 0) || seconds > 30) return fail(EG_ERR_BAD_ARG, "bad argument (0 < seconds <= 30)");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const int waves_per_simd = 3, blocks = c->cus * waves_per_simd, iters = 20000;       // ~30 ms a launch
  const size_t lds = ((size_t)160 * 1024 / waves_per_simd) / 1024 * 1024;
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_selfbench_fmul), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const size_t lanes = (size_t)blocks * NT, waves = lanes / 64;
  void *d_out = nullptr, *d_st = nullptr;
  TRY(prim_bufs(c, {lanes * 16 * sizeof(u32), waves * sizeof(uint2)}, {&d_out, &d_st}));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  HIPCHK(hipEventCreate(&e0));
  ScopeExit ev{[&]() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }};
  HIPCHK(hipEventCreate(&e1));
  const double ops_per_launch = 2.0 * iters * (double)lanes;
  auto burst = [&](int launches, double* g, double* mhz) -> int {
    HIPCHK(hipEventRecord(e0, s));
    for (int l = 0; l < launches; ++l)
      hipLaunchKernelGGL(k_selfbench_fmul, dim3(blocks), dim3(NT), lds, s, (u32*)d_out, (uint2*)d_st, 12345u + (u32)l, iters);
    HIPCHK(hipEventRecord(e1, s));
    HIPCHK(hipEventSynchronize(e1));
    HIPCHK(hipGetLastError());
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint2> st(waves);
    HIPCHK(hipMemcpy(st.data(), d_st, waves * sizeof(uint2), hipMemcpyDeviceToHost));
    std::vector<double> f(waves);
    for (size_t i = 0; i < waves; ++i) f[i] = st[i].y ? (double)st[i].x / ((double)st[i].y * 10.0) * 1e3 : 0.0;     // cycles per 10 ns tick -> MHz
    std::sort(f.begin(), f.end());
    *g = ops_per_launch * launches / (ms * 1e6);
    *mhz = f[waves / 2];
    return EG_OK;
  };
  double g = 0, mhz = 0;
  TRY(burst(1, &g, &mhz));                                   // sizes the halves (and warms the clock up)
  const double one_ms = ops_per_launch / (g * 1e6);
  const int half = std::max(1, (int)(seconds * 500.0 / one_ms));
  TRY(burst(half, &g, &mhz));
  TRY(burst(half, &g, &mhz));
  *fmul_g_per_s = g;
  if (sclk_mhz) *sclk_mhz = mhz;
  return EG_OK;
}

// ---- tally stage (examples/voting.rs:122-177) --------------------------------------------------------------------------
// Params::combine_shares (src/sharing/mod.rs:302-325) with lagrange_coefficients (:139-170): the first `threshold` shares are combined
// by Lagrange interpolation in the exponent.  Every scalar operation (products of the index differences, the batched inversion, the
// scaling) and the multi-scalar multiplication run on the GPU primitives; the host only feeds small integers.
int eg_combine_shares(eg_ctx* c, uint64_t shares, uint64_t threshold, size_t n, const uint64_t* indexes, const uint8_t* dh_elements,
                      uint8_t out[32], int* combined) { EG_LOCK(c);
  if (!c || !combined || !out || (n && (!indexes || !dh_elements)) || threshold < 1 || threshold > shares)
    return fail(EG_ERR_BAD_ARG, "bad argument");
  *combined = 0;
  if (n < threshold) return EG_OK;                         // None: the number of shares is insufficient
  const size_t t = (size_t)threshold;                      // .take(self.threshold)
  for (size_t i = 0; i < t; ++i) {
    if (indexes[i] >= shares) return fail(EG_ERR_BAD_ARG, "share index exceeds the number of participants");   // the reference panics
    for (size_t j = 0; j < i; ++j)
      if (indexes[i] == indexes[j]) return fail(EG_ERR_BAD_ARG, "duplicate share index");
  }
  auto u64_scalar = [](uint64_t v, uint8_t* dst) { memset(dst, 0, 32); memcpy(dst, &v, 8); };
  std::vector<uint8_t> prod(32 * (t + 1)), factor(32 * (t + 1)), zero(32 * (t + 1), 0), tmp(32 * (t + 1));
  // lanes 0..t-1: denominators |prod_j d_ij| with d_ii = index_i + 1; lane t: the scale prod_i (index_i + 1)
  for (size_t i = 0; i <= t; ++i) u64_scalar(1, prod.data() + 32 * i);
  std::vector<int> negative(t, 0);
  for (size_t j = 0; j < t; ++j) {
    for (size_t i = 0; i < t; ++i) {
      const uint64_t a = indexes[i], b = indexes[j];
      if (a > b) negative[i] ^= 1;                         // Ordering::Greater => (true, index - other_index)
      u64_scalar(a == b ? a + 1 : (a > b ? a - b : b - a), factor.data() + 32 * i);
    }
    u64_scalar(indexes[j] + 1, factor.data() + 32 * t);
    TRY(eg_scalar_muladd_batch(c, t + 1, prod.data(), factor.data(), zero.data(), tmp.data()));
    prod.swap(tmp);
  }
  for (size_t i = 0; i < t; ++i)
    if (negative[i]) TRY(eg_scalar_neg_batch(c, 1, prod.data() + 32 * i, prod.data() + 32 * i));
  std::vector<uint8_t> inv(32 * t), coeff(32 * t), scale(32 * t);
  TRY(eg_scalar_invert_batch(c, t, prod.data(), inv.data()));                              // G::invert_scalars
  for (size_t i = 0; i < t; ++i) memcpy(scale.data() + 32 * i, prod.data() + 32 * t, 32);
  TRY(eg_scalar_muladd_batch(c, t, inv.data(), scale.data(), zero.data(), coeff.data()));  // [scale](sum [d_i^-1] S_i) = sum [scale d_i^-1] S_i
  uint8_t ok = 0;
  TRY(eg_vartime_multi_mul_batch(c, 1, t, coeff.data(), dh_elements, out, &ok));
  if (!ok) return fail(EG_ERR_BAD_ARG, "a decryption share is not a valid ristretto255 encoding");
  *combined = 1;
  return EG_OK;
}

// DiscreteLogTable (src/encryption.rs:260-298): [v]G (canonical encoding) -> v for the given values; the products come from the GPU,
// the lookups are a host hash map like the reference's HashMap<Vec<u8>, u64>.
struct eg_dlog_table { std::unordered_map<std::string, uint64_t> map; };
int eg_dlog_table_create(eg_ctx* c, size_t n, const uint64_t* values, eg_dlog_table** out) { EG_LOCK(c);
  if (!c || !out || (n && !values)) return fail(EG_ERR_BAD_ARG, "bad argument");
  std::vector<uint64_t> vals;
  for (size_t i = 0; i < n; ++i) if (values[i] != 0) vals.push_back(values[i]);           // .filter(|&value| value != 0)
  std::vector<uint8_t> sc(32 * vals.size(), 0), enc(32 * vals.size());
  for (size_t i = 0; i < vals.size(); ++i) memcpy(sc.data() + 32 * i, &vals[i], 8);
  if (!vals.empty()) TRY(eg_mul_generator_batch(c, vals.size(), sc.data(), enc.data()));
  std::unique_ptr<eg_dlog_table> t(new eg_dlog_table());
  t->map.reserve(vals.size() * 2);
  for (size_t i = 0; i < vals.size(); ++i) t->map.emplace(std::string(reinterpret_cast<const char*>(enc.data()) + 32 * i, 32), vals[i]);
  *out = t.release();
  return EG_OK;
}
void eg_dlog_table_destroy(eg_dlog_table* t) { delete t; }
// values[i] = discrete log of elements[i] if it is in the table (found[i] = 1); the identity (32 zero bytes) is always 0
int eg_dlog_table_get(const eg_dlog_table* t, size_t n, const uint8_t* elements, uint64_t* values, uint8_t* found) {
  if (!t || (n && (!elements || !values || !found))) return fail(EG_ERR_BAD_ARG, "bad argument");
  static const char zero[32] = {0};
  for (size_t i = 0; i < n; ++i) {
    const char* e = reinterpret_cast<const char*>(elements) + 32 * i;
    if (memcmp(e, zero, 32) == 0) { values[i] = 0; found[i] = 1; continue; }
    auto it = t->map.find(std::string(e, 32));
    found[i] = it != t->map.end();
    values[i] = found[i] ? it->second : 0;
  }
  return EG_OK;
}

// ---- batch tier: choice ---------------------------------------------------------------------------------------------------
// rings per group of the ring-group walk: EG_RING_GROUP overrides the plan's default (0 = every table of a ballot at once)
static int choice_ring_group() { return read_knobs().ring_group; }     // at the creation of a params object (the plan is built before the engine)
size_t eg_choice_ballot_size(int n_options, int single) { return eghost::choice_ballot_size(n_options, single != 0); }

int eg_choice_params_create(eg_ctx* c, const uint8_t pk[32], int n_options, int single, eg_choice_params** out) { EG_LOCK(c);
  if (!c || !pk || !out) return fail(EG_ERR_BAD_ARG, "bad argument");
  if (n_options < 1 || n_options > 4000) return fail(EG_ERR_BAD_ARG, "n_options must be in 1..4000");
  Engine* e = nullptr;
  TRY(engine_create(c, eghost::build_choice_plan(n_options, single != 0, choice_ring_group()), pk, n_options, &e));
  *out = new eg_choice_params{e, n_options, single};
  return EG_OK;
}
void eg_choice_params_destroy(eg_choice_params* p) { params_destroy(p); }
int eg_verify_choice_batch(eg_choice_params* p, size_t n, const uint8_t* ballots, uint32_t* status, uint8_t* tally_out) { EG_LOCK_P(p);
  if (!p || (n && (!ballots || !status))) return fail(EG_ERR_BAD_ARG, "bad argument");
  EG_WAIT_JSON(p->eng);
  return engine_verify_host(p->eng, n, ballots, status, tally_out);
}
int eg_verify_choice_batch_device(eg_choice_params* p, size_t n, const void* d_ballots, void* d_status, void* stream) { EG_LOCK_P(p);
  if (!p || (n && (!d_ballots || !d_status))) return fail(EG_ERR_BAD_ARG, "bad argument");
  EG_WAIT_JSON(p->eng);
  HIPCHK(hipSetDevice(p->eng->ctx->device));
  return engine_verify_device(p->eng, n, d_ballots, d_status, (hipStream_t)stream);
}
static int tally_reset(Engine* e, hipStream_t s, bool wait) {
  if (wait) HIPCHK(hipDeviceSynchronize());   // host form: order after anything still running on caller streams
  hipLaunchKernelGGL(k_tally_init, dim3(blocks_of(e->plan.tally_slots.size())), dim3(NT), 0, s, e->tally, (int)e->plan.tally_slots.size());
  HIPCHK(hipGetLastError());
  TRY(engine_touch(e, s));
  if (wait) HIPCHK(hipStreamSynchronize(s));
  return EG_OK;
}
static int tally_encode_device(Engine* e, void* d_out, hipStream_t s) {
  hipLaunchKernelGGL(k_tally_encode, dim3(blocks_of(e->plan.tally_slots.size())), dim3(NT), 0, s, e->tally, (int)e->plan.tally_slots.size(), (u32*)d_out);
  HIPCHK(hipGetLastError());
  return EG_OK;
}
int eg_choice_tally_encode_device(eg_choice_params* p, void* d_out, void* stream) { EG_LOCK_P(p);
  if (!p || !d_out) return fail(EG_ERR_BAD_ARG, "bad argument");
  EG_WAIT_JSON(p->eng);
  return tally_encode_device(p->eng, d_out, (hipStream_t)stream);
}
int eg_qv_tally_encode_device(eg_qv_params* p, void* d_out, void* stream) { EG_LOCK_P(p);
  if (!p || !d_out) return fail(EG_ERR_BAD_ARG, "bad argument");
  EG_WAIT_JSON(p->eng);
  return tally_encode_device(p->eng, d_out, (hipStream_t)stream);
}
int eg_points_sum_device(eg_ctx* c, int n_ranks, int n_points, const void* d_in, void* d_out, void* d_bad, void* stream) { EG_LOCK(c);
  if (!c || n_ranks < 1 || n_points < 1 || !d_in || !d_out) return fail(EG_ERR_BAD_ARG, "bad argument");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_points_sum, dim3((n_points + 63) / 64), dim3(64), 0, s, (const u32*)d_in, n_ranks, n_points, (u32*)d_out,
                     (u32*)d_bad);
  HIPCHK(hipGetLastError());
  return EG_OK;
}
// builds the wide comb tables now (synchronously, ~12 GB each for G and K) instead of inside the first large verify call
static int prepare_wide(Engine* e) {
  HIPCHK(hipSetDevice(e->ctx->device));
  HIPCHK(hipDeviceSynchronize());
  const int rc = ensure_big_tables(e, e->ctx->stream);
  if (rc) return rc;
  if (e->ctx->big_bits && (!e->ctx->tabG_big || !e->d_tabK_big)) return fail(EG_ERR_NOMEM, "the wide comb tables do not fit the device memory");
  return EG_OK;
}
int eg_choice_prepare_wide_tables(eg_choice_params* p) { EG_LOCK_P(p); return p ? prepare_wide(p->eng) : fail(EG_ERR_BAD_ARG, "null"); }
int eg_qv_prepare_wide_tables(eg_qv_params* p) { EG_LOCK_P(p); return p ? prepare_wide(p->eng) : fail(EG_ERR_BAD_ARG, "null"); }
int eg_choice_tally_reset(eg_choice_params* p) { EG_LOCK_P(p); if (!p) return fail(EG_ERR_BAD_ARG, "null"); EG_WAIT_JSON(p->eng); return tally_reset(p->eng, p->eng->ctx->stream, true); }
int eg_choice_tally_reset_async(eg_choice_params* p, void* stream) { EG_LOCK_P(p); if (!p) return fail(EG_ERR_BAD_ARG, "null"); EG_WAIT_JSON(p->eng); return tally_reset(p->eng, (hipStream_t)stream, false); }
int eg_choice_tally_add(eg_choice_params* p, const uint8_t* in) { EG_LOCK_P(p);
  if (!p || !in) return fail(EG_ERR_BAD_ARG, "bad argument");
  EG_WAIT_JSON(p->eng);
  return engine_tally_add(p->eng, in);
}
int eg_qv_tally_add(eg_qv_params* p, const uint8_t* in) { EG_LOCK_P(p);
  if (!p || !in) return fail(EG_ERR_BAD_ARG, "bad argument");
  EG_WAIT_JSON(p->eng);
  return engine_tally_add(p->eng, in);
}
int eg_choice_tally_encode(eg_choice_params* p, uint8_t* out) { EG_LOCK_P(p);
  if (!p || !out) return fail(EG_ERR_BAD_ARG, "bad argument");
  EG_WAIT_JSON(p->eng);
  return engine_tally_encode(p->eng, out);
}

// ---- batch tier: quadratic voting --------------------------------------------------------------------------------------------
int eg_qv_params_create(eg_ctx* c, const uint8_t pk[32], int n_options, uint64_t credits, eg_qv_params** out) { EG_LOCK(c);
  if (!c || !pk || !out) return fail(EG_ERR_BAD_ARG, "bad argument");
  if (n_options < 1 || n_options > 256 || credits < 1 || credits > 100000) return fail(EG_ERR_BAD_ARG, "options in 1..256, credits in 1..100000");
  Engine* e = nullptr;
  TRY(engine_create(c, eghost::build_qv_plan(n_options, credits), pk, n_options, &e));
  *out = new eg_qv_params{e, n_options, credits, eghost::qv_shape(n_options, credits)};
  return EG_OK;
}
void eg_qv_params_destroy(eg_qv_params* p) { params_destroy(p); }
size_t eg_qv_ballot_size(const eg_qv_params* p) { return p ? p->shape.ballot_size : 0; }
int eg_verify_qv_batch(eg_qv_params* p, size_t n, const uint8_t* ballots, uint32_t* status, uint8_t* tally_out) { EG_LOCK_P(p);
  if (!p || (n && (!ballots || !status))) return fail(EG_ERR_BAD_ARG, "bad argument");
  EG_WAIT_JSON(p->eng);
  return engine_verify_host(p->eng, n, ballots, status, tally_out);
}
int eg_verify_qv_batch_device(eg_qv_params* p, size_t n, const void* d_ballots, void* d_status, void* stream) { EG_LOCK_P(p);
  if (!p || (n && (!d_ballots || !d_status))) return fail(EG_ERR_BAD_ARG, "bad argument");
  EG_WAIT_JSON(p->eng);
  HIPCHK(hipSetDevice(p->eng->ctx->device));
  return engine_verify_device(p->eng, n, d_ballots, d_status, (hipStream_t)stream);
}
int eg_qv_tally_reset(eg_qv_params* p) { EG_LOCK_P(p); if (!p) return fail(EG_ERR_BAD_ARG, "null"); EG_WAIT_JSON(p->eng); return tally_reset(p->eng, p->eng->ctx->stream, true); }
int eg_qv_tally_reset_async(eg_qv_params* p, void* stream) { EG_LOCK_P(p); if (!p) return fail(EG_ERR_BAD_ARG, "null"); EG_WAIT_JSON(p->eng); return tally_reset(p->eng, (hipStream_t)stream, false); }
int eg_qv_tally_encode(eg_qv_params* p, uint8_t* out) { EG_LOCK_P(p);
  if (!p || !out) return fail(EG_ERR_BAD_ARG, "bad argument");
  EG_WAIT_JSON(p->eng);
  return engine_tally_encode(p->eng, out);
}

// ---- batch tier, several GPUs in ONE process (the `device_mask` of the batch entry; examples/voting.rs:179-213 is one
// single-threaded host process) ------------------------------------------------------------------------------------------------
// per_device[d] is a params object of the same election created on its own context (normally one context per GPU).  The batch is
// cut into contiguous slabs, slab d = [n d / n_dev, n (d + 1) / n_dev) (the split of elastic_elgamal_amd/distributed.py:
// shard_range), and one host thread per params object runs the ordinary host entry on its slab: its own device, streams, uploads and
// running tally.  The per-slab tallies (64 n_options bytes each) are merged on the host side of the ABI with the element addition
// of the primitive tier on the first context - in one process there is nothing for RCCL to do.
extern "C++" {
// the checks every multi entry shares: n_dev objects, all different, all of one election
template <class Params>
static int multi_check(Params* const* per_device, int n_dev) {
  if (!per_device || n_dev < 1 || n_dev > 64) return fail(EG_ERR_BAD_ARG, "bad argument");
  for (int d = 0; d < n_dev; ++d) {
    if (!per_device[d]) return fail(EG_ERR_BAD_ARG, "null params object");
    for (int k = 0; k < d; ++k)
      if (per_device[k] == per_device[d]) return fail(EG_ERR_BAD_ARG, "the same params object given twice: one object (and context) per slab");
    const eghost::Plan &A = per_device[0]->eng->plan, &B = per_device[d]->eng->plan;
    if (B.stride != A.stride || per_device[d]->n_options != per_device[0]->n_options || B.tally_slots.size() != A.tally_slots.size() ||
        memcmp(per_device[d]->eng->key_bytes, per_device[0]->eng->key_bytes, 32) != 0)
      return fail(EG_ERR_BAD_ARG, "params objects of different elections");
  }
  return EG_OK;
}
// One host thread per slab (the caller's thread takes slab 0), nothing escapes the C ABI: an exception inside a slab's work becomes that
// slab's error code.
template <class Work>
static void multi_run(int n_dev, std::vector<int>& rcs, std::vector<std::string>& errs, Work work) {
  auto guarded = [&](int d) {
    try {
      rcs[d] = work(d);
      if (rcs[d]) errs[d] = g_err;          // eg_last_error is per thread: carry the text over to the caller's
    } catch (const std::bad_alloc&) { rcs[d] = EG_ERR_NOMEM; errs[d] = "out of host memory";
    } catch (const std::exception& ex) { rcs[d] = EG_ERR_HIP; errs[d] = std::string("exception: ") + ex.what();
    } catch (...) { rcs[d] = EG_ERR_HIP; errs[d] = "unknown exception"; }
  };
  std::vector<std::thread> threads;
  int started = 0;
  try {
    for (int d = 1; d < n_dev; ++d) { threads.emplace_back(guarded, d); started = d; }
  } catch (const std::exception&) {          // no thread for a slab (resource limits): the caller's thread takes it over
    for (int d = started + 1; d < n_dev; ++d) guarded(d);
  }
  guarded(0);
  for (auto& t : threads) t.join();
}
// A multi call HOLDS its params objects from its first check to its merge or roll-back (Engine::long_call_mu + Engine::reserved): it waits
// for a one-shot JSON call that is running on any of them, REFUSES when an explicitly opened JSON stream owns one (nothing has been touched
// at that point: the stream's tally share and its set-aside running tally are as they were), and every call of another thread on a held
// object waits until the multi call is over (EG_WAIT_LONG_ONLY) - so nothing can interleave between save() and restore() (ADVICE r5, medium:
// a snapshot taken in the middle of another call, or put back over one, lost or misattributed ballots).  The mutexes are taken in address
// order, so two multi calls over the same objects in different orders cannot deadlock.
template <class Params>
struct MultiHold {
  std::vector<Engine*> held;
  int acquire(Params* const* per_device, int n_dev) {
    std::vector<Engine*> order;
    for (int d = 0; d < n_dev; ++d) order.push_back(per_device[d]->eng);
    std::sort(order.begin(), order.end(), std::less<Engine*>());
    for (Engine* e : order) {
      e->long_call_mu.lock();
      std::lock_guard<std::recursive_mutex> g(e->ctx->mu);
      if (e->stream_open) {            // (a one-shot stream cannot be: its call holds long_call_mu)
        e->long_call_mu.unlock();
        release();
        return fail(EG_ERR_BAD_ARG, "a JSON stream is open on one of the params objects (eg_verify_json_end or _abort it first)");
      }
      e->reserved = true;
      held.push_back(e);
    }
    return EG_OK;
  }
  void release() {
    for (auto it = held.rbegin(); it != held.rend(); ++it) {
      { std::lock_guard<std::recursive_mutex> g((*it)->ctx->mu); (*it)->reserved = false; }
      (*it)->long_call_mu.unlock();
    }
    held.clear();
  }
  ~MultiHold() { release(); }
};
// The running tallies before a multi call and the way back to them: a multi call that fails in ANY slab leaves every running tally as it
// found it (a host that retries the batch must not count slabs twice).  The tallies stay ON their devices (Engine::tally_saved3: a
// device-to-device copy of 2n points, no allocation, no encoding), and nothing here drains a device (round 5 did, twice per GPU and call):
// save() orders the copy after the engine's last asynchronous call with an event (Engine::last_done) and leaves an event behind
// (Engine::saved_ev) that the slab's stream waits for; restore() waits for the slab's own stream.  The engines are held (MultiHold), so
// no other call can touch the tallies in between.
template <class Params>
struct TallyRollback {
  Params* const* per_device; int n_dev; size_t bytes;
  std::vector<hipStream_t> slab_stream;      // device form: the stream slab d was enqueued on; host form: empty (the contexts' own streams)
  int save() {
    if (!bytes) return EG_OK;
    for (int d = 0; d < n_dev; ++d) {
      Engine* e = per_device[d]->eng;
      std::lock_guard<std::recursive_mutex> g(e->ctx->mu);
      HIPCHK(hipSetDevice(e->ctx->device));
      hipStream_t s = e->ctx->stream;
      if (e->last_used) HIPCHK(hipStreamWaitEvent(s, e->last_done, 0));      // earlier _device calls on caller streams land in the tally that is set aside
      HIPCHK(hipMemcpyAsync(e->tally_saved3, e->tally, e->plan.tally_slots.size() * PT_WORDS * sizeof(u32), hipMemcpyDeviceToDevice, s));
      HIPCHK(hipEventRecord(e->saved_ev, s));
    }
    return EG_OK;
  }
  // the slab's work starts after the copy above (device form; the host form runs on the context's own stream, behind the copy)
  int order_slab_after_save(int d, hipStream_t s) {
    if (!bytes) return EG_OK;
    HIPCHK(hipStreamWaitEvent(s, per_device[d]->eng->saved_ev, 0));
    return EG_OK;
  }
  int encoded_before(int d, uint8_t* out) {
    Engine* e = per_device[d]->eng;
    std::lock_guard<std::recursive_mutex> g(e->ctx->mu);
    HIPCHK(hipSetDevice(e->ctx->device));
    return engine_tally_encode_from(e, e->tally_saved3, out, false);
  }
  int restore() {
    if (!bytes) return EG_OK;
    int first = EG_OK;
    for (int d = 0; d < n_dev; ++d) {
      Engine* e = per_device[d]->eng;
      std::lock_guard<std::recursive_mutex> g(e->ctx->mu);
      hipStream_t s = e->ctx->stream;
      hipError_t he = hipSetDevice(e->ctx->device);
      // whatever the failed slab still has in flight sits on its stream (engine_verify_device joins its work sets back into it on every
      // way out) or, host form, on the context's own: the copy back goes behind it
      if (he == hipSuccess && !slab_stream.empty() && slab_stream[d] != s) {
        he = hipEventRecord(e->last_done, slab_stream[d]);
        if (he == hipSuccess) he = hipStreamWaitEvent(s, e->last_done, 0);
      }
      if (he == hipSuccess) he = hipMemcpyAsync(e->tally, e->tally_saved3, e->plan.tally_slots.size() * PT_WORDS * sizeof(u32), hipMemcpyDeviceToDevice, s);
      if (he == hipSuccess && e->n_sets == 2)        // a share of the failed call that set 1 still holds must not reach the tally later (this call holds the engine)
        hipLaunchKernelGGL(k_tally_init, dim3(blocks_of(e->plan.tally_slots.size())), dim3(NT), 0, s, e->set[1].tally, (int)e->plan.tally_slots.size());
      if (he == hipSuccess) he = hipEventRecord(e->last_done, s);
      if (he == hipSuccess) { e->last_used = true; he = hipStreamSynchronize(s); }
      if (he != hipSuccess && !first) first = EG_ERR_HIP;
    }
    return first;
  }
};
template <class Params>
static int multi_fail(Params* const* per_device, int n_dev, const std::vector<int>& rcs, const std::vector<std::string>& errs, TallyRollback<Params>& rb) {
  for (int d = 0; d < n_dev; ++d)
    if (rcs[d]) {
      const bool back = rb.restore() == EG_OK;
      return fail(rcs[d], "slab " + std::to_string(d) + " of " + std::to_string(n_dev) + ": " + errs[d] +
                              (back ? " (every running tally is as it was before the call)" : " (AND the running tallies could not be restored: reset them)"));
    }
  return EG_OK;
}
// merges per-slab tally encodings (host) into out with the element addition of the primitive tier on the first context
template <class Params>
static int multi_merge(Params* const* per_device, int n_dev, const std::vector<std::vector<uint8_t>>& tallies, size_t tally_bytes, uint8_t* out) {
  if (!tally_bytes) return EG_OK;
  memcpy(out, tallies[0].data(), tally_bytes);
  std::vector<uint8_t> ok(tally_bytes / 32);
  for (int d = 1; d < n_dev; ++d) {
    TRY(eg_point_add_batch(per_device[0]->eng->ctx, tally_bytes / 32, out, tallies[d].data(), 0, out, ok.data()));
    for (uint8_t o : ok) if (!o) return fail(EG_ERR_HIP, "the tally of slab " + std::to_string(d) + " does not decode");
  }
  return EG_OK;
}
// the entry points' bodies without their wait for long calls: what a multi call, which HOLDS the objects, runs on its slabs
template <class Params>
static int held_verify_host(Params* p, size_t n, const uint8_t* ballots, uint32_t* status, uint8_t* tally_out) { EG_LOCK_P(p);
  return engine_verify_host(p->eng, n, ballots, status, tally_out);
}
template <class Params>
static int held_verify_device(Params* p, size_t n, const void* d_ballots, void* d_status, hipStream_t s) { EG_LOCK_P(p);
  HIPCHK(hipSetDevice(p->eng->ctx->device));
  return engine_verify_device(p->eng, n, d_ballots, d_status, s);
}
template <class Params>
static int verify_batch_multi(Params* const* per_device, int n_dev, size_t n, const uint8_t* ballots, uint32_t* status, uint8_t* tally_out) {
  TRY(multi_check(per_device, n_dev));
  KeepDevice keep;
  if (n && (!ballots || !status)) return fail(EG_ERR_BAD_ARG, "bad argument");
  MultiHold<Params> hold;
  TRY(hold.acquire(per_device, n_dev));
  const size_t stride = per_device[0]->eng->plan.stride;
  const size_t tally_bytes = (size_t)per_device[0]->eng->plan.tally_slots.size() * 32;
  std::vector<std::vector<uint8_t>> tallies(n_dev, std::vector<uint8_t>(tally_bytes));
  std::vector<int> rcs(n_dev, EG_OK);
  std::vector<std::string> errs(n_dev);
  TallyRollback<Params> rb{per_device, n_dev, tally_bytes, {}};
  TRY(rb.save());
  multi_run(n_dev, rcs, errs, [&](int d) {
    const size_t b = n * (size_t)d / (size_t)n_dev, e = n * (size_t)(d + 1) / (size_t)n_dev;
    return held_verify_host(per_device[d], e - b, ballots + b * stride, status + b, tally_out ? tallies[d].data() : nullptr);
  });
  TRY(multi_fail(per_device, n_dev, rcs, errs, rb));
  if (tally_out && tally_bytes) {
    const int rc = multi_merge(per_device, n_dev, tallies, tally_bytes, tally_out);
    if (rc) { const std::string why = g_err; (void)rb.restore(); return fail(rc, why); }
  }
  return EG_OK;
}
// The same with every slab ALREADY RESIDENT on its GPU: d_ballots[d] / d_status[d] are device pointers on per_device[d]'s device, n_per_dev[d]
// ballots each, streams[d] (may be NULL = all null streams) a stream of that device.  One host thread per slab enqueues
// eg_verify_*_batch_device and waits for its stream, so that the call returns with every verdict written and every running tally advanced;
// no byte of a ballot crosses PCIe or xGMI.
template <class Params>
static int verify_batch_multi_device(Params* const* per_device, int n_dev, const size_t* n_per_dev, const void* const* d_ballots,
                                     void* const* d_status, void* const* streams, uint8_t* tally_out) {
  TRY(multi_check(per_device, n_dev));
  KeepDevice keep;
  if (!n_per_dev || !d_ballots || !d_status) return fail(EG_ERR_BAD_ARG, "bad argument");
  for (int d = 0; d < n_dev; ++d)
    if (n_per_dev[d] && (!d_ballots[d] || !d_status[d])) return fail(EG_ERR_BAD_ARG, "null device pointer for a non-empty slab");
  MultiHold<Params> hold;
  TRY(hold.acquire(per_device, n_dev));
  const size_t tally_bytes = (size_t)per_device[0]->eng->plan.tally_slots.size() * 32;
  std::vector<int> rcs(n_dev, EG_OK);
  std::vector<std::string> errs(n_dev);
  TallyRollback<Params> rb{per_device, n_dev, tally_bytes, std::vector<hipStream_t>(n_dev, nullptr)};
  for (int d = 0; d < n_dev; ++d) rb.slab_stream[d] = streams ? (hipStream_t)streams[d] : nullptr;
  TRY(rb.save());
  multi_run(n_dev, rcs, errs, [&](int d) {
    hipStream_t s = rb.slab_stream[d];
    HIPCHK(hipSetDevice(per_device[d]->eng->ctx->device));
    TRY(rb.order_slab_after_save(d, s));
    TRY(held_verify_device(per_device[d], n_per_dev[d], d_ballots[d], d_status[d], s));
    HIPCHK(hipStreamSynchronize(s));
    return (int)EG_OK;
  });
  TRY(multi_fail(per_device, n_dev, rcs, errs, rb));
  if (tally_out && tally_bytes) {
    // the batch's own tally = (running tally after) - (running tally before), slab by slab, merged: all on 64 n_options bytes per device
    std::vector<std::vector<uint8_t>> tallies(n_dev, std::vector<uint8_t>(tally_bytes));
    std::vector<uint8_t> ok(tally_bytes / 32), before(tally_bytes);
    int rc = EG_OK;
    for (int d = 0; d < n_dev && !rc; ++d) {
      { std::lock_guard<std::recursive_mutex> g(per_device[d]->eng->ctx->mu);       // (every slab's stream has been waited for above)
        rc = hipSetDevice(per_device[d]->eng->ctx->device) == hipSuccess ? engine_tally_encode_from(per_device[d]->eng, per_device[d]->eng->tally, tallies[d].data(), false)
                                                                         : fail(EG_ERR_HIP, "hipSetDevice"); }
      if (!rc) rc = rb.encoded_before(d, before.data());
      if (!rc) rc = eg_point_add_batch(per_device[d]->eng->ctx, tally_bytes / 32, tallies[d].data(), before.data(), 1, tallies[d].data(), ok.data());
      if (!rc) for (uint8_t o : ok) if (!o) rc = fail(EG_ERR_HIP, "the tally of slab " + std::to_string(d) + " does not decode");
    }
    if (!rc) rc = multi_merge(per_device, n_dev, tallies, tally_bytes, tally_out);
    if (rc) { const std::string why = g_err; (void)rb.restore(); return fail(rc, why); }
  }
  return EG_OK;
}
// sum of the RUNNING tallies of the params objects (each keeps the tally of the slabs it verified); the objects are held while they are
// read, so the sum is the tally of whole calls only
template <class Params>
static int tally_encode_multi(Params* const* per_device, int n_dev, uint8_t* out) {
  TRY(multi_check(per_device, n_dev));
  KeepDevice keep;
  if (!out) return fail(EG_ERR_BAD_ARG, "bad argument");
  MultiHold<Params> hold;
  TRY(hold.acquire(per_device, n_dev));
  const size_t tally_bytes = (size_t)per_device[0]->eng->plan.tally_slots.size() * 32;
  std::vector<std::vector<uint8_t>> tallies(n_dev, std::vector<uint8_t>(tally_bytes));
  for (int d = 0; d < n_dev; ++d) {
    Params* p = per_device[d];
    EG_LOCK_P(p);
    TRY(engine_tally_encode(p->eng, tallies[d].data()));
  }
  return multi_merge(per_device, n_dev, tallies, tally_bytes, out);
}
}  // extern "C++"
int eg_verify_choice_batch_multi(eg_choice_params* const* per_device, int n_dev, size_t n, const uint8_t* ballots, uint32_t* status,
                                 uint8_t* tally_out) {
  return verify_batch_multi(per_device, n_dev, n, ballots, status, tally_out);
}
int eg_verify_qv_batch_multi(eg_qv_params* const* per_device, int n_dev, size_t n, const uint8_t* ballots, uint32_t* status,
                             uint8_t* tally_out) {
  return verify_batch_multi(per_device, n_dev, n, ballots, status, tally_out);
}
int eg_verify_choice_batch_multi_device(eg_choice_params* const* per_device, int n_dev, const size_t* n_per_dev, const void* const* d_ballots,
                                        void* const* d_status, void* const* streams, uint8_t* tally_out) {
  return verify_batch_multi_device(per_device, n_dev, n_per_dev, d_ballots, d_status, streams, tally_out);
}
int eg_verify_qv_batch_multi_device(eg_qv_params* const* per_device, int n_dev, const size_t* n_per_dev, const void* const* d_ballots,
                                    void* const* d_status, void* const* streams, uint8_t* tally_out) {
  return verify_batch_multi_device(per_device, n_dev, n_per_dev, d_ballots, d_status, streams, tally_out);
}
int eg_choice_tally_encode_multi(eg_choice_params* const* per_device, int n_dev, uint8_t* out) { return tally_encode_multi(per_device, n_dev, out); }
int eg_qv_tally_encode_multi(eg_qv_params* const* per_device, int n_dev, uint8_t* out) { return tally_encode_multi(per_device, n_dev, out); }

// ---- PublicKey::verify_zero / verify_bool / verify_range in batches ---------------------------------------
int eg_proof_params_create(eg_ctx* c, const uint8_t pk[32], int kind, uint64_t upper_bound, eg_proof_params** out) { EG_LOCK(c);
  if (!c || !pk || !out) return fail(EG_ERR_BAD_ARG, "bad argument");
  Engine* e = nullptr;
  size_t item = 0;
  if (kind == EG_PROOF_ZERO) { TRY(engine_create(c, eghost::build_zero_plan(), pk, 0, &e)); item = 128; }
  else if (kind == EG_PROOF_BOOL) { TRY(engine_create(c, eghost::build_bool_plan(), pk, 0, &e)); item = 160; }
  else if (kind == EG_PROOF_RANGE) {
    if (upper_bound < 2 || upper_bound > 1000000) return fail(EG_ERR_BAD_ARG, "upper_bound must be in 2..1000000");
    TRY(engine_create(c, eghost::build_range_plan(upper_bound, &item), pk, 0, &e));
  } else return fail(EG_ERR_BAD_ARG, "unknown proof kind");
  *out = new eg_proof_params{e, kind, item};
  return EG_OK;
}
int eg_share_params_create(eg_ctx* c, const uint8_t shared_key[32], uint64_t shares, uint64_t threshold, uint64_t index,
                           const uint8_t participant_key[32], eg_proof_params** out) { EG_LOCK(c);
  if (!c || !shared_key || !participant_key || !out) return fail(EG_ERR_BAD_ARG, "bad argument");
  if (shares < 1 || threshold < 1 || threshold > shares || index >= shares) return fail(EG_ERR_BAD_ARG, "bad sharing parameters");
  Engine* e = nullptr;
  TRY(engine_create(c, eghost::build_share_plan(shares, threshold, shared_key, index), participant_key, 0, &e));
  *out = new eg_proof_params{e, EG_PROOF_SHARE, 128};
  return EG_OK;
}
int eg_sumsq_params_create(eg_ctx* c, const uint8_t pk[32], int n_values, const char* label, size_t label_len, eg_proof_params** out) { EG_LOCK(c);
  if (!c || !pk || !out || (!label && label_len)) return fail(EG_ERR_BAD_ARG, "bad argument");
  if (n_values < 1 || n_values > 1000 || label_len > 255) return fail(EG_ERR_BAD_ARG, "n_values in 1..1000, label up to 255 bytes");
  Engine* e = nullptr;
  size_t item = 0;
  TRY(engine_create(c, eghost::build_sumsq_plan(n_values, std::string(label ? label : "", label_len), &item), pk, 0, &e));
  *out = new eg_proof_params{e, EG_PROOF_SUMSQ, item};
  return EG_OK;
}
void eg_proof_params_destroy(eg_proof_params* p) { params_destroy(p); }
size_t eg_proof_item_size(const eg_proof_params* p) { return p ? p->item_size : 0; }
int eg_verify_proof_batch(eg_proof_params* p, size_t n, const uint8_t* items, uint32_t* status) { EG_LOCK_P(p);
  if (!p || (n && (!items || !status))) return fail(EG_ERR_BAD_ARG, "bad argument");
  return engine_verify_host(p->eng, n, items, status, nullptr);
}
int eg_verify_proof_batch_device(eg_proof_params* p, size_t n, const void* d_items, void* d_status, void* stream) { EG_LOCK_P(p);
  if (!p || (n && (!d_items || !d_status))) return fail(EG_ERR_BAD_ARG, "bad argument");
  HIPCHK(hipSetDevice(p->eng->ctx->device));
  return engine_verify_device(p->eng, n, d_items, d_status, (hipStream_t)stream);
}

// ---- host-only introspection (no GPU needed): the product's own RangeDecomposition and plan builders ----------------------------
int eg_range_decomposition(uint64_t upper_bound, char* buf, size_t cap) {
  if (upper_bound < 2 || !buf || cap == 0) return fail(EG_ERR_BAD_ARG, "upper_bound must be >= 2");   // range.rs:160
  const std::string t = eghost::optimal_range(upper_bound).to_string();
  if (t.size() + 1 > cap) return fail(EG_ERR_BAD_ARG, "buffer too small");
  memcpy(buf, t.c_str(), t.size() + 1);
  return EG_OK;
}
int eg_plan_describe(int kind, int n_options, uint64_t credits_or_bound, char* buf, size_t cap) {
  if (!buf || cap == 0) return fail(EG_ERR_BAD_ARG, "bad argument");
  eghost::Plan P;
  size_t item = 0;
  switch (kind) {
    case 0: P = eghost::build_choice_plan(n_options, true, choice_ring_group()); break;
    case 1: P = eghost::build_choice_plan(n_options, false, choice_ring_group()); break;
    case 2: P = eghost::build_qv_plan(n_options, credits_or_bound); break;
    case 3: P = eghost::build_zero_plan(); break;
    case 4: P = eghost::build_bool_plan(); break;
    case 5: P = eghost::build_range_plan(credits_or_bound, &item); break;
    case 6: P = eghost::build_sumsq_plan(n_options, "test", &item); break;
    default: return fail(EG_ERR_BAD_ARG, "unknown plan kind");
  }
  size_t jobs = 0, insts = 0, var_terms = P.vterms.size(), table_terms = 0, derived = 0;
  size_t combs = 0, deferred = 0, plain_encodes = 0, inversion_groups = 0, derive_terms = P.dterms.size();
  size_t jobs_table1 = 0, chains = 0, chain_extra_terms = 0, loose_table_terms = 0, direct_terms = 0;
  std::string per_stage;
  for (auto& st : P.stages) {
    jobs += st.jobs.size(); insts += st.insts.size();
    per_stage += (per_stage.empty() ? "" : ",") + std::to_string(st.jobs.size());
    for (auto& j : st.jobs) {
      combs += (j.g.kind != egplan::SRC_NONE) + (j.k.kind != egplan::SRC_NONE);
      if (j.defer) ++deferred; else ++plain_encodes;
      switch (job_family(j, P.vterms)) {
        case FAM_TABLE1: ++jobs_table1; break;
        case FAM_TABLEN:
          for (int t0 = 0; t0 < (int)j.term_count; t0 += EG_MULTI_GROUP) {
            ++chains;
            chain_extra_terms += std::min<int>(EG_MULTI_GROUP, j.term_count - t0) - 1;
          }
          break;
        case FAM_DIRECT1: ++direct_terms; break;
        case FAM_GENERIC:
          for (unsigned t = 0; t < j.term_count; ++t)
            if (P.vterms[j.term_first + t].base == 0xffff) ++direct_terms; else ++loose_table_terms;
          break;
        default: break;
      }
    }
    inversion_groups += (st.deferred.size() + 31) / 32;
  }
  for (auto& t : P.vterms) table_terms += t.base != 0xffff;
  for (auto& l : P.derive_levels) derived += l.size();
  char tmp[2048];
  snprintf(tmp, sizeof tmp,
           "{\"stride\": %zu, \"wire_points\": %zu, \"wire_scalars\": %zu, \"derived_points\": %zu, \"derive_terms\": %zu, \"bases\": %zu, "
           "\"stages\": %zu, \"jobs\": %zu, \"jobs_per_stage\": [%s], \"var_terms\": %zu, \"table_terms\": %zu, "
           "\"combs\": %zu, \"deferred\": %zu, \"plain_encodes\": %zu, \"inversion_groups\": %zu, "
           "\"single_table_jobs\": %zu, \"chains\": %zu, \"chain_extra_terms\": %zu, \"loose_table_terms\": %zu, \"direct_terms\": %zu, "
           "\"sum_tables\": %zu, \"sum_table_members\": %zu, "
           "\"hash_programs\": %zu, \"prefixes\": %d, \"flags\": %d, \"rules\": %zu, \"tally_slots\": %zu, "
           "\"pt_slots\": %d, \"cmp_slots\": %d, \"chal_slots\": %d, \"state_slots\": %d, \"tables\": %zu, \"teeth\": %d, "
           "\"ring_group\": %d, \"table_groups\": %zu}",
           P.stride, P.pt_items.size(), P.sc_items.size(), derived, derive_terms, P.base_slots.size(), P.stages.size(), jobs, per_stage.c_str(),
           var_terms, table_terms, combs, deferred, plain_encodes, inversion_groups, jobs_table1, chains, chain_extra_terms,
           loose_table_terms, direct_terms, P.sum_bases.size(), P.sum_members.size(), insts, P.n_prefixes, P.n_flag_slots, P.rules.size(),
           P.tally_slots.size(), P.n_pt_slots, P.n_cmp_slots, P.n_chal_slots, P.n_state_slots, (size_t)P.n_tables(), eghost::plan_teeth(P),
           P.ring_group, P.group_stage.size());
  if (strlen(tmp) + 1 > cap) return fail(EG_ERR_BAD_ARG, "buffer too small");
  memcpy(buf, tmp, strlen(tmp) + 1);
  return EG_OK;
}

// ---- wire ingest: serde's human-readable (JSON, base64url) layout -> packed ballots; host only (wire_json.hpp) --------------------
static int pack_json_common(const char* json, size_t json_len, int threads, size_t max_objects, size_t* n_objects,
                            std::vector<std::pair<size_t, size_t>>& spans) {
  if (!json_len) { if (n_objects) *n_objects = 0; return EG_OK; }
  if (!egwire::split_objects_parallel(json, json_len, threads, spans))
    return fail(EG_ERR_BAD_ARG, "the text is neither a JSON array of objects nor a sequence of JSON objects");
  if (n_objects) *n_objects = spans.size();
  if (spans.size() > max_objects) return fail(EG_ERR_BAD_ARG, "more objects in the text than max_objects (*n_objects says how many)");
  return EG_OK;
}
int eg_choice_pack_json(int n_options, int single, const char* json, size_t json_len, int threads, size_t max_objects,
                        uint8_t* packed, uint32_t* status, size_t* n_objects) {
  if (n_options < 1 || n_options > 4000 || (json_len && !json) || (max_objects && (!packed || !status)))
    return fail(EG_ERR_BAD_ARG, "bad argument");
  std::vector<std::pair<size_t, size_t>> spans;
  TRY(pack_json_common(json, json_len, threads, max_objects, n_objects, spans));
  const size_t stride = eghost::choice_ballot_size(n_options, single != 0);
  egwire::pack_parallel(json, spans, stride, threads, packed, status,
                        [&](egwire::Cursor& c, uint8_t* dst) { return egwire::pack_choice(c, n_options, single != 0, dst); });
  return EG_OK;
}
int eg_qv_pack_json(int n_options, uint64_t credits, const char* json, size_t json_len, int threads, size_t max_objects,
                    uint8_t* packed, uint32_t* status, size_t* n_objects) {
  if (n_options < 1 || n_options > 256 || credits < 1 || credits > 100000 || (json_len && !json) || (max_objects && (!packed || !status)))
    return fail(EG_ERR_BAD_ARG, "bad argument");
  std::vector<std::pair<size_t, size_t>> spans;
  TRY(pack_json_common(json, json_len, threads, max_objects, n_objects, spans));
  const eghost::QvShape sh = eghost::qv_shape(n_options, credits);
  const egwire::RangeShape vote{sh.vote_range.rings.size(), (size_t)sh.vote_range.rings_size()};
  const egwire::RangeShape credit{sh.credit_range.rings.size(), (size_t)sh.credit_range.rings_size()};
  egwire::pack_parallel(json, spans, sh.ballot_size, threads, packed, status, [&](egwire::Cursor& c, uint8_t* dst) {
    return egwire::pack_qv(c, n_options, vote, credit, sh.ballot_size, dst);
  });
  return EG_OK;
}
// JSON text -> verdicts + tally in one call.  The text is cut into windows (egwire::split_next); a producer thread splits and packs
// them on a pool of `threads` host threads into a pinned ring of packed ballots, while this thread uploads the finished windows and
// enqueues their verification, two submissions in flight (verify_json_common below).  Objects that do not pack keep their pack verdict;
// their zeroed slots are verified like any other ballot (an all-zero ballot never verifies, so nothing of it reaches the tally) and the
// verdict is overwritten afterwards.
typedef std::function<void(const char*, const std::vector<std::pair<size_t, size_t>>&, int, uint8_t*, uint32_t*, egwire::WorkerPool*)> PackPieceFn;
// verdicts of the objects that deserialise but do not have the election's shape (egwire::resolve_*_objects): false = a GPU call failed
typedef std::function<bool(const char*, const std::vector<std::pair<size_t, size_t>>&, std::vector<uint32_t>&)> ReshapeFn;
// the two GPU services of the object path (wire_json.hpp): validity of 32-byte items, and the batch verifier on substitute ballots
static egwire::CheckItemsFn make_check_items(eg_ctx* c) {
  return [c](const std::string& kinds, const egwire::Bytes& data, std::vector<uint8_t>& ok) {
    egwire::Bytes pts, scs;
    for (size_t i = 0; i < kinds.size(); ++i) {
      egwire::Bytes& dst = kinds[i] == 'P' ? pts : scs;
      dst.insert(dst.end(), data.begin() + 32 * i, data.begin() + 32 * (i + 1));
    }
    std::vector<uint8_t> pok(pts.size() / 32), sok(scs.size() / 32), tmp(pts.size());
    if (!pok.empty() && eg_point_roundtrip_batch(c, pok.size(), pts.data(), tmp.data(), pok.data())) return false;
    if (!sok.empty() && eg_scalar_is_canonical_batch(c, sok.size(), scs.data(), sok.data())) return false;
    ok.resize(kinds.size());
    size_t pi = 0, si = 0;
    for (size_t i = 0; i < kinds.size(); ++i) ok[i] = kinds[i] == 'P' ? pok[pi++] : sok[si++];
    return true;
  };
}
static int engine_verify_host(Engine* e, size_t n, const uint8_t* ballots, uint32_t* status, uint8_t* tally_out);
static egwire::VerifyPackedFn make_verify_packed(Engine* e) {
  return [e](size_t n, const egwire::Bytes& packed, std::vector<uint32_t>& status) {
    status.assign(n, 0);
    return engine_verify_host(e, n, packed.data(), status.data(), nullptr) == EG_OK;
  };
}
// ---- the JSON text in PIECES: eg_verify_{choice,qv}_json_begin[_multi] / eg_verify_json_feed / _feed_owned / _take / _end / _abort --------------
// (examples/voting.rs:195-198 emits ballots one at a time; src/serde.rs:19-80 is the layout.)  ONE worker thread per stream takes the
// pieces in order (stream_worker): it cuts a piece (egwire::StreamSplitter: values may straddle pieces), packs its complete ballots on
// the stream's pool of host threads into a pinned ring (stream_emit), and then PUMPS the GPU side without ever waiting for it
// (stream_pump) - retire the submissions that have landed, enqueue what has piled up (upload, verification on the two work sets, download
// of the verdicts).  The first submission goes once 2^14 ballots are packed (EG_JSON_FIRST_MIN; measured, profiles/r05_json_stream_probe.txt:
// 2^13 ... 2^14 ballots 0.91 of the HBM-resident rate, 2^15 0.90, 2^17 0.86 - waiting longer idles the GPU for longer than the small first
// launches cost), later ones when they are 1.5 x the one in flight, at most two in flight; the worker only waits for the GPU when the ring
// is full.  The one-shot entries (eg_verify_*_json[_multi]) are this pipeline fed with the whole text in place.  Between begin and end the
// params objects belong to the stream (other verify / tally calls on them are refused).
// SEVERAL GPUs (round 6; VERDICT r5 task 4a: the parser delivers 10 M ballots/s, one GPU takes 6): a stream has one LANE per params object -
// its own pinned ring, device staging, control streams and submissions in flight - and ONE splitter, pool and worker; every packed window
// goes to the lane with the fewest ballots waiting or in flight (a faster GPU, or one that started earlier, simply gets more windows), the
// verdicts are kept by object index (text order), and every params object tallies the windows it verified.
struct eg_json_stream {
  struct Region { size_t first, off, m; bool submitted; };
  struct Group { size_t n_regions, first, off, m; hipEvent_t uploaded, done; };
  struct Lane {
    Engine* e = nullptr;
    size_t cap = 0, first_min = 0, n_submitted = 0, pending = 0;   // pending: ballots in regions that have not been retired yet
    int n_ctl = 1;
    bool set_aside = false;
    std::deque<Region> regions;          // in text order; the front is the oldest one not yet retired
    std::deque<Group> groups;            // submissions in flight, oldest first
  };
  std::vector<Lane> lanes;               // one per params object; lanes[0]'s engine also resolves the ballots of another shape
  std::vector<eg_ctx*> ctxs;             // the lanes' contexts, each once, in address order: locked together (LockAll)
  Engine* e = nullptr;                   // = lanes[0].e
  int threads = 1, ns = 0;
  PackPieceFn pack_piece;
  ReshapeFn reshape;
  std::unique_ptr<egwire::WorkerPool> pool;
  std::unique_ptr<egwire::StreamSplitter> split;
  size_t stride = 0, growth = 150, emitted = 0;
  std::vector<uint32_t> verdicts;      // by object: its pack verdict until (if it packed) the GPU's verdict lands
  std::vector<uint32_t> pack_tmp;
  std::vector<std::string> odd_text;   // objects of another shape than the election's: resolved at the end (the object path needs the GPU to itself)
  std::vector<size_t> odd_at;
  size_t taken = 0;
  bool flushed = false, trace = false;
  bool one_shot = false;               // opened by eg_verify_*_json for the length of that call: other calls on the params objects wait for it
  size_t max_objects = (size_t)-1;     // one-shot: the room in the caller's status buffer; a text with more objects fails at once (stream_emit)
  size_t odd_bytes = 0, odd_max = (size_t)256 << 20;     // text of the ballots of another shape kept for the object path, and its bound (EG_JSON_ODD_MAX_MB)
  std::atomic<int> failed{EG_OK};      // set once (stream_fail), after err has been written
  std::string err;
  // Front end: the caller's pieces reach the worker thread through a short queue.  Pieces below `direct_min` are copied into blocks of
  // `block_bytes` (the caller's thread pays one memcpy and returns; cutting, packing and GPU submission happen on the worker thread, in
  // parallel with the caller producing the next piece); larger pieces are handed over in place and feed waits until the worker is through
  // with them.  Everything above (lanes, verdicts, splitter) is touched by the worker thread only, under the contexts' locks, until the
  // worker has been joined (end / abort) - take() takes those locks too.
  struct Item { std::vector<char> own; const char* ptr = nullptr; size_t len = 0; uint64_t id = 0; bool finish = false;
                eg_json_release_fn release = nullptr; void* user = nullptr; };      // release: a block handed over by eg_verify_json_feed_owned
  std::thread worker;
  std::mutex qmu;
  std::condition_variable q_push, q_pop;
  std::deque<Item> queue;
  std::vector<std::vector<char>> spare;    // emptied blocks
  std::vector<char> acc;                   // the block being filled by feed
  uint64_t next_id = 1, consumed_id = 0;
  size_t queued_bytes = 0;                 // text waiting in the queue (copied blocks and handed-over ones): at most ~64 MB, then feed waits
  bool stop = false, worker_joined = false;
  std::atomic<size_t> objects{0};          // complete objects cut so far (by the worker)
  size_t block_bytes = (size_t)16 << 20, direct_min = (size_t)8 << 20;
  // handed-over blocks joined per piece, and the text that may wait in the queue (8 / 32 / 64 MB pieces measured the same within the
  // +-4 % run-to-run noise of the stream: gpurun_out/json_owned_join_ab.txt, round 6)
  size_t join_bytes = (size_t)16 << 20, queue_bytes = (size_t)64 << 20;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();      // EG_JSON_TRACE: the timeline on stderr, ms since begin
  double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};
// every context of a stream's lanes, locked together in address order (one context: exactly the lock every entry point takes); the calling
// thread's current device, which the work on the lanes changes, is put back when the locks go
struct LockAll {
  KeepDevice keep;
  std::vector<std::unique_lock<std::recursive_mutex>> held;
  explicit LockAll(const std::vector<eg_ctx*>& ctxs) { for (eg_ctx* c : ctxs) held.emplace_back(c->mu); }
};
static bool stream_is_one_shot(const eg_json_stream* S) { return S->one_shot; }
static int stream_fail(eg_json_stream* S, int code, const std::string& msg) {
  if (!S->failed.load(std::memory_order_acquire)) { S->err = msg; S->failed.store(code, std::memory_order_release); }
  return fail(S->failed.load(), S->err);
}
typedef eg_json_stream::Lane StreamLane;
static void stream_retire_oldest(eg_json_stream* S, StreamLane& L) {     // the lane's oldest submission has landed: its verdicts, its part of the ring
  Engine* e = L.e;
  const eg_json_stream::Group g = L.groups.front();
  L.groups.pop_front();
  for (size_t k = 0; k < g.n_regions; ++k) {
    const eg_json_stream::Region r = L.regions.front();
    L.regions.pop_front();
    L.pending -= r.m;
    for (size_t i = 0; i < r.m; ++i)
      if (S->verdicts[r.first + i] == EG_ST_OK) S->verdicts[r.first + i] = e->json_status_ring[r.off + i];
  }
  (void)hipEventDestroy(g.uploaded); (void)hipEventDestroy(g.done);
  if (S->trace) fprintf(stderr, "[json stream] %8.2f ms  landed    %zu ballots from %zu (lane %d)\n", S->ms(), g.m, g.first, (int)(&L - S->lanes.data()));
}
// objects whose verdicts are final as far as the GPUs are concerned: everything before the oldest region any lane still holds
static size_t stream_landed(const eg_json_stream* S) {
  size_t upto = S->emitted;
  for (const StreamLane& L : S->lanes)
    if (!L.regions.empty()) upto = std::min(upto, L.regions.front().first);
  return upto;
}
// enqueue what has piled up on a lane, if it is time (or `force`: the ring is full, or the text has ended); never waits for the GPU
static int stream_pump(eg_json_stream* S, StreamLane& L, bool force) {
  Engine* e = L.e;
  if (hipSetDevice(e->ctx->device) != hipSuccess) return stream_fail(S, EG_ERR_HIP, "hipSetDevice");
  for (;;) {
    while (!L.groups.empty() && hipEventQuery(L.groups.front().done) == hipSuccess) stream_retire_oldest(S, L);
    (void)hipGetLastError();                           // a submission still running reads as hipErrorNotReady: not an error to keep
    if (L.groups.size() >= 2) return EG_OK;
    eg_json_stream::Group g{0, 0, 0, 0, nullptr, nullptr};
    for (auto& r : L.regions) {                        // the run of packed windows behind the submitted ones, contiguous in the ring
      if (r.submitted) continue;
      if (g.n_regions && r.off != g.off + g.m) break;
      if (!g.n_regions) { g.first = r.first; g.off = r.off; }
      ++g.n_regions; g.m += r.m;
    }
    if (!g.n_regions) return EG_OK;
    bool go = force;
    if (!go && L.groups.empty()) go = L.n_submitted ? true : g.m >= L.first_min;       // an idle GPU takes whatever there is - except the very first time
    if (!go && L.groups.size() == 1) go = g.m * 100 >= L.groups.back().m * S->growth && g.m >= std::min(L.first_min, L.cap / 8);
    if (!go) return EG_OK;
    hipStream_t ctl = e->json_ctl[L.n_submitted % (size_t)L.n_ctl];
    const char* what = "window upload: ";
    hipError_t he = hipEventCreateWithFlags(&g.uploaded, hipEventDisableTiming);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&g.done, hipEventDisableTiming | hipEventBlockingSync);
    if (he == hipSuccess) he = hipMemcpyAsync(e->d_wire + g.off * S->stride, e->json_ring + g.off * S->stride, g.m * S->stride, hipMemcpyHostToDevice, e->copy_stream);
    if (he == hipSuccess) he = hipEventRecord(g.uploaded, e->copy_stream);
    if (he == hipSuccess) he = hipStreamWaitEvent(ctl, g.uploaded, 0);
    int rc = EG_OK;
    if (he == hipSuccess) {
      rc = engine_verify_device(e, g.m, e->d_wire + g.off * S->stride, e->d_status + g.off, ctl, VD_FORCE_SETS | VD_KEEP_SET_TALLY);
      what = "verdict download: ";
      if (rc == EG_OK) he = hipMemcpyAsync(e->json_status_ring + g.off, e->d_status + g.off, g.m * sizeof(u32), hipMemcpyDeviceToHost, ctl);
      if (rc == EG_OK && he == hipSuccess) he = hipEventRecord(g.done, ctl);
    }
    if (he != hipSuccess) rc = fail(EG_ERR_HIP, std::string(what) + hipGetErrorString(he));
    if (rc) {
      if (g.uploaded) (void)hipEventDestroy(g.uploaded);
      if (g.done) (void)hipEventDestroy(g.done);
      return stream_fail(S, rc, g_err);
    }
    size_t k = 0;
    for (auto& r : L.regions) { if (r.submitted) continue; if (k++ == g.n_regions) break; r.submitted = true; }
    L.groups.push_back(g);
    ++L.n_submitted;
    if (S->trace) fprintf(stderr, "[json stream] %8.2f ms  submitted %zu windows, %zu ballots from %zu (lane %d, %zu in flight)\n", S->ms(), g.n_regions, g.m, g.first, (int)(&L - S->lanes.data()), L.groups.size());
  }
}
static int stream_pump_all(eg_json_stream* S, bool force) {
  for (StreamLane& L : S->lanes) TRY(stream_pump(S, L, force));
  return EG_OK;
}
// room for m packed ballots in a lane's ring (behind its newest region, or from the start again once the oldest regions there have been
// retired); waits for the GPU only when there is none
static int stream_place(eg_json_stream* S, StreamLane& L, size_t m, size_t* off) {
  for (;;) {
    if (L.regions.empty()) { *off = 0; return EG_OK; }
    const size_t head = L.regions.back().off + L.regions.back().m, tail = L.regions.front().off;
    if (head > tail) {                            // the occupied part does not wrap
      if (head + m <= L.cap) { *off = head; return EG_OK; }
      if (m <= tail) { *off = 0; return EG_OK; }
    } else if (head + m <= tail) { *off = head; return EG_OK; }
    if (L.groups.empty()) {
      TRY(stream_pump(S, L, true));               // the ring is full of packed ballots nobody has submitted yet
      if (L.groups.empty()) return stream_fail(S, EG_ERR_NOMEM, "a window of ballots does not fit the staging ring");
    }
    (void)hipSetDevice(L.e->ctx->device);
    const hipError_t he = hipEventSynchronize(L.groups.front().done);
    if (he != hipSuccess) return stream_fail(S, EG_ERR_HIP, std::string("window: ") + hipGetErrorString(he));
    stream_retire_oldest(S, L);
  }
}
// the complete values of one window of the text: pack them into a lane's ring, remember the ones that need the object path, pump the GPUs
static bool stream_emit(eg_json_stream* S, const char* base, const std::vector<std::pair<size_t, size_t>>& spans, size_t first) {
  if (first + spans.size() > S->max_objects) {
    stream_fail(S, EG_ERR_BAD_ARG, "more objects in the text than max_objects");
    return false;
  }
  // the lane with the least work waiting or in flight whose ring has room without waiting; if none has, the least loaded one (it waits)
  StreamLane* best = nullptr;
  for (StreamLane& L : S->lanes)
    if (L.pending + spans.size() <= L.cap && (!best || L.pending < best->pending)) best = &L;
  if (!best)
    for (StreamLane& L : S->lanes)
      if (!best || L.pending < best->pending) best = &L;
  StreamLane& L = *best;
  Engine* e = L.e;
  size_t off = 0;
  if (stream_place(S, L, spans.size(), &off)) return false;
  S->pack_tmp.resize(spans.size());
  S->pack_piece(base, spans, S->threads, e->json_ring + off * S->stride, S->pack_tmp.data(), S->pool.get());
  if (S->verdicts.size() < first + spans.size()) S->verdicts.resize(first + spans.size());
  for (size_t i = 0; i < spans.size(); ++i) {
    S->verdicts[first + i] = S->pack_tmp[i];
    if (S->pack_tmp[i] == EG_PACK_RESHAPE) {
      // a ballot of another shape than the election's keeps its TEXT until the end (the object path needs the GPU to itself): bounded, so
      // that a text made of such objects cannot make the streaming entry buffer its whole input in host memory
      S->odd_bytes += spans[i].second;
      if (S->odd_bytes > S->odd_max) {
        stream_fail(S, EG_ERR_NOMEM, "the ballots whose shape is not the election's exceed " + std::to_string(S->odd_max >> 20) +
                                         " MB of text (EG_JSON_ODD_MAX_MB): verify such a text through the object path in smaller pieces");
        return false;
      }
      S->odd_text.emplace_back(base + spans[i].first, spans[i].second); S->odd_at.push_back(first + i);
    }
  }
  L.regions.push_back({first, off, spans.size(), false});
  L.pending += spans.size();
  S->emitted = first + spans.size();
  return stream_pump_all(S, false) == EG_OK;
}
static void stream_release(eg_json_stream* S, bool keep_partial_tally) {      // the engines go back to their owners; the stream is deleted
  for (StreamLane& L : S->lanes) {
    Engine* e = L.e;
    (void)hipSetDevice(e->ctx->device);
    (void)hipDeviceSynchronize();                         // nothing may still read the ring or the work sets
    for (auto& g : L.groups) { if (g.uploaded) (void)hipEventDestroy(g.uploaded); if (g.done) (void)hipEventDestroy(g.done); }
    L.groups.clear();
    hipStream_t s = e->ctx->stream;
    if (S->ns && e->n_sets == 2) {
      if (keep_partial_tally) hipLaunchKernelGGL(k_tally_add_points, dim3(blocks_of((size_t)S->ns)), dim3(NT), 0, s, e->set[1].tally, S->ns, e->set[0].tally);
      hipLaunchKernelGGL(k_tally_init, dim3(blocks_of((size_t)S->ns)), dim3(NT), 0, s, e->set[1].tally, S->ns);
    }
    if (L.set_aside && S->ns) {
      // a finished stream: running tally = what it was + the stream's ballots on this lane; an aborted or failed one: what it was
      if (keep_partial_tally) hipLaunchKernelGGL(k_tally_add_points, dim3(blocks_of((size_t)S->ns)), dim3(NT), 0, s, e->tally_saved2, S->ns, e->tally);
      else (void)hipMemcpyAsync(e->tally, e->tally_saved2, (size_t)S->ns * PT_WORDS * sizeof(u32), hipMemcpyDeviceToDevice, s);
    }
    (void)hipStreamSynchronize(s);
    e->stream_open = nullptr;
  }
  delete S;
}
static void stream_worker(eg_json_stream* S);
// size_hint: bytes of text to come if the caller knows (the one-shot entry does: a short text gets a short ring), else 0.  Called with every
// engine's context locked (LockAll) and no stream open on any of them.
static int stream_begin(const std::vector<Engine*>& engines, int threads, PackPieceFn pack_piece, ReshapeFn reshape, size_t size_hint, eg_json_stream** out) {
  if (!out || engines.empty()) return fail(EG_ERR_BAD_ARG, "bad argument");
  *out = nullptr;
  for (Engine* e : engines)
    if (e->stream_open) return fail(EG_ERR_BAD_ARG, "a JSON stream is already open on this params object");
  std::unique_ptr<eg_json_stream> S(new eg_json_stream());
  Engine* e0 = engines[0];
  S->e = e0; S->threads = std::max(threads, 1); S->pack_piece = std::move(pack_piece); S->reshape = std::move(reshape);
  S->stride = e0->plan.stride; S->ns = (int)e0->plan.tally_slots.size();
  S->growth = e0->knobs.json_growth; S->trace = e0->knobs.json_trace;
  S->odd_max = e0->knobs.json_odd_max_mb << 20;
  const size_t n_lanes = engines.size();
  S->lanes.resize(n_lanes);
  // if anything below fails, the lanes that were already set up get their running tallies back
  size_t lanes_ready = 0;
  ScopeExit undo{[&]() {
    if (!S) return;                              // released to the caller: success
    for (size_t k = 0; k < lanes_ready; ++k) {
      Engine* e = S->lanes[k].e;
      if (!S->lanes[k].set_aside || !S->ns) continue;
      (void)hipSetDevice(e->ctx->device);
      (void)hipMemcpyAsync(e->tally, e->tally_saved2, (size_t)S->ns * PT_WORDS * sizeof(u32), hipMemcpyDeviceToDevice, e->ctx->stream);
      (void)hipStreamSynchronize(e->ctx->stream);
    }
  }};
  for (size_t k = 0; k < n_lanes; ++k) {
    Engine* e = engines[k];
    StreamLane& L = S->lanes[k];
    L.e = e;
    HIPCHK(hipSetDevice(e->ctx->device));
    hipStream_t s = e->ctx->stream;
    HIPCHK(hipDeviceSynchronize());
    // one GiB of pinned staging for one GPU; several GPUs share that budget (at least 128 MiB each)
    const size_t ring_max = std::max((e->knobs.json_ring_kb ? e->knobs.json_ring_kb << 10 : (size_t)1 << 30) / n_lanes,
                                     e->knobs.json_ring_kb ? (size_t)0 : (size_t)128 << 20);
    const size_t hint = size_hint ? size_hint / n_lanes : 0;
    const size_t ring_bytes = std::max(hint ? std::min(hint / 4 * 3 + S->stride, ring_max) : ring_max, 64 * S->stride);
    L.cap = ring_bytes / S->stride;
    L.first_min = std::min<size_t>(e->knobs.json_first_min ? e->knobs.json_first_min : (size_t)1 << 14, L.cap / 4);
    if (L.cap * S->stride > e->json_ring_bytes || L.cap > e->json_ring_ballots) {
      if (e->json_ring) (void)hipHostFree(e->json_ring);
      if (e->json_status_ring) (void)hipHostFree(e->json_status_ring);
      e->json_ring = nullptr; e->json_status_ring = nullptr; e->json_ring_bytes = 0; e->json_ring_ballots = 0;
      if (hipHostMalloc((void**)&e->json_ring, L.cap * S->stride, hipHostMallocPortable) != hipSuccess ||
          hipHostMalloc((void**)&e->json_status_ring, L.cap * sizeof(u32), hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        return fail(EG_ERR_NOMEM, "pinned staging allocation failed");
      }
      e->json_ring_bytes = L.cap * S->stride; e->json_ring_ballots = L.cap;
    } else L.cap = std::min(L.cap, e->json_ring_ballots);
    TRY(engine_stage_reserve(e, L.cap));
    if (!e->copy_stream) HIPCHK(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
    L.n_ctl = e->n_sets == 2 ? 2 : 1;
    for (int c = 0; c < L.n_ctl; ++c)
      if (!e->json_ctl[c]) HIPCHK(hipStreamCreateWithFlags(&e->json_ctl[c], hipStreamNonBlocking));
    {
      const size_t per_set = (L.cap + e->n_sets - 1) / e->n_sets + NT;
      const int rr = engine_reserve(e, (u32)std::min<size_t>(per_set, e->max_cap));
      if (rr != EG_OK && rr != EG_ERR_NOMEM) return rr;
    }
    if (S->ns) {        // the stream's own tally is reported at the end; the running tally keeps accumulating (eg_hip.h)
      HIPCHK(hipMemcpyAsync(e->tally_saved2, e->tally, (size_t)S->ns * PT_WORDS * sizeof(u32), hipMemcpyDeviceToDevice, s));
      hipLaunchKernelGGL(k_tally_init, dim3(blocks_of((size_t)S->ns)), dim3(NT), 0, s, e->tally, S->ns);
      L.set_aside = true;
    }
    lanes_ready = k + 1;
    HIPCHK(hipStreamSynchronize(s));
    if (std::find(S->ctxs.begin(), S->ctxs.end(), e->ctx) == S->ctxs.end()) S->ctxs.push_back(e->ctx);
  }
  std::sort(S->ctxs.begin(), S->ctxs.end(), std::less<eg_ctx*>());
  S->pool.reset(new egwire::WorkerPool(S->threads));
  eg_json_stream* raw = S.get();
  const size_t window = e0->knobs.json_window_kb ? e0->knobs.json_window_kb << 10 : (size_t)32 << 20;
  size_t min_cap = S->lanes[0].cap;
  for (const StreamLane& L : S->lanes) min_cap = std::min(min_cap, L.cap);
  S->split.reset(new egwire::StreamSplitter(S->threads, S->pool.get(),
                                            [raw](const char* base, const std::vector<std::pair<size_t, size_t>>& spans, size_t first) {
                                              return stream_emit(raw, base, spans, first);
                                            },
                                            window, std::max<size_t>(1, min_cap / 8)));
  for (Engine* e : engines) e->stream_open = raw;
  if (S->trace) fprintf(stderr, "[json stream] %8.2f ms  begun (%zu lane(s), rings of %zu ballots, first submission from %zu)\n", S->ms(), n_lanes, S->lanes[0].cap, S->lanes[0].first_min);
  S->worker = std::thread(stream_worker, raw);
  *out = S.release();
  return EG_OK;
}
// the worker thread of a stream: pieces in order through the splitter (-> stream_emit -> rings, GPUs), each under the contexts' locks
static void stream_worker(eg_json_stream* S) {
  std::vector<char> joined;                 // small handed-over blocks that were waiting together, copied into one piece (below)
  for (;;) {
    eg_json_stream::Item it;
    std::vector<eg_json_stream::Item> more;      // the handed-over blocks behind `it` that are processed with it
    {
      std::unique_lock<std::mutex> lk(S->qmu);
      S->q_push.wait(lk, [&]() { return S->stop || !S->queue.empty(); });
      if (S->stop) return;
      it = std::move(S->queue.front());
      S->queue.pop_front();
      // Small blocks handed over by eg_verify_json_feed_owned are cheap to take one by one only when they arrive slowly: every piece costs
      // a hand-over, a window of the splitter, two dispatches on the pool (a ballot straddles almost every boundary) and a look at the
      // GPU - 1 MB pieces at the rate of the verifier are 1 400 of those in 160 ms (measured: 0.48 of the resident rate).  Blocks that are
      // WAITING TOGETHER are therefore joined into one piece of up to 16 MB, copied by the pool's threads (not by the caller's, and not one
      // after the other), and given back at once: 0.85-0.91.
      if (it.release && !it.finish && it.len < S->direct_min) {
        size_t total = it.len;
        while (!S->queue.empty() && S->queue.front().release && !S->queue.front().finish && S->queue.front().len < S->direct_min &&
               total + S->queue.front().len <= S->join_bytes) {
          total += S->queue.front().len;
          more.push_back(std::move(S->queue.front()));
          S->queue.pop_front();
        }
      }
    }
    const char* text = it.ptr ? it.ptr : it.own.data();
    size_t len = it.len;
    for (auto& m : more) len += m.len;
    // nothing thrown on this thread may escape it (std::terminate): out of host memory while joining blocks, growing the verdict vector or
    // keeping the text of odd ballots becomes the stream's error, reported by the next feed / take / end like every other
    try {
      if (!more.empty()) {
        std::vector<const eg_json_stream::Item*> parts{&it};
        for (auto& m : more) parts.push_back(&m);
        if (joined.size() < len) joined.resize(len);
        std::vector<size_t> at(parts.size(), 0);
        for (size_t k = 1; k < parts.size(); ++k) at[k] = at[k - 1] + parts[k - 1]->len;
        S->pool->run(parts.size(), 1, [&](size_t lo, size_t hi) { for (size_t k = lo; k < hi; ++k) if (parts[k]->len) memcpy(joined.data() + at[k], parts[k]->ptr, parts[k]->len); });
        text = joined.data();
        if (it.release) { it.release(it.user, it.ptr, it.len); it.release = nullptr; }     // copied: the blocks go back before they are parsed
        for (auto& m : more) { m.release(m.user, m.ptr, m.len); m.release = nullptr; }
      }
      if (!it.finish && !S->failed.load(std::memory_order_acquire)) {
        LockAll lk(S->ctxs);
        if (!S->split->feed(text, len)) {
          if (!S->failed.load()) stream_fail(S, EG_ERR_BAD_ARG, S->split->error().empty() ? std::string("the piece could not be packed") : S->split->error());
        } else (void)stream_pump_all(S, false);
        S->objects.store(S->split->count(), std::memory_order_release);
      }
    } catch (const std::bad_alloc&) { stream_fail(S, EG_ERR_NOMEM, "out of host memory in the stream's worker");
    } catch (const std::exception& ex) { stream_fail(S, EG_ERR_HIP, std::string("exception in the stream's worker: ") + ex.what());
    } catch (...) { stream_fail(S, EG_ERR_HIP, "unknown exception in the stream's worker"); }
    for (auto& m : more)                           // (only after an exception above: blocks that were not given back yet)
      if (m.release) { m.release(m.user, m.ptr, m.len); m.release = nullptr; }
    {
      std::lock_guard<std::mutex> lk(S->qmu);
      S->consumed_id = more.empty() ? it.id : more.back().id;
      S->queued_bytes -= std::min(S->queued_bytes, len);
      if (!it.own.empty() || it.own.capacity()) { it.own.clear(); S->spare.push_back(std::move(it.own)); }
    }
    S->q_pop.notify_all();
    if (it.release) it.release(it.user, it.ptr, it.len);      // the caller's block goes back: nothing of it is referenced any more
    if (it.finish) return;
  }
}
// hands an item to the worker (at most ~64 MB of text wait in the queue); wait = until the worker is through with it
static void stream_enqueue(eg_json_stream* S, std::unique_lock<std::mutex>& lk, eg_json_stream::Item&& it, bool wait) {
  S->q_pop.wait(lk, [&]() { return S->queue.empty() || S->queued_bytes < S->queue_bytes; });
  const uint64_t id = it.id = S->next_id++;
  S->queued_bytes += it.len;
  S->queue.push_back(std::move(it));
  S->q_push.notify_one();
  if (wait) S->q_pop.wait(lk, [&]() { return S->consumed_id >= id; });
}
static void stream_flush_acc(eg_json_stream* S, std::unique_lock<std::mutex>& lk) {
  if (S->acc.empty()) return;
  eg_json_stream::Item it;
  it.len = S->acc.size();
  it.own = std::move(S->acc);
  S->acc = std::vector<char>();
  if (!S->spare.empty()) { S->acc = std::move(S->spare.back()); S->spare.pop_back(); }
  stream_enqueue(S, lk, std::move(it), false);
}
static void stream_join_worker(eg_json_stream* S, bool finish) {      // never called with a context's lock held: the worker takes them per piece
  if (S->worker_joined) return;
  {
    std::unique_lock<std::mutex> lk(S->qmu);
    if (finish) {
      stream_flush_acc(S, lk);
      eg_json_stream::Item fin;
      fin.finish = true;
      stream_enqueue(S, lk, std::move(fin), false);
    } else {
      S->stop = true;
      S->q_push.notify_all();
    }
  }
  if (S->worker.joinable()) S->worker.join();
  S->worker_joined = true;
  for (auto& it : S->queue)                 // an aborted stream: handed-over blocks the worker never got to go back to their owner
    if (it.release) it.release(it.user, it.ptr, it.len);
  S->queue.clear();
}
int eg_verify_json_feed(eg_json_stream* S, const char* text, size_t len, size_t* n_objects) {
  if (!S || (len && !text)) return fail(EG_ERR_BAD_ARG, "bad argument");
  {
    std::unique_lock<std::mutex> lk(S->qmu);
    if (S->failed.load(std::memory_order_acquire)) return fail(S->failed.load(), S->err);
    if (S->flushed || S->worker_joined) return fail(EG_ERR_BAD_ARG, "the stream has been ended");
    if (len >= S->direct_min) {                    // a large piece: in place, and feed returns when the worker is through with it
      stream_flush_acc(S, lk);
      eg_json_stream::Item it;
      it.ptr = text; it.len = len;
      stream_enqueue(S, lk, std::move(it), true);
    } else {
      while (len) {                                // small pieces: copied into the block that is being filled
        if (S->acc.capacity() < S->block_bytes) S->acc.reserve(S->block_bytes);
        const size_t take = std::min(len, S->block_bytes - S->acc.size());
        S->acc.insert(S->acc.end(), text, text + take);
        text += take; len -= take;
        if (S->acc.size() >= S->block_bytes) stream_flush_acc(S, lk);
      }
    }
    if (S->failed.load(std::memory_order_acquire)) return fail(S->failed.load(), S->err);
  }
  if (n_objects) *n_objects = S->objects.load(std::memory_order_acquire);
  return EG_OK;
}
// The next piece WITHOUT a copy and without waiting: the library reads the caller's block in place and calls release(user, text, len) -
// from the stream's worker thread, or from the thread that ends / aborts the stream - when nothing of it is referenced any more.
// (eg_verify_json_feed copies a small piece on the caller's thread: 1.5 GB of text in 1 MB pieces is 150 ms of memcpy, as long as the GPU
// needs for the million ballots in it - json_stream.pieces.1MB of the bench line sat at 0.83 of the resident rate for that reason.)
int eg_verify_json_feed_owned(eg_json_stream* S, const char* text, size_t len, eg_json_release_fn release, void* user, size_t* n_objects) {
  if (!S || !release || (len && !text)) return fail(EG_ERR_BAD_ARG, "bad argument");
  {
    std::unique_lock<std::mutex> lk(S->qmu);
    if (S->failed.load(std::memory_order_acquire)) return fail(S->failed.load(), S->err);
    if (S->flushed || S->worker_joined) return fail(EG_ERR_BAD_ARG, "the stream has been ended");
    stream_flush_acc(S, lk);                        // pieces copied earlier come first
    eg_json_stream::Item it;
    it.ptr = text; it.len = len; it.release = release; it.user = user;
    stream_enqueue(S, lk, std::move(it), false);    // from here on the block is the library's: release will be called, whatever happens
  }
  if (n_objects) *n_objects = S->objects.load(std::memory_order_acquire);
  return EG_OK;
}
// A ready-made release function for eg_verify_json_feed_owned: counts the blocks that came back in the size_t that `user` points to.  For
// bindings whose own callbacks are expensive - a Python (ctypes) callback takes the interpreter lock on the stream's worker thread, once per
// block, in competition with the thread that is feeding - and for tests that only want to know that every block was given back.
void eg_json_release_count(void* user, const char*, size_t) {
  if (user) __atomic_fetch_add(static_cast<size_t*>(user), (size_t)1, __ATOMIC_RELAXED);
}
// verdicts that are final so far, in order, from where the last take stopped: every ballot before the first one that is still on a GPU
// or waits for the object path (a ballot of another shape than the election's gets its verdict at the end)
static size_t stream_final_upto(const eg_json_stream* S) {
  size_t upto = S->flushed ? S->verdicts.size() : stream_landed(S);
  if (!S->flushed && !S->odd_at.empty()) upto = std::min(upto, S->odd_at.front());
  return upto;
}
int eg_verify_json_take(eg_json_stream* S, uint32_t* status, size_t cap, size_t* n_taken) {
  if (!S || !n_taken || (cap && !status)) return fail(EG_ERR_BAD_ARG, "bad argument");
  *n_taken = 0;
  if (S->failed.load(std::memory_order_acquire)) return fail(S->failed.load(), S->err);
  {   // pieces copied so far go to the worker now: a slow source gets its verdicts without waiting for a block to fill up
    std::unique_lock<std::mutex> lk(S->qmu);
    if (!S->flushed && !S->worker_joined) stream_flush_acc(S, lk);
  }
  LockAll lk(S->ctxs);                              // the worker holds them while it cuts and packs a piece
  if (!S->flushed)
    for (StreamLane& L : S->lanes) TRY(stream_pump(S, L, L.groups.empty()));     // an idle GPU takes what has been packed, however little
  const size_t upto = stream_final_upto(S);
  const size_t n = std::min(cap, upto > S->taken ? upto - S->taken : 0);
  if (n) memcpy(status, S->verdicts.data() + S->taken, n * sizeof(uint32_t));
  S->taken += n;
  *n_taken = n;
  return EG_OK;
}
int eg_verify_json_end(eg_json_stream* S, uint32_t* status, size_t cap, size_t* n_taken, size_t* n_objects, uint8_t* tally_out) {
  if (!S || (cap && !status)) return fail(EG_ERR_BAD_ARG, "bad argument");
  stream_join_worker(S, true);                      // every piece cut and packed; from here on this thread owns the stream
  LockAll lk(S->ctxs);
  if (S->trace) fprintf(stderr, "[json stream] %8.2f ms  every piece packed (%zu objects)\n", S->ms(), S->split->count());
  if (n_taken) *n_taken = 0;
  Engine* e = S->e;
  if (!S->flushed && !S->failed.load()) {
    if (!S->split->finish()) stream_fail(S, EG_ERR_BAD_ARG, S->split->error());
    for (;;) {                                     // every lane: submit what is left, wait for what is in flight, oldest first
      if (S->failed.load()) break;
      bool busy = false;
      for (StreamLane& L : S->lanes) {
        if (L.regions.empty() && L.groups.empty()) continue;
        busy = true;
        if (stream_pump(S, L, true)) break;
        if (L.groups.empty()) continue;
        (void)hipSetDevice(L.e->ctx->device);
        const hipError_t he = hipEventSynchronize(L.groups.front().done);
        if (he != hipSuccess) { stream_fail(S, EG_ERR_HIP, std::string("window: ") + hipGetErrorString(he)); break; }
        stream_retire_oldest(S, L);
      }
      if (!busy) break;
    }
    if (!S->failed.load())
      for (StreamLane& L : S->lanes) {
        Engine* le = L.e;
        (void)hipSetDevice(le->ctx->device);
        (void)hipDeviceSynchronize();
        hipStream_t s = le->ctx->stream;
        if (S->ns && le->n_sets == 2) {                           // the sets' shares of the tally, once
          hipLaunchKernelGGL(k_tally_add_points, dim3(blocks_of((size_t)S->ns)), dim3(NT), 0, s, le->set[1].tally, S->ns, le->set[0].tally);
          hipLaunchKernelGGL(k_tally_init, dim3(blocks_of((size_t)S->ns)), dim3(NT), 0, s, le->set[1].tally, S->ns);
          if (hipStreamSynchronize(s) != hipSuccess) stream_fail(S, EG_ERR_HIP, "tally merge failed");
        }
      }
    if (!S->failed.load() && !S->odd_at.empty()) {   // OptionsLenMismatch / LenMismatch territory: the object path, in the reference's order of checks
      std::string all;
      std::vector<std::pair<size_t, size_t>> spans;
      for (auto& t : S->odd_text) { spans.push_back({all.size(), t.size()}); all += t; all += '\n'; }
      std::vector<uint32_t> v;
      e->stream_open = nullptr;               // the object path verifies substitute ballots through the ordinary host entry of the first lane's engine
      const bool ok = S->reshape(all.data(), spans, v);
      e->stream_open = S;
      if (!ok) stream_fail(S, EG_ERR_HIP, g_err.empty() ? std::string("object path: a GPU call failed") : g_err);
      else for (size_t i = 0; i < S->odd_at.size(); ++i) S->verdicts[S->odd_at[i]] = v[i];
    }
    S->flushed = true;
  }
  if (S->failed.load()) {
    const int rc = S->failed.load();
    const std::string why = S->err;
    stream_release(S, false);
    return fail(rc, why);
  }
  const size_t left = S->verdicts.size() - S->taken;
  if (left > cap) {            // the stream stays open (and flushed): *n_taken says how much room the caller must come back with
    if (n_taken) *n_taken = left;
    return fail(EG_ERR_BAD_ARG, "status holds " + std::to_string(cap) + " verdicts, " + std::to_string(left) + " are left: call again with room for them (or eg_verify_json_abort)");
  }
  if (left) memcpy(status, S->verdicts.data() + S->taken, left * sizeof(uint32_t));
  if (n_taken) *n_taken = left;
  if (n_objects) *n_objects = S->verdicts.size();
  int rc = EG_OK;
  if (S->trace) fprintf(stderr, "[json stream] %8.2f ms  verdicts copied\n", S->ms());
  if (tally_out && S->ns) {       // the stream's own tally = the sum of its lanes' (every running tally gets its lane's share added below)
    const size_t bytes = (size_t)S->ns * 32;
    std::vector<uint8_t> part(bytes), ok(S->ns);
    for (size_t k = 0; k < S->lanes.size() && !rc; ++k) {
      (void)hipSetDevice(S->lanes[k].e->ctx->device);
      rc = engine_tally_encode(S->lanes[k].e, k == 0 ? tally_out : part.data());
      if (!rc && k) {
        rc = eg_point_add_batch(e->ctx, (size_t)S->ns, tally_out, part.data(), 0, tally_out, ok.data());
        if (!rc) for (uint8_t o : ok) if (!o) rc = fail(EG_ERR_HIP, "the tally of lane " + std::to_string(k) + " does not decode");
      }
    }
  }
  const std::string why = g_err;
  const bool trace = S->trace;
  const auto t0 = S->t0;
  stream_release(S, true);
  if (trace) fprintf(stderr, "[json stream] %8.2f ms  released\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  return rc ? fail(rc, why) : EG_OK;
}
void eg_verify_json_abort(eg_json_stream* S) {
  if (!S) return;
  stream_join_worker(S, false);
  const std::vector<eg_ctx*> ctxs = S->ctxs;      // (the stream is deleted under the locks)
  LockAll lk(ctxs);
  stream_release(S, false);
}
// Opens a stream over one or several params objects.  one_shot: for the length of an eg_verify_*_json[_multi] call, whose caller holds the
// engines' long_call_mu; an explicit begin takes them only while it opens the stream (so that it waits for a long call that is running on
// any of the objects, as every entry point does).
static int stream_open_on(const std::vector<Engine*>& engines, int threads, PackPieceFn pack_piece, ReshapeFn reshape, size_t size_hint, bool one_shot,
                          size_t max_objects, eg_json_stream** out) {
  std::vector<Engine*> order(engines);
  std::sort(order.begin(), order.end(), std::less<Engine*>());
  std::vector<std::unique_lock<std::mutex>> waits;
  if (!one_shot) for (Engine* e : order) waits.emplace_back(e->long_call_mu);
  std::vector<eg_ctx*> ctxs;
  for (Engine* e : engines) if (std::find(ctxs.begin(), ctxs.end(), e->ctx) == ctxs.end()) ctxs.push_back(e->ctx);
  std::sort(ctxs.begin(), ctxs.end(), std::less<eg_ctx*>());
  LockAll lk(ctxs);
  for (Engine* e : engines)
    if (e->stream_open) return fail(EG_ERR_BAD_ARG, one_shot ? "a JSON stream is open on this params object (eg_verify_json_end or _abort it first)"
                                                             : "a JSON stream is already open on this params object");
  TRY(stream_begin(engines, threads, std::move(pack_piece), std::move(reshape), size_hint, out));
  (*out)->one_shot = one_shot;
  (*out)->max_objects = max_objects;              // one-shot: the worker breaks off as soon as the text holds more (nothing further is parsed or verified)
  return EG_OK;
}
// The one-shot entry = the streaming entry fed with the whole text (in place, 64 MB at a time) - one pipeline for both (round 5; rounds 3-4
// had a second one here, a producer thread and a consumer loop over the same ring).
static int verify_json_common(const std::vector<Engine*>& engines, const char* json, size_t json_len, int threads, size_t max_objects, uint32_t* status,
                              size_t* n_objects, uint8_t* tally_out, const PackPieceFn& pack_piece, const ReshapeFn& reshape) {
  if ((json_len && !json) || (max_objects && !status)) return fail(EG_ERR_BAD_ARG, "bad argument");
  if (n_objects) *n_objects = 0;
  eg_json_stream* S = nullptr;
  // concurrent long calls on a params object are serialised, as every entry point is: the call holds every engine's long_call_mu (address order)
  std::vector<Engine*> order(engines);
  std::sort(order.begin(), order.end(), std::less<Engine*>());
  std::vector<std::unique_lock<std::mutex>> one_at_a_time;
  for (Engine* e : order) one_at_a_time.emplace_back(e->long_call_mu);
  // the contexts' locks only while the stream is opened: its worker thread takes them piece by piece, and so does end
  TRY(stream_open_on(engines, threads, pack_piece, reshape, json_len, true, max_objects, &S));
  const size_t piece = (size_t)64 << 20;
  for (size_t at = 0; at < json_len; at += piece) {
    if (eg_verify_json_feed(S, json + at, std::min(piece, json_len - at), nullptr)) break;       // end reports it and cleans up
  }
  size_t taken = 0, total = 0;
  const int rc = eg_verify_json_end(S, status, max_objects, &taken, &total, tally_out);
  // end leaves the stream OPEN in exactly one case - more verdicts are left than `status` has room for - and says so through its
  // out-values: *n_taken = the number left (> cap).  Every other failure has destroyed the stream and reports *n_taken = 0.  (The worker
  // normally fails the stream as soon as it has cut more than max_objects values, long before this point: stream_emit.)
  if (rc == EG_ERR_BAD_ARG && taken > max_objects) {
    eg_verify_json_abort(S);
    return fail(EG_ERR_BAD_ARG, "more objects in the text than max_objects");
  }
  if (rc) return rc;
  if (n_objects) *n_objects = total;
  return EG_OK;
}
// how a choice / QV election packs its ballots and resolves the ones of another shape (the object path runs on the FIRST engine)
static void choice_json_fns(const eg_choice_params* p, PackPieceFn* pack, ReshapeFn* reshape) {
  const int n_options = p->n_options, single = p->single;
  const size_t stride = p->eng->plan.stride;
  Engine* e = p->eng;
  *pack = [=](const char* text, const std::vector<std::pair<size_t, size_t>>& sub, int th, uint8_t* dst, uint32_t* st, egwire::WorkerPool* pool) {
    egwire::pack_parallel(text, sub, stride, th, dst, st, [&](egwire::Cursor& c, uint8_t* d) { return egwire::pack_choice(c, n_options, single != 0, d); }, pool);
  };
  *reshape = [=](const char* text, const std::vector<std::pair<size_t, size_t>>& odd, std::vector<uint32_t>& res) {
    return egwire::resolve_choice_objects(text, odd, n_options, single != 0, stride, make_check_items(e->ctx), make_verify_packed(e), res);
  };
}
static void qv_json_fns(const eg_qv_params* p, PackPieceFn* pack, ReshapeFn* reshape) {
  const int n_options = p->n_options;
  const eghost::QvShape sh = p->shape;
  const egwire::RangeShape vote{sh.vote_range.rings.size(), (size_t)sh.vote_range.rings_size()};
  const egwire::RangeShape credit{sh.credit_range.rings.size(), (size_t)sh.credit_range.rings_size()};
  Engine* e = p->eng;
  *pack = [=](const char* text, const std::vector<std::pair<size_t, size_t>>& sub, int th, uint8_t* dst, uint32_t* st, egwire::WorkerPool* pool) {
    egwire::pack_parallel(text, sub, sh.ballot_size, th, dst, st, [&](egwire::Cursor& c, uint8_t* d) { return egwire::pack_qv(c, n_options, vote, credit, sh.ballot_size, d); }, pool);
  };
  *reshape = [=](const char* text, const std::vector<std::pair<size_t, size_t>>& odd, std::vector<uint32_t>& res) {
    return egwire::resolve_qv_objects(text, odd, n_options, vote, credit, sh.ballot_size, make_check_items(e->ctx), make_verify_packed(e), res);
  };
}
extern "C++" {
template <class Params>
static int engines_of(Params* const* per_device, int n_dev, std::vector<Engine*>* out) {
  TRY(multi_check(per_device, n_dev));
  for (int d = 0; d < n_dev; ++d) out->push_back(per_device[d]->eng);
  return EG_OK;
}
}
int eg_verify_choice_json(eg_choice_params* p, const char* json, size_t json_len, int threads, size_t max_objects, uint32_t* status,
                          size_t* n_objects, uint8_t* tally_out) {
  if (!p) return fail(EG_ERR_BAD_ARG, "bad argument");
  PackPieceFn pack; ReshapeFn reshape;
  choice_json_fns(p, &pack, &reshape);
  return verify_json_common({p->eng}, json, json_len, threads, max_objects, status, n_objects, tally_out, pack, reshape);
}
int eg_verify_qv_json(eg_qv_params* p, const char* json, size_t json_len, int threads, size_t max_objects, uint32_t* status,
                      size_t* n_objects, uint8_t* tally_out) {
  if (!p) return fail(EG_ERR_BAD_ARG, "bad argument");
  PackPieceFn pack; ReshapeFn reshape;
  qv_json_fns(p, &pack, &reshape);
  return verify_json_common({p->eng}, json, json_len, threads, max_objects, status, n_objects, tally_out, pack, reshape);
}
int eg_verify_choice_json_multi(eg_choice_params* const* per_device, int n_dev, const char* json, size_t json_len, int threads, size_t max_objects,
                                uint32_t* status, size_t* n_objects, uint8_t* tally_out) {
  std::vector<Engine*> engines;
  TRY(engines_of(per_device, n_dev, &engines));
  PackPieceFn pack; ReshapeFn reshape;
  choice_json_fns(per_device[0], &pack, &reshape);
  return verify_json_common(engines, json, json_len, threads, max_objects, status, n_objects, tally_out, pack, reshape);
}
int eg_verify_qv_json_multi(eg_qv_params* const* per_device, int n_dev, const char* json, size_t json_len, int threads, size_t max_objects,
                            uint32_t* status, size_t* n_objects, uint8_t* tally_out) {
  std::vector<Engine*> engines;
  TRY(engines_of(per_device, n_dev, &engines));
  PackPieceFn pack; ReshapeFn reshape;
  qv_json_fns(per_device[0], &pack, &reshape);
  return verify_json_common(engines, json, json_len, threads, max_objects, status, n_objects, tally_out, pack, reshape);
}
int eg_verify_choice_json_begin(eg_choice_params* p, int threads, eg_json_stream** out) {
  if (!p) return fail(EG_ERR_BAD_ARG, "bad argument");
  PackPieceFn pack; ReshapeFn reshape;
  choice_json_fns(p, &pack, &reshape);
  return stream_open_on({p->eng}, threads, pack, reshape, 0, false, (size_t)-1, out);
}
int eg_verify_qv_json_begin(eg_qv_params* p, int threads, eg_json_stream** out) {
  if (!p) return fail(EG_ERR_BAD_ARG, "bad argument");
  PackPieceFn pack; ReshapeFn reshape;
  qv_json_fns(p, &pack, &reshape);
  return stream_open_on({p->eng}, threads, pack, reshape, 0, false, (size_t)-1, out);
}
int eg_verify_choice_json_begin_multi(eg_choice_params* const* per_device, int n_dev, int threads, eg_json_stream** out) {
  std::vector<Engine*> engines;
  TRY(engines_of(per_device, n_dev, &engines));
  PackPieceFn pack; ReshapeFn reshape;
  choice_json_fns(per_device[0], &pack, &reshape);
  return stream_open_on(engines, threads, pack, reshape, 0, false, (size_t)-1, out);
}
int eg_verify_qv_json_begin_multi(eg_qv_params* const* per_device, int n_dev, int threads, eg_json_stream** out) {
  std::vector<Engine*> engines;
  TRY(engines_of(per_device, n_dev, &engines));
  PackPieceFn pack; ReshapeFn reshape;
  qv_json_fns(per_device[0], &pack, &reshape);
  return stream_open_on(engines, threads, pack, reshape, 0, false, (size_t)-1, out);
}
size_t eg_qv_ballot_size_for(int n_options, uint64_t credits) {
  if (n_options < 1 || n_options > 256 || credits < 1 || credits > 100000) return 0;
  return eghost::qv_shape(n_options, credits).ballot_size;
}

// ---- synthetic ballots ---------------------------------------------------------------------------------------------------------
static int choice_encrypt_device(eg_choice_params* p, uint64_t base_seed, size_t first, size_t n, int n_selected,
                                 const void* d_selection, uint64_t rng_skip, void* d_out, hipStream_t s) {
  Engine* e = p->eng;
  HIPCHK(hipSetDevice(e->ctx->device));
  if (!d_selection && !p->single && (n_selected < 0 || n_selected > p->n_options)) return fail(EG_ERR_BAD_ARG, "n_selected out of range");
  if (n == 0) return EG_OK;
  int blocks = 0;
  TRY(gen_workspace(e, n, eg_gen_choice_ws_words(p->n_options), &blocks));
  eg_launch_choice_encrypt(blocks, s, base_seed + first, n, p->n_options, p->single, n_selected,
                           reinterpret_cast<const u32*>(d_selection), rng_skip, e->ctx->tabG,
                           e->d_tabK, e->d_prefixes, e->plan.gen_pre_main, e->plan.gen_pre_ring, e->plan.gen_pre_logeq,
                           reinterpret_cast<u32*>(d_out), (u32)(e->plan.stride / 4), e->gen_ws);
  HIPCHK(hipGetLastError());
  return EG_OK;
}
int eg_choice_encrypt_batch_device(eg_choice_params* p, uint64_t base_seed, size_t first, size_t n, int n_selected, void* d_out,
                                   void* stream) { EG_LOCK_P(p);
  if (!p || (n && !d_out)) return fail(EG_ERR_BAD_ARG, "bad argument");
  return choice_encrypt_device(p, base_seed, first, n, n_selected, nullptr, 0, d_out, (hipStream_t)stream);
}
int eg_choice_encrypt_batch(eg_choice_params* p, uint64_t base_seed, size_t first, size_t n, int n_selected, uint8_t* out) { EG_LOCK_P(p);
  if (!p || (n && !out)) return fail(EG_ERR_BAD_ARG, "bad argument");
  Engine* e = p->eng;
  HIPCHK(hipSetDevice(e->ctx->device));
  DevBuf d;
  TRY(d.alloc(n * e->plan.stride));
  TRY(choice_encrypt_device(p, base_seed, first, n, n_selected, nullptr, 0, d.p, e->ctx->stream));
  TRY(d.get(out, n * e->plan.stride, e->ctx->stream));
  HIPCHK(hipStreamSynchronize(e->ctx->stream));
  return EG_OK;
}
int eg_choice_encrypt_selected_batch_device(eg_choice_params* p, uint64_t base_seed, size_t first, size_t n, uint64_t rng_skip,
                                            const void* d_selection, void* d_out, void* stream) { EG_LOCK_P(p);
  if (!p || (n && (!d_out || !d_selection))) return fail(EG_ERR_BAD_ARG, "bad argument");
  return choice_encrypt_device(p, base_seed, first, n, 0, d_selection, rng_skip, d_out, (hipStream_t)stream);
}
int eg_choice_encrypt_selected_batch(eg_choice_params* p, uint64_t base_seed, size_t first, size_t n, uint64_t rng_skip,
                                     const uint32_t* selection, uint8_t* out) { EG_LOCK_P(p);
  if (!p || (n && (!out || !selection))) return fail(EG_ERR_BAD_ARG, "bad argument");
  Engine* e = p->eng;
  HIPCHK(hipSetDevice(e->ctx->device));
  const size_t sw = ((size_t)p->n_options + 31) / 32;
  for (size_t i = 0; i < n; ++i) {      // what EncryptedChoice::single / ::new would refuse or mis-prove
    int chosen = 0;
    for (size_t w = 0; w < sw; ++w) {
      const uint32_t word = selection[i * sw + w];
      const int valid_bits = (int)std::min<size_t>(32, (size_t)p->n_options - 32 * w);
      if (valid_bits < 32 && (word >> valid_bits)) return fail(EG_ERR_BAD_ARG, "selection has bits beyond the options");
      chosen += __builtin_popcount(word);
    }
    if (p->single && chosen != 1) return fail(EG_ERR_BAD_ARG, "a single-choice ballot selects exactly one option");
  }
  DevBuf d, sel;
  TRY(d.alloc(n * e->plan.stride)); TRY(sel.alloc(n * sw * 4));
  TRY(sel.put(selection, n * sw * 4, e->ctx->stream));
  TRY(choice_encrypt_device(p, base_seed, first, n, 0, sel.p, rng_skip, d.p, e->ctx->stream));
  TRY(d.get(out, n * e->plan.stride, e->ctx->stream));
  HIPCHK(hipStreamSynchronize(e->ctx->stream));
  return EG_OK;
}
static int qv_encrypt_device(eg_qv_params* p, uint64_t base_seed, size_t first, size_t n, const void* d_votes, uint64_t rng_skip,
                             void* d_out, hipStream_t s) {
  Engine* e = p->eng;
  HIPCHK(hipSetDevice(e->ctx->device));
  const eghost::QvShape& sh = p->shape;
  if (n == 0) return EG_OK;
  const size_t vr = sh.vote_range.rings.size(), cr = sh.credit_range.rings.size();
  if (!e->d_gen_desc) {     // ring shapes (size, step) of the vote range, then of the credit range
    std::vector<u32> desc;
    for (auto* d : {&sh.vote_range, &sh.credit_range})
      for (auto& r : d->rings) { desc.push_back((u32)r.size); desc.push_back((u32)r.step); }
    HIPCHK(hipMalloc((void**)&e->d_gen_desc, desc.size() * sizeof(u32)));
    HIPCHK(hipMemcpy(e->d_gen_desc, desc.data(), desc.size() * sizeof(u32), hipMemcpyHostToDevice));
  }
  int blocks = 0;
  const unsigned words = eg_gen_qv_ws_words(p->n_options, (unsigned)std::max(vr, cr),
                                            (unsigned)std::max(sh.vote_range.rings_size(), sh.credit_range.rings_size()));
  TRY(gen_workspace(e, n, words, &blocks));
  eg_launch_qv_encrypt(blocks, s, base_seed + first, n, p->n_options, p->credits, reinterpret_cast<const u32*>(d_votes), rng_skip,
                       (int)vr, e->plan.gen_vote_main, e->plan.gen_vote_ring, e->d_gen_desc, (int)cr, e->plan.gen_credit_main,
                       e->plan.gen_credit_ring, e->d_gen_desc + 2 * vr, e->plan.gen_pre_sumsq, e->ctx->tabG, e->d_tabK,
                       e->d_prefixes, reinterpret_cast<u32*>(d_out), (u32)(sh.ballot_size / 4), (u32)(sh.vote_size / 4),
                       (u32)(sh.credit_size / 4), e->gen_ws);
  HIPCHK(hipGetLastError());
  return EG_OK;
}
int eg_qv_encrypt_batch_device(eg_qv_params* p, uint64_t base_seed, size_t first, size_t n, void* d_out, void* stream) { EG_LOCK_P(p);
  if (!p || (n && !d_out)) return fail(EG_ERR_BAD_ARG, "bad argument");
  return qv_encrypt_device(p, base_seed, first, n, nullptr, 0, d_out, (hipStream_t)stream);
}
int eg_qv_encrypt_votes_batch_device(eg_qv_params* p, uint64_t base_seed, size_t first, size_t n, uint64_t rng_skip,
                                     const void* d_votes, void* d_out, void* stream) { EG_LOCK_P(p);
  if (!p || (n && (!d_out || !d_votes))) return fail(EG_ERR_BAD_ARG, "bad argument");
  return qv_encrypt_device(p, base_seed, first, n, d_votes, rng_skip, d_out, (hipStream_t)stream);
}
int eg_qv_encrypt_votes_batch(eg_qv_params* p, uint64_t base_seed, size_t first, size_t n, uint64_t rng_skip, const uint32_t* votes,
                              uint8_t* out) { EG_LOCK_P(p);
  if (!p || (n && (!out || !votes))) return fail(EG_ERR_BAD_ARG, "bad argument");
  Engine* e = p->eng;
  HIPCHK(hipSetDevice(e->ctx->device));
  const uint64_t max_vote = eghost::isqrt(p->credits);
  for (size_t i = 0; i < n; ++i) {      // the assertions of QuadraticVotingBallot::new (quadratic_voting.rs:240-253)
    uint64_t credit = 0;
    for (int k = 0; k < p->n_options; ++k) {
      const uint64_t v = votes[i * (size_t)p->n_options + k];
      if (v > max_vote) return fail(EG_ERR_BAD_ARG, "a vote exceeds isqrt(credits)");
      credit += v * v;
    }
    if (credit > p->credits) return fail(EG_ERR_BAD_ARG, "votes exceed the credit amount");
  }
  DevBuf d, vv;
  TRY(d.alloc(n * e->plan.stride)); TRY(vv.alloc(n * (size_t)p->n_options * 4));
  TRY(vv.put(votes, n * (size_t)p->n_options * 4, e->ctx->stream));
  TRY(qv_encrypt_device(p, base_seed, first, n, vv.p, rng_skip, d.p, e->ctx->stream));
  TRY(d.get(out, n * e->plan.stride, e->ctx->stream));
  HIPCHK(hipStreamSynchronize(e->ctx->stream));
  return EG_OK;
}

// ---- Merlin transcripts as a primitive (merlin 3.0.0 via src/proofs/mod.rs:39-57): known-answer tests on the device ----------
int eg_merlin_challenge_batch(eg_ctx* c, size_t n, const char* proto, size_t proto_len, const char* msg_label, size_t msg_label_len,
                              const uint8_t* msgs, size_t msg_len, const char* chal_label, size_t chal_label_len, uint8_t* out,
                              size_t out_len) { EG_LOCK(c);
  if (!c || !proto || !msg_label || !chal_label || (n && (!out || (msg_len && !msgs)))) return fail(EG_ERR_BAD_ARG, "bad argument");
  if (proto_len > 255 || msg_label_len > 255 || chal_label_len > 255 || out_len == 0 || out_len > 1024 || msg_len > (1u << 20))
    return fail(EG_ERR_BAD_ARG, "labels up to 255 bytes, challenges of 1..1024 bytes, messages up to 1 MiB");
  if (n == 0) return EG_OK;
  HIPCHK(hipSetDevice(c->device));
  std::vector<uint8_t> labels;
  labels.insert(labels.end(), proto, proto + proto_len);
  labels.insert(labels.end(), msg_label, msg_label + msg_label_len);
  labels.insert(labels.end(), chal_label, chal_label + chal_label_len);
  DevBuf l, m, o;
  TRY(l.alloc(labels.size())); TRY(m.alloc(n * msg_len)); TRY(o.alloc(n * out_len));
  TRY(l.put(labels.data(), labels.size(), c->stream)); TRY(m.put(msgs, n * msg_len, c->stream));
  hipLaunchKernelGGL(k_prim_merlin, dim3(blocks_of(n)), dim3(NT), 0, c->stream, n, (const unsigned char*)l.p, (int)proto_len,
                     (int)msg_label_len, (int)chal_label_len, (const unsigned char*)m.p, (int)msg_len, (unsigned char*)o.p, (int)out_len);
  TRY(o.get(out, n * out_len, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipGetLastError());
  return EG_OK;
}

}  // extern "C"

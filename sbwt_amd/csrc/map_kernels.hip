// map_kernels.hip -- the read-mapping hot path on gfx950.
//
// Upstream per-read path (searcher.hpp:386-539): reverse-complement the read, 12-mer table lookup
// (init_exact_match :83-96), backward search (exec_exact_match :102-105, two rank queries per base),
// suppression test (:415), locate (level-0 scan :424-443 + find_possible :460-503), push_position
// filters (:504-539).  Here, per batch:
//   k_search_sm  (search_machine.h): search, suppression, locate and filters of reads with at most one hit;
//                multi-hit reads are queued with their hit slots reserved.
//   k_multi_sort (multi_locate.h): locate of the queued reads by gather + sort in upstream emission order.
//   k_multi      (below): the upstream depth-first locate, statement for statement, for the reads
//                k_multi_sort cannot take.
// This file holds k_multi, the sbwt_ctx object and the batch entry points of the C ABI.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <new>

#include "../../include/sbwt_amd.h"
#include "index_impl.hpp"
#include "rank_device.h"
#include "search_machine.h"
#include "multi_locate.h"
#include "settle.h"
#include "direct.h"
#include "rank_pass.h"
#include "search_rounds.h"

namespace sbwt {

struct Counters {
  unsigned long long hit_count;    // slots handed out in hits[] behind the first n_reads (multi-hit reads)
  unsigned long long multi_count;  // reads queued for k_multi
  unsigned int overflow;           // hits[] too small
  unsigned int pad0;
  unsigned long long index_accesses;  // random index accesses k_search_sm issued (one per 128-byte line it needs)
  unsigned long long dfs_count;       // multi-hit reads k_multi_sort handed on to the depth-first kernel
  unsigned long long hash_stats[2];   // hash_tab lookups of k_search_sm, and those that found their prefix
  unsigned long long wide_count;      // compact output: results that went to wide[]
  unsigned long long pending_count;   // k_settle: reads left to the general kernels
  unsigned long long pending_other;   // ... of which not because their prefix is missing from hash_tab
  unsigned long long shared_count;    // multi-hit reads whose record points at the list of another read with the same interval (k_multi_share)
  unsigned long long bfs_count;       // reads of the depth-first queue located level by level (k_multi_bfs); the rest took k_multi
  unsigned long long absent_count;    // reads the fast pass settled because their prefix occurs nowhere (DeviceIndex::hash_complete)
  unsigned long long chain_count;     // reads of the depth-first queue located from memoised chains (k_multi_chain)
  unsigned long long rows_located;    // raw rows of the multi-hit reads that were located themselves (leaders; a shared list counts once)
};

struct MapArgs {
  const uint32_t* packed;
  const uint32_t* lens;
  uint64_t n_reads;
  uint32_t wpr;    // 32-bit words per read record (multiple of 4)
  int strand;
  uint64_t limit;  // (uint64_t)(int64_t)limitNumber: the upstream comparison is unsigned (searcher.hpp:415)
  ResultIO io;
  sbwt_hit* hits;
  uint64_t hits_cap;
  uint64_t multi_base;  // first slot of hits[] for the lists of multi-hit reads
  Counters* ctr;
  uint32_t* multi_queue;
};

struct Frame {  // one activation of find_possible
  uint64_t pf, pe;
  uint32_t next_c, pad;
};

// ---- multi-hit locate: the upstream DFS, one lane per queued read -----------------------------
struct MultiState {
  const DeviceIndex* ix;
  sbwt_hit* out;
  uint64_t cap;    // room of the list: the slots reserved (raw interval size), or the list's own (extra_slots) on the second walk
  uint64_t nres;   // accepted pushes (VARS.result.size())
  uint32_t L;
  int strand;
  uint32_t truncated;
};

__device__ __forceinline__ void multi_push(MultiState& S, uint64_t position) {
  sbwt_hit h;
  if (!push_position_lane(*S.ix, position, S.L, S.strand, &h)) return;
  if (S.nres < S.cap) S.out[S.nres] = h; else S.truncated = 1;
  S.nres++;
}

// for i in [f, e): if fbwt[i] == 0 and i is in location_table -> push(len + SA[i]); returns true when the
// upstream loop returned early (interval of size 1 whose position was sampled, searcher.hpp:495-496)
__device__ __forceinline__ bool scan_sampled(MultiState& S, uint64_t f, uint64_t e, uint64_t len, bool single_returns) {
  const DeviceIndex& ix = *S.ix;
  const uint64_t bf = f >> 6, be = (e - 1) >> 6;
  for (uint64_t bb = bf; bb <= be; ++bb) {
    const Block1* b = ix.blocks1 + bb;
    const uint64_t sampled = b->sampled, sbase = b->sampled_base;
    const uint32_t o_lo = bb == bf ? (uint32_t)(f & 63) : 0u;
    const uint32_t o_hi = bb == be ? (uint32_t)((e - 1) & 63) + 1u : 64u;
    uint64_t bits = sampled & below(o_hi) & ~below(o_lo);
    while (bits) {
      const uint32_t o = (uint32_t)__ffsll((unsigned long long)bits) - 1u;
      bits &= bits - 1;
      const uint64_t sa = ix.loc_vals[sbase + (uint64_t)__popcll(sampled & below(o))];
      multi_push(S, len + sa);
      if (single_returns && e - f == 1) return true;
    }
  }
  return false;
}

// start_sbwt_match's locate (searcher.hpp:424-449) for the rows [lo, hi): level 0, then find_possible (searcher.hpp:460-503)
// with the recursion unrolled onto an explicit stack (frames 1 .. interval-1)
__device__ __forceinline__ void multi_walk(const DeviceIndex& ix, MultiState& S, Frame* stack, uint64_t lo, uint64_t hi) {
  // VARS.maybe_result_number is an int compared against a size_t (searcher.hpp:412,464)
  const uint64_t maybe = (uint64_t)(int64_t)(int32_t)(uint32_t)(hi - lo);
  // level 0 (searcher.hpp:424-443); its break leaves a loop over ONE row after that row
  scan_sampled(S, lo, hi, 0, false);
  uint32_t d = 1;
  stack[1].pf = lo; stack[1].pe = hi; stack[1].next_c = 0;
  while (d >= 1) {
    Frame& f = stack[d];
    if (f.next_c == 4) { --d; continue; }
    const uint32_t c = f.next_c++;
    if ((uint64_t)d >= ix.interval) continue;
    if (S.nres == maybe) continue;
    const uint64_t nf = lf1(ix, f.pf, c), ne = lf1(ix, f.pe, c);
    if (nf >= ne) continue;
    if (scan_sampled(S, nf, ne, d, true)) continue;
    if ((uint64_t)d + 1 < ix.interval) {  // a deeper call would return at once (len >= interval)
      ++d;
      stack[d].pf = nf; stack[d].pe = ne; stack[d].next_c = 0;
    }
  }
}

__global__ __launch_bounds__(256) void k_multi(const DeviceIndex ix, const MapArgs a, Frame* __restrict__ stacks,
                                               uint32_t lanes_total, const uint32_t* __restrict__ queue,
                                               const unsigned long long* __restrict__ queue_count) {
  const uint32_t lane_global = blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t qn = *queue_count;
  Frame* stack = stacks + (uint64_t)lane_global * ix.interval;  // frames 1 .. interval-1
  for (uint64_t q = lane_global; q < qn; q += lanes_total) {
    const uint32_t r = queue[q];
    if (r == kQueueDone) continue;  // k_multi_bfs took it
    const MultiRead res = rio_load(a.io, r);
    const uint64_t lo = res.lo, hi = res.hi;
    if (hi <= lo) continue;  // (a record in a wide[] slot past wide_cap was never written: the batch fails with SBWT_ERR_CAPACITY)
    MultiState S;
    S.ix = &ix;
    S.out = a.hits + res.first_hit;
    S.cap = hi - lo;
    S.nres = 0;
    S.L = a.lens[r];
    S.strand = a.strand;
    S.truncated = 0;
    multi_walk(ix, S, stack, lo, hi);
    uint64_t first = res.first_hit;
    if (S.truncated && list_fits_record(a.io, r, S.nres)) {
      // more hits than rows (multi_locate.h, extra_slots): the walk again, into room of its own
      const uint64_t e = extra_slots(&a.ctr->hit_count, &a.ctr->overflow, a.multi_base, a.hits_cap, S.nres);
      if (e != ~0ull) {
        first = e;
        S.out = a.hits + e;
        S.cap = S.nres;
        S.nres = 0;
        S.truncated = 0;
        multi_walk(ix, S, stack, lo, hi);
      }
    }
    rio_store_multi(a.io, r, first, true, (int32_t)(S.nres < S.cap ? S.nres : S.cap), S.truncated ? SBWT_RF_TRUNCATED : 0u);
    atomicAdd(&a.ctr->rows_located, (unsigned long long)(hi - lo));  // (the few reads nothing else took)
  }
}

}  // namespace sbwt

// ---------------------------------------------------------------------------------------------------
using namespace sbwt;

// test knobs (tests/test_gpu_rare_paths.py): an environment variable lowers a limit so that small inputs reach
// the code large ones would
static uint32_t debug_knob(const char* name, uint32_t dflt) {
  const char* e = getenv(name);
  if (!e || !*e) return dflt;
  const unsigned long v = strtoul(e, nullptr, 10);
  return v < dflt ? (uint32_t)v : dflt;
}

// What one device pass needs besides its inputs and outputs: the stream, the scratch buffers of the kernels and the
// launch shapes.  A sbwt_ctx owns one (hipMalloc at sbwt_ctx_create); sbwt_map_batch_device_static points one at
// the file-scope scratch array of the device.
constexpr int kNumStats = 14;  // sbwt_ctx_last_stats_ex
constexpr uint32_t kModeSkipHash = 1u << 31;  // internal mode bit of map_pass: every read of the batch already missed hash_tab

struct MapWork {
  hipStream_t stream = nullptr;
  Counters* d_ctr = nullptr;
  Counters* h_ctr = nullptr;  // pinned
  uint32_t* d_queue = nullptr;      // max_reads entries
  uint32_t* d_dfs_queue = nullptr;  // max_reads entries
  WaveInfo* d_wave_info = nullptr;  // one per wave of k_search_sm
  Frame* d_stacks = nullptr;        // multi_lanes * interval frames
  uint32_t multi_lanes = 0;
  uint32_t grid_sm_lds = 0, grid_sm_glb = 0;  // resident workgroups of k_search_sm
  uint32_t grid_sort = 0, grid_bfs = 0;       // ... of k_multi_sort, k_multi_bfs
  size_t sm_lds_bytes = 0;                    // 0: the filter tables / sb_abs do not fit the LDS budget
  size_t sm_ring_bytes = 0;                   // result ring of k_search_sm, always in LDS
  size_t sort_lds_bytes = 0;                  // filter tables of k_multi_sort in LDS (0: they stay in global memory)
  unsigned long long* d_group_keys = nullptr; // grouping of multi-hit reads with one interval and length (multi_locate.h)
  uint32_t* d_group_leader = nullptr;
  uint32_t group_slots = 0;                   // a power of two; 0: no grouping (the no-allocation entry point)
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // null: the pass is not timed ([3], [4]: around locate kernels launched in a second phase)
  // test knobs (environment variables, read ONCE: when the ctx is made, or -- the entry point without a ctx -- the index)
  bool no_bfs = false, no_chain = false, no_rounds = false;
  uint32_t grid_rounds_lds = 0, grid_rounds_glb = 0;  // k_search_rounds (search_rounds.h)
  uint32_t no_segments = 0;  // SBWT_NO_SEGMENTS: 1 = wave teams locate one read at a time, 2 = no segments of 8 entries
  uint32_t ring_width_max = (uint32_t)kRingWidthMax, ring_chr_max = 0xFFFFu;
};

constexpr uint32_t kFastGridMax = 16384;  // workgroups of a fast-pass launch (SettleArgs::wg_stats has room for as many)

static void read_pass_knobs(bool* no_bfs, bool* no_chain, bool* no_rounds, uint32_t* no_segments, uint32_t* ring_width_max, uint32_t* ring_chr_max) {
  *no_rounds = getenv("SBWT_NO_ROUNDS") != nullptr;
  const char* seg = getenv("SBWT_NO_SEGMENTS");
  *no_segments = seg && *seg ? (uint32_t)strtoul(seg, nullptr, 10) : 0u;
  *no_bfs = getenv("SBWT_NO_BFS") != nullptr;
  *no_chain = getenv("SBWT_NO_CHAIN") != nullptr;
  *ring_width_max = debug_knob("SBWT_DEBUG_RING_WIDTH", (uint32_t)kRingWidthMax);
  *ring_chr_max = debug_knob("SBWT_DEBUG_RING_CHR", 0xFFFFu);
}

struct sbwt_ctx : MapWork {
  const sbwt_index* ix = nullptr;
  uint64_t max_reads = 0;
  hipStream_t s_in = nullptr, s_out = nullptr;  // copies of the sub-batched host-buffer path
  hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_k = nullptr;
  float last_ms[3] = {0, 0, 0};
  uint64_t last_stats[kNumStats] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  // staging for the host-buffer entry point
  uint32_t *d_packed = nullptr, *d_lens = nullptr;
  sbwt_read_result* d_results = nullptr;
  sbwt_hit* d_hits = nullptr;
  sbwt_wide_result* d_wide = nullptr;
  uint64_t staged_words = 0, staged_reads = 0, staged_hits = 0, staged_wide = 0;
  // hits-only fast pass (settle.h): the list of pending reads, and their dense batch (grown on demand)
  uint32_t* d_pend_list = nullptr;  // max_reads entries
  uint32_t *d_pend_packed = nullptr, *d_pend_lens = nullptr;
  sbwt_compact_result* d_pend_out = nullptr;
  sbwt_read_result* d_pend_res = nullptr;  // 32-byte records and unique hits of the pending reads (sbwt_map_batch_device)
  sbwt_hit* d_pend_hits = nullptr;
  uint64_t pend_words = 0, pend_lens_cap = 0, pend_out_cap = 0, pend_res_cap = 0, pend_hits_cap = 0;
  hipEvent_t ev_s[2] = {nullptr, nullptr};
  uint32_t grid_settle = 0, grid_direct = 0, grid_rank = 0;
  unsigned long long* d_wg_stats = nullptr;  // SettleArgs::wg_stats: kFastGridMax x 5 tallies
  size_t rank_lds_bytes = 0;  // k_rank: the filter tables and sb_abs in LDS (0: from global memory)
  bool no_rank = false;       // SBWT_NO_RANK_PASS
  float last_ms_ex[2] = {0, 0};  // k_settle; the pass over the pending reads
  bool no_settle = false, no_direct = false, no_spec = false;  // test knobs, read when the ctx is made
  uint64_t subbatch = 2u << 20;               // reads per sub-batch of the host-buffer entry points (SBWT_SUBBATCH; 0 = never split)
};

// Launch shapes of the kernels for this index on its device: resident workgroups per kernel variant and the LDS
// each takes.  Allocates nothing.
static hipError_t plan_launches(const sbwt_index* ix, MapWork* w) {
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, ix->device);
  if (e != hipSuccess) return e;
  const uint64_t n_super = ((ix->d.n_blocks - 1) >> (kSuperShift - 6)) + 1;
  const size_t ring_bytes = (size_t)4 * (kSmRing + kSmStage) * 16;  // four waves per workgroup: the result ring and the stage of the Block2 lines
  w->sm_ring_bytes = ring_bytes;
  w->sm_lds_bytes = 0;
  w->sort_lds_bytes = 0;
  const uint32_t max_keys = debug_knob("SBWT_DEBUG_LDS_KEYS", kSmMaxKeys);
  if (ix->d.n_chr + ix->d.n_npos <= max_keys && n_super <= kSmMaxSuper)
    w->sm_lds_bytes = (((size_t)ix->d.n_chr + 2 * (size_t)ix->d.n_npos + 16 * n_super + 1) & ~(size_t)1) * sizeof(uint64_t);
  int sm_lds = 0, sm_glb = 0, sort_per_cu = 0;
  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&sm_lds, k_search_sm<true, true>, 256, w->sm_lds_bytes + ring_bytes);
  if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&sm_glb, k_search_sm<false, true>, 256, ring_bytes);
  if (ix->d.n_chr + ix->d.n_npos <= max_keys) w->sort_lds_bytes = ((size_t)ix->d.n_chr + 2 * (size_t)ix->d.n_npos) * sizeof(uint64_t);
  if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&sort_per_cu, k_multi_sort<256, true>, 256, w->sort_lds_bytes);
  if (e != hipSuccess) return e;
  w->grid_sort = (uint32_t)(prop.multiProcessorCount * (sort_per_cu > 0 ? sort_per_cu : 1));
  int bfs_per_cu = 0;
  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&bfs_per_cu, k_multi_bfs<256, 512>, 256, w->sort_lds_bytes);
  if (e != hipSuccess) return e;
  w->grid_bfs = (uint32_t)(prop.multiProcessorCount * (bfs_per_cu > 0 ? bfs_per_cu : 1));
  int rd_lds = 0, rd_glb = 0;
  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&rd_lds, k_search_rounds<2, true>, 256, w->sm_lds_bytes);
  if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&rd_glb, k_search_rounds<2, false>, 256, 0);
  if (e != hipSuccess) return e;
  w->grid_rounds_lds = (uint32_t)(prop.multiProcessorCount * (rd_lds > 0 ? rd_lds : 1));
  w->grid_rounds_glb = (uint32_t)(prop.multiProcessorCount * (rd_glb > 0 ? rd_glb : 1));
  w->grid_sm_lds = (uint32_t)(prop.multiProcessorCount * (sm_lds > 0 ? sm_lds : 1));
  w->grid_sm_glb = (uint32_t)(prop.multiProcessorCount * (sm_glb > 0 ? sm_glb : 1));
  return hipSuccess;
}

extern "C" int sbwt_ctx_create(const sbwt_index* ix, uint64_t max_reads, sbwt_ctx** out) {
  if (!ix || !out || max_reads == 0 || max_reads > 0xffffffffull) {
    set_error("sbwt_ctx_create: bad argument (max_reads must be in 1..2^32-1)");
    return SBWT_ERR_ARG;
  }
  SBWT_HIP(enter_device(ix->device));
  sbwt_ctx* c = new (std::nothrow) sbwt_ctx;
  if (!c) return SBWT_ERR_NOMEM;
  c->ix = ix;
  c->max_reads = max_reads;
  read_pass_knobs(&c->no_bfs, &c->no_chain, &c->no_rounds, &c->no_segments, &c->ring_width_max, &c->ring_chr_max);
  c->no_settle = getenv("SBWT_NO_SETTLE") != nullptr;
  c->no_direct = getenv("SBWT_NO_DIRECT") != nullptr;
  c->no_rank = getenv("SBWT_NO_RANK_PASS") != nullptr;
  c->no_spec = getenv("SBWT_NO_SPECULATIVE_PENDING") != nullptr;
  if (const char* sb = getenv("SBWT_SUBBATCH")) c->subbatch = strtoull(sb, nullptr, 10);
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc(&c->d_ctr, 2 * sizeof(Counters));  // [0] the general kernels, [1] the fast pass
  if (e == hipSuccess) e = hipHostMalloc(&c->h_ctr, 2 * sizeof(Counters));
  if (e == hipSuccess) e = hipMalloc(&c->d_queue, max_reads * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMalloc(&c->d_dfs_queue, max_reads * sizeof(uint32_t));
  if (e == hipSuccess && (ix->d.hash_text || ix->d.pos_lines)) e = hipMalloc(&c->d_pend_list, max_reads * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMalloc(&c->d_wg_stats, (size_t)kFastGridMax * 5 * sizeof(unsigned long long));
  if (e == hipSuccess && !getenv("SBWT_NO_SHARED_LISTS")) {
    uint32_t slots = 1u << 12;
    while (slots < max_reads && slots < (1u << 23)) slots <<= 1;
    e = hipMalloc(&c->d_group_keys, (size_t)slots * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc(&c->d_group_leader, (size_t)slots * sizeof(uint32_t));
    if (e == hipSuccess) c->group_slots = slots;
  }
  for (int k = 0; k < 2 && e == hipSuccess; ++k) e = hipEventCreate(&c->ev_s[k]);
  if (e == hipSuccess) {
    int per_cu = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_settle<true>, 256, ((size_t)ix->d.n_chr + 2 * (size_t)ix->d.n_npos) * sizeof(uint64_t));
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, ix->device);
    if (e == hipSuccess) c->grid_settle = (uint32_t)(prop.multiProcessorCount * (per_cu > 0 ? per_cu : 1));
    int per_cu_d = 0;
    if (e == hipSuccess)
      e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_d, k_direct<3, true>, 256, ((size_t)ix->d.n_chr + 2 * (size_t)ix->d.n_npos) * sizeof(uint64_t));
    if (e == hipSuccess) c->grid_direct = (uint32_t)(prop.multiProcessorCount * (per_cu_d > 0 ? per_cu_d : 1));
  }
  // k_multi runs a fixed number of lanes, each with its own DFS stack of `interval` frames
  c->multi_lanes = 256u * 1024u;
  if (c->multi_lanes > max_reads) c->multi_lanes = (uint32_t)((max_reads + 255) / 256 * 256);
  if (e == hipSuccess) e = hipMalloc(&c->d_stacks, (uint64_t)c->multi_lanes * ix->d.interval * sizeof(Frame));
  for (int k = 0; k < 5 && e == hipSuccess; ++k) e = hipEventCreate(&c->ev[k]);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->s_in, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->s_out, hipStreamNonBlocking);
  for (int k = 0; k < 2 && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&c->ev_in[k], hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_k, hipEventDisableTiming);
  if (e == hipSuccess) e = plan_launches(ix, c);
  if (e == hipSuccess) {
    // k_rank (rank_pass.h): the filter tables (sort_lds_bytes, set by plan_launches) and sb_abs
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, ix->device);
    const size_t n_super = (size_t)(((ix->d.n_blocks - 1) >> (kSuperShift - 6)) + 1);
    c->rank_lds_bytes = c->sort_lds_bytes ? c->sort_lds_bytes + 16 * n_super * sizeof(uint64_t) : 0;
    if (c->rank_lds_bytes > 48 * 1024) c->rank_lds_bytes = 0;
    int per_cu_r = 0;
    if (e == hipSuccess) {
      e = c->rank_lds_bytes ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_r, k_rank<2, true>, 256, c->rank_lds_bytes)
                            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_r, k_rank<2, false>, 256, 0);
    }
    if (e == hipSuccess) c->grid_rank = (uint32_t)(prop.multiProcessorCount * (per_cu_r > 0 ? per_cu_r : 1));
  }
  if (e == hipSuccess)
    e = hipMalloc(&c->d_wave_info, (size_t)std::max(c->grid_sm_lds, c->grid_sm_glb) * 4 * sizeof(WaveInfo));
  if (e != hipSuccess) {
    sbwt_ctx_free(c);
    return hip_fail(e, "sbwt_ctx_create", __FILE__, __LINE__);
  }
  *out = c;
  return SBWT_OK;
}

extern "C" void sbwt_ctx_free(sbwt_ctx* c) {
  if (!c) return;
  hipSetDevice(c->ix->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  hipFree(c->d_ctr);
  if (c->h_ctr) hipHostFree(c->h_ctr);
  hipFree(c->d_queue);
  hipFree(c->d_dfs_queue);
  hipFree(c->d_group_keys);
  hipFree(c->d_group_leader);
  hipFree(c->d_wave_info);
  hipFree(c->d_stacks);
  hipFree(c->d_packed); hipFree(c->d_lens); hipFree(c->d_results); hipFree(c->d_hits); hipFree(c->d_wide);
  hipFree(c->d_wg_stats);
  hipFree(c->d_pend_list); hipFree(c->d_pend_packed); hipFree(c->d_pend_lens); hipFree(c->d_pend_out); hipFree(c->d_pend_res); hipFree(c->d_pend_hits);
  for (int k = 0; k < 2; ++k)
    if (c->ev_s[k]) hipEventDestroy(c->ev_s[k]);
  for (int k = 0; k < 5; ++k)
    if (c->ev[k]) hipEventDestroy(c->ev[k]);
  for (int k = 0; k < 2; ++k)
    if (c->ev_in[k]) hipEventDestroy(c->ev_in[k]);
  if (c->ev_k) hipEventDestroy(c->ev_k);
  if (c->s_in) hipStreamDestroy(c->s_in);
  if (c->s_out) hipStreamDestroy(c->s_out);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
}

extern "C" void* sbwt_ctx_stream(const sbwt_ctx* c) { return c ? (void*)c->stream : nullptr; }

extern "C" int sbwt_ctx_last_stats(const sbwt_ctx* c, uint64_t stats[4]) {
  if (!c || !stats) return SBWT_ERR_ARG;
  stats[0] = c->last_stats[0];
  stats[1] = c->last_stats[1];
  stats[2] = c->last_stats[2];
  stats[3] = c->last_stats[3];
  return SBWT_OK;
}

extern "C" int sbwt_ctx_last_stats_ex(const sbwt_ctx* c, uint64_t* stats, uint32_t n) {
  if (!c || (!stats && n)) return SBWT_ERR_ARG;
  for (uint32_t k = 0; k < n; ++k) stats[k] = k < (uint32_t)kNumStats ? c->last_stats[k] : 0;
  return SBWT_OK;
}

extern "C" int sbwt_ctx_last_timing_ex(const sbwt_ctx* c, float* ms, uint32_t n) {
  if (!c || (!ms && n)) return SBWT_ERR_ARG;
  for (uint32_t k = 0; k < n; ++k) ms[k] = k < 3 ? c->last_ms[k] : (k < 5 ? c->last_ms_ex[k - 3] : 0.f);
  return SBWT_OK;
}

extern "C" int sbwt_ctx_last_timing(const sbwt_ctx* c, float ms[3]) {
  if (!c || !ms) return SBWT_ERR_ARG;
  for (int k = 0; k < 3; ++k) ms[k] = c->last_ms[k];
  return SBWT_OK;
}

// first_hit of every read with hits += bias (a sub-batch numbers its reads and its hit slots from 0)
__global__ __launch_bounds__(256) void k_bias_first_hit(sbwt_read_result* __restrict__ results, uint64_t n, uint64_t bias) {
  const uint64_t r = global_thread();
  if (r < n && results[r].nh > 0) results[r].first_hit += bias;
}

// One device pass over n_reads reads.  Unique hits go to d_hits[read index], multi-hit lists to d_hits[multi_base ..)
// (multi_base = n_reads for a whole batch); *multi_used = slots the lists took.  Synchronises the stream of `w`.
// ms (3 floats) / stats (4 counters) may be null; with `accumulate` they are added to.
static int map_pass(const sbwt_index* ix, const MapWork& w, const uint32_t* d_packed, const uint32_t* d_lens, uint64_t n_reads,
                    uint32_t words_per_read, int limit, int strand, const ResultIO& io, uint32_t mode, sbwt_hit* d_hits,
                    uint64_t hits_cap, uint64_t multi_base, uint64_t* multi_used, uint64_t* wide_used, bool accumulate,
                    float* ms_out, uint64_t* stats_out);
static int map_pass_compact(sbwt_ctx* c, const uint32_t* d_packed, const uint32_t* d_lens, uint64_t n_reads, uint32_t words_per_read,
                            int limit, int strand, const ResultIO& io, uint32_t mode, sbwt_hit* d_hits, uint64_t hits_cap,
                            uint64_t multi_base, uint64_t* multi_used, uint64_t* wide_used, bool accumulate);
static int map_pass(sbwt_ctx* c, const uint32_t* d_packed, const uint32_t* d_lens, uint64_t n_reads,
                    uint32_t words_per_read, int limit, int strand, sbwt_read_result* d_results, sbwt_hit* d_hits,
                    uint64_t hits_cap, uint64_t multi_base, uint64_t* multi_used, bool accumulate) {
  const ResultIO io{d_results, nullptr, nullptr, 0};
  // (through the two-access fast pass where the index has what it needs, else the general kernels alone)
  return map_pass_compact(c, d_packed, d_lens, n_reads, words_per_read, limit, strand, io, 0u, d_hits, hits_cap, multi_base, multi_used,
                          nullptr, accumulate);
}

extern "C" int sbwt_map_batch_device(sbwt_ctx* c, const uint32_t* d_packed, const uint32_t* d_lens, uint64_t n_reads,
                                     uint32_t words_per_read, int limit, int strand, sbwt_read_result* d_results,
                                     sbwt_hit* d_hits, uint64_t hits_cap, uint64_t* n_hits_host) {
  if (!c || (n_reads && (!d_packed || !d_lens || !d_results)) || (hits_cap && !d_hits)) {
    set_error("sbwt_map_batch_device: null pointer");
    return SBWT_ERR_ARG;
  }
  if (n_reads > c->max_reads || words_per_read == 0 || (words_per_read & 3u)) {
    set_error("sbwt_map_batch_device: n_reads exceeds the ctx capacity or words_per_read is not a multiple of 4");
    return SBWT_ERR_ARG;
  }
  if (n_hits_host) *n_hits_host = 0;
  if (n_reads == 0) return SBWT_OK;
  if (hits_cap < n_reads) {
    // unique hits are written at their read's index: hits[] holds one slot per read, multi-hit lists behind them
    char msg[160];
    snprintf(msg, sizeof msg, "hits buffer too small: at least one slot per read (%llu) is needed, %llu provided",
             (unsigned long long)n_reads, (unsigned long long)hits_cap);
    set_error(msg);
    if (n_hits_host) *n_hits_host = n_reads;
    return SBWT_ERR_CAPACITY;
  }
  SBWT_HIP(enter_device(c->ix->device));
  uint64_t multi_used = 0;
  const int rc = map_pass(c, d_packed, d_lens, n_reads, words_per_read, limit, strand, d_results, d_hits, hits_cap, n_reads,
                          &multi_used, false);
  const uint64_t used = n_reads + multi_used;  // one slot per read + the multi-hit lists
  c->last_stats[2] = used;
  if (n_hits_host) *n_hits_host = used;        // on SBWT_ERR_CAPACITY: what a retry needs
  return rc;
}

// One device pass over a batch, in three pieces so that the pending pass of a fast pass can be launched before the
// host knows how many reads it holds:
//   pass_search   counters zeroed, k_search_sm (the batch size possibly read from device memory), k_multi_plan
//   pass_multi    the locate kernels of the queued multi-hit reads (the host knows the batch size by then)
//   pass_collect  the counters (copied back and synchronised by the caller) into timings, statistics and the verdict
struct PassState {
  MapArgs a;
  MultiSortArgs ms;
  uint32_t n_waves = 0;
};

static int pass_search(const sbwt_index* ix, const MapWork& w, const uint32_t* d_packed, const uint32_t* d_lens, uint64_t n_reads,
                       uint32_t words_per_read, int limit, int strand, const ResultIO& io, uint32_t mode, sbwt_hit* d_hits,
                       uint64_t hits_cap, uint64_t multi_base, const unsigned long long* n_reads_dev,
                       const unsigned long long* hash_skip_dev, PassState* P) {
  MapArgs& a = P->a;
  a.packed = d_packed;
  a.lens = d_lens;
  a.n_reads = n_reads;
  a.wpr = words_per_read;
  a.strand = strand;
  a.limit = (uint64_t)(int64_t)limit;
  a.io = io;
  a.hits = d_hits;
  a.hits_cap = hits_cap;
  a.multi_base = multi_base;
  a.ctr = w.d_ctr;
  a.multi_queue = w.d_queue;
  SBWT_HIP(hipMemsetAsync(w.d_ctr, 0, sizeof(Counters), w.stream));
  const bool timed = w.ev[0] != nullptr;
  if (timed) SBWT_HIP(hipEventRecord(w.ev[0], w.stream));
  const DeviceIndex& d = ix->d;
  {
    SmArgs sa;
    sa.packed = d_packed;
    sa.lens = d_lens;
    sa.n_reads = n_reads;
    sa.wpr = words_per_read;
    sa.strand = strand;
    sa.limit = a.limit;
    sa.results = io.results;
    sa.compact = io.compact;
    sa.wide = io.wide;
    sa.wide_cap = io.wide_cap;
    sa.wide_count = &w.d_ctr->wide_count;
    sa.skip_hash = (mode & kModeSkipHash) ? 1u : 0u;
    sa.n_reads_dev = n_reads_dev;
    sa.hash_skip_dev = hash_skip_dev;
    sa.hits = io.unique_hits ? io.unique_hits : d_hits;
    sa.hits_cap = hits_cap;
    sa.multi_queue = w.d_queue;
    sa.wave_info = w.d_wave_info;
    sa.index_accesses = &w.d_ctr->index_accesses;
    sa.hash_stats = w.d_ctr->hash_stats;
    sa.ring_width_max = w.ring_width_max;
    sa.ring_chr_max = w.ring_chr_max;
    const bool lds = w.sm_lds_bytes != 0;
    const uint64_t need = (n_reads + 255) / 256;  // a wave with fewer than 64 reads has nothing to overlap
    // records of up to 128 bases: the search in lockstep rounds (search_rounds.h); the state machine for longer ones
    const bool rounds = words_per_read <= 8 && d.pos_lines && !w.no_rounds;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(need, rounds ? (lds ? w.grid_rounds_lds : w.grid_rounds_glb) : lds ? w.grid_sm_lds : w.grid_sm_glb);
    // records of more than 64 bases: the variant that holds two 64-base pieces and compares up to 128 bases at once
    const bool long_reads = words_per_read > 4;
    if (rounds && lds && long_reads) k_search_rounds<2, true><<<grid, 256, w.sm_lds_bytes, w.stream>>>(d, sa);
    else if (rounds && lds) k_search_rounds<1, true><<<grid, 256, w.sm_lds_bytes, w.stream>>>(d, sa);
    else if (rounds && long_reads) k_search_rounds<2, false><<<grid, 256, 0, w.stream>>>(d, sa);
    else if (rounds) k_search_rounds<1, false><<<grid, 256, 0, w.stream>>>(d, sa);
    else if (lds && long_reads) k_search_sm<true, true><<<grid, 256, w.sm_lds_bytes + w.sm_ring_bytes, w.stream>>>(d, sa);
    else if (lds) k_search_sm<true, false><<<grid, 256, w.sm_lds_bytes + w.sm_ring_bytes, w.stream>>>(d, sa);
    else if (long_reads) k_search_sm<false, true><<<grid, 256, w.sm_ring_bytes, w.stream>>>(d, sa);
    else k_search_sm<false, false><<<grid, 256, w.sm_ring_bytes, w.stream>>>(d, sa);
    SBWT_LAUNCHED();
    P->n_waves = grid * 4;
  }
  if (timed) SBWT_HIP(hipEventRecord(w.ev[1], w.stream));
  {
    k_multi_plan<<<1, 1024, 0, w.stream>>>(w.d_wave_info, P->n_waves, multi_base, hits_cap, &w.d_ctr->hit_count,
                                           &w.d_ctr->multi_count, &w.d_ctr->overflow);
    SBWT_LAUNCHED();
    MultiSortArgs& ms = P->ms;
    ms.lens = d_lens;
    ms.strand = strand;
    ms.io = io;
    ms.hits = d_hits;
    ms.multi_base = multi_base;
    ms.queue = w.d_queue;
    ms.wave_info = w.d_wave_info;
    ms.n_waves = P->n_waves;
    ms.chunk = 0;  // pass_multi
    ms.queue_count = &w.d_ctr->multi_count;
    ms.overflow = &w.d_ctr->overflow;
    ms.hit_count = &w.d_ctr->hit_count;
    ms.overflow_flag = &w.d_ctr->overflow;
    ms.hits_cap = hits_cap;
    ms.dfs_queue = w.d_dfs_queue;
    ms.dfs_count = &w.d_ctr->dfs_count;
    ms.lds_tables = w.sort_lds_bytes ? 1u : 0u;
    ms.shared_count = &w.d_ctr->shared_count;
    ms.bfs_count = &w.d_ctr->bfs_count;
    ms.rows_count = &w.d_ctr->rows_located;
    ms.no_segments = w.no_segments;
    ms.grp.keys = nullptr;
    ms.grp.leader = nullptr;
    ms.grp.mask = 0;
  }
  return SBWT_OK;
}

// multi-hit reads: gather + sort where the tables allow it, memoised chains or the upstream depth-first walk for the rest
static int pass_multi(const sbwt_index* ix, const MapWork& w, PassState& P, uint64_t n_reads) {
  const DeviceIndex& d = ix->d;
  MultiSortArgs& ms = P.ms;
  const MapArgs& a = P.a;
  const uint32_t n_waves = P.n_waves;
  ms.chunk = ((n_reads + n_waves - 1) / n_waves + 63) & ~63ull;  // as k_search_sm split the batch
  {
    if (w.group_slots) {
      uint32_t slots = 1u << 12;  // as many as this batch can fill, at most what the ctx holds
      while (slots < n_reads && slots < w.group_slots) slots <<= 1;
      if (slots > w.group_slots) slots = w.group_slots;
      ms.grp.keys = w.d_group_keys;
      ms.grp.leader = w.d_group_leader;
      ms.grp.mask = slots - 1u;
      SBWT_HIP(hipMemsetAsync(w.d_group_keys, 0xFF, (size_t)slots * sizeof(unsigned long long), w.stream));
      SBWT_HIP(hipMemsetAsync(w.d_group_leader, 0xFF, (size_t)slots * sizeof(uint32_t), w.stream));
      k_multi_group<<<n_waves, 256, 0, w.stream>>>(ms);
      SBWT_LAUNCHED();
    }
    if (w.sort_lds_bytes) {
      k_multi_sort<64, true><<<w.grid_sort, 256, w.sort_lds_bytes, w.stream>>>(d, ms);
      k_multi_sort<256, true><<<w.grid_sort, 256, w.sort_lds_bytes, w.stream>>>(d, ms);
    } else {
      k_multi_sort<64, false><<<w.grid_sort, 256, 0, w.stream>>>(d, ms);
      k_multi_sort<256, false><<<w.grid_sort, 256, 0, w.stream>>>(d, ms);
    }
    SBWT_LAUNCHED();
    if (d.chain_tab && !w.no_chain) {  // the reads k_multi_sort handed on, from memoised chains (multi_locate.h); what it leaves stays queued
      if (w.sort_lds_bytes) {
        k_multi_chain<64, true><<<w.grid_sort, 256, w.sort_lds_bytes, w.stream>>>(d, ms, w.d_dfs_queue, &w.d_ctr->dfs_count, &w.d_ctr->chain_count);
        k_multi_chain<256, true><<<w.grid_sort, 256, w.sort_lds_bytes, w.stream>>>(d, ms, w.d_dfs_queue, &w.d_ctr->dfs_count, &w.d_ctr->chain_count);
      } else {
        k_multi_chain<64, false><<<w.grid_sort, 256, 0, w.stream>>>(d, ms, w.d_dfs_queue, &w.d_ctr->dfs_count, &w.d_ctr->chain_count);
        k_multi_chain<256, false><<<w.grid_sort, 256, 0, w.stream>>>(d, ms, w.d_dfs_queue, &w.d_ctr->dfs_count, &w.d_ctr->chain_count);
      }
      SBWT_LAUNCHED();
    }
    if (!w.no_bfs) {  // the same walk level by level, a team per read (multi_locate.h); k_multi takes what is left
      k_multi_bfs<64, 64><<<w.grid_bfs, 256, w.sort_lds_bytes, w.stream>>>(d, ms, w.d_dfs_queue, &w.d_ctr->dfs_count);
      k_multi_bfs<256, 512><<<w.grid_bfs, 256, w.sort_lds_bytes, w.stream>>>(d, ms, w.d_dfs_queue, &w.d_ctr->dfs_count);
      k_multi_bfs<256, 1024><<<w.grid_bfs / 2, 256, w.sort_lds_bytes, w.stream>>>(d, ms, w.d_dfs_queue, &w.d_ctr->dfs_count);  // frontiers the second outgrew
      SBWT_LAUNCHED();
    }
    k_multi<<<w.multi_lanes / 256, 256, 0, w.stream>>>(d, a, w.d_stacks, w.multi_lanes, w.d_dfs_queue, &w.d_ctr->dfs_count);
    SBWT_LAUNCHED();
    if (w.group_slots) {
      k_multi_share<<<n_waves, 256, 0, w.stream>>>(ms);
      SBWT_LAUNCHED();
    }
  }
  return SBWT_OK;
}

static int pass_collect(const MapWork& w, const ResultIO& io, uint64_t hits_cap, uint64_t multi_base, uint64_t* multi_used,
                        uint64_t* wide_used, bool accumulate, float* ms_out, uint64_t* stats_out, bool second_phase) {
  const bool timed = w.ev[0] != nullptr;
#ifdef SBWT_PHASE_CLOCKS
  {
    unsigned long long ph[24] = {0}, zero[24] = {0};
    SBWT_HIP(hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_multi_phase), sizeof(ph)));
    SBWT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_multi_phase), zero, sizeof(zero)));
    fprintf(stderr, "k_multi_sort phase clocks (scan, record, sa_full, text, keys, decide, radix, compare-sort, write | reads, radix reads)\n");
    for (int t = 0; t < 2; ++t) {
      fprintf(stderr, t ? "  k_multi_sort<256>:" : "  k_multi_chain<256> (rows + entries, prune, emit + filters, sort, cut + write, -, -, -, -, entries | reads, radix reads):");
      for (int k = 0; k < 10; ++k) fprintf(stderr, " %.3g", (double)ph[12 * t + k]);
      fprintf(stderr, " | %llu %llu\n", ph[12 * t + 10], ph[12 * t + 11]);
    }
  }
#endif
  if (ms_out) {
    float ms[3] = {0, 0, 0};
    if (timed) {
      hipEventElapsedTime(&ms[0], w.ev[0], w.ev[2]);
      hipEventElapsedTime(&ms[1], w.ev[0], w.ev[1]);
      hipEventElapsedTime(&ms[2], w.ev[1], w.ev[2]);
      if (second_phase) {  // the locate kernels ran after the counters had been read once
        float more = 0.f;
        hipEventElapsedTime(&more, w.ev[3], w.ev[4]);
        ms[0] += more;
        ms[2] += more;
      }
    }
    for (int k = 0; k < 3; ++k) ms_out[k] = accumulate ? ms_out[k] + ms[k] : ms[k];
  }
  if (stats_out) {
    stats_out[0] = (accumulate ? stats_out[0] : 0) + w.h_ctr->index_accesses;
    stats_out[1] = (accumulate ? stats_out[1] : 0) + w.h_ctr->multi_count;
    stats_out[3] = (accumulate ? stats_out[3] : 0) + w.h_ctr->dfs_count;
    stats_out[4] = (accumulate ? stats_out[4] : 0) + w.h_ctr->hash_stats[0];
    stats_out[5] = (accumulate ? stats_out[5] : 0) + w.h_ctr->hash_stats[1];
    stats_out[8] = (accumulate ? stats_out[8] : 0) + w.h_ctr->shared_count;
    stats_out[9] = (accumulate ? stats_out[9] : 0) + w.h_ctr->bfs_count;
    stats_out[11] = (accumulate ? stats_out[11] : 0) + w.h_ctr->chain_count;
    stats_out[12] = (accumulate ? stats_out[12] : 0) + w.h_ctr->rows_located;
  }
  *multi_used = w.h_ctr->hit_count;
  if (wide_used) *wide_used = w.h_ctr->wide_count;
  if (io.compact && w.h_ctr->wide_count > io.wide_cap) {
    char msg[200];
    snprintf(msg, sizeof msg, "wide[] too small: %llu results do not fit a 16-byte record, room for %llu",
             (unsigned long long)w.h_ctr->wide_count, (unsigned long long)io.wide_cap);
    set_error(msg);
    return SBWT_ERR_CAPACITY;
  }
  if (w.h_ctr->overflow) {
    char msg[160];
    snprintf(msg, sizeof msg, "hits buffer too small: %llu slots needed, %llu provided",
             (unsigned long long)(multi_base + w.h_ctr->hit_count), (unsigned long long)hits_cap);
    set_error(msg);
    return SBWT_ERR_CAPACITY;
  }
  return SBWT_OK;
}

static int map_pass(const sbwt_index* ix, const MapWork& w, const uint32_t* d_packed, const uint32_t* d_lens, uint64_t n_reads,
                    uint32_t words_per_read, int limit, int strand, const ResultIO& io, uint32_t mode, sbwt_hit* d_hits,
                    uint64_t hits_cap, uint64_t multi_base, uint64_t* multi_used, uint64_t* wide_used, bool accumulate,
                    float* ms_out, uint64_t* stats_out) {
  PassState P;
  if (int rc = pass_search(ix, w, d_packed, d_lens, n_reads, words_per_read, limit, strand, io, mode, d_hits, hits_cap, multi_base,
                           nullptr, nullptr, &P)) return rc;
  if (int rc = pass_multi(ix, w, P, n_reads)) return rc;
  if (w.ev[0]) SBWT_HIP(hipEventRecord(w.ev[2], w.stream));
  SBWT_HIP(hipMemcpyAsync(w.h_ctr, w.d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, w.stream));
  SBWT_HIP(hipStreamSynchronize(w.stream));
  SBWT_HIP(hipGetLastError());
  return pass_collect(w, io, hits_cap, multi_base, multi_used, wide_used, accumulate, ms_out, stats_out, false);
}

template <typename T>
static int ensure(T** p, uint64_t* have, uint64_t need);

// One device pass with a fast pass in front where the index allows one.  Compact records, hits only, on an index whose
// hash slots carry text, records of 64 bases: k_settle first (one access per read), then the general kernels over the
// reads it left pending, gathered into a dense batch; the other compact calls and the 32-byte records (io.compact
// null): k_direct (two accesses); otherwise the general kernels alone.
static int map_pass_compact(sbwt_ctx* c, const uint32_t* d_packed, const uint32_t* d_lens, uint64_t n_reads, uint32_t words_per_read,
                            int limit, int strand, const ResultIO& io, uint32_t mode, sbwt_hit* d_hits, uint64_t hits_cap,
                            uint64_t multi_base, uint64_t* multi_used, uint64_t* wide_used, bool accumulate) {
  const DeviceIndex& d = c->ix->d;
  // the fast passes need the text-carrying hash table; k_settle (one access) where the caller wants hits only and the
  // records hold 64 bases, k_direct (two accesses) for the other compact calls with records of 64 or 128 bases
  const bool fast = d.hash_text && d.hash_order && c->d_pend_list && !c->no_settle;
  const bool plain = io.compact == nullptr;
  const bool settle = fast && !plain && (mode & SBWT_MAP_HITS_ONLY) && words_per_read == 4;
  const bool direct = fast && !settle && words_per_read <= 12 && d.pos_lines && d.hash_order <= 31 && !c->no_direct;
  // an index without the text-carrying table: the lockstep rank pass (rank_pass.h) for records of 64 or 128 bases
  const bool rank = !d.hash_text && (words_per_read == 4 || words_per_read == 8) && c->d_pend_list && d.pos_lines && !c->no_rank;
  if (!settle && !direct && !rank) {
    if (!accumulate) {
      c->last_ms_ex[0] = c->last_ms_ex[1] = 0.f;
      c->last_stats[6] = c->last_stats[7] = c->last_stats[10] = c->last_stats[13] = 0;
    }
    return map_pass(c->ix, *c, d_packed, d_lens, n_reads, words_per_read, limit, strand, io, mode, d_hits, hits_cap, multi_base,
                    multi_used, wide_used, accumulate, c->last_ms, c->last_stats);
  }
  SettleArgs sa;
  sa.packed = d_packed;
  sa.lens = d_lens;
  sa.n_reads = n_reads;
  sa.strand = strand;
  sa.limit = (uint64_t)(int64_t)limit;
  sa.out = io.compact;
  sa.results = io.results;
  sa.hits = d_hits;
  sa.pending_list = c->d_pend_list;
  sa.pending_count = &c->d_ctr[1].pending_count;
  sa.index_accesses = &c->d_ctr[1].index_accesses;
  sa.hash_stats = c->d_ctr[1].hash_stats;
  sa.pending_other = &c->d_ctr[1].pending_other;
  sa.absent_count = &c->d_ctr[1].absent_count;
  sa.lds_tables = c->sort_lds_bytes ? 1u : 0u;
  sa.ring_chr_max = c->ring_chr_max;
  sa.hits_only = (mode & SBWT_MAP_HITS_ONLY) ? 1u : 0u;
  sa.wg_stats = c->d_wg_stats;
  SBWT_HIP(hipMemsetAsync(c->d_ctr + 1, 0, sizeof(Counters), c->stream));
  SBWT_HIP(hipEventRecord(c->ev_s[0], c->stream));
  const uint32_t grid = (uint32_t)std::min<uint64_t>(std::min<uint64_t>((n_reads + 255) / 256, kFastGridMax), settle ? c->grid_settle : rank ? c->grid_rank : c->grid_direct);
  if (rank) {
    const bool lds = c->rank_lds_bytes != 0, two = words_per_read == 8;
    const size_t lb = c->rank_lds_bytes;
    if (plain) {
      if (lds && two) k_rank<2, true, true><<<grid, 256, lb, c->stream>>>(d, sa);
      else if (lds) k_rank<1, true, true><<<grid, 256, lb, c->stream>>>(d, sa);
      else if (two) k_rank<2, false, true><<<grid, 256, 0, c->stream>>>(d, sa);
      else k_rank<1, false, true><<<grid, 256, 0, c->stream>>>(d, sa);
    } else {
      if (lds && two) k_rank<2, true><<<grid, 256, lb, c->stream>>>(d, sa);
      else if (lds) k_rank<1, true><<<grid, 256, lb, c->stream>>>(d, sa);
      else if (two) k_rank<2, false><<<grid, 256, 0, c->stream>>>(d, sa);
      else k_rank<1, false><<<grid, 256, 0, c->stream>>>(d, sa);
    }
  } else if (settle && c->sort_lds_bytes) k_settle<true><<<grid, 256, c->sort_lds_bytes, c->stream>>>(d, sa);
  else if (settle) k_settle<false><<<grid, 256, 0, c->stream>>>(d, sa);
  else if (plain && c->sort_lds_bytes) {
    if (words_per_read == 4) k_direct<1, true, true><<<grid, 256, c->sort_lds_bytes, c->stream>>>(d, sa);
    else if (words_per_read == 8) k_direct<2, true, true><<<grid, 256, c->sort_lds_bytes, c->stream>>>(d, sa);
    else k_direct<3, true, true><<<grid, 256, c->sort_lds_bytes, c->stream>>>(d, sa);
  } else if (plain) {
    if (words_per_read == 4) k_direct<1, false, true><<<grid, 256, 0, c->stream>>>(d, sa);
    else if (words_per_read == 8) k_direct<2, false, true><<<grid, 256, 0, c->stream>>>(d, sa);
    else k_direct<3, false, true><<<grid, 256, 0, c->stream>>>(d, sa);
  } else if (c->sort_lds_bytes) {
    if (words_per_read == 4) k_direct<1, true><<<grid, 256, c->sort_lds_bytes, c->stream>>>(d, sa);
    else if (words_per_read == 8) k_direct<2, true><<<grid, 256, c->sort_lds_bytes, c->stream>>>(d, sa);
    else k_direct<3, true><<<grid, 256, c->sort_lds_bytes, c->stream>>>(d, sa);
  } else {
    if (words_per_read == 4) k_direct<1, false><<<grid, 256, 0, c->stream>>>(d, sa);
    else if (words_per_read == 8) k_direct<2, false><<<grid, 256, 0, c->stream>>>(d, sa);
    else k_direct<3, false><<<grid, 256, 0, c->stream>>>(d, sa);
  }
  SBWT_LAUNCHED();
  k_fast_stats<<<1, 256, 0, c->stream>>>(c->d_wg_stats, grid, sa);  // the tallies of the launch's workgroups into the counters
  SBWT_LAUNCHED();
  SBWT_HIP(hipEventRecord(c->ev_s[1], c->stream));
  Counters* const hf = c->h_ctr + 1;  // the fast pass counts in the second Counters of the ctx: the general kernels zero the first
  int rc = SBWT_OK;
  *multi_used = 0;
  if (wide_used) *wide_used = 0;
  float ms3[3] = {0, 0, 0};
  uint64_t st[kNumStats] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t n_pending = 0;
  bool pending_done = false;
  auto scatter = [&](uint64_t n, const unsigned long long* n_dev) -> hipError_t {
    if (plain)
      k_scatter_pending_plain<<<grid_items(n, 256), 256, 0, c->stream>>>(c->d_pend_res, c->d_pend_hits, c->d_pend_list, n, io.results, d_hits, n_dev);
    else
      k_scatter_pending<<<grid_items(n, 256), 256, 0, c->stream>>>(c->d_pend_out, c->d_pend_list, n, io.compact, n_dev);
    return hipGetLastError();
  };
  // The pending pass WITHOUT waiting for the fast pass: gather, search kernel and scatter take the number of pending reads
  // from device memory (the buffers of an earlier call bound it), the counters of both passes come back in one copy and the
  // stream is synchronised once.  Only when the batch turns out to hold multi-hit reads do the locate kernels run, with the
  // records scattered again behind them; a batch that outgrew the buffers is done over below.
  uint64_t room = std::min<uint64_t>(std::min(c->pend_lens_cap, c->pend_words / 12), plain ? std::min(c->pend_res_cap, c->pend_hits_cap) : c->pend_out_cap);
  room = std::min(room, n_reads);
  if (room && !c->no_spec) {
    const unsigned long long* n_dev = &c->d_ctr[1].pending_count;
    k_gather_pending<<<grid_items(room, 256), 256, 0, c->stream>>>(d_packed, d_lens, c->d_pend_list, room, words_per_read / 4,
                                                                    c->d_pend_packed, c->d_pend_lens, n_dev);
    SBWT_LAUNCHED();
    const ResultIO io2 = plain ? ResultIO{c->d_pend_res, nullptr, nullptr, 0, c->d_pend_hits} : ResultIO{nullptr, c->d_pend_out, io.wide, io.wide_cap};
    PassState P;
    if (int e = pass_search(c->ix, *c, c->d_pend_packed, c->d_pend_lens, room, words_per_read, limit, strand, io2, mode, d_hits, hits_cap,
                            multi_base, n_dev, &c->d_ctr[1].pending_other, &P)) return e;
    SBWT_HIP(hipEventRecord(c->ev[2], c->stream));
    SBWT_HIP(scatter(room, n_dev));
    SBWT_HIP(hipMemcpyAsync(c->h_ctr, c->d_ctr, 2 * sizeof(Counters), hipMemcpyDeviceToHost, c->stream));
    SBWT_HIP(hipStreamSynchronize(c->stream));
    SBWT_HIP(hipGetLastError());
    n_pending = hf->pending_count;
    if (n_pending <= room) {
      pending_done = true;
      bool second = false;
      if (n_pending && c->h_ctr->multi_count && !c->h_ctr->overflow) {
        second = true;
        SBWT_HIP(hipEventRecord(c->ev[3], c->stream));
        if (int e = pass_multi(c->ix, *c, P, n_pending)) return e;
        SBWT_HIP(hipEventRecord(c->ev[4], c->stream));
        SBWT_HIP(scatter(n_pending, nullptr));
        SBWT_HIP(hipMemcpyAsync(c->h_ctr, c->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, c->stream));
        SBWT_HIP(hipStreamSynchronize(c->stream));
        SBWT_HIP(hipGetLastError());
      }
      if (n_pending) {
        rc = pass_collect(*c, io2, hits_cap, multi_base, multi_used, wide_used, false, ms3, st, second);
        if (rc != SBWT_OK && rc != SBWT_ERR_CAPACITY) return rc;
      }
    }
  } else {
    SBWT_HIP(hipMemcpyAsync(hf, c->d_ctr + 1, sizeof(Counters), hipMemcpyDeviceToHost, c->stream));
    SBWT_HIP(hipStreamSynchronize(c->stream));
    n_pending = hf->pending_count;
  }
  const bool all_missed = hf->pending_other == 0;  // then the general kernels skip the hash table for this batch
  const uint64_t s_acc = hf->index_accesses, s_look = hf->hash_stats[0], s_found = hf->hash_stats[1];
  const uint64_t s_absent = hf->absent_count;
  float ms_settle = 0.f;
  hipEventElapsedTime(&ms_settle, c->ev_s[0], c->ev_s[1]);
  float ms_pending = 0.f;
  if (n_pending && !pending_done) {
    // the first batch of a ctx, or one with more pending reads than the buffers hold: sized now, then the same pass
    if (c->pend_out_cap < n_pending) {  // grown with a quarter to spare, so that batches of similar make-up reuse it
      const uint64_t want = std::min<uint64_t>(c->max_reads, n_pending + (n_pending >> 2));
      if (int e = ensure(&c->d_pend_packed, &c->pend_words, want * 12)) return e;  // (records of up to twelve words)
      if (int e = ensure(&c->d_pend_lens, &c->pend_lens_cap, want)) return e;
      if (int e = ensure(&c->d_pend_out, &c->pend_out_cap, want)) return e;
    }
    if (plain && c->pend_res_cap < n_pending) {
      const uint64_t want = std::min<uint64_t>(c->max_reads, n_pending + (n_pending >> 2));
      if (int e = ensure(&c->d_pend_res, &c->pend_res_cap, want)) return e;
      if (int e = ensure(&c->d_pend_hits, &c->pend_hits_cap, want)) return e;
    }
    k_gather_pending<<<grid_items(n_pending, 256), 256, 0, c->stream>>>(d_packed, d_lens, c->d_pend_list, n_pending, words_per_read / 4,
                                                                         c->d_pend_packed, c->d_pend_lens, nullptr);
    SBWT_LAUNCHED();
    const ResultIO io2 = plain ? ResultIO{c->d_pend_res, nullptr, nullptr, 0, c->d_pend_hits} : ResultIO{nullptr, c->d_pend_out, io.wide, io.wide_cap};
    rc = map_pass(c->ix, *c, c->d_pend_packed, c->d_pend_lens, n_pending, words_per_read, limit, strand, io2,
                  mode | (all_missed ? kModeSkipHash : 0u), d_hits, hits_cap, multi_base,
                  multi_used, wide_used, false, ms3, st);
    if (rc != SBWT_OK && rc != SBWT_ERR_CAPACITY) return rc;
    SBWT_HIP(scatter(n_pending, nullptr));
    SBWT_HIP(hipStreamSynchronize(c->stream));
  }
  if (n_pending) {
    ms_pending = ms3[0];
    const float add[3] = {ms_settle + ms3[0], ms_settle + ms3[1], ms3[2]};
    for (int k = 0; k < 3; ++k) c->last_ms[k] = accumulate ? c->last_ms[k] + add[k] : add[k];
    const uint64_t sadd[kNumStats] = {s_acc + st[0], st[1], 0, st[3], s_look + st[4], s_found + st[5], s_acc, n_pending, st[8], st[9], s_absent, st[11], st[12], pending_done ? 1u : 0u};
    for (int k = 0; k < kNumStats; ++k)
      if (k != 2) c->last_stats[k] = (accumulate ? c->last_stats[k] : 0) + sadd[k];
  } else {
    const float add[3] = {ms_settle, ms_settle, 0.f};
    for (int k = 0; k < 3; ++k) c->last_ms[k] = accumulate ? c->last_ms[k] + add[k] : add[k];
    const uint64_t sadd[kNumStats] = {s_acc, 0, 0, 0, s_look, s_found, s_acc, 0, 0, 0, s_absent, 0, 0, pending_done ? 1u : 0u};
    for (int k = 0; k < kNumStats; ++k)
      if (k != 2) c->last_stats[k] = (accumulate ? c->last_stats[k] : 0) + sadd[k];
  }
  c->last_ms_ex[0] = (accumulate ? c->last_ms_ex[0] : 0.f) + ms_settle;
  c->last_ms_ex[1] = (accumulate ? c->last_ms_ex[1] : 0.f) + ms_pending;
  return rc;
}

extern "C" int sbwt_map_batch_device_compact(sbwt_ctx* c, const uint32_t* d_packed, const uint32_t* d_lens, uint64_t n_reads,
                                             uint32_t words_per_read, int limit, int strand, uint32_t mode,
                                             sbwt_compact_result* d_out, sbwt_hit* d_hits, uint64_t hits_cap,
                                             sbwt_wide_result* d_wide, uint64_t wide_cap, uint64_t counts_host[2]) {
  if (!c || (n_reads && (!d_packed || !d_lens || !d_out)) || (hits_cap && !d_hits) || (wide_cap && !d_wide)) {
    set_error("sbwt_map_batch_device_compact: null pointer");
    return SBWT_ERR_ARG;
  }
  if (n_reads > c->max_reads || words_per_read == 0 || (words_per_read & 3u) || (mode & ~SBWT_MAP_HITS_ONLY)) {
    set_error("sbwt_map_batch_device_compact: n_reads exceeds the ctx capacity, words_per_read is not a multiple of 4, or unknown mode bits");
    return SBWT_ERR_ARG;
  }
  if (counts_host) counts_host[0] = counts_host[1] = 0;
  if (n_reads == 0) return SBWT_OK;
  SBWT_HIP(enter_device(c->ix->device));
  const ResultIO io{nullptr, d_out, d_wide, wide_cap};
  uint64_t multi_used = 0, wide_used = 0;
  const int rc = map_pass_compact(c, d_packed, d_lens, n_reads, words_per_read, limit, strand, io, mode, d_hits, hits_cap, 0,
                                  &multi_used, &wide_used, false);
  c->last_stats[2] = multi_used;
  if (counts_host) {  // on SBWT_ERR_CAPACITY: what a retry needs
    counts_host[0] = multi_used;
    counts_host[1] = wide_used;
  }
  return rc;
}

extern "C" int sbwt_host_alloc(uint64_t bytes, void** out) {
  if (!out) return SBWT_ERR_ARG;
  *out = nullptr;
  SBWT_HIP(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
  return SBWT_OK;
}

extern "C" void sbwt_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

template <typename T>
static int ensure(T** p, uint64_t* have, uint64_t need) {
  if (*have >= need) return 0;
  if (*p) hipFree(*p);
  *p = nullptr;
  *have = 0;
  hipError_t e = hipMalloc(p, need * sizeof(T));
  if (e != hipSuccess) return hip_fail(e, "hipMalloc(staging)", __FILE__, __LINE__);
  *have = need;
  return 0;
}

extern "C" int sbwt_map_batch(sbwt_ctx* c, const uint32_t* packed, const uint32_t* lens, uint64_t n_reads,
                              uint32_t words_per_read, int limit, int strand, sbwt_read_result* results,
                              sbwt_hit* hits, uint64_t hits_cap, uint64_t* n_hits) {
  if (!c || (n_reads && (!packed || !lens || !results)) || (hits_cap && !hits)) {
    set_error("sbwt_map_batch: null pointer");
    return SBWT_ERR_ARG;
  }
  // the same checks as sbwt_map_batch_device, before anything is staged or split: the queues of the ctx hold
  // max_reads entries whichever way the batch is cut
  if (n_reads > c->max_reads || words_per_read == 0 || (words_per_read & 3u)) {
    set_error("sbwt_map_batch: n_reads exceeds the ctx capacity or words_per_read is not a multiple of 4");
    return SBWT_ERR_ARG;
  }
  if (n_hits) *n_hits = 0;
  if (n_reads == 0) return SBWT_OK;
  SBWT_HIP(enter_device(c->ix->device));
  const uint64_t words = n_reads * words_per_read;
  uint64_t lens_have = c->staged_reads, res_have = c->staged_reads;
  if (int rc = ensure(&c->d_packed, &c->staged_words, words)) return rc;
  if (int rc = ensure(&c->d_lens, &lens_have, n_reads)) return rc;
  if (int rc = ensure(&c->d_results, &res_have, n_reads)) return rc;
  c->staged_reads = lens_have < res_have ? lens_have : res_have;
  if (int rc = ensure(&c->d_hits, &c->staged_hits, hits_cap ? hits_cap : 1)) return rc;
  // Large batches go in sub-batches so that the copy-in of the next one and the copy-out of the previous one run
  // while one is being mapped (the link, not the kernels, bounds this entry point).  SBWT_SUBBATCH = reads per
  // sub-batch (0 = never split).
  uint64_t sub = c->subbatch;
  sub = (sub + 63) & ~63ull;  // whole result lines per sub-batch
  if (sub && n_reads >= 2 * sub && hits_cap >= n_reads && c->s_in && c->s_out) {
    uint64_t used_multi = 0;
    bool ok = true;
    int rc = SBWT_OK;
    hipError_t he = hipSuccess;  // first HIP failure inside the loop: the loop stops, the streams drain, then it is reported
    auto copy_in = [&](uint64_t a0, uint64_t n0, int slot) -> hipError_t {
      hipError_t e = hipMemcpyAsync(c->d_packed + a0 * words_per_read, packed + a0 * words_per_read, n0 * words_per_read * 4,
                                    hipMemcpyHostToDevice, c->s_in);
      if (e == hipSuccess) e = hipMemcpyAsync(c->d_lens + a0, lens + a0, n0 * 4, hipMemcpyHostToDevice, c->s_in);
      if (e == hipSuccess) e = hipEventRecord(c->ev_in[slot], c->s_in);
      return e;
    };
    he = copy_in(0, std::min(sub, n_reads), 0);
    int slot = 0;
    for (uint64_t a0 = 0; a0 < n_reads && ok && he == hipSuccess; a0 += sub, slot ^= 1) {
      const uint64_t n0 = std::min(sub, n_reads - a0);
      if (a0 + sub < n_reads) he = copy_in(a0 + sub, std::min(sub, n_reads - a0 - sub), slot ^ 1);
      if (he == hipSuccess) he = hipStreamWaitEvent(c->stream, c->ev_in[slot], 0);
      if (he != hipSuccess) break;
      uint64_t mu = 0;
      // this sub-batch sees hits[] from its first read on: unique hits at its own read indices, lists behind the
      // slots of ALL reads and the lists of the sub-batches before it
      rc = map_pass(c, c->d_packed + a0 * words_per_read, c->d_lens + a0, n0, words_per_read, limit, strand,
                    c->d_results + a0, c->d_hits + a0, hits_cap - a0, (n_reads - a0) + used_multi, &mu, a0 != 0);
      if (rc != SBWT_OK) { ok = false; break; }
      if (a0) {
        k_bias_first_hit<<<grid_items(n0, 256), 256, 0, c->stream>>>(c->d_results + a0, n0, a0);
        he = hipGetLastError();
      }
      if (he == hipSuccess) he = hipEventRecord(c->ev_k, c->stream);
      if (he == hipSuccess) he = hipStreamWaitEvent(c->s_out, c->ev_k, 0);
      if (he == hipSuccess) he = hipMemcpyAsync(results + a0, c->d_results + a0, n0 * sizeof(sbwt_read_result), hipMemcpyDeviceToHost, c->s_out);
      if (he == hipSuccess) he = hipMemcpyAsync(hits + a0, c->d_hits + a0, n0 * sizeof(sbwt_hit), hipMemcpyDeviceToHost, c->s_out);
      if (he == hipSuccess && mu)
        he = hipMemcpyAsync(hits + n_reads + used_multi, c->d_hits + n_reads + used_multi, mu * sizeof(sbwt_hit),
                            hipMemcpyDeviceToHost, c->s_out);
      used_multi += mu;
    }
    // whatever happened, no copy may still be aimed at the caller's buffers when this call returns
    const hipError_t e1 = hipStreamSynchronize(c->s_in), e2 = hipStreamSynchronize(c->stream), e3 = hipStreamSynchronize(c->s_out);
    if (he == hipSuccess) he = e1 != hipSuccess ? e1 : (e2 != hipSuccess ? e2 : e3);
    if (he != hipSuccess) return hip_fail(he, "sbwt_map_batch (sub-batches)", __FILE__, __LINE__);
    if (ok) {
      c->last_stats[2] = n_reads + used_multi;
      if (n_hits) *n_hits = n_reads + used_multi;
      return SBWT_OK;
    }
    if (rc != SBWT_ERR_CAPACITY) return rc;
    // the lists did not fit: the whole batch in one pass below reports what a retry needs
  }
  SBWT_HIP(hipMemcpyAsync(c->d_packed, packed, words * 4, hipMemcpyHostToDevice, c->stream));
  SBWT_HIP(hipMemcpyAsync(c->d_lens, lens, n_reads * 4, hipMemcpyHostToDevice, c->stream));
  uint64_t nh = 0;
  int rc = sbwt_map_batch_device(c, c->d_packed, c->d_lens, n_reads, words_per_read, limit, strand, c->d_results,
                                 c->d_hits, hits_cap, &nh);
  if (rc != SBWT_OK) {
    if (n_hits) *n_hits = nh;
    return rc;
  }
  SBWT_HIP(hipMemcpyAsync(results, c->d_results, n_reads * sizeof(sbwt_read_result), hipMemcpyDeviceToHost, c->stream));
  if (nh) SBWT_HIP(hipMemcpyAsync(hits, c->d_hits, nh * sizeof(sbwt_hit), hipMemcpyDeviceToHost, c->stream));
  SBWT_HIP(hipStreamSynchronize(c->stream));
  if (n_hits) *n_hits = nh;
  return SBWT_OK;
}

// ---- sbwt_map_batch_device_static: the device-pointer entry point for callers that cannot allocate ------------
// The scratch of the kernels (counters, the two queues, per-wave totals, the depth-first stacks) lives in a
// file-scope device array instead of buffers a sbwt_ctx allocates.  Every GPU the library runs on holds its own
// copy of the array until the process ends; all calls on one GPU share it, so calls that overlap on one GPU race.
namespace sbwt {
constexpr uint64_t kStaticMaxReads = 1ull << 20;   // reads per call
constexpr uint32_t kStaticMaxWaves = 8192;         // 256 CUs x 8 resident workgroups x 4 waves
constexpr uint64_t kStaticStackBytes = 256ull * 1024 * sizeof(Frame);  // 256 lanes at interval 1024, 4096 at 64
constexpr uint64_t kStaticOffCtr = 0;
constexpr uint64_t kStaticOffWave = 256;
constexpr uint64_t kStaticOffQueue = kStaticOffWave + kStaticMaxWaves * sizeof(WaveInfo);
constexpr uint64_t kStaticOffDfs = kStaticOffQueue + kStaticMaxReads * sizeof(uint32_t);
constexpr uint64_t kStaticOffStacks = kStaticOffDfs + kStaticMaxReads * sizeof(uint32_t);
constexpr uint64_t kStaticBytes = kStaticOffStacks + kStaticStackBytes;  // 14.1 MiB
static_assert(sizeof(Counters) <= 256 && kStaticOffStacks % 256 == 0, "scratch layout");
}  // namespace sbwt

static __device__ __align__(256) unsigned char scratch[sbwt::kStaticBytes];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_static_mutex;
unsigned char* g_scratch_ptr[kMaxDevices];  // device address of `scratch` on each device (the name itself is a host-side stand-in)
Counters* g_static_hctr[kMaxDevices];       // pinned read-back buffer, one per device
}  // namespace

// Set-up, once per device, from the calls that create an index on it (the library's init for that device): the
// device address of the scratch array and a pinned buffer for the counters.  The only allocation of this path
// happens here, never in the entry point.
int sbwt::static_scratch_setup(int device) {
  if (device < 0 || device >= kMaxDevices) return SBWT_OK;  // such a device simply has no static path
  std::lock_guard<std::mutex> lk(g_static_mutex);
  if (g_scratch_ptr[device]) return SBWT_OK;
  SBWT_HIP(enter_device(device));
  unsigned char* ptr = nullptr;
  SBWT_HIP(hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(scratch)));
  if (!ptr) {
    set_error("hipGetSymbolAddress(scratch) returned a null address");
    return SBWT_ERR_HIP;
  }
  Counters* h = nullptr;
  SBWT_HIP(hipHostMalloc(reinterpret_cast<void**>(&h), sizeof(Counters)));
  g_static_hctr[device] = h;
  g_scratch_ptr[device] = ptr;
  return SBWT_OK;
}

int sbwt::finish_index(sbwt_index* ix) {
  if (int rc = static_scratch_setup(ix->device)) return rc;
  SBWT_HIP(enter_device(ix->device));
  MapWork w;
  SBWT_HIP(plan_launches(ix, &w));
  ix->plan.grid_sm_lds = w.grid_sm_lds;
  ix->plan.grid_sm_glb = w.grid_sm_glb;
  ix->plan.grid_rounds_lds = w.grid_rounds_lds;
  ix->plan.grid_rounds_glb = w.grid_rounds_glb;
  ix->plan.grid_sort = w.grid_sort;
  ix->plan.grid_bfs = w.grid_bfs;
  ix->plan.sm_lds_bytes = w.sm_lds_bytes;
  ix->plan.sm_ring_bytes = w.sm_ring_bytes;
  ix->plan.sort_lds_bytes = w.sort_lds_bytes;
  read_pass_knobs(&ix->plan.no_bfs, &ix->plan.no_chain, &ix->plan.no_rounds, &ix->plan.no_segments, &ix->plan.ring_width_max, &ix->plan.ring_chr_max);
  return SBWT_OK;
}

extern "C" int sbwt_map_batch_device_static(const sbwt_index* ix, void* stream, const uint32_t* d_packed,
                                            const uint32_t* d_lens, uint64_t n_reads, uint32_t words_per_read, int limit,
                                            int strand, sbwt_read_result* d_results, sbwt_hit* d_hits, uint64_t hits_cap,
                                            uint64_t* n_hits_host) {
  if (!ix || (n_reads && (!d_packed || !d_lens || !d_results)) || (hits_cap && !d_hits)) {
    set_error("sbwt_map_batch_device_static: null pointer");
    return SBWT_ERR_ARG;
  }
  if (n_reads > kStaticMaxReads) {
    char msg[160];
    snprintf(msg, sizeof msg, "sbwt_map_batch_device_static: %llu reads in one call, the static scratch holds %llu at most",
             (unsigned long long)n_reads, (unsigned long long)kStaticMaxReads);
    set_error(msg);
    return SBWT_ERR_ARG;
  }
  if (words_per_read == 0 || (words_per_read & 3u)) {
    set_error("sbwt_map_batch_device_static: words_per_read is not a multiple of 4");
    return SBWT_ERR_ARG;
  }
  if (n_hits_host) *n_hits_host = 0;
  if (n_reads == 0) return SBWT_OK;
  if (hits_cap < n_reads) {
    set_error("hits buffer too small: at least one slot per read is needed");
    if (n_hits_host) *n_hits_host = n_reads;
    return SBWT_ERR_CAPACITY;
  }
  const int dev = ix->device;
  unsigned char* base = nullptr;
  Counters* h_ctr = nullptr;
  if (dev >= 0 && dev < kMaxDevices) {
    std::lock_guard<std::mutex> lk(g_static_mutex);
    base = g_scratch_ptr[dev];
    h_ctr = g_static_hctr[dev];
  }
  if (!base || !h_ctr) {
    set_error("sbwt_map_batch_device_static: the static scratch of this device was not set up");
    return SBWT_ERR_HIP;
  }
  SBWT_HIP(enter_device(dev));
  MapWork w;
  w.stream = static_cast<hipStream_t>(stream);
  w.d_ctr = reinterpret_cast<Counters*>(base + kStaticOffCtr);
  w.h_ctr = h_ctr;
  w.d_wave_info = reinterpret_cast<WaveInfo*>(base + kStaticOffWave);
  w.d_queue = reinterpret_cast<uint32_t*>(base + kStaticOffQueue);
  w.d_dfs_queue = reinterpret_cast<uint32_t*>(base + kStaticOffDfs);
  w.d_stacks = reinterpret_cast<Frame*>(base + kStaticOffStacks);
  // launch shapes: computed when the index was made (plan_launches allocates nothing, but asks the runtime)
  w.grid_sm_lds = ix->plan.grid_sm_lds;
  w.grid_sm_glb = ix->plan.grid_sm_glb;
  w.grid_sort = ix->plan.grid_sort;
  w.grid_bfs = ix->plan.grid_bfs;
  w.sm_lds_bytes = ix->plan.sm_lds_bytes;
  w.sm_ring_bytes = ix->plan.sm_ring_bytes;
  w.sort_lds_bytes = ix->plan.sort_lds_bytes;
  w.no_bfs = ix->plan.no_bfs;
  w.no_chain = ix->plan.no_chain;
  w.no_rounds = ix->plan.no_rounds;
  w.grid_rounds_lds = ix->plan.grid_rounds_lds;
  w.grid_rounds_glb = ix->plan.grid_rounds_glb;
  w.no_segments = ix->plan.no_segments;
  w.ring_width_max = ix->plan.ring_width_max;
  w.ring_chr_max = ix->plan.ring_chr_max;
  if (!w.grid_sort || !w.sm_ring_bytes) {
    set_error("sbwt_map_batch_device_static: the index carries no launch plan");
    return SBWT_ERR_HIP;
  }
  // the scratch bounds the shapes: waves of the search kernel, lanes of the depth-first kernel
  w.grid_sm_lds = std::min<uint32_t>(w.grid_sm_lds, kStaticMaxWaves / 4);
  w.grid_sm_glb = std::min<uint32_t>(w.grid_sm_glb, kStaticMaxWaves / 4);
  const uint64_t lanes = kStaticStackBytes / (ix->d.interval * sizeof(Frame)) / 256 * 256;  // >= 256: interval <= 1024
  w.multi_lanes = (uint32_t)std::min<uint64_t>(lanes, (n_reads + 255) / 256 * 256);
  uint64_t multi_used = 0;
  const ResultIO io{d_results, nullptr, nullptr, 0};
  const int rc = map_pass(ix, w, d_packed, d_lens, n_reads, words_per_read, limit, strand, io, 0u, d_hits, hits_cap, n_reads,
                          &multi_used, nullptr, false, nullptr, nullptr);
  if (n_hits_host) *n_hits_host = n_reads + multi_used;  // on SBWT_ERR_CAPACITY: what a retry needs
  return rc;
}

// ---- compact records through host buffers ---------------------------------------------------------------------
// payload of every record of a sub-batch that points into wide[] += bias (the sub-batch numbered its slots from 0)
__global__ __launch_bounds__(256) void k_bias_wide(sbwt_compact_result* __restrict__ out, uint64_t n, uint64_t bias) {
  const uint64_t r = global_thread();
  if (r >= n) return;
  const uint64_t w1 = out[r].w1;
  if ((uint32_t)(w1 >> 56) & SBWT_CR_WIDE) out[r].w1 = (w1 & ~kPosMask) | (((w1 & kPosMask) + bias) & kPosMask);
}

extern "C" int sbwt_map_batch_compact(sbwt_ctx* c, const uint32_t* packed, const uint32_t* lens, uint64_t n_reads,
                                      uint32_t words_per_read, int limit, int strand, uint32_t mode, sbwt_compact_result* out,
                                      sbwt_hit* hits, uint64_t hits_cap, sbwt_wide_result* wide, uint64_t wide_cap,
                                      uint64_t counts[2]) {
  if (!c || (n_reads && (!packed || !lens || !out)) || (hits_cap && !hits) || (wide_cap && !wide)) {
    set_error("sbwt_map_batch_compact: null pointer");
    return SBWT_ERR_ARG;
  }
  if (n_reads > c->max_reads || words_per_read == 0 || (words_per_read & 3u) || (mode & ~SBWT_MAP_HITS_ONLY)) {
    set_error("sbwt_map_batch_compact: n_reads exceeds the ctx capacity, words_per_read is not a multiple of 4, or unknown mode bits");
    return SBWT_ERR_ARG;
  }
  if (counts) counts[0] = counts[1] = 0;
  if (n_reads == 0) return SBWT_OK;
  SBWT_HIP(enter_device(c->ix->device));
  // staging: the 32-byte result staging of the plain entry point holds two compact records per slot
  const uint64_t words = n_reads * words_per_read;
  uint64_t lens_have = c->staged_reads, res_have = c->staged_reads;
  if (int rc = ensure(&c->d_packed, &c->staged_words, words)) return rc;
  if (int rc = ensure(&c->d_lens, &lens_have, n_reads)) return rc;
  if (int rc = ensure(&c->d_results, &res_have, n_reads)) return rc;
  c->staged_reads = lens_have < res_have ? lens_have : res_have;
  if (int rc = ensure(&c->d_hits, &c->staged_hits, hits_cap ? hits_cap : 1)) return rc;
  if (int rc = ensure(&c->d_wide, &c->staged_wide, wide_cap ? wide_cap : 1)) return rc;
  sbwt_compact_result* d_out = reinterpret_cast<sbwt_compact_result*>(c->d_results);
  uint64_t sub = c->subbatch;
  sub = (sub + 63) & ~63ull;
  if (!sub || n_reads < 2 * sub || !c->s_in || !c->s_out) sub = n_reads;  // one piece
  uint64_t used_multi = 0, used_wide = 0;
  int rc = SBWT_OK;
  hipError_t he = hipSuccess;
  auto copy_in = [&](uint64_t a0, uint64_t n0, int slot) -> hipError_t {
    hipError_t e = hipMemcpyAsync(c->d_packed + a0 * words_per_read, packed + a0 * words_per_read, n0 * words_per_read * 4,
                                  hipMemcpyHostToDevice, c->s_in);
    if (e == hipSuccess) e = hipMemcpyAsync(c->d_lens + a0, lens + a0, n0 * 4, hipMemcpyHostToDevice, c->s_in);
    if (e == hipSuccess) e = hipEventRecord(c->ev_in[slot], c->s_in);
    return e;
  };
  he = copy_in(0, std::min(sub, n_reads), 0);
  int slot = 0;
  for (uint64_t a0 = 0; a0 < n_reads && he == hipSuccess; a0 += sub, slot ^= 1) {
    const uint64_t n0 = std::min(sub, n_reads - a0);
    if (a0 + sub < n_reads) he = copy_in(a0 + sub, std::min(sub, n_reads - a0 - sub), slot ^ 1);
    if (he == hipSuccess) he = hipStreamWaitEvent(c->stream, c->ev_in[slot], 0);
    if (he != hipSuccess) break;
    // the lists of this piece go behind those of the pieces before it; its wide slots likewise
    const ResultIO io{nullptr, d_out + a0, c->d_wide + std::min(used_wide, wide_cap), wide_cap - std::min(used_wide, wide_cap)};
    uint64_t mu = 0, wu = 0;
    const int prc = map_pass_compact(c, c->d_packed + a0 * words_per_read, c->d_lens + a0, n0, words_per_read, limit, strand, io,
                                     mode, c->d_hits, hits_cap, used_multi, &mu, &wu, a0 != 0);
    if (prc != SBWT_OK && prc != SBWT_ERR_CAPACITY) { rc = prc; break; }
    if (prc == SBWT_ERR_CAPACITY) rc = prc;  // keep counting: the caller learns what the whole batch needs
    if (rc == SBWT_OK) {
      if (wu && used_wide) {
        k_bias_wide<<<grid_items(n0, 256), 256, 0, c->stream>>>(d_out + a0, n0, used_wide);
        he = hipGetLastError();
      }
      if (he == hipSuccess) he = hipEventRecord(c->ev_k, c->stream);
      if (he == hipSuccess) he = hipStreamWaitEvent(c->s_out, c->ev_k, 0);
      if (he == hipSuccess) he = hipMemcpyAsync(out + a0, d_out + a0, n0 * sizeof(sbwt_compact_result), hipMemcpyDeviceToHost, c->s_out);
      if (he == hipSuccess && mu)
        he = hipMemcpyAsync(hits + used_multi, c->d_hits + used_multi, mu * sizeof(sbwt_hit), hipMemcpyDeviceToHost, c->s_out);
      if (he == hipSuccess && wu)
        he = hipMemcpyAsync(wide + used_wide, c->d_wide + used_wide, wu * sizeof(sbwt_wide_result), hipMemcpyDeviceToHost, c->s_out);
    }
    used_multi += mu;
    used_wide += wu;
  }
  const hipError_t e1 = hipStreamSynchronize(c->s_in), e2 = hipStreamSynchronize(c->stream), e3 = hipStreamSynchronize(c->s_out);
  if (he == hipSuccess) he = e1 != hipSuccess ? e1 : (e2 != hipSuccess ? e2 : e3);
  if (he != hipSuccess) return hip_fail(he, "sbwt_map_batch_compact", __FILE__, __LINE__);
  c->last_stats[2] = used_multi;
  if (counts) { counts[0] = used_multi; counts[1] = used_wide; }
  return rc;
}

extern "C" int sbwt_compact_expand(const sbwt_compact_result* compact, uint64_t n_reads, const sbwt_wide_result* wide,
                                   uint64_t n_wide, const sbwt_hit* list_hits, uint64_t n_list_hits, sbwt_read_result* results,
                                   sbwt_hit* hits) {
  if ((n_reads && (!compact || !results || !hits)) || (n_wide && !wide) || (n_list_hits && !list_hits)) {
    set_error("sbwt_compact_expand: null pointer");
    return SBWT_ERR_ARG;
  }
  for (uint64_t k = 0; k < n_list_hits; ++k) hits[n_reads + k] = list_hits[k];
  for (uint64_t r = 0; r < n_reads; ++r) {
    const uint64_t w0 = compact[r].w0, w1 = compact[r].w1;
    const uint32_t meta = (uint32_t)(w1 >> 56);
    const uint64_t payload = w1 & kPosMask;
    sbwt_read_result o;
    if (meta & SBWT_CR_WIDE) {
      if (payload >= n_wide) {
        set_error("sbwt_compact_expand: a record points past wide[]");
        return SBWT_ERR_ARG;
      }
      o = wide[payload].r;
      if (o.nh == 1 && !(meta & SBWT_CR_MULTI)) { hits[r] = wide[payload].hit; o.first_hit = r; }
      else if (o.nh > 0) o.first_hit += n_reads;
      else o.first_hit = 0;
    } else {
      o.sa_lo = w0 & kPosMask;
      o.sa_hi = o.sa_lo + (w0 >> kPosBits);
      o.flags = ((meta >> 3) & 3u) | ((meta & SBWT_CR_INVALID) ? 0x80000000u : 0u);
      o.nh = 0;
      o.first_hit = 0;
      if (meta & SBWT_CR_UNIQUE) {
        o.nh = 1;
        o.first_hit = r;
        hits[r].pos0 = payload;
        hits[r].chr_id = (uint32_t)(w1 >> kPosBits) & 0xFFFFu;
        hits[r].is_rc = (meta & SBWT_CR_RC) ? 1u : 0u;
      } else if (meta & SBWT_CR_MULTI) {
        o.nh = (int32_t)((w1 >> kPosBits) & 0xFFFFu);
        o.first_hit = o.nh > 0 ? n_reads + payload : 0;
      }
    }
    results[r] = o;
  }
  return SBWT_OK;
}

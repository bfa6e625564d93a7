// settle.h -- k_settle: the hits-only fast pass.
//
// With SBWT_MAP_HITS_ONLY the caller wants what the SAM output needs -- the hits, NH and flags of every read
// (searcher.hpp:386-458 returns nothing else) -- not the SA interval.  On an index whose hash slots carry the text
// behind the located prefix (device_layout.h, hash_text) a read of hash_order + 1 .. hash_order + 32 bases whose
// first hash_order bases are in the table is SETTLED by its bucket alone: that prefix is ONE located row, so under
// exec_exact_match (searcher.hpp:102-105) the remaining bases either equal the text in front of the row's position
// (the interval stays that one row's LF chain: one hit, `rem` positions further down) or the interval dies.  One
// random access per read, no dependent chain: a streaming kernel, one read per lane, instead of the state machine.
//
// Everything else -- prefix not in the table, run of LF-consistent positions too short, a read of another length, a
// result no 16-byte record holds -- is left PENDING: its index goes to a list, the host gathers those reads into a
// dense batch and the general kernels (k_search_sm, k_multi_sort, k_multi) map it exactly as they map any batch;
// k_scatter_pending puts their records in place.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sbwt_amd.h"
#include "rank_device.h"
#include "search_machine.h"

namespace sbwt {

struct SettleArgs {
  const uint32_t* packed;  // records of 4 words (64 bases)
  const uint32_t* lens;
  uint64_t n_reads;
  int strand;
  uint64_t limit;
  sbwt_compact_result* out;
  sbwt_read_result* results;            // k_direct<.., kPlain>: 32-byte records instead of `out`, the hit of a settled read at
  sbwt_hit* hits;                       // hits[its own index] (sbwt_map_batch_device)
  uint32_t* pending_list;               // read indices left to the general kernels
  unsigned long long* pending_count;
  unsigned long long* index_accesses;   // += 128-byte lines of the index read
  unsigned long long* hash_stats;       // [0] lookups, [1] found
  unsigned long long* pending_other;    // pending reads that were not looked up and missed (long, short run, wide result)
  unsigned long long* absent_count;     // += reads settled because their prefix occurs nowhere (DeviceIndex::hash_complete); may be null
  uint32_t lds_tables;                  // the launch carries (n_chr + 2 n_npos) * 8 bytes of LDS for the filter tables
  uint32_t ring_chr_max;                // chr_id a record holds (test knob, as in SmArgs)
  uint32_t hits_only;                   // k_direct: SBWT_MAP_HITS_ONLY (a read that differs from the text is settled: no hit)
  unsigned long long* wg_stats = nullptr;  // [gridDim.x][5]: the tallies of every workgroup, summed by k_fast_stats (null: atomics)
};

// The four lanes of a quad look their buckets up together.  Round k: the quad loads the bucket of its lane k, lane
// `sub` the slot `sub`, in ONE instruction (a lane that reads its 64-byte bucket through four loads of its own makes
// three of them wait in the vector cache behind the first one's miss) -- an LDS-DMA load: the 16 bytes go straight
// to LDS (wave base + 16 * loading lane, masked-off lanes write nothing), where lane k then finds its whole bucket
// and compares the four tags itself.  No exchange between lanes, which is what the kernel's VALU budget wants: a
// fifth of k_settle's instructions used to be the DPP reduction that brought every lane its slot.
// A prefix that found its bucket full when the table was built sits in another bucket of its group
// (device_index.hip, hash_insert: b ^ 1 in the same 128-byte line, then b ^ 2, ...): further passes look there for the
// reads whose buckets so far have the tag nowhere and no free slot.  Every lane of the wave calls this; `look` says
// who asks.  s_bkt: kBktLds uint4 of LDS owned by this wave (a round's 64 slots, then one slot of padding: the four
// lanes of a quad read their buckets 1040 bytes apart, in different banks).
// slot / text: the position word and the 32 bases behind the prefix; slot stays ~0 when the prefix is not there.
// slot2 / text2: the other slot of the bucket with the same tag (twins, device_layout.h), else ~0.
// absent: the prefix is in none of the buckets looked at and the last of them has a free slot -- it never came by when
// the table was built (slots are never emptied), which on a hash_complete index means it occurs nowhere.
// lines += the 128-byte lines of the table this lane's lookup touched (two buckets a line).
constexpr uint32_t kBktRound = 65, kBktLds = 4 * kBktRound;
__device__ __forceinline__ void settle_lookup(const DeviceIndex& ix, const uint4* __restrict__ tab, uint64_t bucket, uint32_t tag,
                                              uint32_t tmask, bool look, uint32_t lane, uint4* s_bkt, uint64_t& slot, uint64_t& text,
                                              uint64_t& slot2, uint64_t& text2, bool& absent, uint32_t& lines) {
  const uint32_t sub = lane & 3u;
  bool again = false;
#pragma unroll 1
  for (uint32_t pass = 0; pass < ix.hash_group; ++pass) {
    const bool go = pass == 0 ? look : again;
    if (!__any(go)) break;
    const uint64_t bk = bucket ^ (uint64_t)pass;
    if (go && !(pass & 1u)) ++lines;
    const uintptr_t ba = reinterpret_cast<uintptr_t>(tab + bk * 4u);
    const uint32_t ba_lo = (uint32_t)ba, ba_hi = (uint32_t)(ba >> 32);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t want = quad_bcast(go ? 1u : 0u, k);
      if (!__any(want)) continue;  // wave-uniform: the later passes serve few lanes, most of their rounds nobody
      const uintptr_t q = u64_of(quad_bcast(ba_lo, k), quad_bcast(ba_hi, k)) + 16u * sub;
      if (want)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)q,
                                         (__attribute__((address_space(3))) void*)(s_bkt + kBktRound * k), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (go) {
      const uint4* im = s_bkt + kBktRound * sub + (lane & ~3u);  // round `sub` served this lane; its quad's four loads lie together
      bool found = false, free_slot = false;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint4 e = im[q];
        const bool used = e.y != 0xFFFFFFFFu;
        free_slot = free_slot || !used;
        if (used && ((e.x ^ tag) & tmask) == 0) {  // one slot of the group carries the tag, or two twins
          slot2 = slot;
          text2 = text;
          slot = u64_of(e.x, e.y);
          text = u64_of(e.z, e.w);
          found = true;
        }
      }
      again = !found && !free_slot;  // full, and the tag nowhere: on to the next bucket of the group, if there is one
      absent = !found && free_slot;
    }
    // the next pass (and the caller) may load into s_bkt again: not before these reads have returned
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int d) {
  return ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), d) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)v, d);
}

constexpr uint32_t kSettleBuf = 512;  // pending indices a wave holds in LDS before it reserves list slots for them

// The end of a fast-pass kernel (k_settle, k_direct, k_rank; blocks of four waves, every thread gets here): what the
// waves still hold of the pending list, and the tallies, leave with ONE set of atomics per workgroup.  All waves of a
// launch finish within microseconds of each other, and atomics on one cache line run at about 1.2e8 per second: five per
// wave -- 30,000 for the 6,144 waves of a launch -- were a quarter of a millisecond at the end of EVERY launch, whatever
// its size (k_settle: 260 us for 10^6 reads, 570 us for 10^7).
// tallies: {hash lookups, found, pending reads not looked up and missed, reads settled by a miss, index lines}
__device__ __forceinline__ void settle_epilogue(const SettleArgs& a, const uint32_t* s_pend, uint32_t n_buf, const uint32_t (&tallies)[5],
                                                uint32_t lane) {
  __shared__ uint32_t s_left[4];
  __shared__ unsigned long long s_sum[4][5];
  __shared__ unsigned long long s_list_base;
  const uint32_t wave = threadIdx.x >> 6;
  uint32_t t[5];
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    t[q] = tallies[q];
    for (int d = 32; d > 0; d >>= 1) t[q] += __shfl_xor(t[q], d);
  }
  if (lane == 0) {
    s_left[wave] = n_buf;
#pragma unroll
    for (int q = 0; q < 5; ++q) s_sum[wave][q] = t[q];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t left = s_left[0] + s_left[1] + s_left[2] + s_left[3];
    s_list_base = left ? atomicAdd(a.pending_count, (unsigned long long)left) : 0ull;
    unsigned long long sum[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) sum[q] = s_sum[0][q] + s_sum[1][q] + s_sum[2][q] + s_sum[3][q];
    if (a.wg_stats) {
      // no atomics at all for the tallies: k_fast_stats adds the workgroups up behind this kernel
#pragma unroll
      for (int q = 0; q < 5; ++q) a.wg_stats[(size_t)blockIdx.x * 5 + q] = sum[q];
    } else {
    if (sum[0] && a.hash_stats) atomicAdd(a.hash_stats, sum[0]);
    if (sum[1] && a.hash_stats) atomicAdd(a.hash_stats + 1, sum[1]);
    if (sum[2]) atomicAdd(a.pending_other, sum[2]);
    if (sum[3] && a.absent_count) atomicAdd(a.absent_count, sum[3]);
    if (sum[4]) atomicAdd(a.index_accesses, sum[4]);
    }
  }
  __syncthreads();
  unsigned long long base = s_list_base;
  for (uint32_t v = 0; v < wave; ++v) base += s_left[v];
  for (uint32_t k = lane; k < n_buf; k += 64) a.pending_list[base + k] = s_pend[k];
}

// kLds: the filter tables are staged in LDS (the launch carries (n_chr + 2 n_npos) * 8 bytes of it): a compile-time
// choice, so that their reads are LDS instructions and not flat loads
template <bool kLds>
__global__ __launch_bounds__(256) void k_settle(const DeviceIndex ix, const SettleArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint64_t s_tab[];
  const uint64_t *t_chr = ix.chr_start, *t_nk = ix.npos_key, *t_nv = ix.npos_val;
  if (kLds) {
    uint64_t* s_chr = s_tab;
    uint64_t* s_nk = s_chr + ix.n_chr;
    uint64_t* s_nv = s_nk + ix.n_npos;
    for (uint32_t k = threadIdx.x; k < ix.n_chr; k += blockDim.x) s_chr[k] = ix.chr_start[k];
    for (uint32_t k = threadIdx.x; k < ix.n_npos; k += blockDim.x) { s_nk[k] = ix.npos_key[k]; s_nv[k] = ix.npos_val[k]; }
    __syncthreads();
    t_chr = s_chr; t_nk = s_nk; t_nv = s_nv;
  }
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t H = ix.hash_order;
  const uint64_t hmask = (1ull << (2 * H)) - 1ull;
  const uint32_t tmask = (1u << ix.hash_tag_bits) - 1u;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint4* const recs = reinterpret_cast<const uint4*>(a.packed);
  const uint4* const tab = reinterpret_cast<const uint4*>(ix.hash_tab);
  // pending read indices collect per wave in LDS and leave in batches: one atomic on the shared counter per batch
  // (same-address atomics run at about 1.2e8 per second on this device; one per wave and step would cost more than the
  // whole pass)
  __shared__ uint32_t s_pend_all[4][kSettleBuf];
  __shared__ __attribute__((aligned(16))) uint4 s_bkt_all[4][kBktLds];  // settle_lookup: the buckets of a wave's 64 reads
  uint32_t* const s_pend = s_pend_all[threadIdx.x >> 6];
  uint4* const s_bkt = s_bkt_all[threadIdx.x >> 6];
  uint32_t n_buf = 0;  // wave-uniform
  auto flush_pending = [&]() {
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(a.pending_count, (unsigned long long)n_buf);
    base = ((unsigned long long)__shfl((int)(base >> 32), 0) << 32) | (uint32_t)__shfl((int)(uint32_t)base, 0);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t k = lane; k < n_buf; k += 64) a.pending_list[base + k] = s_pend[k];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    n_buf = 0;
  };
  uint32_t looked = 0, found_n = 0, other_n = 0, absent_n = 0, lines_n = 0;

  // ONE load instruction per line.  A lane that reads its 64-byte bucket through four 16-byte loads makes the second
  // to fourth wait in the vector cache behind the first one's miss (measured: TCP_PENDING_STALL_CYCLES over the whole
  // run of the kernel, 0.035 lines per cycle and CU against the 0.078 of the random-access probe).  So the four lanes
  // of a quad fetch one bucket together, 16 bytes each in the same instruction -- four instructions serve the four
  // reads of the quad, each line touched by exactly one -- and DPP moves inside the quad bring every read its slot.
  const uint64_t n_round = (a.n_reads + 63) & ~63ull;  // whole waves stay in the loop together (ballots, DPP)
  uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool valid = r < a.n_reads;
  uint4 rec = make_uint4(0, 0, 0, 0);
  uint32_t L = 0;
  if (valid) { rec = recs[r]; L = a.lens[r]; }
  for (; r < n_round; r += stride) {
    // the record of the read after this one is on its way while this one's bucket is awaited
    const uint64_t rn = r + stride;
    const bool nvalid = rn < a.n_reads;
    uint4 nrec = make_uint4(0, 0, 0, 0);
    uint32_t nL = 0;
    if (nvalid) { nrec = recs[rn]; nL = a.lens[rn]; }

    const uint64_t w_lo = u64_of(rec.x, rec.y), w_hi = u64_of(rec.z, rec.w);
    bool pending = valid;
    uint64_t e1 = 0;  // record word 1; word 0 (the SA interval) is 0 for every read settled here
    if (valid && (L < (uint32_t)kKmer || L > 64u)) {
      e1 = (uint64_t)R_INVALID << 56;  // as k_search_sm reports a read its record cannot hold: nothing mapped
      pending = false;
    }
    const uint32_t rem = L - H;
    const bool look = pending && L >= H && rem <= 32u;  // (L == H: the located row IS the result)
    const uint64_t vv = hash_mix((~w_lo) & hmask, H);
    const uint64_t bucket = look ? __umul64hi(vv << (64u - 2u * H), ix.hash_buckets) : 0ull;
    const uint32_t tag = (uint32_t)vv & tmask;
    if (look) ++looked;
    uint64_t slot = ~0ull, text = 0, slot2 = ~0ull, text2 = 0;
    bool absent = false;
    settle_lookup(ix, tab, bucket, tag, tmask, look, lane, s_bkt, slot, text, slot2, text2, absent, lines_n);
    bool missed = false;  // looked up and not in the table as a located prefix: the general kernels need not look again
    if (look) {
      uint64_t pos = (slot >> 24) & kHashPosMask;
      uint64_t run = (slot & 0xFFFFFFull) >> ix.hash_tag_bits;
      const bool twin = slot != ~0ull && (slot & kHashTwin) != 0;
      // (run 0: a marker -- the prefix occurs, on several rows or on one not located; so does a twin without its partner)
      missed = slot == ~0ull || run == 0 || (twin && (slot2 == ~0ull || !(slot2 & kHashTwin)));
      if (slot == ~0ull && absent && ix.hash_complete) {
        // the first H bases occur nowhere in the text: the upstream search ends on an empty interval (searcher.hpp:406)
        e1 = (uint64_t)R_NOSA << 56;
        pending = false;
        ++absent_n;
      } else if (!missed) {
        ++found_n;
        // bases H .. L-1 of the read against the text behind the prefix (search_machine.h, S_CMP: same orientation)
        uint64_t r0, r1;
        sm_funnel128(w_lo, w_hi, 0, 0, H, r0, r1);
        const uint64_t cmask = rem < 32u ? (1ull << (2 * rem)) - 1ull : ~0ull;
        uint64_t diff = (r0 ^ text) & cmask;
        bool usable = run > (uint64_t)rem && pos >= (uint64_t)rem;
        if (twin) {
          // two rows hold the prefix: the one whose text the read goes on with is the row the search ends on, alone
          const uint64_t pos2 = (slot2 >> 24) & kHashPosMask, run2 = (slot2 & 0xFFFFFFull) >> ix.hash_tag_bits;
          const uint64_t diff2 = (r0 ^ text2) & cmask;
          usable = usable && run2 > (uint64_t)rem && pos2 >= (uint64_t)rem && (diff != 0 || diff2 != 0);  // (both: two hits, not here)
          if (diff != 0 && diff2 == 0) { pos = pos2; run = run2; diff = 0; }
        }
        if (usable) {
          uint32_t meta = R_NOSA;
          uint64_t payload = 0;
          uint32_t chr_id = 0;
          bool fits = true;
          if (diff == 0) {  // raw interval of one row
            if (1ull > a.limit) {
              meta |= 1u << R_FLAGS_SHIFT;  // SBWT_RF_SUPPRESSED (searcher.hpp:415)
            } else {
              sbwt_hit h;
              if (push_position_keys(ix, t_chr, t_nk, t_nv, pos - rem, L, a.strand, &h)) {  // searcher.hpp:504-539
                payload = h.pos0;
                chr_id = h.chr_id;
                meta |= R_NH | (h.is_rc ? R_RC : 0u);
                fits = chr_id <= a.ring_chr_max && payload <= kPosMask;
              } else if (walk_may_go_on(ix, pos - rem, run - rem)) {
                fits = false;  // (rank_device.h: the general kernels; `run` stops counting at its field's width: more reads, no fewer)
              }
            }
          }
          if (fits) {  // (a result no record holds stays pending: the general kernels own the wide path)
            e1 = payload | ((uint64_t)chr_id << kPosBits) | ((uint64_t)meta << 56);
            pending = false;
          }
        }
      }
    }
    if (pending && !missed) ++other_n;
    // records leave as whole lines: a pending read's slot is written too (zeros) and overwritten by k_scatter_pending
    if (valid) reinterpret_cast<uint4*>(a.out)[r] = make_uint4(0, 0, (uint32_t)e1, (uint32_t)(e1 >> 32));
    const uint64_t pm = __ballot(pending);
    if (pm) {
      if (pending) s_pend[n_buf + (uint32_t)__popcll(pm & ((1ull << lane) - 1ull))] = (uint32_t)r;
      n_buf += (uint32_t)__popcll(pm);
      if (n_buf > kSettleBuf - 64) flush_pending();
    }
    valid = nvalid;
    rec = nrec;
    L = nL;
  }
  const uint32_t tallies[5] = {looked, found_n, other_n, absent_n, lines_n};
  settle_epilogue(a, s_pend, n_buf, tallies, lane);
}

// the tallies of a fast-pass launch (SettleArgs::wg_stats), added into the counters the host reads: one workgroup
__global__ __launch_bounds__(256) void k_fast_stats(const unsigned long long* __restrict__ wg, uint32_t n_wg, SettleArgs a) {
  __shared__ unsigned long long s_part[4][5];
  unsigned long long t[5] = {0, 0, 0, 0, 0};
  for (uint32_t i = threadIdx.x; i < n_wg; i += blockDim.x)
#pragma unroll
    for (int q = 0; q < 5; ++q) t[q] += wg[(size_t)i * 5 + q];
#pragma unroll
  for (int q = 0; q < 5; ++q)
    for (int d = 32; d > 0; d >>= 1) t[q] += shfl_xor_u64(t[q], d);
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int q = 0; q < 5; ++q) s_part[threadIdx.x >> 6][q] = t[q];
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long sum[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) sum[q] = s_part[0][q] + s_part[1][q] + s_part[2][q] + s_part[3][q];
    if (a.hash_stats) { a.hash_stats[0] += sum[0]; a.hash_stats[1] += sum[1]; }
    *a.pending_other += sum[2];
    if (a.absent_count) *a.absent_count += sum[3];
    *a.index_accesses += sum[4];
  }
}

// the pending reads as a dense batch: record (`pieces` 16-byte pieces) and length of read list[i] at slot i
__global__ __launch_bounds__(256) void k_gather_pending(const uint32_t* __restrict__ packed, const uint32_t* __restrict__ lens,
                                                        const uint32_t* __restrict__ list, uint64_t n_pending, uint32_t pieces,
                                                        uint32_t* __restrict__ packed2, uint32_t* __restrict__ lens2,
                                                        const unsigned long long* __restrict__ n_dev) {
  const uint64_t i = (uint64_t)blockIdx.y * gridDim.x * blockDim.x + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n_dev && *n_dev < n_pending) n_pending = *n_dev;  // (launched before the host knows the count: n_pending is the room)
  if (i >= n_pending) return;
  const uint32_t r = list[i];
  for (uint32_t k = 0; k < pieces; ++k)
    reinterpret_cast<uint4*>(packed2)[i * pieces + k] = reinterpret_cast<const uint4*>(packed)[(uint64_t)r * pieces + k];
  lens2[i] = lens[r];
}

// the same for 32-byte records: a unique hit lives at its read's own index, in the pending batch (hits_from) as in
// the caller's (hits); the lists of multi-hit reads are where the caller's first_hit already points
__global__ __launch_bounds__(256) void k_scatter_pending_plain(const sbwt_read_result* __restrict__ from, const sbwt_hit* __restrict__ hits_from,
                                                               const uint32_t* __restrict__ list, uint64_t n_pending,
                                                               sbwt_read_result* __restrict__ out, sbwt_hit* __restrict__ hits,
                                                               const unsigned long long* __restrict__ n_dev) {
  const uint64_t i = (uint64_t)blockIdx.y * gridDim.x * blockDim.x + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n_dev && *n_dev < n_pending) n_pending = *n_dev;
  if (i >= n_pending) return;
  sbwt_read_result res = from[i];
  const uint32_t r = list[i];
  if (res.nh == 1 && res.first_hit == i) {  // (a list of one hit starts behind the slots of all reads: never at i)
    hits[r] = hits_from[i];
    res.first_hit = r;
  }
  out[r] = res;
}

__global__ __launch_bounds__(256) void k_scatter_pending(const sbwt_compact_result* __restrict__ from, const uint32_t* __restrict__ list,
                                                         uint64_t n_pending, sbwt_compact_result* __restrict__ out,
                                                         const unsigned long long* __restrict__ n_dev) {
  const uint64_t i = (uint64_t)blockIdx.y * gridDim.x * blockDim.x + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n_dev && *n_dev < n_pending) n_pending = *n_dev;
  if (i >= n_pending) return;
  reinterpret_cast<uint4*>(out)[list[i]] = reinterpret_cast<const uint4*>(from)[i];
}

}  // namespace sbwt

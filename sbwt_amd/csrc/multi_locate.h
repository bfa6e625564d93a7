// multi_locate.h -- k_multi_sort: locate of multi-hit reads without the depth-first walk.
//
// Upstream (searcher.hpp:424-503) reports the hits of an interval [lo, hi) by extending the pattern to the left one
// base at a time, depth first in A<C<G<T order, and testing every row of every child interval for a sampled SA
// value.  Row i of the interval surfaces exactly once: at the depth t_i where its LF chain reaches a sampled row,
// with position SA = sa_full[i], after the bases (text[p-1], text[p-2], ..., text[p-t_i]) have been prepended.
// The emission order is therefore: rows sorted by that prepended string in pre-order (a string before its
// extensions, A<C<G<T), rows with the same string in row order.  Where the tables are LF-consistent over the
// whole depth of the walk (run >= min(interval, p + 1), device_index.hip) this kernel produces exactly that:
// (Elsewhere a row-by-row walk would NOT do: on tables with unordered ties the depth-first walk keeps following a
// chain after it has surfaced -- a node is left only when it holds ONE row and that row is sampled, searcher.hpp:495 --
// so one chain can surface twice and another never, and the count-based early exit, searcher.hpp:464, depends on the
// order of all of it.  Measured on a tie-shuffled index: per-row walks lose hits the upstream walk reports.)
// gather sa_full[lo..hi) (contiguous), read the up to 63 preceding bases of each hit from the text, sort, filter
// (push_position, searcher.hpp:504-539), write.  One wave per read up to 256 rows, one workgroup above.  Reads with a row outside such a region, with more than
// kSortMax rows, or on an index sampled wider than 64 go to the depth-first kernel (k_multi) instead.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

#include "../../include/sbwt_amd.h"
#include "rank_device.h"
#include "search_machine.h"

namespace sbwt {

constexpr uint32_t kSortMax = 1024;
#ifndef SBWT_RADIX_MAX
#define SBWT_RADIX_MAX 8
#endif
constexpr uint32_t kRadixMaxPasses = SBWT_RADIX_MAX;  // differing key digits up to which the stable radix passes beat the comparison sort

// Reads with the same interval and length have the same hit list (the list is a function of the rows, the read
// length, the strand option and the limit): small-RNA batches hold the same molecule many times over.  The queued
// reads are grouped through a hash table keyed by (lo, rows, length); one member -- the
// leader -- is located, and k_multi_share points the records of the others at its list.  A read the table has no room for (kGroupProbes
// slots taken by other keys) simply leads itself.  Upstream locates every read (searcher.hpp:424-503): same output.
struct GroupArgs {
  unsigned long long* keys;  // null: no grouping
  uint32_t* leader;          // of the key in the same slot: group_scramble(read index) of its leader
  uint32_t mask;             // slots - 1 (a power of two)
};
constexpr uint32_t kGroupProbes = 16;
// The leader is the member with the smallest SCRAMBLED index (a bijection of the 32-bit indices): leaders chosen by
// smallest index would crowd the first part of the batch, i.e. the queue chunks of a few workgroups of k_multi_sort.
__device__ __forceinline__ uint32_t group_scramble(uint32_t r) { return r * 0x9E3779B1u; }
__device__ __forceinline__ uint32_t group_unscramble(uint32_t v) { return v * 0x0E8B2F51u; }
constexpr unsigned long long kGroupEmpty = ~0ull;

__device__ __forceinline__ bool group_key(uint64_t lo, uint64_t rows, uint32_t L, unsigned long long& key) {
  if (rows >= 4096u || L >= 4095u) return false;
  key = lo | (rows << kPosBits) | ((unsigned long long)L << (kPosBits + 12));
  return true;
}
__device__ __forceinline__ uint32_t group_hash(unsigned long long z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return (uint32_t)(z ^ (z >> 31));
}
// slot of `key`, ~0u when it is not in the table (keys are only ever added: an empty slot ends the probe)
__device__ __forceinline__ uint32_t group_find(const GroupArgs& g, unsigned long long key) {
  const uint32_t h0 = group_hash(key);
  for (uint32_t i = 0; i < kGroupProbes; ++i) {
    const uint32_t h = (h0 + i) & g.mask;
    const unsigned long long k = g.keys[h];
    if (k == key) return h;
    if (k == kGroupEmpty) break;
  }
  return ~0u;
}
// read index of the leader of read r's group (r itself when it has none)
__device__ __forceinline__ uint32_t group_leader(const GroupArgs& g, uint32_t r, uint64_t lo, uint64_t rows, uint32_t L) {
  unsigned long long key;
  if (!g.keys || !group_key(lo, rows, L, key)) return r;
  const uint32_t h = group_find(g, key);
  return h == ~0u ? r : group_unscramble(g.leader[h]);
}

struct MultiSortArgs {
  const uint32_t* lens;
  int strand;
  ResultIO io;
  sbwt_hit* hits;
  uint64_t multi_base;                   // first slot of hits[] for multi-hit lists (n_reads, unless the caller splits a batch)
  const uint32_t* queue;                 // wave w of the search kernel queued its reads at queue[w * chunk ..)
  const WaveInfo* wave_info;             // after k_multi_plan: prefix sums over waves
  uint32_t n_waves;
  uint64_t chunk;
  const unsigned long long* queue_count; // total number of queued reads (k_multi_plan)
  const unsigned int* overflow;          // set by k_multi_plan when the hit slots do not fit hits_cap
  unsigned long long* hit_count;         // slots handed out behind multi_base, and the flag above to write, and the room: extra_slots
  unsigned int* overflow_flag;
  uint64_t hits_cap;
  uint32_t lds_tables;                   // the launch carries (n_chr + 2 n_npos) * 8 bytes of dynamic LDS for the filter tables
  uint32_t* dfs_queue;                   // reads this kernel hands on to k_multi
  unsigned long long* dfs_count;
  GroupArgs grp;
  unsigned long long* shared_count;      // reads whose record points at their leader's list (k_multi_share)
  unsigned long long* bfs_count;         // reads of the depth-first queue that k_multi_bfs located
  unsigned long long* rows_count;        // raw rows of the reads the locate kernels located themselves (measurement: a shared list counts once)
  uint32_t no_segments;                  // wave teams take one read at a time whatever its size (SBWT_NO_SEGMENTS: measurement and tests)
};

// Room for a list that is longer than its read has rows.  On tables whose suffix order has unordered ties, a read of L
// bases with L + interval - 1 beyond the depth that order vouches for can surface one chain twice (multi_locate.h top);
// the search kernels reserved one slot per row.  n slots behind everything handed out so far; ~0: hits[] has no room
// (the batch fails with SBWT_ERR_CAPACITY, and the count the caller is told includes this list).
__device__ __forceinline__ uint64_t extra_slots(unsigned long long* hit_count, unsigned int* overflow, uint64_t multi_base,
                                                uint64_t hits_cap, uint64_t n) {
  const uint64_t off = atomicAdd(hit_count, (unsigned long long)n);
  if (multi_base + off + n > hits_cap) {
    atomicExch(overflow, 1u);
    return ~0ull;
  }
  return multi_base + off;
}
// does read r's record hold a list of n hits?  (a 16-byte record has 16 bits for NH)
__device__ __forceinline__ bool list_fits_record(const ResultIO& io, uint64_t r, uint64_t n) {
  if (!io.compact) return n <= 0x7FFFFFFFull;
  if ((uint32_t)(io.compact[r].w1 >> 56) & SBWT_CR_WIDE) return n <= 0x7FFFFFFFull;
  return n <= 0xFFFFull;
}

// Turns the per-wave totals of k_search_sm into global offsets: exclusive prefix sums of the hit slots and of the
// queued reads over waves, the grand totals, and the capacity verdict.  One workgroup.
__global__ __launch_bounds__(1024) void k_multi_plan(WaveInfo* __restrict__ wi, uint32_t n_waves, uint64_t multi_base,
                                                     uint64_t hits_cap, unsigned long long* total_slots,
                                                     unsigned long long* total_count, unsigned int* overflow) {
  __shared__ uint64_t s_slots[16];
  __shared__ uint32_t s_count[16];
  const uint32_t per = (n_waves + 1023u) / 1024u;
  const uint32_t a = threadIdx.x * per < n_waves ? threadIdx.x * per : n_waves;
  const uint32_t b = a + per < n_waves ? a + per : n_waves;
  uint64_t slots = 0;
  uint32_t count = 0;
  for (uint32_t w = a; w < b; ++w) { slots += wi[w].slots; count += wi[w].count; }
  // exclusive prefix over the 1,024 threads: shuffles inside a wave, the sixteen wave totals through LDS
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint64_t is = slots;
  uint32_t ic = count;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t us = ((uint64_t)(uint32_t)__shfl_up((int)(uint32_t)(is >> 32), d) << 32) | (uint32_t)__shfl_up((int)(uint32_t)is, d);
    const uint32_t uc = (uint32_t)__shfl_up((int)ic, d);
    if ((int)lane >= d) { is += us; ic += uc; }
  }
  if (lane == 63) { s_slots[wave] = is; s_count[wave] = ic; }
  __syncthreads();
  uint64_t ps = is - slots;
  uint32_t pc = ic - count;
  for (uint32_t v = 0; v < wave; ++v) { ps += s_slots[v]; pc += s_count[v]; }
  for (uint32_t w = a; w < b; ++w) {
    const uint64_t sl = wi[w].slots;
    const uint32_t cn = wi[w].count;
    wi[w].slots = ps;
    wi[w].queue_base = pc;
    ps += sl;
    pc += cn;
  }
  if (threadIdx.x == 1023) {
    *total_slots = ps;
    *total_count = pc;
    if (multi_base + ps > hits_cap) *overflow = 1u;
  }
}

// every queued read enters its key; the slot's leader becomes the member with the smallest scrambled index
__global__ __launch_bounds__(256) void k_multi_group(const MultiSortArgs a) {
  if (*a.overflow) return;
  for (uint32_t w = blockIdx.x; w < a.n_waves; w += gridDim.x) {
    const uint32_t count = a.wave_info[w].count;
    for (uint32_t q = threadIdx.x; q < count; q += blockDim.x) {
      const uint32_t r = a.queue[(uint64_t)w * a.chunk + q];
      const MultiRead mr = rio_load(a.io, r);
      unsigned long long key;
      if (!group_key(mr.lo, mr.hi - mr.lo, a.lens[r], key)) continue;
      const uint32_t h0 = group_hash(key);
      for (uint32_t i = 0; i < kGroupProbes; ++i) {
        const uint32_t h = (h0 + i) & a.grp.mask;
        const unsigned long long prev = atomicCAS(&a.grp.keys[h], kGroupEmpty, key);
        if (prev == kGroupEmpty || prev == key) {
          atomicMin(&a.grp.leader[h], group_scramble(r));
          break;
        }
      }
    }
  }
}

// where a located read's list is, how many hits it holds, and the flag the locate may have added
__device__ __forceinline__ void rio_load_list(const ResultIO& io, uint64_t r, uint64_t& first_hit, int32_t& nh, uint32_t& truncated) {
  if (!io.compact) {
    const sbwt_read_result res = io.results[r];
    first_hit = res.first_hit; nh = res.nh; truncated = res.flags & SBWT_RF_TRUNCATED;
    return;
  }
  const uint64_t w1 = io.compact[r].w1;
  if ((uint32_t)(w1 >> 56) & SBWT_CR_WIDE) {
    first_hit = 0; nh = 0; truncated = 0;
    const uint64_t slot = w1 & kPosMask;
    if (slot < io.wide_cap) {
      const sbwt_read_result res = io.wide[slot].r;
      first_hit = res.first_hit; nh = res.nh; truncated = res.flags & SBWT_RF_TRUNCATED;
    }
    return;
  }
  first_hit = w1 & kPosMask;
  nh = (int32_t)((w1 >> kPosBits) & 0xFFFFull);
  truncated = ((uint32_t)(w1 >> 56) & SBWT_CR_TRUNCATED) ? SBWT_RF_TRUNCATED : 0u;
}

// after the locate kernels: the members of a group that are not its leader point their record at the leader's list
// (hits[first_hit .. first_hit + nh) of reads with one interval and length is one range; the slots reserved for the
// followers stay unused)
__global__ __launch_bounds__(256) void k_multi_share(const MultiSortArgs a) {
  if (*a.overflow) return;
  uint32_t shared = 0;
  for (uint32_t w = blockIdx.x; w < a.n_waves; w += gridDim.x) {
    const uint32_t count = a.wave_info[w].count;
    for (uint32_t q = threadIdx.x; q < count; q += blockDim.x) {
      const uint32_t r = a.queue[(uint64_t)w * a.chunk + q];
      const MultiRead mr = rio_load(a.io, r);
      const uint32_t lead = group_leader(a.grp, r, mr.lo, mr.hi - mr.lo, a.lens[r]);
      if (lead == r) continue;
      uint64_t first;
      int32_t nh;
      uint32_t truncated;
      rio_load_list(a.io, lead, first, nh, truncated);
      rio_store_multi(a.io, r, first, true, nh, truncated);
      ++shared;
    }
  }
  for (int d = 32; d > 0; d >>= 1) shared += (uint32_t)__shfl_xor((int)shared, d);
  if ((threadIdx.x & 63) == 0 && shared) atomicAdd(a.shared_count, (unsigned long long)shared);
}

// 2-bit groups of x in reverse order (first base most significant)
__device__ __forceinline__ uint64_t rev_bases(uint64_t x) {
  x = __brevll(x);
  return ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
}


// A team of T threads (one wave, or the whole workgroup) handles one read at a time; reads with up to
// kSortMax / (256 / T) rows.  Launched twice: waves take the reads with at most 256 rows (four at a time per
// workgroup, no workgroup barriers), workgroups the larger ones.
template <int T>
__device__ __forceinline__ void team_sync() {
  if (T == 256) {
    __syncthreads();
  } else {
    // one wave: LDS operations of a wave execute in order; only the compiler must not move them
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
}

__device__ __forceinline__ uint64_t shfl_xor64(uint64_t v, uint32_t lane_mask) {
  const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, (int)lane_mask), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), (int)lane_mask);
  return ((uint64_t)hi << 32) | lo;
}

// hit of one row as it waits in LDS for its place in the emission order:
// pos0 (40 bits) | chr_id << 40 (16 bits) | is_rc << 56 | passed the filters << 57 | t << 58 (6 bits)
constexpr uint64_t kPayOk = 1ull << 57;

// What the sort of one read's entries works on (LDS of the team).  An entry is one reported hit of the read; its sort key
// is 16 bytes (sort_key below): the bases its node prepended and its depth.
struct SortLds {
  ulonglong2 *ka, *kb;  // the keys of entries 0 .. m-1; a second buffer of the same size (merge passes; the radix path's counters)
  uint16_t *ta, *tb;    // entry indices in order so far, ping-pong
  uint64_t (*red)[4];
  uint32_t* scan;
};
// The emission order of two hits: the strings their nodes prepended, zero padded (a string before its extensions,
// A < C < G < T), then the depth t, then -- hits of ONE node -- row order.  (h0, h1) = the up to 63 prepended bases, first
// base most significant, zero beyond t; t <= 63.  The key is (x, y) compared as one 128-bit number: x = bases 0..31,
// y = bases 32..59 over an 8-bit code that orders (bases 60..62, t) the same way -- t itself when t <= 60, else the
// pre-order number of the node below depth 60 (a 4-ary trie of depth 3: 21 nodes under every first base, 5 under every
// second).  One 16-byte LDS read per comparison, and no tie-break in the key: the sort is stable and entries are
// numbered in row order.
__device__ __forceinline__ ulonglong2 sort_key(uint64_t h0, uint64_t h1, uint32_t t) {
  uint32_t code = t;
  if (t > 60u) {
    const uint32_t c60 = (uint32_t)(h1 >> 6) & 3u, c61 = (uint32_t)(h1 >> 4) & 3u, c62 = (uint32_t)(h1 >> 2) & 3u;
    code = 61u + c60 * 21u + (t >= 62u ? 1u + c61 * 5u + (t >= 63u ? 1u + c62 : 0u) : 0u);
  }
  return make_ulonglong2(h0, (h1 & ~0xFFull) | code);
}
struct SortedRows {
  const uint16_t* ord;     // entry at place k = ord[k]
  const ulonglong2* keys;  // its key = keys[k]; null: ka[ord[k]] (the merge passes move the keys, the radix passes do not)
};
__device__ __forceinline__ ulonglong2 sorted_key(const SortedRows& R, const SortLds& S, uint32_t k) {
  return R.keys ? R.keys[k] : S.ka[R.ord[k]];
}

// The merge passes of team_sort_rows: sorted runs of 1, 2, 4 ... entries of src merged pairwise until the runs hold r_end
// entries (r_end >= m: the whole array; smaller: every aligned segment of r_end entries for itself).  ts / td: the entry
// indices, moved along with the keys.  Entries m .. (next power of two) - 1 of BOTH key buffers hold the largest key; no
// entry that is merged itself (x < m) has an all-ones second word.
template <int T>
__device__ __forceinline__ SortedRows merge_by_rank(const ulonglong2* src, ulonglong2* dst, uint16_t* ts, uint16_t* td, uint32_t m,
                                                    uint32_t r_end, uint32_t tid) {
  for (uint32_t R = 1; R < m && R < r_end; R <<= 1) {  // sorted runs of R elements -> of 2R
    // (entries m .. 2^k - 1 of both buffers hold the largest key, see above: the run an element is merged with is always
    // R entries long, and what lies beyond the real ones never goes before anything)
    uint64_t ex[4], ey[4], ey_cmp[4];
    uint32_t et[4], ob1[4], cnt[4], dbase[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const uint32_t x = tid + (uint32_t)kk * T;
      ob1[kk] = ~0u; cnt[kk] = 0; dbase[kk] = 0; et[kk] = 0;
      ex[kk] = 0; ey[kk] = 0; ey_cmp[kk] = 0;
      if (x < m) {
        const ulonglong2 e = src[x];
        et[kk] = ts[x];
        const uint32_t pair_base = x & ~(2u * R - 1u);
        const bool right = (x & R) != 0;
        ob1[kk] = pair_base + (right ? 0u : R) - 1u;  // the run this element is merged with starts at ob1 + 1
        dbase[kk] = pair_base + (x & (R - 1u));
        ex[kk] = e.x;
        ey[kk] = e.y;
        // elements of the other run that go before this one: those below it, and -- stability -- for an element of the
        // right run those equal to it too: v <= e  <=>  v < (e.x, e.y + 1) (the low byte of y is the depth code: no carry)
        ey_cmp[kk] = e.y + (right ? 1ull : 0ull);
      }
    }
    // binary lifting, the four searches of a thread side by side: four reads in flight, then four comparisons (written
    // without a branch: one between a read and the next read makes every read wait for the one before it)
    // (steps R/2 ... 1 find min(count, R - 1); a last look at the entry behind those, "step" 1 again, makes it the count:
    // no step ever looks beyond the R entries of the run)
    for (uint32_t steps = R; steps; steps >>= 1) {
      const uint32_t step = steps > 1u ? steps >> 1 : 1u;
      ulonglong2 v[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) v[kk] = src[ob1[kk] + cnt[kk] + step];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const bool before = (v[kk].x < ex[kk]) | ((v[kk].x == ex[kk]) & (v[kk].y < ey_cmp[kk]));
        cnt[kk] += before ? step : 0u;
      }
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const uint32_t x = tid + (uint32_t)kk * T;
      if (x < m) { dst[dbase[kk] + cnt[kk]] = make_ulonglong2(ex[kk], ey[kk]); td[dbase[kk] + cnt[kk]] = (uint16_t)et[kk]; }
    }
    team_sync<T>();
    const ulonglong2* const s1 = src; src = dst; dst = const_cast<ulonglong2*>(s1);
    uint16_t* const t1 = ts; ts = td; td = t1;
  }
  return SortedRows{ts, src};
}

// Puts the m entries of one read in emission order: keys ascending, equal keys in entry order.  or0 .. and1: OR and AND
// of the key words over the calling thread's entries.  Returns the order (SortedRows); ka and kb do not keep their
// contents.  Every thread of the team calls it.
//
// Two ways.  (1) Few 8-bit digits of the key differ between the entries (copies of a repeat share most of their left
// context): one stable counting pass per differing digit.  (2) Otherwise a merge sort whose merges are done by RANK:
// every element of two sorted runs finds its place in the merged run by a binary search over the other run -- log2(run)
// + 1 dependent 16-byte LDS reads, four independent searches per thread in flight, ONE synchronisation per level and no
// data-dependent branch.  (The bitonic network this replaces made log^2 passes of compare-EXCHANGES, each a dependent
// read-compare-write round trip through LDS with a second round trip for entries whose first 24 bases tie -- diverged
// repeat copies tie all the time.  Tried beside it, measured on the repeat-rich genome: merges along the merge path, eight
// places per thread after runs of eight sorted in registers -- a third of the LDS reads, but half the threads idle and a
// chain of dependent reads per level: k_multi_sort 39 ms against 26.5.)
template <int T>
__device__ __forceinline__ SortedRows team_sort_rows(const SortLds& S, uint32_t m, uint32_t tid, uint32_t lane,
                                                     uint64_t or0, uint64_t and0, uint64_t or1, uint64_t and1, bool& used_radix) {
  constexpr uint32_t kCap = kSortMax / (256 / T);
  constexpr uint32_t kWaveRows = kCap / 64;  // 64-entry groups of a team's entries: the unit the radix passes rank in
  ulonglong2* const ka = S.ka;
  const uint64_t lane_lt = (1ull << lane) - 1ull;
  // Which 8-bit digits of the key differ between the entries of this read?  The emission order is what a STABLE sort by
  // those digits alone gives.
  bool radix;
  uint32_t dmask = 0;  // bit d: digit d differs (0 = the first four bases ... 15 = the code)
  {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      or0 |= shfl_xor64(or0, (uint32_t)d); and0 &= shfl_xor64(and0, (uint32_t)d);
      or1 |= shfl_xor64(or1, (uint32_t)d); and1 &= shfl_xor64(and1, (uint32_t)d);
    }
    if (T == 256) {
      uint64_t* red = S.red[threadIdx.x >> 6];
      if (lane == 0) { red[0] = or0; red[1] = and0; red[2] = or1; red[3] = and1; }
      __syncthreads();
      for (uint32_t v = 0; v < 4; ++v) { or0 |= S.red[v][0]; and0 &= S.red[v][1]; or1 |= S.red[v][2]; and1 &= S.red[v][3]; }
    }
    const uint64_t v0 = or0 & ~and0, v1 = or1 & ~and1;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      if ((v0 >> (56 - 8 * b)) & 0xFFull) dmask |= 1u << b;
      if ((v1 >> (56 - 8 * b)) & 0xFFull) dmask |= 1u << (8 + b);
    }
    dmask = (uint32_t)__builtin_amdgcn_readfirstlane((int)dmask);
    radix = __popc(dmask) <= (int)kRadixMaxPasses;
  }
  used_radix = radix;
  uint16_t* ts = S.ta;
  uint16_t* td = S.tb;
  for (uint32_t k = tid; k < m; k += T) ts[k] = (uint16_t)k;
  if (!radix && m > 1u) {
    // the merge passes: both buffers padded with the largest key up to the next power of two
    const uint32_t padded = 1u << (32 - __clz((int)(m - 1u)));
    for (uint32_t k = m + tid; k < padded; k += T) S.ka[k] = S.kb[k] = make_ulonglong2(~0ull, ~0ull);
  }
  team_sync<T>();
  if (radix) {
    uint16_t* const h16 = reinterpret_cast<uint16_t*>(S.kb);  // [256 digits][kWaveRows] counters
    for (int d = 15; d >= 0; --d) {
      if (!((dmask >> d) & 1u)) continue;
      for (uint32_t k = tid; k < kCap * 2u; k += T) reinterpret_cast<uint32_t*>(S.kb)[k] = 0u;
      team_sync<T>();
      uint32_t dig[4], rnk[4], idx[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const uint32_t j = tid + (uint32_t)kk * T, wr = j >> 6;
        dig[kk] = 0; rnk[kk] = 0; idx[kk] = 0;
        if (wr * 64u >= m) continue;  // wave-uniform
        const bool valid = j < m;
        if (valid) {
          idx[kk] = ts[j];
          dig[kk] = (uint32_t)((d >= 8 ? ka[idx[kk]].y : ka[idx[kk]].x) >> (56 - 8 * (d & 7))) & 0xFFu;
        }
        // lanes of this 64-entry group with the same digit; an entry's place among them
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const bool bit = (dig[kk] >> b) & 1u;
          const uint64_t bal = __ballot(bit);
          peers &= bit ? bal : ~bal;
        }
        rnk[kk] = (uint32_t)__popcll(peers & lane_lt);
        if (valid && rnk[kk] == 0) h16[dig[kk] * kWaveRows + wr] = (uint16_t)__popcll(peers);
      }
      team_sync<T>();
      {
        // exclusive prefix sums in (digit, group) order: 16 consecutive counters per thread, then over threads
        uint4* hq = reinterpret_cast<uint4*>(S.kb) + 2u * tid;
        uint4 qa = hq[0], qb = hq[1];
        uint32_t w8[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const uint32_t c_lo = w8[k] & 0xFFFFu, c_hi = w8[k] >> 16;
          w8[k] = run | ((run + c_lo) << 16);
          run += c_lo + c_hi;
        }
        uint32_t inc = run;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
          const uint32_t o = (uint32_t)__shfl_up((int)inc, s);
          if (lane >= (uint32_t)s) inc += o;
        }
        uint32_t base = inc - run;
        if (T == 256) {
          const uint32_t wave = threadIdx.x >> 6;
          if (lane == 63) S.scan[wave] = inc;
          __syncthreads();
          for (uint32_t v = 0; v < wave; ++v) base += S.scan[v];
        }
        const uint32_t b2 = base | (base << 16);
        hq[0] = make_uint4(w8[0] + b2, w8[1] + b2, w8[2] + b2, w8[3] + b2);
        hq[1] = make_uint4(w8[4] + b2, w8[5] + b2, w8[6] + b2, w8[7] + b2);
      }
      team_sync<T>();
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const uint32_t j = tid + (uint32_t)kk * T, wr = j >> 6;
        if (j < m) td[(uint32_t)h16[dig[kk] * kWaveRows + wr] + rnk[kk]] = (uint16_t)idx[kk];
      }
      team_sync<T>();
      uint16_t* const sw = ts; ts = td; td = sw;
    }
    return SortedRows{ts, nullptr};
  }
  // ---- merge sort by rank
  const ulonglong2* src = S.ka;
  ulonglong2* dst = S.kb;
  auto key_less = [](const ulonglong2& a, const ulonglong2& b) -> bool { return a.x < b.x || (a.x == b.x && a.y < b.y); };
  {
    // Entries whose left contexts agree over their whole length (copies of a repeat at the same phase of the sampling,
    // say) are in emission order as they stand: one pass finds out
    bool in_order = true;
    for (uint32_t i = tid; i + 1 < m; i += T) in_order = in_order && !key_less(src[i + 1], src[i]);
    const bool wave_ok = __all(in_order);
    if (T == 256) {
      if (lane == 0) S.scan[threadIdx.x >> 6] = wave_ok ? 1u : 0u;
      __syncthreads();
      in_order = (S.scan[0] & S.scan[1] & S.scan[2] & S.scan[3]) != 0;
      __syncthreads();
    } else {
      in_order = wave_ok;
    }
    if (in_order) return SortedRows{ts, src};
  }
  return merge_by_rank<T>(src, dst, ts, td, m, m, tid);
}

// -DSBWT_PHASE_CLOCKS: thread 0 of every team adds the clocks each phase of a read took to g_multi_phase[] (a
// development aid; map_pass prints the totals)
#ifdef SBWT_PHASE_CLOCKS
__device__ unsigned long long g_multi_phase[24];  // [0..12) wave teams, [12..24) workgroup teams
#define SBWT_PHASE(k) do { const unsigned long long c_ = clock64(); if (tid == 0) s_phase[team][k] += c_ - ph_t; ph_t = c_; } while (0)
#define SBWT_PHASE_MEM(k) do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); SBWT_PHASE(k); } while (0)
#else
#define SBWT_PHASE(k) do { } while (0)
#define SBWT_PHASE_MEM(k) do { } while (0)
#endif

// ---- k_multi_bfs: the depth-first locate of searcher.hpp:424-503, level by level --------------------------------
//
// For the reads k_multi_sort hands on (rows in LF-inconsistent regions: an index whose ties beyond sort_len are
// unordered, as upstream's own builder leaves them).  There the walk cannot be replaced by per-row arithmetic (see the
// header), but it CAN be run breadth first: the nodes of the walk are the intervals (f, e) reached by prepending a
// string of bases, a node's children are lf1 of both ends for A, C, G, T, every node scans its rows for sampled ones
// and pushes them, and a node of one sampled row has no children (searcher.hpp:495).  The frontier of one level is
// expanded by all threads of a team at once -- one (node, base) per thread -- where the depth-first kernel has ONE lane
// walk the whole tree.  What the depth-first ORDER decides is restored afterwards:
//  * the emission order is pre-order, A < C < G < T, rows of one node in row order: the hits are sorted by the string
//    their node prepended (the same keys k_multi_sort sorts by); hits of one node keep their slot order, which is
//    row order;
//  * the early exit (searcher.hpp:464: a child is skipped when the number of accepted hits EQUALS the raw interval
//    size) compares a count that only grows: once it holds at the entry of a node it holds for every later node, so
//    the list is cut at the first node boundary (in emission order) where the accepted hits before it equal the
//    interval size -- and nowhere if the count jumps over that value inside one node;
//  * a list longer than the interval has rows goes into slots of its own (extra_slots above); SBWT_RF_TRUNCATED only
//    where the record cannot hold its length.
// Teams are waves for reads of up to 64 rows (four per workgroup), workgroups up to kSortMax rows; a read whose
// frontier or hits outgrow the team's LDS stays in the queue for k_multi.  A read done here leaves ~0 in its entry.
constexpr uint32_t kQueueDone = ~0u;

template <int T, uint32_t kNodes>  // kNodes: frontier a team holds
__global__ __launch_bounds__(256, kNodes > 512 ? 1 : 3) void k_multi_bfs(const DeviceIndex ix, const MultiSortArgs a, uint32_t* __restrict__ queue,
                                                                        const unsigned long long* __restrict__ queue_count) {
  constexpr uint32_t kTeams = 256 / T;
  constexpr uint32_t kCap = kSortMax / kTeams;       // hits a team holds
  __shared__ __attribute__((aligned(16))) ulonglong2 s_ka[kSortMax];
  __shared__ __attribute__((aligned(16))) uint64_t s_pay[kSortMax];
  // Per team, one region used twice: while the walk runs, the two frontiers -- a node is its first row | rows << 40 and
  // the bases prepended so far (first base most significant) --; once it is over, what the sort and the cut need
  // (the second key buffer, two order buffers, the row at every place, accepted hits before every place).  The kernel is
  // bound by the latency of 63 dependent levels per read: LDS per team decides how many reads a CU has in flight.
  constexpr uint32_t kFrontierBytes = 2u * 3u * kNodes * 8u, kSortBytes = kCap * (16u + 4u * 2u);
  constexpr uint32_t kRawTeam = kFrontierBytes > kSortBytes ? kFrontierBytes : kSortBytes;
  __shared__ __attribute__((aligned(16))) unsigned char s_raw[kTeams * kRawTeam];
  __shared__ uint64_t s_red_all[4][4];
  __shared__ uint32_t s_scan_all[4];
  __shared__ uint32_t s_cnt_all[kTeams][4];  // hits, nodes of the next level, overflow, cut
  __shared__ unsigned long long s_extra[kTeams];  // extra_slots of the read at hand
  // workgroup teams: children of more than kBigRows rows are scanned by a whole wave (a lane per 64-row block), not by
  // the thread that made them -- on tie-shuffled tables the copies of a repeat stay in ONE node level after level
  constexpr uint32_t kBigRows = 128, kBigMax = 16;
  __shared__ uint32_t s_bigi[2][kBigMax];    // the big nodes of this level and of the next: their places in the frontier
  __shared__ uint32_t s_nbig[2];
  __shared__ uint8_t s_pruned_all[kTeams][3 * kNodes];  // per depth (three in rotation): the node was one sampled row
  extern __shared__ __attribute__((aligned(16))) uint64_t s_tab[];
  if (*a.overflow) return;
  const uint64_t *t_chr = ix.chr_start, *t_nk = ix.npos_key, *t_nv = ix.npos_val;
  if (a.lds_tables) {
    uint64_t* s_chr = s_tab;
    uint64_t* s_nk = s_chr + ix.n_chr;
    uint64_t* s_nv = s_nk + ix.n_npos;
    for (uint32_t k = threadIdx.x; k < ix.n_chr; k += blockDim.x) s_chr[k] = ix.chr_start[k];
    for (uint32_t k = threadIdx.x; k < ix.n_npos; k += blockDim.x) { s_nk[k] = ix.npos_key[k]; s_nv[k] = ix.npos_val[k]; }
    __syncthreads();
    t_chr = s_chr; t_nk = s_nk; t_nv = s_nv;
  }
  const uint32_t team = threadIdx.x / T, tid = threadIdx.x % T, lane = threadIdx.x & 63;
  ulonglong2* ka = s_ka + team * kCap;
  unsigned char* const raw = s_raw + team * kRawTeam;
  uint64_t* const fr = reinterpret_cast<uint64_t*>(raw);  // [2][3][kNodes]
  uint16_t* const ord0 = reinterpret_cast<uint16_t*>(raw + kCap * 16u);
  uint64_t* pay = s_pay + team * kCap;
  uint16_t* idx_of = ord0 + 2u * kCap;
  uint16_t* acc_of = ord0 + 3u * kCap;
  uint32_t* cnt = s_cnt_all[team];
  uint8_t* const pruned = s_pruned_all[team];
  const SortLds sort_lds{ka, reinterpret_cast<ulonglong2*>(raw), ord0, ord0 + kCap, s_red_all, s_scan_all};
  if (ix.interval > 64 || ix.n_chr > 0xFFFFu) return;  // key / payload fields too narrow: k_multi
  const uint64_t qn = *queue_count;
#ifdef SBWT_PHASE_CLOCKS
  __shared__ unsigned long long s_phase[4][12];
  if (tid < 12) s_phase[team][tid] = 0;
  team_sync<T>();
  unsigned long long ph_t = clock64();
#endif

  // sampled rows among [f, e)
  auto count_sampled = [&](uint64_t f, uint64_t e) -> uint32_t {
    const uint64_t bf = f >> 6, be = (e - 1) >> 6;
    uint32_t n = 0;
    for (uint64_t bb = bf; bb <= be; ++bb) {
      const uint32_t o_lo = bb == bf ? (uint32_t)(f & 63) : 0u;
      const uint32_t o_hi = bb == be ? (uint32_t)((e - 1) & 63) + 1u : 64u;
      n += (uint32_t)__popcll(ix.blocks1[bb].sampled & below(o_hi) & ~below(o_lo));
    }
    return n;
  };
  // rows [f, e) of one node at depth `depth`, reached by prepending the bases in (p0, p1), n > 0 of them sampled:
  // push those (scan_sampled + multi_push of map_kernels.hip).  Returns true for a node of one (sampled) row.
  auto emit_node = [&](uint64_t f, uint64_t e, uint32_t depth, uint64_t p0, uint64_t p1, uint32_t L, uint32_t n) -> bool {
    const uint64_t bf = f >> 6, be = (e - 1) >> 6;
    uint32_t at = atomicAdd(&cnt[0], n);
    if (at + n > kCap) { cnt[2] = 1u; return false; }
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
        sbwt_hit h;
        uint64_t py = (uint64_t)depth << 58;
        if (push_position_keys(ix, t_chr, t_nk, t_nv, (uint64_t)depth + sa, L, a.strand, &h)) {
          if (h.pos0 > kPosMask) cnt[2] = 1u;  // (an upstream-accepted wrapped position: k_multi writes whole hits)
          py |= kPayOk | (h.pos0 & kPosMask) | ((uint64_t)h.chr_id << kPosBits) | ((uint64_t)(h.is_rc & 1u) << 56);
        }
        ka[at] = sort_key(p0, p1, depth); pay[at] = py;
        ++at;
      }
    }
    return e - f == 1;
  };

  for (uint64_t q = (uint64_t)blockIdx.x * kTeams + team; q < qn; q += (uint64_t)gridDim.x * kTeams) {
    const uint32_t r = queue[q];
    if (r == kQueueDone) continue;  // (team-uniform)
    const MultiRead res = rio_load(a.io, r);  // first_hit is global by now (k_multi_sort)
    const uint64_t lo = res.lo, m = res.hi - res.lo;
    if (m == 0 || m > (T == 64 ? kNodes : kSortMax)) continue;  // (0 rows: a record past wide_cap, never written -- the batch fails)
    const uint32_t L = a.lens[r];
    team_sync<T>();  // the previous read is done with the team's LDS
    if (tid < 4) cnt[tid] = 0;
    for (uint32_t k = tid; k < 3u * kNodes; k += T) pruned[k] = 0;
    if (tid == 0) {
      fr[0] = lo | (m << kPosBits); fr[kNodes] = 0; fr[2 * kNodes] = 0;
      if (T == 256) { s_nbig[0] = m > kBigRows ? 1u : 0u; s_nbig[1] = 0; s_bigi[0][0] = 0; }
    }
    team_sync<T>();
    SBWT_PHASE(0);
    // (the overflow flag is read where every thread of the team sees the same value: between two syncs, like n_cur --
    // a wave ahead of the others may set it for the NEXT level while they still evaluate the loop condition)
    bool over = false;
    // Level d: the nodes of depth d are scanned for sampled rows AND expanded into the nodes of depth d + 1, by
    // different threads at the same time -- a node's own scan is not needed to expand it, except that a node of one
    // sampled row has no children (searcher.hpp:495): its scan marks it, and the children made meanwhile are dropped
    // at the next level.  So a level costs the longer of the two dependent chains (rank query; bitmap -> sampled
    // position), not their sum.
    uint32_t n_cur = 1, cur = 0;
    for (uint32_t d = 0; d < ix.interval && n_cur && !over; ++d) {
      const uint64_t *cfe = fr + cur * 3u * kNodes, *c0 = cfe + kNodes, *c1 = cfe + 2 * kNodes;
      uint64_t *xfe = fr + (cur ^ 1u) * 3u * kNodes, *x0 = xfe + kNodes, *x1 = xfe + 2 * kNodes;
      const uint8_t* dead_of = pruned + ((d + 2u) % 3u) * kNodes;  // marks of depth d - 1
      uint8_t* mark = pruned + (d % 3u) * kNodes;                  // marks of depth d
      uint8_t* wipe = pruned + ((d + 1u) % 3u) * kNodes;           // ... of depth d + 1, written at the next level
      for (uint32_t k = tid; k < kNodes; k += T) wipe[k] = 0;
      const uint32_t n_big = T == 256 ? (s_nbig[d & 1u] < kBigMax ? s_nbig[d & 1u] : kBigMax) : 0u;
      const uint32_t nexp = d + 1 < ix.interval ? 4u * n_cur : 0u, nitems = nexp + n_cur;
      const bool split = T == 256 && n_big > 0;  // the last wave takes the big nodes, the others the items
      const uint32_t W = split ? 192u : (uint32_t)T;
      if (!split || tid < 192u) {
        constexpr int U = 4;
        for (uint32_t it0 = 0; it0 < nitems; it0 += W * U) {
          uint64_t vf[U], ve[U];   // expand: the child's rows; scan: the node's rows
          uint32_t vn[U], node[U]; // scan: sampled rows among them
          uint32_t kind[U];        // 0: nothing, 1: expand, 2: scan
#pragma unroll
          for (int j = 0; j < U; ++j) {
            const uint32_t it = it0 + (uint32_t)j * W + tid;
            kind[j] = it < nexp ? 1u : (it < nitems ? 2u : 0u);
            node[j] = kind[j] == 1u ? it >> 2 : (kind[j] == 2u ? it - nexp : 0u);
            const uint64_t fe = cfe[node[j]];
            const uint64_t f = fe & kPosMask, e = f + ((fe >> kPosBits) & 0x7FFull);
            if (d > 0 && dead_of[(uint32_t)(fe >> 51) & 0x3FFu]) kind[j] = 0;  // its parent was one sampled row
            vn[j] = 0;
            if (kind[j] == 1u) {
              vf[j] = lf1(ix, f, it & 3u);
              ve[j] = lf1(ix, e, it & 3u);
              if (vf[j] >= ve[j]) kind[j] = 0;
            } else {
              vf[j] = f;
              ve[j] = e;
              if (kind[j] == 2u && (T == 64 || e - f <= kBigRows)) vn[j] = count_sampled(f, e);
            }
          }
#pragma unroll
          for (int j = 0; j < U; ++j) {
            if (kind[j] == 1u) {
              const uint32_t c = (it0 + (uint32_t)j * W + tid) & 3u, dd = d + 1;
              uint64_t p0 = c0[node[j]], p1 = c1[node[j]];
              if (dd <= 32u) p0 |= (uint64_t)c << (64u - 2u * dd); else p1 |= (uint64_t)c << (64u - 2u * (dd - 32u));
              const uint32_t at = atomicAdd(&cnt[1], 1u);
              if (at < kNodes) {
                xfe[at] = vf[j] | ((ve[j] - vf[j]) << kPosBits) | ((uint64_t)node[j] << 51);
                x0[at] = p0; x1[at] = p1;
                if (T == 256 && ve[j] - vf[j] > kBigRows) {
                  const uint32_t bt = atomicAdd(&s_nbig[(d + 1u) & 1u], 1u);
                  if (bt < kBigMax) s_bigi[(d + 1u) & 1u][bt] = at; else cnt[2] = 1u;
                }
              } else {
                cnt[2] = 1u;
              }
            } else if (kind[j] == 2u && vn[j]) {
              const bool single = emit_node(vf[j], ve[j], d, c0[node[j]], c1[node[j]], L, vn[j]);
              if (single && d > 0) mark[node[j]] = 1;  // (level 0, searcher.hpp:424-443, has no early return)
            }
          }
        }
      } else {
        for (uint32_t b = 0; b < n_big; ++b) {  // a wave per big node, a lane per 64-row block
          const uint32_t bn = s_bigi[d & 1u][b];
          const uint64_t fe = cfe[bn], f = fe & kPosMask, e = f + ((fe >> kPosBits) & 0x7FFull), p0 = c0[bn], p1 = c1[bn];
          const uint64_t bf = f >> 6, be = (e - 1) >> 6;
          for (uint64_t b0 = bf; b0 <= be; b0 += 64) {
            const uint64_t bb = b0 + lane;
            uint64_t bits = 0, sampled = 0, sbase = 0;
            if (bb <= be) {
              const Block1* blk = ix.blocks1 + bb;
              sampled = blk->sampled;
              sbase = blk->sampled_base;
              const uint32_t o_lo = bb == bf ? (uint32_t)(f & 63) : 0u;
              const uint32_t o_hi = bb == be ? (uint32_t)((e - 1) & 63) + 1u : 64u;
              bits = sampled & below(o_hi) & ~below(o_lo);
            }
            const uint32_t mine = (uint32_t)__popcll(bits);
            uint32_t inc = mine;  // hits of the blocks up to and including this lane's: slots in row order
#pragma unroll
            for (int sft = 1; sft < 64; sft <<= 1) {
              const uint32_t o = (uint32_t)__shfl_up((int)inc, sft);
              if (lane >= (uint32_t)sft) inc += o;
            }
            const uint32_t total = (uint32_t)__shfl((int)inc, 63);
            if (total == 0) continue;
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&cnt[0], total);
            base = (uint32_t)__shfl((int)base, 0);
            if (base + total > kCap) { cnt[2] = 1u; continue; }
            uint32_t at = base + inc - mine;
            while (bits) {
              const uint32_t o = (uint32_t)__ffsll((unsigned long long)bits) - 1u;
              bits &= bits - 1;
              const uint64_t sa = ix.loc_vals[sbase + (uint64_t)__popcll(sampled & below(o))];
              sbwt_hit h;
              uint64_t py = (uint64_t)d << 58;
              if (push_position_keys(ix, t_chr, t_nk, t_nv, (uint64_t)d + sa, L, a.strand, &h)) {
                if (h.pos0 > kPosMask) cnt[2] = 1u;
                py |= kPayOk | (h.pos0 & kPosMask) | ((uint64_t)h.chr_id << kPosBits) | ((uint64_t)(h.is_rc & 1u) << 56);
              }
              ka[at] = sort_key(p0, p1, d); pay[at] = py;
              ++at;
            }
          }
        }
      }
      SBWT_PHASE(1);
      team_sync<T>();
      n_cur = cnt[1] < kNodes ? cnt[1] : kNodes;
      over = cnt[2] != 0;
      team_sync<T>();
      if (tid == 0) { cnt[1] = 0; if (T == 256) s_nbig[d & 1u] = 0; }
      cur ^= 1;
      team_sync<T>();
      SBWT_PHASE(3);
#ifdef SBWT_PHASE_CLOCKS
      if (tid == 0) s_phase[team][11] += 1;
#endif
    }
    if (over) continue;  // outgrew the team: the read stays queued for k_multi (team-uniform: snapshotted between syncs)
    const uint32_t H = cnt[0];
    // ---- emission order
    uint64_t or0 = 0, and0 = ~0ull, or1 = 0, and1 = ~0ull;
    for (uint32_t i = tid; i < H; i += T) {
      const ulonglong2 h = ka[i];
      or0 |= h.x; and0 &= h.x; or1 |= h.y; and1 &= h.y;
    }
    team_sync<T>();  // (the frontiers in the team's region are dead: the sort takes it over)
    bool radix = false;
    const SortedRows sorted = team_sort_rows<T>(sort_lds, H, tid, lane, or0, and0, or1, and1, radix);
    SBWT_PHASE(4);
    // ---- accepted hits before every place, node boundaries, the cut
    uint32_t base = 0;
    for (uint32_t c = 0; c < H; c += T) {
      const uint32_t k = c + tid;
      uint32_t i = 0;
      bool ok = false;
      if (k < H) {
        i = sorted.ord[k];
        ok = (pay[i] & kPayOk) != 0;
        idx_of[k] = (uint16_t)i;
      }
      const uint64_t bal = __ballot(ok);
      uint32_t off = base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
      if (T == 256) {
        const uint32_t wave = threadIdx.x >> 6;
        if (lane == 0) s_scan_all[wave] = (uint32_t)__popcll(bal);
        __syncthreads();
        for (uint32_t v = 0; v < wave; ++v) off += s_scan_all[v];
        base += s_scan_all[0] + s_scan_all[1] + s_scan_all[2] + s_scan_all[3];
        __syncthreads();
      } else {
        base += (uint32_t)__popcll(bal);
      }
      if (k < H) acc_of[k] = (uint16_t)off;  // accepted hits before place k
    }
    if (tid == 0) cnt[3] = H;
    team_sync<T>();
    const uint64_t maybe = (uint64_t)(int64_t)(int32_t)(uint32_t)m;  // (searcher.hpp:412,464: an int against a size_t)
    for (uint32_t k = tid + 1; k < H; k += T) {
      const ulonglong2 qa = sorted_key(sorted, sort_lds, k - 1), qb = sorted_key(sorted, sort_lds, k);
      const bool boundary = qa.x != qb.x || qa.y != qb.y;  // (the depth is part of the key)
      if (boundary && (uint64_t)acc_of[k] == maybe) atomicMin(&cnt[3], k);
    }
    team_sync<T>();
    const uint32_t cut = cnt[3];
    const uint32_t nres = cut < H ? (uint32_t)acc_of[cut] : base;
    uint64_t first = res.first_hit;
    uint32_t room = (uint32_t)m;
    if ((uint64_t)nres > m) {  // more hits than rows (extra_slots above): the list goes behind everything else
      if (tid == 0) s_extra[team] = list_fits_record(a.io, r, nres) ? extra_slots(a.hit_count, a.overflow_flag, a.multi_base, a.hits_cap, nres) : ~0ull;
      team_sync<T>();
      if (s_extra[team] != ~0ull) { first = s_extra[team]; room = nres; }
    }
    sbwt_hit* out = a.hits + first;
    for (uint32_t k = tid; k < cut; k += T) {
      const uint64_t py = pay[idx_of[k]];
      if ((py & kPayOk) && (uint32_t)acc_of[k] < room) {
        sbwt_hit h;
        h.pos0 = py & kPosMask;
        h.chr_id = (uint32_t)(py >> kPosBits) & 0xFFFFu;
        h.is_rc = (uint32_t)(py >> 56) & 1u;
        out[acc_of[k]] = h;
      }
    }
    if (tid == 0) {
      rio_store_multi(a.io, r, first, true, (int32_t)(nres < room ? nres : room), nres > room ? SBWT_RF_TRUNCATED : 0u);
      queue[q] = kQueueDone;
      atomicAdd(a.bfs_count, 1ull);
      atomicAdd(a.rows_count, (unsigned long long)m);
    }
    SBWT_PHASE(5);
#ifdef SBWT_PHASE_CLOCKS
    if (tid == 0) { s_phase[team][10] += 1; s_phase[team][9] += m; }
#endif
  }
#ifdef SBWT_PHASE_CLOCKS
  team_sync<T>();
  (void)s_phase;  // (the slots of g_multi_phase[0..12) now report k_multi_chain)
#endif
}


template <int T, bool kLds>  // kLds: the filter tables are staged in LDS (a compile-time choice: LDS reads, not flat loads)
__global__ __launch_bounds__(256, 3) void k_multi_sort(const DeviceIndex ix, const MultiSortArgs a) {
  constexpr uint32_t kTeams = 256 / T;
  constexpr uint32_t kCap = kSortMax / kTeams;  // rows a team can sort
  // what is sorted: per row the full key (the up to 63 prepended bases, 16 bytes) and a 16-bit tail (depth | row); two
  // buffers of each (team_sort_rows)
  __shared__ __attribute__((aligned(16))) ulonglong2 s_ka[kSortMax], s_kb[kSortMax];
  __shared__ __attribute__((aligned(16))) uint64_t s_pay[kSortMax];
  __shared__ uint32_t s_list_all[256];
  __shared__ uint32_t s_scan_all[4];
  __shared__ uint32_t s_bad_all[4];
  __shared__ uint16_t s_ord_all[2][kSortMax];
  __shared__ uint64_t s_red_all[4][4];          // OR / AND of the key words over a wave's rows
  // wave teams, several small reads at a time (seg_pass below): per segment the read's interval, slots, length
  constexpr uint32_t kSegMax = 32;
  __shared__ uint64_t s_seg_lo_all[4][kSegMax], s_seg_first_all[4][kSegMax];
  __shared__ uint32_t s_seg_m_all[4][kSegMax], s_seg_len_all[4][kSegMax], s_seg_bad_all[4][kSegMax];
  __shared__ uint32_t s_seg_cnt[4][4];  // workgroup teams: accepted hits per (piece of 256 entries, wave)
  extern __shared__ __attribute__((aligned(16))) uint64_t s_tab[];  // (n_chr + 2 n_npos) words when a.lds_tables
  if (*a.overflow) return;  // the hit lists do not fit hits[]: the batch fails with SBWT_ERR_CAPACITY
  // push_position tables: LDS copies when they fit
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
  const uint32_t team = threadIdx.x / T, tid = threadIdx.x % T;
  const uint32_t lane = threadIdx.x & 63;
#ifdef SBWT_PHASE_CLOCKS
  __shared__ unsigned long long s_phase[4][12];  // same-address global atomics per read would dominate what is measured
  if (tid < 12) s_phase[team][tid] = 0;
  team_sync<T>();
  unsigned long long ph_t = clock64();
#endif
  ulonglong2* ka = s_ka + team * kCap;
  uint64_t* pay = s_pay + team * kCap;
  uint32_t* list = s_list_all + team * T;
  uint32_t* s_bad = s_bad_all + team;
  const SortLds sort_lds{ka, s_kb + team * kCap, s_ord_all[0] + team * kCap, s_ord_all[1] + team * kCap, s_red_all, s_scan_all};
  const uint64_t m_text = ix.n - 1;
  const bool no_sort_path = ix.interval > 64 || ix.n_chr > 0xFFFFu;  // payload fields too narrow: depth-first walk
  unsigned long long rows_done = 0;  // (thread 0 of a team; seg_pass: the first lane of a segment)
  // ---- wave teams: SEVERAL small reads at a time.  Most multi-hit reads have a handful of rows (on the repeat-rich bench
  // genome 960,000 of the 1.34 x 10^6 reads of up to 256 rows have at most 8): a wave per read spends its time on the
  // latency of three dependent accesses with a few lanes at work.  The 256 entries of a wave team are cut into segments of
  // kS entries, a read each (32 reads of <= 8 rows, or 4 of <= 64): the same loads, filters and keys for all of them at once,
  // unused entries padded with the largest key, and the merge passes stopped at runs of kS entries sort every segment for itself.
  uint64_t* const seg_lo = s_seg_lo_all[team];
  uint64_t* const seg_first = s_seg_first_all[team];
  uint32_t* const seg_m = s_seg_m_all[team];
  uint32_t* const seg_len = s_seg_len_all[team];
  uint32_t* const seg_bad = s_seg_bad_all[team];
  auto seg_pass = [&](auto seg_size, const uint32_t* lst, uint32_t nseg, const WaveInfo& wi) {
    constexpr uint32_t kS = decltype(seg_size)::value;
    static_assert(kCap / kS <= kSegMax && kCap % kS == 0, "segment descriptors");
    if (tid < nseg) {
      const uint32_t r = lst[tid];
      const MultiRead mr = rio_load(a.io, r);
      seg_lo[tid] = mr.lo;
      seg_m[tid] = (uint32_t)(mr.hi - mr.lo);
      seg_first[tid] = mr.first_hit + a.multi_base + wi.slots;  // wave-relative until now
      seg_len[tid] = a.lens[r];
      seg_bad[tid] = 0;
    }
    team_sync<T>();
    uint64_t e4[4];
    uint32_t len4[4];
    bool live[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const uint32_t x = tid + (uint32_t)T * (uint32_t)kk, g = x / kS, i = x % kS;
      live[kk] = g < nseg && i < seg_m[g];
      len4[kk] = g < nseg ? seg_len[g] : 0u;
      e4[kk] = live[kk] ? ix.sa_full[seg_lo[g] + i] : kNoPosition;
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const uint32_t g = (tid + (uint32_t)T * (uint32_t)kk) / kS;
      const uint64_t p = e4[kk] & kPosMask;
      // a row the text does not answer for: its read goes to k_multi_chain / the walk
      if (live[kk] && (e4[kk] == kNoPosition || (e4[kk] >> kPosBits) < (p + 1 < ix.interval ? p + 1 : ix.interval))) seg_bad[g] = 1u;
    }
    team_sync<T>();
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const uint32_t g = (tid + (uint32_t)T * (uint32_t)kk) / kS;
      live[kk] = live[kk] && !seg_bad[g];
    }
    // the text windows, pair-cooperative (as below)
    uint4 tw[4][2];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const uint64_t p = e4[kk] & kPosMask;
      const uint64_t x0 = !live[kk] || p > m_text ? 0 : m_text - p;
      const uintptr_t tx = reinterpret_cast<uintptr_t>(reinterpret_cast<const uint4*>(ix.textrc) + (x0 >> 6));
      const uint32_t tx_lo = (uint32_t)tx, tx_hi = (uint32_t)(tx >> 32);
      const uintptr_t t0 = u64_of(pair_bcast<0>(tx_lo), pair_bcast<0>(tx_hi)), t1 = u64_of(pair_bcast<1>(tx_lo), pair_bcast<1>(tx_hi));
      const uint4 r0 = reinterpret_cast<const uint4*>(t0)[lane & 1u];
      const uint4 r1 = reinterpret_cast<const uint4*>(t1)[lane & 1u];
      const uint4 s0 = make_uint4(quad_swap1(r0.x), quad_swap1(r0.y), quad_swap1(r0.z), quad_swap1(r0.w));
      const uint4 s1 = make_uint4(quad_swap1(r1.x), quad_swap1(r1.y), quad_swap1(r1.z), quad_swap1(r1.w));
      tw[kk][0] = (lane & 1u) ? s1 : r0;
      tw[kk][1] = (lane & 1u) ? r1 : s0;
    }
    uint64_t fpos[4] = {0, 0, 0, 0};
    uint32_t c1[4], n1[4];
    bool frc[4], fgo[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      frc[kk] = false;
      fgo[kk] = false;
      if (live[kk]) {
        fpos[kk] = e4[kk] & kPosMask;
        fgo[kk] = push_position_orient(ix, fpos[kk], len4[kk], a.strand, frc[kk]);
        if (!fgo[kk]) fpos[kk] = 0;
      }
    }
    upper_bound_minus1_x4(t_chr, ix.n_chr, fpos, c1);
    upper_bound_minus1_x4(t_nk, ix.n_npos, fpos, n1);
    const bool pow2 = (ix.interval & (ix.interval - 1u)) == 0;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const uint32_t x = tid + (uint32_t)T * (uint32_t)kk, g = x / kS;
      // (an unused entry: above every key of a row; not all ones, because these entries ARE merged -- an element of a right
      // run is compared through y + 1, merge_by_rank)
      ulonglong2 key = make_ulonglong2(~0ull, ~0ull - 1ull);
      uint64_t py = 63ull << 58;
      if (live[kk]) {
        const uint64_t p = e4[kk] & kPosMask;
        const uint32_t t = pow2 ? (uint32_t)p & (ix.interval - 1u) : (uint32_t)(p % ix.interval);
        const uint64_t x0 = m_text - p;
        sbwt_hit h;
        const bool ok = fgo[kk] && push_position_finish(ix, t_chr, t_nk, t_nv, fpos[kk], len4[kk], frc[kk], c1[kk], n1[kk], &h);
        uint64_t c0, cc1;
        sm_funnel128(u64_of(tw[kk][0].x, tw[kk][0].y), u64_of(tw[kk][0].z, tw[kk][0].w), u64_of(tw[kk][1].x, tw[kk][1].y),
                     u64_of(tw[kk][1].z, tw[kk][1].w), (uint32_t)(x0 & 63), c0, cc1);
        c0 = ~c0;
        cc1 = ~cc1;
        if (t < 32u) { c0 &= t ? ((1ull << (2 * t)) - 1ull) : 0ull; cc1 = 0; }
        else if (t < 64u) cc1 &= t > 32u ? ((1ull << (2 * (t - 32u))) - 1ull) : 0ull;
        key = sort_key(rev_bases(c0), rev_bases(cc1), t);
        py = ((uint64_t)t << 58);
        if (ok && h.pos0 > kPosMask) seg_bad[g] = 1u;  // (an upstream-accepted wrapped position: the depth-first kernel writes whole hits)
        if (ok) py |= kPayOk | (h.pos0 & kPosMask) | ((uint64_t)h.chr_id << kPosBits) | ((uint64_t)(h.is_rc & 1u) << 56);
      }
      ka[x] = key;
      pay[x] = py;
      sort_lds.ta[x] = (uint16_t)x;
    }
    team_sync<T>();
    const SortedRows sorted = merge_by_rank<T>(sort_lds.ka, sort_lds.kb, sort_lds.ta, sort_lds.tb, kCap, kS, tid);
    // every segment's rows in emission order: compact those that passed the filters, write
    uint64_t bal4[4];
    uint64_t py4[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const uint32_t k = tid + (uint32_t)T * (uint32_t)kk, g = k / kS;
      const bool seg = g < nseg, bad = seg && seg_bad[g] != 0;
      py4[kk] = seg && !bad ? pay[sorted.ord[k]] : 0ull;
      bal4[kk] = __ballot((py4[kk] & kPayOk) != 0);
    }
    if (T == 256) {
      // a segment spans waves: the accepted hits of every (kk, wave) piece, through LDS
      static_assert(T != 256 || kS % 256u == 0, "workgroup teams: a segment is whole pieces of 256 entries");
      const uint32_t wave = threadIdx.x >> 6;
      if (lane == 0) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) s_seg_cnt[kk][wave] = (uint32_t)__popcll(bal4[kk]);
      }
      __syncthreads();
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const uint32_t k = tid + (uint32_t)T * (uint32_t)kk, g = k / kS, i = k % kS;
      const bool seg = g < nseg, bad = seg && seg_bad[g] != 0;
      const uint64_t py = py4[kk];
      const bool ok = (py & kPayOk) != 0;
      uint32_t before, total;  // accepted hits of this segment in front of this entry; in all of it
      if (T == 64) {
        const uint64_t mine_bits = kS == 64 ? bal4[kk] : (bal4[kk] >> (lane & ~(kS - 1u))) & ((1ull << (kS & 63u)) - 1ull);
        before = (uint32_t)__popcll(mine_bits & ((1ull << (i & 63u)) - 1ull));
        total = (uint32_t)__popcll(mine_bits);
      } else {
        constexpr uint32_t kPer = kS / 256u > 0 ? kS / 256u : 1u;  // pieces (kk) of a segment
        const uint32_t wave = threadIdx.x >> 6, k0 = ((uint32_t)kk / kPer) * kPer;
        before = (uint32_t)__popcll(bal4[kk] & ((1ull << lane) - 1ull));
        total = 0;
        for (uint32_t q = k0; q < k0 + kPer; ++q)
          for (uint32_t v = 0; v < 4; ++v) {
            const uint32_t c = s_seg_cnt[q][v];
            total += c;
            if (q < (uint32_t)kk || (q == (uint32_t)kk && v < wave)) before += c;
          }
      }
      if (ok) {
        sbwt_hit h;
        h.pos0 = py & kPosMask;
        h.chr_id = (uint32_t)(py >> kPosBits) & 0xFFFFu;
        h.is_rc = (uint32_t)(py >> 56) & 1u;
        a.hits[seg_first[g] + before] = h;
      }
      if (seg && i == 0) {
        const uint32_t r = lst[g];
        if (bad) {
          rio_store_multi(a.io, r, seg_first[g], false, 0, 0u);
          a.dfs_queue[atomicAdd(a.dfs_count, 1ull)] = r;
        } else {
          rio_store_multi(a.io, r, seg_first[g], true, (int32_t)total, 0u);
          rows_done += seg_m[g];
        }
      }
    }
    team_sync<T>();
  };
  // teams stride over the waves of the search kernel; each wave queued its reads at queue[w * chunk ..)
  for (uint32_t w = blockIdx.x * kTeams + team; w < a.n_waves; w += gridDim.x * kTeams) {
    const WaveInfo wi = a.wave_info[w];
    for (uint32_t q0 = 0; q0 < wi.count; q0 += T) {
      // the team looks at T queue entries at once and keeps those this launch takes
      uint32_t my_r = 0;
      bool mine = false;
      uint32_t cls = 2;  // wave teams: 0 = at most 8 rows, 1 = at most 64 (several reads at a time, seg_pass), 2 = one read at a time
      if (q0 + tid < wi.count) {
        my_r = a.queue[(uint64_t)w * a.chunk + q0 + tid];
        const MultiRead mr = rio_load(a.io, my_r);
        const uint64_t m = mr.hi - mr.lo;
        mine = T == 64 ? m <= kCap : m > kSortMax / 4;
        if (mine && a.grp.keys) mine = group_leader(a.grp, my_r, mr.lo, m, a.lens[my_r]) == my_r;
        if (T == 64 && !no_sort_path && a.no_segments != 1u) cls = m <= 8 && a.no_segments != 2u ? 0u : m <= 64 ? 1u : 2u;
        if (T == 256 && !no_sort_path && a.no_segments != 1u) cls = m <= 512 ? 0u : 2u;  // two reads of up to 512 rows at a time
      }
      uint32_t n_list, n_cls0 = 0, n_cls1 = 0;
      {
        const uint64_t bal = __ballot(mine);
        uint32_t off = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        if (T == 64) {
          // the list in the order of the classes
          const uint64_t lt = (1ull << lane) - 1ull;
          const uint64_t b0 = __ballot(mine && cls == 0), b1 = __ballot(mine && cls == 1), b2 = __ballot(mine && cls == 2);
          n_cls0 = (uint32_t)__popcll(b0);
          n_cls1 = (uint32_t)__popcll(b1);
          off = cls == 0 ? (uint32_t)__popcll(b0 & lt) : cls == 1 ? n_cls0 + (uint32_t)__popcll(b1 & lt) : n_cls0 + n_cls1 + (uint32_t)__popcll(b2 & lt);
        }
        if (T == 256) {
          // the list in the order of the classes: those of class 0 of every wave, then the others
          const uint32_t wave = threadIdx.x >> 6;
          const uint64_t lt = (1ull << lane) - 1ull;
          const uint64_t b0 = __ballot(mine && cls == 0), b2 = __ballot(mine && cls != 0);
          __syncthreads();  // the previous batch is done with list and s_seg_cnt
          if (lane == 0) { s_seg_cnt[0][wave] = (uint32_t)__popcll(b0); s_seg_cnt[1][wave] = (uint32_t)__popcll(b2); }
          __syncthreads();
          n_cls0 = s_seg_cnt[0][0] + s_seg_cnt[0][1] + s_seg_cnt[0][2] + s_seg_cnt[0][3];
          n_list = n_cls0 + s_seg_cnt[1][0] + s_seg_cnt[1][1] + s_seg_cnt[1][2] + s_seg_cnt[1][3];
          off = cls == 0 ? (uint32_t)__popcll(b0 & lt) : n_cls0 + (uint32_t)__popcll(b2 & lt);
          for (uint32_t v = 0; v < wave; ++v) off += s_seg_cnt[cls == 0 ? 0 : 1][v];
          __syncthreads();  // (s_seg_cnt is written again by seg_pass)
        } else {
          n_list = (uint32_t)__popcll(bal);
        }
        if (mine) list[off] = my_r;
      }
      team_sync<T>();
      SBWT_PHASE(0);
      if constexpr (T == 256) {
        for (uint32_t at = 0; at < n_cls0; at += 2) seg_pass(std::integral_constant<uint32_t, 512>{}, list + at, n_cls0 - at < 2u ? n_cls0 - at : 2u, wi);
      }
      if constexpr (T == 64) {
        for (uint32_t at = 0; at < n_cls0; at += 32) seg_pass(std::integral_constant<uint32_t, 8>{}, list + at, n_cls0 - at < 32u ? n_cls0 - at : 32u, wi);
        for (uint32_t at = 0; at < n_cls1; at += 4) seg_pass(std::integral_constant<uint32_t, 64>{}, list + n_cls0 + at, n_cls1 - at < 4u ? n_cls1 - at : 4u, wi);
      }
      for (uint32_t li = n_cls0 + n_cls1; li < n_list; ++li) {
        const uint32_t r = list[li];
        MultiRead res = rio_load(a.io, r);
        const uint64_t lo = res.lo, m = res.hi - res.lo;
        const uint32_t L = a.lens[r];
        res.first_hit += a.multi_base + wi.slots;  // wave-relative until now
        const bool pre_bad = m > kCap || no_sort_path;  // (team-uniform, from registers)
        if (tid == 0) *s_bad = pre_bad ? 1u : 0u;
        team_sync<T>();
        SBWT_PHASE_MEM(1);
        uint64_t or0 = 0, and0 = ~0ull, or1 = 0, and1 = ~0ull;  // over this thread's rows: which key bits differ
        if (!pre_bad) {
          // up to four rows per thread, phase by phase, so that the loads of all of them are in flight together
          uint64_t e4[4];
          uint4 tw[4][2];  // 32 bytes of textrc from a 16-byte boundary around the row's left context
          bool row_bad = false;
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const uint32_t i = tid + (uint32_t)kk * T;
            e4[kk] = i < m ? ix.sa_full[lo + i] : kNoPosition;
            const uint64_t p = e4[kk] & kPosMask;
            if (i < m && (e4[kk] == kNoPosition || (e4[kk] >> kPosBits) < (p + 1 < ix.interval ? p + 1 : ix.interval))) row_bad = true;
          }
          // a row the text does not answer for: the read goes to k_multi_chain / the walk, and nothing more is fetched for it here
          if (row_bad) *s_bad = 1u;  // benign race: every writer stores 1
          team_sync<T>();
          SBWT_PHASE_MEM(2);
          if (!*s_bad) {  // (wave-uniform: the lanes of a wave read it together; nothing below synchronises the team)
          // ONE load instruction per line (settle.h): the two 16-byte halves of a row's window are fetched by the two
          // lanes of a pair in the same instruction -- round h serves the row of the pair's lane h -- and a DPP swap
          // hands the partner's half over.  (A lane loading both halves itself keeps the second load waiting in the
          // vector cache behind the first one's miss.)
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const uint64_t p = e4[kk] & kPosMask;
            // text[p - 1 - k] = 3 - textrc[x0 + k]; a row without a position reads the start of the text
            const uint64_t x0 = e4[kk] == kNoPosition || p > m_text ? 0 : m_text - p;
            const uintptr_t tx = reinterpret_cast<uintptr_t>(reinterpret_cast<const uint4*>(ix.textrc) + (x0 >> 6));
            const uint32_t tx_lo = (uint32_t)tx, tx_hi = (uint32_t)(tx >> 32);
            const uintptr_t t0 = u64_of(pair_bcast<0>(tx_lo), pair_bcast<0>(tx_hi)), t1 = u64_of(pair_bcast<1>(tx_lo), pair_bcast<1>(tx_hi));
            const uint4 r0 = reinterpret_cast<const uint4*>(t0)[lane & 1u];
            const uint4 r1 = reinterpret_cast<const uint4*>(t1)[lane & 1u];
            const uint4 s0 = make_uint4(quad_swap1(r0.x), quad_swap1(r0.y), quad_swap1(r0.z), quad_swap1(r0.w));
            const uint4 s1 = make_uint4(quad_swap1(r1.x), quad_swap1(r1.y), quad_swap1(r1.z), quad_swap1(r1.w));
            tw[kk][0] = (lane & 1u) ? s1 : r0;
            tw[kk][1] = (lane & 1u) ? r1 : s0;
          }
          SBWT_PHASE_MEM(3);
          // filters (push_position, searcher.hpp:504-539): the table searches of the four rows run interleaved
          uint64_t fpos[4] = {0, 0, 0, 0};
          uint32_t c1[4], n1[4];
          bool frc[4], fgo[4];
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const uint32_t i = tid + (uint32_t)kk * T;
            const uint64_t e = e4[kk], p = e & kPosMask;
            const uint64_t need = p + 1 < ix.interval ? p + 1 : ix.interval;
            frc[kk] = false;
            fgo[kk] = false;
            if (i < m) {
              if (e == kNoPosition || (e >> kPosBits) < need) {
                *s_bad = 1u;  // benign race: every writer stores 1
              } else {
                fpos[kk] = p;
                fgo[kk] = push_position_orient(ix, fpos[kk], L, a.strand, frc[kk]);
                if (!fgo[kk]) fpos[kk] = 0;
              }
            }
          }
          upper_bound_minus1_x4(t_chr, ix.n_chr, fpos, c1);
          upper_bound_minus1_x4(t_nk, ix.n_npos, fpos, n1);
          const bool pow2 = (ix.interval & (ix.interval - 1u)) == 0;
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const uint32_t i = tid + (uint32_t)kk * T;
            ulonglong2 key = make_ulonglong2(~0ull, ~0ull);
            uint64_t py = 63ull << 58;
            if (i < m) {
              const uint64_t e = e4[kk];
              const uint64_t p = e & kPosMask;
              const uint32_t t = pow2 ? (uint32_t)p & (ix.interval - 1u) : (uint32_t)(p % ix.interval);
              const uint64_t need = p + 1 < ix.interval ? p + 1 : ix.interval;
              if (!(e == kNoPosition || (e >> kPosBits) < need)) {
                const uint64_t x0 = m_text - p;
                sbwt_hit h;
                const bool ok = fgo[kk] && push_position_finish(ix, t_chr, t_nk, t_nv, fpos[kk], L, frc[kk], c1[kk], n1[kk], &h);
                uint64_t c0, cc1;  // the 64 bases from x0 on, out of the 128 loaded from the 64-base boundary below it
                sm_funnel128(u64_of(tw[kk][0].x, tw[kk][0].y), u64_of(tw[kk][0].z, tw[kk][0].w), u64_of(tw[kk][1].x, tw[kk][1].y),
                             u64_of(tw[kk][1].z, tw[kk][1].w), (uint32_t)(x0 & 63), c0, cc1);
                c0 = ~c0;
                cc1 = ~cc1;
                if (t < 32u) { c0 &= t ? ((1ull << (2 * t)) - 1ull) : 0ull; cc1 = 0; }
                else if (t < 64u) cc1 &= t > 32u ? ((1ull << (2 * (t - 32u))) - 1ull) : 0ull;
                key = sort_key(rev_bases(c0), rev_bases(cc1), t);
                py = ((uint64_t)t << 58);
                if (ok && h.pos0 > kPosMask) *s_bad = 1u;  // (an upstream-accepted wrapped position: the depth-first kernel writes whole hits)
                if (ok) py |= kPayOk | (h.pos0 & kPosMask) | ((uint64_t)h.chr_id << kPosBits) | ((uint64_t)(h.is_rc & 1u) << 56);
                or0 |= key.x; and0 &= key.x; or1 |= key.y; and1 &= key.y;
              }
            }
            if (i < m) { ka[i] = key; pay[i] = py; }
          }
          }
        }
        team_sync<T>();
        SBWT_PHASE(4);
        if (*s_bad) {
          if (tid == 0) {
            rio_store_multi(a.io, r, res.first_hit, false, 0, 0u);
            a.dfs_queue[atomicAdd(a.dfs_count, 1ull)] = r;
          }
          team_sync<T>();
          continue;
        }
        bool radix = false;
        const SortedRows sorted = team_sort_rows<T>(sort_lds, (uint32_t)m, tid, lane, or0, and0, or1, and1, radix);
        SBWT_PHASE(6);
        // rows in emission order: compact those that passed the filters, write
        sbwt_hit* out = a.hits + res.first_hit;
        uint32_t base = 0;
        for (uint32_t c = 0; c < m; c += T) {
          const uint32_t k = c + tid;
          uint64_t py = 0;
          if (k < m) py = pay[sorted.ord[k]];
          const bool ok = (py & kPayOk) != 0;
          sbwt_hit h;
          h.pos0 = py & kPosMask;
          h.chr_id = (uint32_t)(py >> kPosBits) & 0xFFFFu;
          h.is_rc = (uint32_t)(py >> 56) & 1u;
          const uint64_t bal = __ballot(ok);
          if (T == 256) {
            const uint32_t wave = threadIdx.x >> 6;
            if (lane == 0) s_scan_all[wave] = (uint32_t)__popcll(bal);
            __syncthreads();
            uint32_t off = base;
            for (uint32_t v = 0; v < wave; ++v) off += s_scan_all[v];
            if (ok) out[off + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = h;
            base += s_scan_all[0] + s_scan_all[1] + s_scan_all[2] + s_scan_all[3];
            __syncthreads();
          } else {
            if (ok) out[base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = h;
            base += (uint32_t)__popcll(bal);
          }
        }
        if (tid == 0) { rio_store_multi(a.io, r, res.first_hit, true, (int32_t)base, 0u); rows_done += m; }
        team_sync<T>();
        SBWT_PHASE(8);
#ifdef SBWT_PHASE_CLOCKS
        if (tid == 0) { s_phase[team][10] += 1; s_phase[team][11] += radix ? 1 : 0; }
#endif
      }
    }
  }
  for (int d = 32; d > 0; d >>= 1) rows_done += shfl_xor64(rows_done, (uint32_t)d);
  if (lane == 0 && rows_done) atomicAdd(a.rows_count, rows_done);
#ifdef SBWT_PHASE_CLOCKS
  team_sync<T>();
  if (tid < 12 && T == 256) atomicAdd(&g_multi_phase[12 + tid], s_phase[team][tid]);
#endif
}


// ---- k_multi_chain: the depth-first locate put together from memoised chains -------------------------------------
//
// For the reads k_multi_sort hands on because a row lies outside an LF-consistent region.  find_possible
// (searcher.hpp:460-503) reaches the rows of a node from the rows of its parent through LF, one base at a time, and LF
// restricted to the rows of one base keeps their order: so the node the walk holds after prepending the bases w is
// exactly { LF^|w|(i) : i a row of the read's interval whose chain spells w }, in the order of the rows i.  A row's
// chain does not depend on the read: chain_tab (device_layout.h) holds it for the rows the text does not answer for,
// the text IS the chain of the others.  What the walk reports is then, per row, its events (depth d, position) --
// sampled rows on its chain -- and its order falls out of the chains:
//  * emission order = pre-order over the prepended strings, A < C < G < T, rows of one node in row order: the events
//    sorted by (first d bases of the chain, d), ties in row order (team_sort_rows; entries are numbered in row order);
//  * a node of ONE row that is sampled has no children (searcher.hpp:495): an event at depth d >= 1 of a chain whose
//    first d bases no other chain of the read shares (no other row passes through that node) ends the chain there;
//  * the early exit (searcher.hpp:464) and the lists longer than their intervals: as in k_multi_bfs, on the sorted list.
// A read whose entries outgrow the team, or with a row chain_tab does not hold, stays queued for k_multi_bfs / k_multi.
template <int T, bool kLds>
__global__ __launch_bounds__(256, 3) void k_multi_chain(const DeviceIndex ix, const MultiSortArgs a, uint32_t* __restrict__ queue,
                                                        const unsigned long long* __restrict__ queue_count,
                                                        unsigned long long* __restrict__ chain_count) {
  constexpr uint32_t kTeams = 256 / T;
  constexpr uint32_t kCap = kSortMax / kTeams;  // rows, and entries, a team holds
  static_assert(kCap == 4u * T, "four rows, and four entries, per thread");
  __shared__ __attribute__((aligned(16))) ulonglong2 s_ka[kSortMax], s_kb[kSortMax];
  __shared__ __attribute__((aligned(16))) uint64_t s_pay[kSortMax];
  __shared__ uint16_t s_ord_all[2][kSortMax];
  __shared__ uint16_t s_acc_all[kSortMax];
  __shared__ uint8_t s_len_all[kSortMax];
  __shared__ uint64_t s_red_all[4][4];
  __shared__ uint32_t s_scan_all[4];
  __shared__ uint32_t s_cnt_all[kTeams][4];  // [0] fail, [1] cut
  __shared__ unsigned long long s_extra[kTeams];  // extra_slots of the read at hand
  extern __shared__ __attribute__((aligned(16))) uint64_t s_tab[];
  if (*a.overflow) return;
  if (!ix.chain_tab || ix.interval > 64 || ix.n_chr > 0xFFFFu) return;
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
  const uint32_t team = threadIdx.x / T, tid = threadIdx.x % T, lane = threadIdx.x & 63;
  ulonglong2* const ka = s_ka + team * kCap;
  ulonglong2* const kb = s_kb + team * kCap;  // before the sort: the whole chain of every row
  uint64_t* const pay = s_pay + team * kCap;
  uint16_t* const acc_of = s_acc_all + team * kCap;
  uint8_t* const len_of = s_len_all + team * kCap;
  uint32_t* const cnt = s_cnt_all[team];
  const SortLds sort_lds{ka, kb, s_ord_all[0] + team * kCap, s_ord_all[1] + team * kCap, s_red_all, s_scan_all};
  const uint64_t qn = *queue_count;
  const uint64_t m_text = ix.n - 1;
  const uint32_t depth_max = (uint32_t)ix.interval - 1u;  // bases a chain can hold
  unsigned long long rows_done = 0, reads_done = 0;  // (thread 0 of a team; one atomic each when the kernel ends)
#ifdef SBWT_PHASE_CLOCKS
  __shared__ unsigned long long s_phase[4][12];
  if (tid < 12) s_phase[team][tid] = 0;
  team_sync<T>();
  unsigned long long ph_t = clock64();
#endif
  // the first d bases of a chain (d <= 63), zero padded
  auto prefix_of = [](ulonglong2 f, uint32_t d) -> ulonglong2 {
    if (d == 0) return make_ulonglong2(0, 0);
    if (d < 32u) return make_ulonglong2(f.x & (~0ull << (64u - 2u * d)), 0);
    if (d == 32u) return make_ulonglong2(f.x, 0);
    return make_ulonglong2(f.x, f.y & (~0ull << (64u - 2u * (d - 32u))));
  };
  // wave-level exclusive scan; *total = sum over the team (T == 256: through s_scan_all, two barriers)
  auto team_excl_scan = [&](uint32_t v, uint32_t* total) -> uint32_t {
    uint32_t inc = v;
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) {
      const uint32_t o = (uint32_t)__shfl_up((int)inc, sft);
      if (lane >= (uint32_t)sft) inc += o;
    }
    uint32_t base = inc - v, tot = (uint32_t)__shfl((int)inc, 63);
    if (T == 256) {
      const uint32_t wave = threadIdx.x >> 6;
      __syncthreads();
      if (lane == 63) s_scan_all[wave] = inc;
      __syncthreads();
      tot = s_scan_all[0] + s_scan_all[1] + s_scan_all[2] + s_scan_all[3];
      for (uint32_t v2 = 0; v2 < wave; ++v2) base += s_scan_all[v2];
    }
    *total = tot;
    return base;
  };

  for (uint64_t q = (uint64_t)blockIdx.x * kTeams + team; q < qn; q += (uint64_t)gridDim.x * kTeams) {
    const uint32_t r = queue[q];
    if (r == kQueueDone) continue;  // (team-uniform)
    const MultiRead res = rio_load(a.io, r);  // first_hit is global by now (k_multi_sort)
    const uint64_t lo = res.lo, m64 = res.hi - res.lo;
    if (m64 == 0 || m64 > kCap) continue;  // (0 rows: a record past wide_cap, never written -- the batch fails)
    const uint32_t m = (uint32_t)m64;
    const uint32_t L = a.lens[r];
    team_sync<T>();  // the previous read is done with the team's LDS
    if (tid < 4) cnt[tid] = 0;
    // ---- the chains of this thread's rows: rows 4 tid .. 4 tid + 3 (consecutive, so that entries numbered thread by
    // thread are numbered in row order)
    ulonglong2 full[4];
    uint64_t ev0[4], meta[4];
    uint32_t n_ev[4], len[4];
    bool fail = false;
    // three rounds of loads, the four rows of a thread side by side in each (no branch between a row's load and the next
    // row's: the compiler would make every load wait for the one before it): sa_full; then the text window of a row the
    // text answers for, or the chain_rows / chain_base words of the other rows' blocks; then their chain_tab entries
    uint64_t e_sa[4];
    bool live[4], good[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const uint32_t i = 4u * tid + (uint32_t)kk;
      live[kk] = i < m;
      e_sa[kk] = ix.sa_full[live[kk] ? lo + i : lo];
    }
    const uint4* a2[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const uint64_t row = lo + 4u * tid + (uint32_t)kk;
      const uint64_t p = e_sa[kk] & kPosMask;
      const uint64_t need = p + 1 < ix.interval ? p + 1 : ix.interval;
      good[kk] = e_sa[kk] != kNoPosition && (e_sa[kk] >> kPosBits) >= need;
      a2[kk] = good[kk] ? reinterpret_cast<const uint4*>(ix.textrc) + ((m_text - p) >> 6)
                        : reinterpret_cast<const uint4*>(&ix.blocks2[(live[kk] ? row : lo) >> 6].chain_rows);
    }
    uint4 w0[4], w1[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { w0[kk] = a2[kk][0]; w1[kk] = a2[kk][1]; }  // (a block's reserved words behind chain_base: in bounds)
    const uint4* a3[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const uint64_t row = lo + 4u * tid + (uint32_t)kk;
      const uint64_t rows = u64_of(w0[kk].x, w0[kk].y), base = u64_of(w0[kk].z, w0[kk].w);
      const uint32_t o = (uint32_t)(row & 63);
      const bool have = live[kk] && !good[kk] && ((rows >> o) & 1ull);
      if (live[kk] && !good[kk] && !have) fail = true;
      a3[kk] = reinterpret_cast<const uint4*>(ix.chain_tab + (have ? base + (uint64_t)__popcll(rows & below(o)) : 0ull));
      good[kk] = good[kk] && live[kk];
      live[kk] = have;  // from here on: the row has a chain_tab entry
    }
    uint4 q0[4], q1[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { q0[kk] = a3[kk][0]; q1[kk] = a3[kk][1]; }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const uint32_t i = 4u * tid + (uint32_t)kk;
      full[kk] = make_ulonglong2(0, 0);
      ev0[kk] = 0; meta[kk] = 0; n_ev[kk] = 0; len[kk] = 0;
      if (good[kk]) {
        // the text is the chain: bases text[p - 1], text[p - 2], ... = 3 - textrc[x0 ..], one event at depth p mod interval
        const uint64_t p = e_sa[kk] & kPosMask;
        const uint64_t x0 = m_text - p;
        uint64_t c0, c1;
        sm_funnel128(u64_of(w0[kk].x, w0[kk].y), u64_of(w0[kk].z, w0[kk].w), u64_of(w1[kk].x, w1[kk].y), u64_of(w1[kk].z, w1[kk].w),
                     (uint32_t)(x0 & 63), c0, c1);
        const uint32_t ln = p < depth_max ? (uint32_t)p : depth_max;
        full[kk] = prefix_of(make_ulonglong2(rev_bases(~c0), rev_bases(~c1)), ln);
        len[kk] = ln;
        n_ev[kk] = 1;
        ev0[kk] = ((uint64_t)(uint32_t)(p % ix.interval) << 58) | p;
      } else if (live[kk]) {
        full[kk] = make_ulonglong2(u64_of(q0[kk].x, q0[kk].y), u64_of(q0[kk].z, q0[kk].w));
        meta[kk] = u64_of(q1[kk].x, q1[kk].y);
        ev0[kk] = u64_of(q1[kk].z, q1[kk].w);
        n_ev[kk] = (uint32_t)meta[kk] & 0xFFu;
        len[kk] = (uint32_t)(meta[kk] >> 8) & 0xFFu;
      }
      if (i < m) {
        kb[i] = full[kk];
        len_of[i] = (uint8_t)len[kk];
      }
    }
    if (fail) cnt[0] = 1u;
    team_sync<T>();
    SBWT_PHASE_MEM(0);
    // ---- a node of one sampled row has no children: an event at depth d >= 1 in a node no other chain passes through
    // ends its chain (only chains with several events can lose any)
    // The rows with several events go into a list, a thread per listed row: the rule is a chain of dependent LDS reads per
    // event (the neighbours, nearest first), and a thread that walks through the events of its own four rows one after
    // the other made the team wait 15,000 clocks a read for its slowest wave -- for 16 events a read.  A list entry =
    // row | events << 10 | the depths of the events the rule looks at (all but the last; six bits each, eight at most).
    {
      uint64_t ev1[4], ev2[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {  // (the second and third event of the thread's four rows: one round of loads)
        const uint64_t* evs = ix.chain_ev + (meta[kk] >> kChainEvOffShift);
        ev1[kk] = 0; ev2[kk] = 0;
        if (n_ev[kk] >= 3u) ev1[kk] = evs[0];
        if (n_ev[kk] >= 4u) ev2[kk] = evs[1];
      }
      uint32_t listed = 0;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) listed += n_ev[kk] >= 2u ? 1u : 0u;
      uint32_t n_listed;
      uint32_t at = team_excl_scan(listed, &n_listed);
      bool too_many = false;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (n_ev[kk] < 2u) continue;
        const uint32_t i = 4u * tid + (uint32_t)kk;
        const uint64_t* evs = ix.chain_ev + (meta[kk] >> kChainEvOffShift);
        uint64_t word = (uint64_t)i | ((uint64_t)n_ev[kk] << 10);
        if (n_ev[kk] > 9u) { too_many = true; word = (uint64_t)i | (1ull << 10); }  // (the level-by-level kernel takes the read)
        else
          for (uint32_t j = 0; j + 1 < n_ev[kk]; ++j) {
            const uint64_t e = j == 0 ? ev0[kk] : j == 1 ? ev1[kk] : j == 2 ? ev2[kk] : evs[j - 1];
            word |= (e >> 58) << (16u + 6u * j);
          }
        pay[at++] = word;  // (pay and acc_of are not in use before the entries are made)
      }
      if (too_many) cnt[0] = 1u;
      team_sync<T>();
      SBWT_PHASE_MEM(5);
      // eight threads per listed row: the sixteen nearest neighbours in one round of LDS reads, the next sixteen in the
      // next (under load an LDS read takes hundreds of clocks: the neighbours one after the other were what was slow)
      const uint32_t sub = tid & 7u;
      for (uint32_t q2 = tid >> 3; q2 < n_listed; q2 += T >> 3) {
        const uint64_t word = pay[q2];
        const uint32_t i = (uint32_t)word & 1023u, n = (uint32_t)(word >> 10) & 63u;
        const ulonglong2 mine = kb[i];
        uint32_t keep = n;
        for (uint32_t j = 0; j + 1 < n; ++j) {
          const uint32_t d = (uint32_t)(word >> (16u + 6u * j)) & 63u;
          if (d == 0) continue;  // (level 0, searcher.hpp:424-443, has no early return)
          const ulonglong2 w = prefix_of(mine, d);
          // (on an index that is k-ordered as far as the walk reaches there IS another chain through the node whenever this one
          // has a later event -- a lone row would be a unique occurrence from here on -- and it is a row nearby: the copies
          // the chain hops between share their right contexts too.  So: the neighbours first, and out at the first sharer.)
          bool alone = true;
          for (uint32_t first = 1; first < m && alone; first += 8u) {
            const uint32_t dl = first + sub;
            uint32_t share = 0;
            if (dl < m) {
              const uint32_t o1 = i + dl;
              if (o1 < m && len_of[o1] >= d) {
                const ulonglong2 v = prefix_of(kb[o1], d);
                share = v.x == w.x && v.y == w.y ? 1u : 0u;
              }
              if (!share && i >= dl && len_of[i - dl] >= d) {
                const ulonglong2 v = prefix_of(kb[i - dl], d);
                share = v.x == w.x && v.y == w.y ? 1u : 0u;
              }
            }
            share |= (uint32_t)__shfl_xor((int)share, 1);
            share |= (uint32_t)__shfl_xor((int)share, 2);
            share |= (uint32_t)__shfl_xor((int)share, 4);
            alone = share == 0;
          }
          if (alone) { keep = j + 1; break; }
        }
        if (sub == 0) acc_of[i] = (uint16_t)keep;
      }
      team_sync<T>();
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        if (n_ev[kk] >= 2u) n_ev[kk] = acc_of[4u * tid + (uint32_t)kk];
      team_sync<T>();  // (pay and acc_of are written again below)
    }
    SBWT_PHASE_MEM(1);
    // ---- entries: numbered thread by thread, i.e. in row order
    uint32_t H;
    uint32_t slot = team_excl_scan(n_ev[0] + n_ev[1] + n_ev[2] + n_ev[3], &H);
    const bool failed = cnt[0] != 0 || H > kCap;  // (team-uniform: cnt[0] was written before the sync above)
    if (failed) continue;
    // ---- the event word of every entry, and its row: written by the thread that holds the row ...
    uint16_t* const row_of = acc_of;  // (acc_of is not in use before the sort)
    constexpr uint64_t kEvIndirect = 1ull << 57;  // (an event word is depth << 58 | position: bit 57 is free)
    {
      uint32_t at = slot;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const uint32_t i = 4u * tid + (uint32_t)kk, n = n_ev[kk];
        if (n) { pay[at] = ev0[kk]; row_of[at] = (uint16_t)i; }
        // (a further event: where it lies in chain_ev; whoever gets the entry below fetches it)
        for (uint32_t j = 1; j < n; ++j) { pay[at + j] = kEvIndirect | ((meta[kk] >> kChainEvOffShift) + j - 1u); row_of[at + j] = (uint16_t)i; }
        at += n;
      }
    }
    team_sync<T>();
    // ... and turned into keys and filtered hits by ALL threads, four entries each (H <= kCap = 4 T): a row with many
    // events costs its thread a few stores above, not a filter run per event.  The filters (push_position,
    // searcher.hpp:504-539): the table searches of a thread's four entries run interleaved
    uint64_t or0 = 0, and0 = ~0ull, or1 = 0, and1 = ~0ull;
    bool wrapped = false;
    {
      uint64_t ev[4], fpos[4];
      ulonglong2 fullk[4];
      uint32_t c1[4], n1[4];
      bool frc[4], fgo[4], have[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const uint32_t k = tid + (uint32_t)kk * T;
        have[kk] = k < H;
        ev[kk] = have[kk] ? pay[k] : 0ull;
        fullk[kk] = kb[have[kk] ? row_of[k] : 0u];
      }
      {
        uint64_t fetched[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) fetched[kk] = ix.chain_ev[(ev[kk] & kEvIndirect) ? ev[kk] & (kEvIndirect - 1ull) : 0ull];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) ev[kk] = (ev[kk] & kEvIndirect) ? fetched[kk] : ev[kk];
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        fpos[kk] = ev[kk] & kPosMask;
        frc[kk] = false;
        fgo[kk] = have[kk] && push_position_orient(ix, fpos[kk], L, a.strand, frc[kk]);
        if (!fgo[kk]) fpos[kk] = 0;
      }
      upper_bound_minus1_x4(t_chr, ix.n_chr, fpos, c1);
      upper_bound_minus1_x4(t_nk, ix.n_npos, fpos, n1);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (!have[kk]) continue;
        const uint32_t k = tid + (uint32_t)kk * T;
        sbwt_hit h;
        const bool ok = fgo[kk] && push_position_finish(ix, t_chr, t_nk, t_nv, fpos[kk], L, frc[kk], c1[kk], n1[kk], &h);
        const uint32_t d = (uint32_t)(ev[kk] >> 58);
        const ulonglong2 pre = prefix_of(fullk[kk], d);
        const ulonglong2 key = sort_key(pre.x, pre.y, d);
        uint64_t py = (uint64_t)d << 58;
        if (ok) {
          if (h.pos0 > kPosMask) wrapped = true;  // (an upstream-accepted wrapped position: k_multi writes whole hits)
          py |= kPayOk | (h.pos0 & kPosMask) | ((uint64_t)h.chr_id << kPosBits) | ((uint64_t)(h.is_rc & 1u) << 56);
        }
        ka[k] = key;
        pay[k] = py;
        or0 |= key.x; and0 &= key.x; or1 |= key.y; and1 &= key.y;
      }
    }
    if (wrapped) cnt[0] = 1u;
    team_sync<T>();
    SBWT_PHASE_MEM(2);
    const bool failed2 = cnt[0] != 0;
    team_sync<T>();  // (nobody writes cnt[0] again before every thread has read it)
    if (failed2) continue;
    bool radix = false;
    const SortedRows sorted = team_sort_rows<T>(sort_lds, H, tid, lane, or0, and0, or1, and1, radix);
    SBWT_PHASE(3);
    // ---- accepted hits before every place, node boundaries, the cut (as k_multi_bfs)
    uint32_t base = 0;
    for (uint32_t c = 0; c < H; c += T) {
      const uint32_t k = c + tid;
      bool ok = false;
      if (k < H) ok = (pay[sorted.ord[k]] & kPayOk) != 0;
      uint32_t tot;
      const uint32_t off = base + team_excl_scan(ok ? 1u : 0u, &tot);
      base += tot;
      if (k < H) acc_of[k] = (uint16_t)off;  // accepted hits before place k
    }
    if (tid == 0) cnt[1] = H;
    team_sync<T>();
    const uint64_t maybe = (uint64_t)(int64_t)(int32_t)m;  // (searcher.hpp:412,464: an int against a size_t)
    for (uint32_t k = tid + 1; k < H; k += T) {
      const ulonglong2 qa = sorted_key(sorted, sort_lds, k - 1), qb = sorted_key(sorted, sort_lds, k);
      const bool boundary = qa.x != qb.x || qa.y != qb.y;  // (the depth is part of the key)
      if (boundary && (uint64_t)acc_of[k] == maybe) atomicMin(&cnt[1], k);
    }
    team_sync<T>();
    const uint32_t cut = cnt[1];
    const uint32_t nres = cut < H ? (uint32_t)acc_of[cut] : base;
    uint64_t first = res.first_hit;
    uint32_t room = m;
    if (nres > m) {  // more hits than rows (extra_slots): the list goes behind everything else
      if (tid == 0) s_extra[team] = list_fits_record(a.io, r, nres) ? extra_slots(a.hit_count, a.overflow_flag, a.multi_base, a.hits_cap, nres) : ~0ull;
      team_sync<T>();
      if (s_extra[team] != ~0ull) { first = s_extra[team]; room = nres; }
    }
    sbwt_hit* out = a.hits + first;
    for (uint32_t k = tid; k < cut; k += T) {
      const uint64_t py = pay[sorted.ord[k]];
      if ((py & kPayOk) && (uint32_t)acc_of[k] < room) {
        sbwt_hit h;
        h.pos0 = py & kPosMask;
        h.chr_id = (uint32_t)(py >> kPosBits) & 0xFFFFu;
        h.is_rc = (uint32_t)(py >> 56) & 1u;
        out[acc_of[k]] = h;
      }
    }
    if (tid == 0) {
      rio_store_multi(a.io, r, first, true, (int32_t)(nres < room ? nres : room), nres > room ? SBWT_RF_TRUNCATED : 0u);
      queue[q] = kQueueDone;
      rows_done += m;
      ++reads_done;
    }
    SBWT_PHASE_MEM(4);
#ifdef SBWT_PHASE_CLOCKS
    if (tid == 0) { s_phase[team][10] += 1; s_phase[team][9] += H; s_phase[team][11] += radix ? 1 : 0; }
#endif
  }
#ifdef SBWT_PHASE_CLOCKS
  team_sync<T>();
  if (tid < 12 && T == 256) atomicAdd(&g_multi_phase[tid], s_phase[team][tid]);
#endif
  if (tid == 0 && reads_done) {
    atomicAdd(chain_count, reads_done);
    atomicAdd(a.rows_count, rows_done);
  }
}

}  // namespace sbwt

// multi_locate.h -- k_multi_sort: locate of multi-hit reads without the depth-first walk.
//
// Upstream (searcher.hpp:424-503) reports the hits of an interval [lo, hi) by extending the pattern to the left one
// base at a time, depth first in A<C<G<T order, and testing every row of every child interval for a sampled SA
// value.  Row i of the interval surfaces exactly once: at the depth t_i where its LF chain reaches a sampled row,
// with position SA = sa_full[i], after the bases (text[p-1], text[p-2], ..., text[p-t_i]) have been prepended.
// The emission order is therefore: rows sorted by that prepended string in pre-order (a string before its
// extensions, A<C<G<T), rows with the same string in row order.  Where the tables are LF-consistent over the
// whole depth of the walk (run >= min(interval, p + 1), device_index.hip) this kernel produces exactly that:
// gather sa_full[lo..hi) (contiguous), read the up to 63 preceding bases of each hit from the text, sort, filter
// (push_position, searcher.hpp:504-539), write.  One wave per read up to 256 rows, one workgroup above.  Reads with a row outside such a region, with more than
// kSortMax rows, or on an index sampled wider than 64 go to the depth-first kernel (k_multi) instead.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sbwt_amd.h"
#include "rank_device.h"
#include "search_machine.h"

namespace sbwt {

constexpr uint32_t kSortMax = 1024;

struct MultiSortArgs {
  const uint32_t* lens;
  int strand;
  sbwt_read_result* results;
  sbwt_hit* hits;
  uint64_t n_reads;
  const uint32_t* queue;                 // wave w of the search kernel queued its reads at queue[w * chunk ..)
  const WaveInfo* wave_info;             // after k_multi_plan: prefix sums over waves
  uint32_t n_waves;
  uint64_t chunk;
  const unsigned long long* queue_count; // total number of queued reads (k_multi_plan)
  const unsigned int* overflow;          // set by k_multi_plan when the hit slots do not fit hits_cap
  uint32_t* dfs_queue;                   // reads this kernel hands on to k_multi
  unsigned long long* dfs_count;
};

// Turns the per-wave totals of k_search_sm into global offsets: exclusive prefix sums of the hit slots and of the
// queued reads over waves, the grand totals, and the capacity verdict.  One workgroup.
__global__ __launch_bounds__(1024) void k_multi_plan(WaveInfo* __restrict__ wi, uint32_t n_waves, uint64_t n_reads,
                                                     uint64_t hits_cap, unsigned long long* total_slots,
                                                     unsigned long long* total_count, unsigned int* overflow) {
  __shared__ uint64_t s_slots[1024];
  __shared__ uint32_t s_count[1024];
  const uint32_t per = (n_waves + 1023u) / 1024u;
  const uint32_t a = threadIdx.x * per < n_waves ? threadIdx.x * per : n_waves;
  const uint32_t b = a + per < n_waves ? a + per : n_waves;
  uint64_t slots = 0;
  uint32_t count = 0;
  for (uint32_t w = a; w < b; ++w) { slots += wi[w].slots; count += wi[w].count; }
  s_slots[threadIdx.x] = slots;
  s_count[threadIdx.x] = count;
  __syncthreads();
  uint64_t ps = 0;
  uint32_t pc = 0;
  for (uint32_t t = 0; t < threadIdx.x; ++t) { ps += s_slots[t]; pc += s_count[t]; }
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
    if (n_reads + ps > hits_cap) *overflow = 1u;
  }
}

// 2-bit groups of x in reverse order (first base most significant)
__device__ __forceinline__ uint64_t rev_bases(uint64_t x) {
  x = __brevll(x);
  return ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
}

__device__ __forceinline__ bool key_less(uint64_t a0, uint64_t a1, uint32_t a2, uint64_t b0, uint64_t b1, uint32_t b2) {
  if (a0 != b0) return a0 < b0;
  if (a1 != b1) return a1 < b1;
  return a2 < b2;
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

template <int T>
__global__ __launch_bounds__(256) void k_multi_sort(const DeviceIndex ix, const MultiSortArgs a) {
  constexpr uint32_t kTeams = 256 / T;
  constexpr uint32_t kCap = kSortMax / kTeams;  // rows a team can sort
  __shared__ uint64_t s_k0[kSortMax], s_k1[kSortMax];
  __shared__ uint32_t s_k2[kSortMax];
  __shared__ uint32_t s_scan_all[4];
  __shared__ uint32_t s_bad_all[4];
  if (*a.overflow) return;  // the hit lists do not fit hits[]: the batch fails with SBWT_ERR_CAPACITY
  const uint32_t team = threadIdx.x / T, tid = threadIdx.x % T;
  uint64_t* k0 = s_k0 + team * kCap;
  uint64_t* k1 = s_k1 + team * kCap;
  uint32_t* k2 = s_k2 + team * kCap;
  uint32_t* s_bad = s_bad_all + team;
  const uint64_t m_text = ix.n - 1;
  // teams stride over the waves of the search kernel; each wave queued its reads at queue[w * chunk ..)
  for (uint32_t w = blockIdx.x * kTeams + team; w < a.n_waves; w += gridDim.x * kTeams) {
    const WaveInfo wi = a.wave_info[w];
    for (uint32_t qi = 0; qi < wi.count; ++qi) {
      const uint32_t r = a.queue[(uint64_t)w * a.chunk + qi];
      sbwt_read_result res = a.results[r];
      const uint64_t lo = res.sa_lo, m = res.sa_hi - res.sa_lo;
      // which launch takes this read
      if (T == 64 ? m > kCap : m <= kSortMax / 4) continue;
      res.first_hit += a.n_reads + wi.slots;  // wave-relative until now
      if (tid == 0) *s_bad = (m > kCap || ix.interval > 64) ? 1u : 0u;
      team_sync<T>();
      uint32_t N = 64;
      while (N < m && N < kCap) N <<= 1;
      if (!*s_bad) {
        // keys: the prepended string of every row (first base most significant, zero padded), then its length
        // and the row's offset in the interval
        for (uint32_t i = tid; i < N; i += T) {
          uint64_t h0 = ~0ull, h1 = ~0ull;
          uint32_t h2 = ~0u;
          if (i < m) {
            const uint64_t e = ix.sa_full[lo + i];
            const uint64_t p = e & kPosMask, run = e >> kPosBits;
            const uint32_t t = (uint32_t)(p % ix.interval);
            const uint64_t need = p + 1 < ix.interval ? p + 1 : ix.interval;
            if (e == kNoPosition || run < need) {
              *s_bad = 1u;  // benign race: every writer stores 1
            } else {
              // text[p - 1 - k] = 3 - textrc[x0 + k]
              const uint64_t x0 = m_text - p;
              const uint64_t wi0 = x0 >> 4;
              const uint32_t sh = 2u * (uint32_t)(x0 & 15);
              const uint32_t* tx = ix.textrc + wi0;
              const uint64_t w0 = tx[0] | ((uint64_t)tx[1] << 32), w1 = tx[2] | ((uint64_t)tx[3] << 32), w2 = tx[4];
              uint64_t c0 = sh ? ((w0 >> sh) | (w1 << (64 - sh))) : w0;
              uint64_t c1 = sh ? ((w1 >> sh) | (w2 << (64 - sh))) : w1;
              c0 = ~c0;
              c1 = ~c1;
              if (t < 32u) { c0 &= t ? ((1ull << (2 * t)) - 1ull) : 0ull; c1 = 0; }
              else if (t < 64u) c1 &= t > 32u ? ((1ull << (2 * (t - 32u))) - 1ull) : 0ull;
              h0 = rev_bases(c0);
              h1 = rev_bases(c1);
              h2 = (t << 16) | i;
            }
          }
          k0[i] = h0; k1[i] = h1; k2[i] = h2;
        }
      }
      team_sync<T>();
      if (*s_bad) {
        if (tid == 0) {
          a.results[r].first_hit = res.first_hit;
          a.dfs_queue[atomicAdd(a.dfs_count, 1ull)] = r;
        }
        team_sync<T>();
        continue;
      }
      // bitonic sort of N keys
      for (uint32_t size = 2; size <= N; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
          for (uint32_t t = tid; t < (N >> 1); t += T) {
            const uint32_t i = 2 * t - (t & (stride - 1));  // lower index of the pair
            const uint32_t j = i + stride;
            const bool up = (i & size) == 0;
            const uint64_t a0 = k0[i], a1 = k1[i], b0 = k0[j], b1 = k1[j];
            const uint32_t a2 = k2[i], b2 = k2[j];
            if (key_less(b0, b1, b2, a0, a1, a2) == up) {
              k0[i] = b0; k1[i] = b1; k2[i] = b2;
              k0[j] = a0; k1[j] = a1; k2[j] = a2;
            }
          }
          team_sync<T>();
        }
      }
      // filter in emission order, compact, write
      const uint32_t L = a.lens[r];
      sbwt_hit* out = a.hits + res.first_hit;
      uint32_t base = 0;
      for (uint32_t c = 0; c < m; c += T) {
        const uint32_t k = c + tid;
        bool ok = false;
        sbwt_hit h;
        if (k < m) {
          const uint64_t e = ix.sa_full[lo + (k2[k] & 0xffffu)];
          ok = push_position_lane(ix, e & kPosMask, L, a.strand, &h);
        }
        const uint64_t bal = __ballot(ok);
        const uint32_t lane = threadIdx.x & 63;
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
      if (tid == 0) {
        a.results[r].first_hit = res.first_hit;
        a.results[r].nh = (int32_t)base;
      }
      team_sync<T>();
    }
  }
}

}  // namespace sbwt

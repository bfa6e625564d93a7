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
// (push_position, searcher.hpp:504-539), write.  Reads with a row outside such a region, with more than
// kSortMax rows, or on an index sampled wider than 64 go to the depth-first kernel (k_multi) instead.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sbwt_amd.h"
#include "rank_device.h"

namespace sbwt {

constexpr uint32_t kSortMax = 1024;

struct MultiSortArgs {
  const uint32_t* lens;
  int strand;
  sbwt_read_result* results;
  sbwt_hit* hits;
  const uint32_t* queue;                 // reads queued by the search kernel
  const unsigned long long* queue_count;
  uint32_t* dfs_queue;                   // reads this kernel hands on to k_multi
  unsigned long long* dfs_count;
};

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

__global__ __launch_bounds__(256) void k_multi_sort(const DeviceIndex ix, const MultiSortArgs a) {
  __shared__ uint64_t k0[kSortMax], k1[kSortMax];
  __shared__ uint32_t k2[kSortMax];
  __shared__ uint32_t s_scan[4];
  __shared__ uint32_t s_bad;
  const uint64_t qn = *a.queue_count;
  const uint64_t m_text = ix.n - 1;
  for (uint64_t q = blockIdx.x; q < qn; q += gridDim.x) {
    const uint32_t r = a.queue[q];
    const sbwt_read_result res = a.results[r];
    const uint64_t lo = res.sa_lo, m = res.sa_hi - res.sa_lo;
    if (threadIdx.x == 0) s_bad = (m > kSortMax || ix.interval > 64) ? 1u : 0u;
    __syncthreads();
    uint32_t N = 64;
    while (N < m && N < kSortMax) N <<= 1;
    if (!s_bad) {
      // keys: the prepended string of every row (first base most significant, zero padded), then its length and
      // the row's offset in the interval
      for (uint32_t i = threadIdx.x; i < N; i += blockDim.x) {
        uint64_t h0 = ~0ull, h1 = ~0ull;
        uint32_t h2 = ~0u;
        if (i < m) {
          const uint64_t e = ix.sa_full[lo + i];
          const uint64_t p = e & kPosMask, run = e >> kPosBits;
          const uint32_t t = (uint32_t)(p % ix.interval);
          const uint64_t need = p + 1 < ix.interval ? p + 1 : ix.interval;
          if (e == kNoPosition || run < need) {
            s_bad = 1u;  // benign race: every writer stores 1
          } else {
            // text[p - 1 - k] = 3 - textrc[x0 + k]
            const uint64_t x0 = m_text - p;
            const uint64_t wi = x0 >> 4;
            const uint32_t sh = 2u * (uint32_t)(x0 & 15);
            const uint32_t* tx = ix.textrc + wi;
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
    __syncthreads();
    if (s_bad) {
      if (threadIdx.x == 0) a.dfs_queue[atomicAdd(a.dfs_count, 1ull)] = r;
      __syncthreads();
      continue;
    }
    // bitonic sort of N keys
    for (uint32_t size = 2; size <= N; size <<= 1) {
      for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
        for (uint32_t t = threadIdx.x; t < (N >> 1); t += blockDim.x) {
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
        __syncthreads();
      }
    }
    // filter in emission order, compact, write
    const uint32_t L = a.lens[r];
    sbwt_hit* out = a.hits + res.first_hit;
    uint32_t base = 0;
    for (uint32_t c = 0; c < m; c += blockDim.x) {
      const uint32_t k = c + threadIdx.x;
      bool ok = false;
      sbwt_hit h;
      if (k < m) {
        const uint64_t e = ix.sa_full[lo + (k2[k] & 0xffffu)];
        ok = push_position_lane(ix, e & kPosMask, L, a.strand, &h);
      }
      const uint64_t bal = __ballot(ok);
      const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      if (lane == 0) s_scan[wave] = (uint32_t)__popcll(bal);
      __syncthreads();
      uint32_t off = base;
      for (uint32_t w = 0; w < wave; ++w) off += s_scan[w];
      if (ok) out[off + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = h;
      base += s_scan[0] + s_scan[1] + s_scan[2] + s_scan[3];
      __syncthreads();
    }
    if (threadIdx.x == 0) a.results[r].nh = (int32_t)base;
    __syncthreads();
  }
}

}  // namespace sbwt

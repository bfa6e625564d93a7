// device_index.hip -- load-time re-layout of the reference-format tables into the HBM layout of
// device_layout.h.  One-time work per index; correctness first, all kernels are plain gfx950 HIP.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <unordered_map>

#include "index_impl.hpp"
#include "rank_device.h"

namespace sbwt {

constexpr int kChunk = 256;  // index blocks per scan chunk = threads per workgroup

struct LayoutErrors {
  unsigned long long loc_fbwt_mismatch;  // location_table key whose fbwt is not 0, or the converse
  unsigned long long loc_not_sorted;
  unsigned long long occ_mismatch;       // occ_function checkpoint differs from the BWT it summarises
};

// ---- blocks1: planes + sampled bitmap ------------------------------------------------------
__global__ __launch_bounds__(256) void k_planes1(Block1* __restrict__ blocks, uint64_t n_blocks, uint64_t n,
                                                 const uint8_t* __restrict__ jbwt, uint64_t jbwt_len,
                                                 const uint8_t* __restrict__ fbwt, const uint64_t* __restrict__ loc,
                                                 uint64_t n_loc, LayoutErrors* err) {
  const uint64_t b = global_thread();
  if (b >= n_blocks) return;
  uint64_t plo = 0, phi = 0;
  for (int k = 0; k < 16; ++k) {
    const uint64_t byte_idx = b * 16 + k;
    const uint32_t byte = byte_idx < jbwt_len ? jbwt[byte_idx] : 0u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // base q of a byte sits at bits 7-2q .. 6-2q (jbit.hpp:193-206)
      const uint32_t code = (byte >> (6 - 2 * q)) & 3u;
      const int j = 4 * k + q;
      plo |= (uint64_t)(code & 1u) << j;
      phi |= (uint64_t)(code >> 1) << j;
    }
  }
  // sampled bitmap: position i is sampled iff fbwt[i] == 0 and i is a key of location_table
  const uint64_t p0 = b * 64;
  uint64_t l = 0, h = n_loc;
  while (l < h) {
    const uint64_t m = (l + h) >> 1;
    if (loc[2 * m] < p0) l = m + 1; else h = m;
  }
  uint64_t sampled = 0;
  unsigned long long bad = 0;
  for (uint64_t k = l; k < n_loc; ++k) {
    const uint64_t i = loc[2 * k];
    if (i >= p0 + 64) break;
    if (i < n && fbwt[i] == 0) sampled |= 1ull << (i - p0); else ++bad;
  }
  uint64_t zeros = 0;
  for (int j = 0; j < 64; ++j)
    if (p0 + j < n && fbwt[p0 + j] == 0) zeros |= 1ull << j;
  if (zeros != sampled) bad += (unsigned long long)__popcll(zeros ^ sampled);
  if (bad) atomicAdd(&err->loc_fbwt_mismatch, bad);
  Block1 out;
  out.cnt[0] = out.cnt[1] = out.cnt[2] = out.cnt[3] = 0;
  out.plane_lo = plo;
  out.plane_hi = phi;
  out.sampled = sampled;
  out.sampled_base = l;
  blocks[b] = out;
}

__global__ __launch_bounds__(256) void k_loc_vals(const uint64_t* __restrict__ loc, uint64_t n_loc, uint64_t n,
                                                  uint64_t* __restrict__ loc_vals, LayoutErrors* err) {
  const uint64_t k = global_thread();
  if (k >= n_loc) return;
  loc_vals[k] = loc[2 * k + 1];
  if (loc[2 * k] >= n || (k + 1 < n_loc && loc[2 * k] >= loc[2 * k + 2])) atomicAdd(&err->loc_not_sorted, 1ull);
}

// ---- chunked exclusive scan of K per-block counters ------------------------------------------
// exclusive scan of one u32 per thread over a 256-thread workgroup; returns the exclusive prefix and
// leaves the workgroup total in *total.  lds must hold 4 u32 and be distinct per concurrent call site.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* lds, uint32_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(inc, d);
    if (lane >= d) inc += t;
  }
  __syncthreads();
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  uint32_t wave_off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const uint32_t s = lds[w];
    if (w < wave) wave_off += s;
    tot += s;
  }
  *total = tot;
  return wave_off + inc - v;
}

template <int K>
struct Counts { uint32_t c[K]; };

__device__ __forceinline__ Counts<4> counts_of(const Block1& b) {
  Counts<4> r;
#pragma unroll
  for (int c = 0; c < 4; ++c) r.c[c] = (uint32_t)__popcll(match1(b.plane_lo, b.plane_hi, (uint32_t)c));
  return r;
}
__device__ __forceinline__ uint64_t match2(const uint64_t* plane, uint32_t code) {
  uint64_t m = ~0ull;
#pragma unroll
  for (int k = 0; k < 4; ++k) m &= ((code >> k) & 1u) ? plane[k] : ~plane[k];
  return m;
}
__device__ __forceinline__ Counts<16> counts_of(const Block2& b) {
  Counts<16> r;
#pragma unroll
  for (int c = 0; c < 16; ++c) r.c[c] = (uint32_t)__popcll(match2(b.plane, (uint32_t)c));
  return r;
}

// totals[chunk][K] = counts of one chunk of 256 blocks
template <int K, typename B>
__global__ __launch_bounds__(256) void k_chunk_totals(const B* __restrict__ blocks, uint64_t n_blocks,
                                                      uint64_t* __restrict__ totals) {
  __shared__ uint32_t lds[4];
  const uint64_t b = (uint64_t)blockIdx.x * kChunk + threadIdx.x;
  Counts<K> cnt;
  if (b < n_blocks) cnt = counts_of(blocks[b]);
  else
    for (int k = 0; k < K; ++k) cnt.c[k] = 0;
  for (int k = 0; k < K; ++k) {
    uint32_t tot;
    block_excl_scan(cnt.c[k], lds, &tot);
    if (threadIdx.x == 0) totals[(uint64_t)blockIdx.x * K + k] = tot;
    __syncthreads();
  }
}

// in place: totals[chunk][k] becomes the exclusive prefix over chunks (one thread per counter)
__global__ void k_scan_chunks(uint64_t* totals, uint64_t n_chunks, int K) {
  const int k = threadIdx.x;
  if (k >= K) return;
  uint64_t acc = 0;
  for (uint64_t c = 0; c < n_chunks; ++c) {
    const uint64_t t = totals[c * K + k];
    totals[c * K + k] = acc;
    acc += t;
  }
}

__global__ __launch_bounds__(256) void k_apply1(Block1* __restrict__ blocks, uint64_t n_blocks,
                                                const uint64_t* __restrict__ chunk_base, uint64_t c0, uint64_t c1,
                                                uint64_t c2, uint64_t c3, uint64_t interval, const uint64_t* occA,
                                                const uint64_t* occC, const uint64_t* occG, const uint64_t* occT,
                                                uint64_t occ_len, LayoutErrors* err) {
  __shared__ uint32_t lds[4];
  const uint64_t b = (uint64_t)blockIdx.x * kChunk + threadIdx.x;
  const bool live = b < n_blocks;
  Counts<4> cnt;
  if (live) cnt = counts_of(blocks[b]);
  else
    for (int k = 0; k < 4; ++k) cnt.c[k] = 0;
  const uint64_t C[4] = {c0, c1, c2, c3};
  const uint64_t* occ[4] = {occA, occC, occG, occT};
  for (int k = 0; k < 4; ++k) {
    uint32_t tot;
    const uint32_t ex = block_excl_scan(cnt.c[k], lds, &tot);
    __syncthreads();
    if (!live) continue;
    const uint64_t raw = chunk_base[(uint64_t)blockIdx.x * 4 + k] + ex;  // #k in bwt[0 .. 64b), '$' as 'A'
    blocks[b].cnt[k] = C[k] + raw;
    // the file's own checkpoints must agree with the BWT they summarise (same values the upstream rank reads)
    const uint64_t pos = b * 64;
    if (occ[k] && pos % interval == 0 && pos / interval < occ_len && occ[k][pos / interval] != raw)
      atomicAdd(&err->occ_mismatch, 1ull);
  }
}

// ---- sa_full: memoised unique-hit locate ---------------------------------------------------------
// sa_full[i] = what the upstream locate reports for the interval [i, i+1): level-0 test
// (searcher.hpp:424-443) then find_possible (searcher.hpp:460-503) on a size-1 interval, i.e. follow LF
// until a sampled position; kNoPosition if depth `interval` or the '$' row is reached first.
__device__ __forceinline__ uint64_t upstream_walk(const DeviceIndex& ix, uint64_t i) {
  uint64_t cur = i, t = 0;
  for (;;) {
    const Block1* b = ix.blocks1 + (cur >> 6);
    const uint4 pl = *reinterpret_cast<const uint4*>(&b->plane_lo);
    const uint4 sm = *reinterpret_cast<const uint4*>(&b->sampled);
    const uint32_t o = (uint32_t)(cur & 63);
    const uint64_t sampled = ((uint64_t)sm.y << 32) | sm.x, base = ((uint64_t)sm.w << 32) | sm.z;
    if ((sampled >> o) & 1ull) return ix.loc_vals[base + (uint64_t)__popcll(sampled & below(o))] + t;
    if (t + 1 >= ix.interval) return kNoPosition;      // find_possible: len >= interval
    if (cur == ix.first_location) return kNoPosition;  // '$' row: every child interval is empty
    const uint64_t plo = ((uint64_t)pl.y << 32) | pl.x, phi = ((uint64_t)pl.w << 32) | pl.z;
    const uint32_t c = (uint32_t)((plo >> o) & 1ull) | ((uint32_t)((phi >> o) & 1ull) << 1);
    uint64_t r = b->cnt[c] + (uint64_t)__popcll(match1(plo, phi, c) & below(o));
    if (c == 0 && cur > ix.first_location) r -= 1;
    cur = r;
    ++t;
  }
}

// the definition, row by row: about interval / 2 dependent gathers per row
__global__ __launch_bounds__(256) void k_sa_full(DeviceIndex ix, uint64_t* __restrict__ sa_full) {
  const uint64_t i = global_thread();
  if (i >= ix.n) return;
  sa_full[i] = upstream_walk(ix, i);
}

// The same values with about two gathers per row, for tables that are LF-consistent (every properly built index
// whose ties beyond sort_len do not matter).  The walk of row x ends on the first sampled row below it; seen from
// a sampled row s the rows LF(s), LF^2(s), ... are, on a consistent index, the rows whose walks end on the NEXT
// sampled row with values loc(s) - 1, loc(s) - 2, ...  k_sa_fast assigns those, k_sa_fill walks the few rows no
// chain reached, and k_sa_check accepts the result only if EVERY row satisfies the recurrence that defines the
// walk (a sampled row reports its own value; any other row reports its LF row's value + 1, and is not a multiple
// of the sampling distance, which bounds the walk below `interval` steps).  Otherwise k_sa_full runs.
struct SaCheck {
  unsigned long long unset;   // rows no chain reached (walked one by one)
  unsigned long long failed;  // rows violating the recurrence
};

__global__ __launch_bounds__(256) void k_sa_fast(DeviceIndex ix, const uint64_t* __restrict__ loc, uint64_t n_loc,
                                                 uint64_t* __restrict__ sa_full, SaCheck* chk) {
  const uint64_t k = global_thread();
  if (k >= n_loc) return;
  uint64_t cur = loc[2 * k];
  const uint64_t val = loc[2 * k + 1];
  if (val % ix.interval != 0 || val >= ix.n) {
    atomicAdd(&chk->failed, 1ull);
    return;
  }
  sa_full[cur] = val;
  for (uint64_t t = 1; t < ix.interval && t <= val; ++t) {
    if (cur == ix.first_location) break;
    const Block1* b = ix.blocks1 + (cur >> 6);
    const uint32_t o = (uint32_t)(cur & 63);
    const uint64_t plo = b->plane_lo, phi = b->plane_hi;
    const uint32_t c = (uint32_t)((plo >> o) & 1ull) | ((uint32_t)((phi >> o) & 1ull) << 1);
    uint64_t r = b->cnt[c] + (uint64_t)__popcll(match1(plo, phi, c) & below(o));
    if (c == 0 && cur > ix.first_location) r -= 1;
    cur = r;
    const Block1* nb = ix.blocks1 + (cur >> 6);
    if ((nb->sampled >> (cur & 63)) & 1ull) break;  // a sampled row answers for itself
    sa_full[cur] = val - t;
  }
}

__global__ __launch_bounds__(256) void k_sa_fill(DeviceIndex ix, uint64_t* __restrict__ sa_full, SaCheck* chk) {
  const uint64_t i = global_thread();
  if (i >= ix.n) return;
  if (sa_full[i] != kNoPosition) return;
  atomicAdd(&chk->unset, 1ull);
  sa_full[i] = upstream_walk(ix, i);
}

__global__ __launch_bounds__(256) void k_sa_check(DeviceIndex ix, const uint64_t* __restrict__ sa_full, SaCheck* chk) {
  const uint64_t j = global_thread();
  if (j >= ix.n) return;
  const Block1* b = ix.blocks1 + (j >> 6);
  const uint32_t o = (uint32_t)(j & 63);
  if ((b->sampled >> o) & 1ull) return;  // k_sa_fast wrote its own value
  const uint64_t v = sa_full[j];
  bool ok;
  if (j == ix.first_location) {
    ok = v == kNoPosition;  // an unsampled '$' row reports nothing
  } else {
    const uint64_t plo = b->plane_lo, phi = b->plane_hi;
    const uint32_t c = (uint32_t)((plo >> o) & 1ull) | ((uint32_t)((phi >> o) & 1ull) << 1);
    const uint64_t nx = sa_full[lf1(ix, j, c)];
    ok = v != kNoPosition && nx != kNoPosition && v == nx + 1 && v % ix.interval != 0;
  }
  if (!ok) atomicAdd(&chk->failed, 1ull);
}

// ---- text-jump tables ---------------------------------------------------------------------------------
// A size-1 interval [i, i+1) evolves under exec_exact_match (searcher.hpp:102-105) as the LF chain of row i: base
// c keeps it alive iff bwt[i] == c (and i is not the '$' row), and the next row is LF(i).  Wherever the tables are
// LF-consistent -- the row LF(j) reports the text position of row j minus one, and positions map back to their
// rows one to one -- r such steps are: compare the next r bases with the text implied by the tables, and land on
// row_tab[p - r].  k_verify_rows finds the rows where that holds; `run` (packed into sa_full) says how far it
// holds going down from each position, so that an index with unordered ties (the upstream k-ordered sort leaves
// suffixes equal over sort_len bases in any order) or a damaged one falls back to rank queries exactly there.
__global__ __launch_bounds__(256) void k_row_scatter(uint64_t n, const uint64_t* __restrict__ sa_full,
                                                     unsigned long long* __restrict__ row_tab,
                                                     uint8_t* __restrict__ badpos) {
  const uint64_t i = global_thread();
  if (i >= n) return;
  const uint64_t p = sa_full[i];
  if (p >= n) return;  // kNoPosition or nonsense: the row claims no position
  const unsigned long long old = atomicCAS(&row_tab[p], ~0ull, (unsigned long long)i);
  if (old != ~0ull) badpos[p] = 1;  // two rows report the same position
}

__global__ __launch_bounds__(256) void k_verify_rows(DeviceIndex ix, const uint64_t* __restrict__ sa_full,
                                                     const unsigned long long* __restrict__ row_tab,
                                                     uint8_t* __restrict__ badpos) {
  const uint64_t j = global_thread();
  if (j >= ix.n) return;
  const uint64_t p = sa_full[j];
  if (p >= ix.n) return;
  bool bad = row_tab[p] != j;
  {
    // a row is sampled exactly when its position is a multiple of the sampling distance (what makes the
    // depth-first locate report every row of an interval once: multi_locate.h)
    const Block1* bj = ix.blocks1 + (j >> 6);
    const bool sampled = (bj->sampled >> (j & 63)) & 1ull;
    if (sampled != (p % ix.interval == 0)) bad = true;
  }
  if (!bad && j == ix.first_location) bad = p != 0;  // the '$' row ends every chain: text position 0
  if (!bad && j != ix.first_location) {
    if (p == 0) bad = true;
    else {
      const Block1* b = ix.blocks1 + (j >> 6);
      const uint32_t o = (uint32_t)(j & 63);
      const uint32_t c = (uint32_t)((b->plane_lo >> o) & 1ull) | ((uint32_t)((b->plane_hi >> o) & 1ull) << 1);
      bad = sa_full[lf1(ix, j, c)] != p - 1;
    }
  }
  if (bad) badpos[p] = 1;
}

// one thread per 16-base word of textrc
__global__ __launch_bounds__(256) void k_textrc(DeviceIndex ix, const unsigned long long* __restrict__ row_tab,
                                                uint32_t* __restrict__ textrc, uint64_t n_words) {
  const uint64_t w = global_thread();
  if (w >= n_words) return;
  const uint64_t m = ix.n - 1;
  uint32_t out = 0;
  for (int t = 0; t < 16; ++t) {
    const uint64_t x = w * 16 + t;
    uint32_t c = 0;
    if (x < m) {
      const uint64_t r = row_tab[m - x];  // text[q] = bwt[row of position q + 1], q = m - 1 - x
      if (r != ~0ull) {
        const Block1* b = ix.blocks1 + (r >> 6);
        const uint32_t o = (uint32_t)(r & 63);
        c = 3u - ((uint32_t)((b->plane_lo >> o) & 1ull) | ((uint32_t)((b->plane_hi >> o) & 1ull) << 1));
      }
    }
    out |= c << (2 * t);
  }
  textrc[w] = out;
}

constexpr uint64_t kRunChunk = 1024;  // text positions per thread in the run computation

// chunk_last[c] = 1 + the largest bad position of chunk c, or 0
__global__ __launch_bounds__(256) void k_run_chunks(uint64_t n, const uint8_t* __restrict__ badpos,
                                                    const unsigned long long* __restrict__ row_tab,
                                                    uint64_t* __restrict__ chunk_last) {
  const uint64_t c = global_thread();
  const uint64_t a = c * kRunChunk;
  if (a >= n) return;
  const uint64_t b = a + kRunChunk < n ? a + kRunChunk : n;
  uint64_t last = 0;
  for (uint64_t q = a; q < b; ++q)
    if (badpos[q] || row_tab[q] == ~0ull) last = q + 1;
  chunk_last[c] = last;
}
// in place, one workgroup: running maximum over chunks (n / 1024 entries); each thread owns a contiguous slice
__global__ __launch_bounds__(1024) void k_run_scan(uint64_t* chunk_last, uint64_t n_chunks) {
  __shared__ uint64_t slice_max[1024];
  const uint64_t per = (n_chunks + 1023) / 1024;
  const uint64_t a = threadIdx.x * per < n_chunks ? threadIdx.x * per : n_chunks;
  const uint64_t b = a + per < n_chunks ? a + per : n_chunks;
  uint64_t mx = 0;
  for (uint64_t c = a; c < b; ++c) mx = chunk_last[c] > mx ? chunk_last[c] : mx;
  slice_max[threadIdx.x] = mx;
  __syncthreads();
  uint64_t acc = 0;
  for (uint32_t t = 0; t < threadIdx.x; ++t) acc = slice_max[t] > acc ? slice_max[t] : acc;
  for (uint64_t c = a; c < b; ++c) {
    const uint64_t v = chunk_last[c];
    acc = v > acc ? v : acc;
    chunk_last[c] = acc;
  }
}
__global__ __launch_bounds__(256) void k_run_apply(uint64_t n, const uint8_t* __restrict__ badpos,
                                                   const unsigned long long* __restrict__ row_tab,
                                                   const uint64_t* __restrict__ chunk_last, uint64_t* __restrict__ sa_full) {
  const uint64_t c = global_thread();
  const uint64_t a = c * kRunChunk;
  if (a >= n) return;
  const uint64_t b = a + kRunChunk < n ? a + kRunChunk : n;
  uint64_t last = c ? chunk_last[c - 1] : 0;
  for (uint64_t q = a; q < b; ++q) {
    const uint64_t r = row_tab[q];
    if (badpos[q] || r == ~0ull) { last = q + 1; continue; }
    uint64_t run = q + 1 - last;  // positions q, q-1, ..., q-run+1 are good
    if (run > kRunMax) run = kRunMax;
    sa_full[r] = q | (run << kPosBits);  // sa_full[r] == q was verified
  }
}

// 16 bases of textrc starting at x (zero where x < 0; the array is zero-padded past its end)
__device__ __forceinline__ uint32_t textrc_word_at(const uint32_t* __restrict__ textrc, int64_t x) {
  if (x >= 0) {
    const uint64_t w = (uint64_t)x >> 4;
    const uint32_t sh = 2 * (uint32_t)(x & 15);
    const uint32_t lo = textrc[w], hi = textrc[w + 1];
    return sh ? ((lo >> sh) | (hi << (32 - sh))) : lo;
  }
  uint32_t out = 0;
  for (int t = 0; t < 16; ++t) {
    const int64_t y = x + t;
    if (y >= 0) out |= ((textrc[(uint64_t)y >> 4] >> (2 * (uint32_t)(y & 15))) & 3u) << (2 * t);
  }
  return out;
}

// one thread per PosLine (device_layout.h)
__global__ __launch_bounds__(256) void k_pos_lines(DeviceIndex ix, const unsigned long long* __restrict__ row_tab,
                                                   const uint32_t* __restrict__ textrc, PosLine* __restrict__ lines,
                                                   uint64_t n_lines) {
  const uint64_t line = global_thread();
  if (line >= n_lines) return;
  const uint64_t q0 = line * kPosPerLine;
  const int64_t xs = (int64_t)ix.n - 2 - (int64_t)q0 - 127;
  uint32_t text[8];
#pragma unroll
  for (int w = 0; w < 8; ++w) text[w] = textrc_word_at(textrc, xs + 16 * w);
  uint64_t rows[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) rows[k] = 0;
#pragma unroll
  for (int s = 0; s < (int)kPosPerLine; ++s) {
    const uint64_t q = q0 + (uint64_t)s;
    const uint64_t v = (q < ix.n ? (uint64_t)row_tab[q] : ~0ull) & kPosMask;
    const int word = (40 * s) >> 6, sh = (40 * s) & 63;
    rows[word] |= v << sh;
    if (sh > 24) rows[word + 1] |= v >> (64 - sh);
  }
  uint4* out = reinterpret_cast<uint4*>(lines + line);
  out[0] = make_uint4(text[0], text[1], text[2], text[3]);
  out[1] = make_uint4(text[4], text[5], text[6], text[7]);
#pragma unroll
  for (int k = 0; k < 6; ++k)
    out[2 + k] = make_uint4((uint32_t)rows[2 * k], (uint32_t)(rows[2 * k] >> 32), (uint32_t)rows[2 * k + 1],
                            (uint32_t)(rows[2 * k + 1] >> 32));
}

// ---- blocks2: two-step planes ----------------------------------------------------------------------
// one wave per index block; lane j owns position 64b + j
__global__ __launch_bounds__(256) void k_planes2(DeviceIndex ix, Block2* __restrict__ blocks2, uint64_t* special) {
  const uint64_t gid = global_thread();
  const uint64_t b = gid >> 6;
  if (b >= ix.n_blocks) return;  // wave-uniform
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t j = b * 64 + lane;
  const Block1* blk = ix.blocks1 + b;
  uint32_t code = 0;
  if (j < ix.n && j != ix.first_location) {
    const uint32_t x = (uint32_t)((blk->plane_lo >> lane) & 1ull) | ((uint32_t)((blk->plane_hi >> lane) & 1ull) << 1);
    const uint64_t j1 = lf1(ix, j, x);
    uint32_t y = 0;
    if (j1 == ix.first_location) {
      special[0] = j;       // the one position whose second base is '$'
      special[1] = 4 * x;   // stored with y = 0; rank2 takes it back out
    } else {
      const Block1* b1 = ix.blocks1 + (j1 >> 6);
      const uint32_t o1 = (uint32_t)(j1 & 63);
      y = (uint32_t)((b1->plane_lo >> o1) & 1ull) | ((uint32_t)((b1->plane_hi >> o1) & 1ull) << 1);
    }
    code = 4 * x + y;
  }
  uint64_t planes[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) planes[k] = __ballot((code >> k) & 1u);
  if (lane == 0) {
    Block2* o = blocks2 + b;
#pragma unroll
    for (int k = 0; k < 4; ++k) o->plane[k] = planes[k];
    o->chain_rows = 0;
    o->chain_base = 0;
    o->reserved[0] = o->reserved[1] = 0;
  }
}

__global__ void k_k2(DeviceIndex ix, uint64_t* k2) {
  const uint32_t code = threadIdx.x;
  if (code >= 16) return;
  const uint32_t x = code >> 2, y = code & 3;
  // C[y] + Occ(y, C[x]): where the suffixes starting with y x begin
  k2[code] = lf1(ix, ix.c_function[x], y);
}

__global__ __launch_bounds__(256) void k_apply2(Block2* __restrict__ blocks, uint64_t n_blocks,
                                                const uint64_t* __restrict__ chunk_base, const uint64_t* __restrict__ k2,
                                                uint64_t* __restrict__ sb_abs) {
  __shared__ uint32_t lds[4];
  const uint64_t b = (uint64_t)blockIdx.x * kChunk + threadIdx.x;
  const bool live = b < n_blocks;
  Counts<16> cnt;
  if (live) cnt = counts_of(blocks[b]);
  else
    for (int k = 0; k < 16; ++k) cnt.c[k] = 0;
  // superblock of this chunk: 2^31 positions = 2^25 blocks = 2^17 chunks (chunks never straddle one)
  const uint64_t sb = ((uint64_t)blockIdx.x * kChunk) >> (kSuperShift - 6);
  const uint64_t sb_chunk = sb << (kSuperShift - 6 - 8);
  for (int k = 0; k < 16; ++k) {
    uint32_t tot;
    const uint32_t ex = block_excl_scan(cnt.c[k], lds, &tot);
    __syncthreads();
    const uint64_t base = chunk_base[(uint64_t)blockIdx.x * 16 + k], sbase = chunk_base[sb_chunk * 16 + k];
    if (live) blocks[b].cnt[k] = (uint32_t)(base - sbase + ex);
    if (blockIdx.x == sb_chunk && threadIdx.x == 0) sb_abs[sb * 16 + k] = k2[k] + sbase;
  }
}

// ---- chain_tab: the depth-first locate of one row, memoised (device_layout.h) ---------------------------------
// Which rows need an entry: those whose locate sa_full and the text do not answer for -- no position, or a run too
// short for the text to stand in for the chain over the whole depth of the walk (the test k_multi_sort applies).
__device__ __forceinline__ bool chain_row_needed(const DeviceIndex& ix, uint64_t e) {
  if (e == kNoPosition) return true;
  const uint64_t p = e & kPosMask;
  return (e >> kPosBits) < (p + 1 < ix.interval ? p + 1 : ix.interval);
}

// one wave per index block, lane j owns row 64b + j
__global__ __launch_bounds__(256) void k_chain_rows(DeviceIndex ix, Block2* __restrict__ blocks2) {
  const uint64_t gid = global_thread();
  const uint64_t b = gid >> 6;
  if (b >= ix.n_blocks) return;  // wave-uniform
  const uint64_t i = gid;
  const bool need = i < ix.n && chain_row_needed(ix, ix.sa_full[i]);
  const uint64_t bits = __ballot(need);
  if ((threadIdx.x & 63) == 0) blocks2[b].chain_rows = bits;
}
// totals[chunk] = rows with an entry in one chunk of 256 blocks; then (after k_scan_chunks) chain_base of every block
__global__ __launch_bounds__(256) void k_chain_totals(const Block2* __restrict__ blocks2, uint64_t n_blocks, uint64_t* __restrict__ totals) {
  __shared__ uint32_t lds[4];
  const uint64_t b = (uint64_t)blockIdx.x * kChunk + threadIdx.x;
  uint32_t tot;
  block_excl_scan(b < n_blocks ? (uint32_t)__popcll(blocks2[b].chain_rows) : 0u, lds, &tot);
  if (threadIdx.x == 0) totals[blockIdx.x] = tot;
}
__global__ __launch_bounds__(256) void k_chain_bases(Block2* __restrict__ blocks2, uint64_t n_blocks, const uint64_t* __restrict__ chunk_base) {
  __shared__ uint32_t lds[4];
  const uint64_t b = (uint64_t)blockIdx.x * kChunk + threadIdx.x;
  uint32_t tot;
  const uint32_t ex = block_excl_scan(b < n_blocks ? (uint32_t)__popcll(blocks2[b].chain_rows) : 0u, lds, &tot);
  if (b < n_blocks) blocks2[b].chain_base = chunk_base[blockIdx.x] + ex;
}
__global__ __launch_bounds__(256) void k_chain_clear(Block2* __restrict__ blocks2, uint64_t n_blocks) {
  const uint64_t b = global_thread();
  if (b < n_blocks) { blocks2[b].chain_rows = 0; blocks2[b].chain_base = 0; }
}

// The chain of row i (device_layout.h): up to interval - 1 LF steps, every sampled row on the way an event.  kWrite
// false: only counts the events beyond the first (total_extra); true: writes the entry and its further events (slots
// handed out through next_extra).  One wave per index block, lane j owns row 64b + j.
template <bool kWrite>
__global__ __launch_bounds__(256) void k_chain_walk(DeviceIndex ix, ChainEntry* __restrict__ tab, uint64_t* __restrict__ ev,
                                                    unsigned long long* __restrict__ extra_counter) {
  const uint64_t gid = global_thread();
  const uint64_t b = gid >> 6;
  if (b >= ix.n_blocks) return;  // wave-uniform
  const uint32_t lane = threadIdx.x & 63;
  const Block2* blk2 = ix.blocks2 + b;
  const uint64_t rows = blk2->chain_rows;
  if (!rows) return;  // wave-uniform
  const bool mine = (rows >> lane) & 1ull;
  uint64_t k0 = 0, k1 = 0, ev0 = 0, first_extra = 0;
  uint32_t n_ev = 0, len = 0;
  for (int pass = 0; pass < (kWrite ? 2 : 1); ++pass) {
    // (writing: the first pass counts the chain's events so that its slots in chain_ev can be taken in one piece)
    if (kWrite && pass == 1) {
      if (mine && n_ev > 1) first_extra = atomicAdd(extra_counter, (unsigned long long)(n_ev - 1));
      n_ev = 0;
    }
    if (mine) {
      uint64_t cur = gid;
      for (uint64_t d = 0; d < ix.interval; ++d) {
        const Block1* bl = ix.blocks1 + (cur >> 6);
        const uint32_t o = (uint32_t)(cur & 63);
        const uint64_t sampled = bl->sampled;
        if ((sampled >> o) & 1ull) {
          const uint64_t e = (d << 58) | ((ix.loc_vals[bl->sampled_base + (uint64_t)__popcll(sampled & below(o))] + d) & kPosMask);
          if (n_ev == 0) ev0 = e;
          else if (kWrite && pass == 1) ev[first_extra + n_ev - 1] = e;
          ++n_ev;
        }
        if (d + 1 >= ix.interval || cur == ix.first_location) break;  // find_possible: len >= interval; the '$' row has no children
        const uint64_t plo = bl->plane_lo, phi = bl->plane_hi;
        const uint32_t c = (uint32_t)((plo >> o) & 1ull) | ((uint32_t)((phi >> o) & 1ull) << 1);
        if (pass == 0) {
          if (d < 32) k0 |= (uint64_t)c << (62 - 2 * d); else k1 |= (uint64_t)c << (62 - 2 * (d - 32));
          len = (uint32_t)d + 1;
        }
        uint64_t r = bl->cnt[c] + (uint64_t)__popcll(match1(plo, phi, c) & below(o));
        if (c == 0 && cur > ix.first_location) r -= 1;
        cur = r;
      }
    }
  }
  if (!kWrite) {
    unsigned long long extra = mine && n_ev > 1 ? n_ev - 1 : 0;
    for (int d = 32; d > 0; d >>= 1) extra += ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)extra, d));
    if (lane == 0 && extra) atomicAdd(extra_counter, extra);
    return;
  }
  if (mine) {
    ChainEntry e;
    e.k0 = k0;
    e.k1 = k1;
    e.meta = (uint64_t)n_ev | ((uint64_t)len << 8) | (first_extra << kChainEvOffShift);
    e.ev0 = ev0;
    tab[blk2->chain_base + (uint64_t)__popcll(rows & below(lane))] = e;
  }
}

// ---- kmer_tab: memoised search prefix -------------------------------------------------------------------
// entry v (the complemented codes of read[0..K), base t at bits 2t) = the interval the upstream search holds after
// init_exact_match on the first 12 bases (searcher.hpp:83-96) and exec_exact_match on the next K - 12
// (searcher.hpp:406-410, which stops as soon as the interval is empty): computed by running exactly those steps.
__global__ __launch_bounds__(256) void k_kmer_table(DeviceIndex ix, uint32_t K, uint64_t* __restrict__ tab) {
  const uint64_t v = global_thread();
  if (v >> (2 * K)) return;
  const uint32_t kmer = (uint32_t)v & 0xFFFFFFu;
  uint64_t lo = ix.c_functions[kmer], hi = ix.c_functions[kmer + 1];
  bool alive = lo < hi;
  uint32_t t = kKmer;
  for (; alive && t + 2 <= K; t += 2)
    alive = step_two_bases(ix, ((uint32_t)(v >> (2 * t)) & 15u) ^ 15u, lo, hi);
  if (alive && t < K) {  // odd order: the last base alone
    const uint32_t c = (uint32_t)(v >> (2 * t)) & 3u;  // v holds complemented codes: the query is RC(read)
    const uint64_t l1 = lf1(ix, lo, c), h1 = lf1(ix, hi, c);
    lo = l1;
    hi = h1;
  }
  uint64_t width = hi > lo ? hi - lo : 0;
  if (width >= kKmerWide) { width = kKmerWide; lo = 0; }
  uint64_t entry = lo | (width << kPosBits);
  if (width == 1) {
    const uint64_t e = ix.sa_full[lo];
    const uint64_t run = e >> kPosBits;
    if (e != kNoPosition && run >= 1)
      entry = kKmerLocated | (e & kPosMask) | ((run < kKmerRunMax ? run : kKmerRunMax) << kPosBits);
  }
  tab[v] = entry;
}

// ---- hash_tab: located prefixes of order K (device_layout.h) ----------------------------------------------
// the slot of the mixed value vv goes into the first free slot of its bucket b, then (with text) of the other buckets
// of its GROUP, b ^ 1, b ^ 2, ... b ^ (group - 1): the other half of its 128-byte line first, then the neighbouring
// lines.  The values of the buckets of one group differ by less than 2^tag_bits, so the tag still tells them all
// apart, and the lookups go on to the next bucket only when the one before has no free slot.  Slots are never emptied
// again: a slot free at the end was free to every prefix that came by.
__device__ __forceinline__ bool hash_insert(uint64_t* tab, uint64_t n_buckets, uint32_t with_text, uint32_t group, uint64_t vv, uint32_t K,
                                            uint64_t slot, uint64_t text) {
  const uint64_t b = __umul64hi(vv << (64 - 2 * K), n_buckets);
  const uint32_t stride = with_text ? 2u : 1u;  // words per slot
  for (uint32_t s = 0; s < group * kHashSlots; ++s) {
    unsigned long long* bk = reinterpret_cast<unsigned long long*>(tab) + ((b ^ (uint64_t)(s / kHashSlots)) * kHashSlots + (s & (kHashSlots - 1))) * stride;
    if (atomicCAS(&bk[0], ~0ull, (unsigned long long)slot) == ~0ull) {
      if (with_text) bk[1] = text;
      return true;
    }
  }
  return false;  // every bucket of the group full: this K-mer stays with the fallback path
}


// One thread per row: the K-mer in front of the row's text position and the upstream steps for that K-mer (exactly as
// k_kmer_table runs them).  When they end on this very row, alone: an entry with the position.  With text, when they
// end on an interval that holds this row among others, its first row stores a MARKER (run = 0: the prefix occurs,
// k_settle leaves the read to the general kernels), and a row the steps for ITS K-mer do not end on -- no K bases
// behind it, not located, an order of suffixes the text does not explain -- goes to `uncovered` for k_hash_cover.
__global__ __launch_bounds__(256) void k_hash_build(DeviceIndex ix, uint32_t K, uint64_t* __restrict__ tab, uint64_t n_buckets,
                                                    uint32_t tag_bits, uint32_t with_text, uint32_t group, uint32_t twins,
                                                    unsigned long long* __restrict__ counts, uint64_t* __restrict__ uncovered,
                                                    uint64_t uncovered_cap) {
  const uint64_t i = global_thread();
  uint32_t outcome = 0;  // 1 stored, 2 left out, 3 marker stored, 4 marker left out, 7 twin slots stored
  bool covered = false;
  const uint64_t e = i < ix.n ? ix.sa_full[i] : kNoPosition;
  const uint64_t p = e & kPosMask, run = e >> kPosBits;
  const uint64_t m = ix.n - 1;
  // (with text every located row is looked at: one whose run is 0 gets no entry of its own, but its K-mer a marker)
  if (e != kNoPosition && (run >= 1 || with_text) && p + K <= m) {  // K real bases behind the position
    // digit t of v = complemented code of read[t] = 3 - textrc[x0 + t], x0 = m - p - K
    const uint64_t x0 = m - p - K;
    const uint32_t* tx = ix.textrc + (x0 >> 4);
    const uint32_t sh = 2u * (uint32_t)(x0 & 15);
    const uint64_t w0 = tx[0] | ((uint64_t)tx[1] << 32), w1 = tx[2];
    const uint64_t bits = sh ? ((w0 >> sh) | (w1 << (64 - sh))) : w0;
    const uint64_t v = (~bits) & ((1ull << (2 * K)) - 1ull);
    const uint32_t kmer = (uint32_t)v & 0xFFFFFFu;
    uint64_t lo = ix.c_functions[kmer], hi = ix.c_functions[kmer + 1];
    bool alive = lo < hi;
    uint32_t t = kKmer;
    for (; alive && t + 2 <= K; t += 2)
      alive = step_two_bases(ix, ((uint32_t)(v >> (2 * t)) & 15u) ^ 15u, lo, hi);
    if (alive && t < K) {
      const uint32_t c = (uint32_t)(v >> (2 * t)) & 3u;
      const uint64_t l1 = lf1(ix, lo, c), h1 = lf1(ix, hi, c);
      lo = l1;
      hi = h1;
      alive = lo < hi;
    }
    covered = alive && lo <= i && i < hi;
    const uint64_t vv = hash_mix(v, K);
    const uint64_t tag = vv & ((1ull << tag_bits) - 1ull);
    const uint64_t run_max = (1ull << (24 - tag_bits)) - 1ull;
    // what the read's bases K .. K+31 are compared with: textrc[m - q ..) (search_machine.h, S_CMP: x0 = n - 1 - q)
    auto text_behind = [&](uint64_t q) -> uint64_t {
      const uint64_t x1 = m - q;
      const uint32_t* ty = ix.textrc + (x1 >> 4);
      const uint32_t sh1 = 2u * (uint32_t)(x1 & 15);
      const uint64_t u0 = ty[0] | ((uint64_t)ty[1] << 32), u1 = ty[2];
      return sh1 ? ((u0 >> sh1) | (u1 << (64 - sh1))) : u0;
    };
    if (covered && hi - lo == 1 && run >= 1) {
      const uint64_t slot = (p << 24) | ((run < run_max ? run : run_max) << tag_bits) | tag;
      outcome = hash_insert(tab, n_buckets, with_text, group, vv, K, slot, with_text ? text_behind(p) : 0) ? 1 : 2;
    } else if (covered && with_text && i == lo) {
      // TWINS (device_layout.h): a K-mer on exactly two rows, both located, both with the K bases behind them in the text:
      // each gets a slot of its own (kHashTwin), and a read whose further bases match ONE of the two texts is settled
      // like any other.  A twin whose partner found no slot in the same bucket reads as a marker.
      uint64_t e2 = kNoPosition;
      if (twins && hi - lo == 2 && run >= 1) e2 = ix.sa_full[lo + 1];
      const uint64_t p2 = e2 & kPosMask, run2 = e2 >> kPosBits;
      bool twin = e2 != kNoPosition && run2 >= 1 && p2 + K <= m;
      if (twin) {
        const uint64_t y0 = m - p2 - K;
        const uint32_t* tz = ix.textrc + (y0 >> 4);
        const uint32_t sh2 = 2u * (uint32_t)(y0 & 15);
        const uint64_t z0 = tz[0] | ((uint64_t)tz[1] << 32), z1 = tz[2];
        const uint64_t bits2 = sh2 ? ((z0 >> sh2) | (z1 << (64 - sh2))) : z0;
        twin = ((~bits2) & ((1ull << (2 * K)) - 1ull)) == v;
      }
      if (twin) {
        const uint64_t s1 = kHashTwin | (p << 24) | ((run < run_max ? run : run_max) << tag_bits) | tag;
        const uint64_t s2 = kHashTwin | (p2 << 24) | ((run2 < run_max ? run2 : run_max) << tag_bits) | tag;
        const bool in1 = hash_insert(tab, n_buckets, with_text, group, vv, K, s1, text_behind(p));
        const bool in2 = in1 && hash_insert(tab, n_buckets, with_text, group, vv, K, s2, text_behind(p2));
        outcome = in2 ? 7 : in1 ? 3 : 4;
      } else {
        outcome = hash_insert(tab, n_buckets, with_text, group, vv, K, tag, 0) ? 3 : 4;  // position 0, run 0
      }
    }
  }
  if (with_text && i < ix.n && !covered) {
    const unsigned long long k = atomicAdd(&counts[4], 1ull);  // (a handful of rows on a fully sorted index: the ends of the text)
    if (k < uncovered_cap) uncovered[k] = i;
  }
  // statistics: a few atomics per wave
  for (uint32_t o = 1; o <= 7; ++o) {
    if (o == 5 || o == 6) continue;  // (counts[4], counts[5]: uncovered rows, failed checks)
    const uint64_t mk = __ballot(outcome == o);
    if ((threadIdx.x & 63) == 0 && mk) atomicAdd(&counts[o - 1], (unsigned long long)__popcll(mk));
  }
}

// What a miss may prove rests on three facts about the tables, checked here (counts[5] += violations): c_functions
// ascends and ends inside the rows (the intervals of different 12-mers are disjoint), and the rows LF reaches with
// base c lie before those it reaches with base c + 1.  Together with rank being a count (lf1 never decreases in i)
// they make the intervals of different K-mers disjoint at every order: a row lies in the interval of ONE K-mer at most.
__global__ __launch_bounds__(256) void k_hash_checks(DeviceIndex ix, unsigned long long* __restrict__ counts) {
  const uint64_t v = global_thread();
  uint32_t bad = 0;
  if (v < kNumKmer) bad += ix.c_functions[v] > ix.c_functions[v + 1];
  if (v == 0) {
    bad += ix.c_functions[kNumKmer] > ix.n;
    for (uint32_t c = 0; c < 4; ++c) {
      const uint64_t top = lf1(ix, ix.n, c), next = c < 3 ? lf1(ix, 0, c + 1) : ix.n;
      bad += top > next || lf1(ix, 0, c) > top;
    }
  }
  if (bad) atomicAdd(&counts[5], (unsigned long long)bad);
}

// One thread per uncovered row u: the ONE K-mer whose interval can hold u, found backwards.  The last step of the
// search brought the interval into the rows of base c = digit K-1; u lies among the rows of one base only
// (k_hash_checks), and the row j with LF_c(j) = u is unique, so the interval before that step held j: K - 12 such
// steps back, then the 12-mer whose c_functions interval holds the row arrived at.  That K-mer gets a marker (if it
// has no slot yet); a chain that breaks (no base reaches the row, no 12-mer holds it) means no K-mer ends on u.
__global__ __launch_bounds__(64) void k_hash_cover(DeviceIndex ix, uint32_t K, uint64_t* __restrict__ tab, uint64_t n_buckets, uint32_t tag_bits,
                                                   uint32_t group, const uint64_t* __restrict__ uncovered, uint64_t n_uncovered,
                                                   unsigned long long* __restrict__ counts) {
  const uint64_t k = global_thread();
  if (k >= n_uncovered) return;
  uint64_t u = uncovered[k], v = 0;
  uint64_t first[4], end[4];  // the rows LF reaches with each base
  for (uint32_t q = 0; q < 4; ++q) { first[q] = lf1(ix, 0, q); end[q] = lf1(ix, ix.n, q); }
  for (uint32_t t = K; t-- > (uint32_t)kKmer;) {
    uint32_t c = 4;
    for (uint32_t q = 0; q < 4; ++q)
      if (first[q] <= u && u < end[q]) c = q;
    if (c == 4) return;
    // smallest i in [1, n] with lf1(i, c) >= u + 1; the row is i - 1
    uint64_t a = 1, b = ix.n;
    while (a < b) {
      const uint64_t mid = a + (b - a) / 2;
      if (lf1(ix, mid, c) >= u + 1) b = mid; else a = mid + 1;
    }
    u = a - 1;
    v |= (uint64_t)c << (2 * t);
  }
  // the 12-mer: last entry of c_functions that is <= u
  uint64_t a = 0, b = kNumKmer;  // c_functions[0] <= u is not given: checked below
  while (a < b) {
    const uint64_t mid = a + (b - a + 1) / 2;
    if (ix.c_functions[mid] <= u) a = mid; else b = mid - 1;
  }
  if (a >= kNumKmer || ix.c_functions[a] > u || u >= ix.c_functions[a + 1]) return;
  v |= a;
  const uint64_t vv = hash_mix(v, K);
  const uint64_t tag = vv & ((1ull << tag_bits) - 1ull);
  // already there?  (a row of an interval whose first row stored the marker; markers of the same K-mer written by
  // two threads of this kernel are equal words, so a lookup that meets both reads the same thing twice)
  const uint64_t bkt = __umul64hi(vv << (64 - 2 * K), n_buckets);
  for (uint32_t s = 0; s < group * kHashSlots; ++s) {
    const uint64_t w = tab[((bkt ^ (uint64_t)(s / kHashSlots)) * kHashSlots + (s & (kHashSlots - 1))) * 2u];
    if ((w >> 32) != 0xFFFFFFFFull && ((w ^ tag) & ((1ull << tag_bits) - 1ull)) == 0) return;
  }
  atomicAdd(&counts[hash_insert(tab, n_buckets, 1u, group, vv, K, tag, 0) ? 2 : 3], 1ull);
}

// ---------------------------------------------------------------------------------------------------
static inline dim3 grid_for(uint64_t items, uint32_t per_block) { return grid_items(items, per_block); }

int build_device_layout(sbwt_index* ix, const RawDeviceTables& raw) {
  DeviceIndex& d = ix->d;
  d.n = raw.n;
  d.first_location = raw.first_location;
  d.real_size = raw.real_size;
  d.interval = raw.interval;
  d.n_blocks = raw.n / 64 + 1;
  for (int c = 0; c < 4; ++c) d.c_function[c] = raw.c_function4[c];
  d.c_functions = raw.c_functions;
  d.special_pos[0] = raw.first_location;
  d.special_code[0] = 0;
  d.special_pos[1] = ~0ull;
  d.special_code[1] = 0;
  ix->n_loc = raw.n_loc;

  const uint64_t nb = d.n_blocks;
  const uint64_t n_chunks = (nb + kChunk - 1) / kChunk;
  // SBWT_TIMING: seconds per part of the layout (stderr); each mark waits for the device
  const bool timing = getenv("SBWT_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_mark = now();
  auto mark = [&](const char* what) {
    if (!timing) return;
    (void)hipDeviceSynchronize();
    const double t = now();
    fprintf(stderr, "[sbwt timing]   layout: %-34s %.2f s\n", what, t - t_mark);
    t_mark = t;
  };
  Block1* blocks1 = nullptr;
  Block2* blocks2 = nullptr;
  uint64_t *loc_vals = nullptr, *sa_full = nullptr, *sb_abs = nullptr;
  if (int rc = dev_alloc(ix, &blocks1, nb)) return rc;
  if (int rc = dev_alloc(ix, &loc_vals, raw.n_loc)) return rc;
  if (int rc = dev_alloc(ix, &sa_full, raw.n)) return rc;
  if (int rc = dev_alloc(ix, &blocks2, nb)) return rc;
  const uint64_t n_super = ((nb - 1) >> (kSuperShift - 6)) + 1;
  if (int rc = dev_alloc(ix, &sb_abs, n_super * 16)) return rc;

  // temporaries
  DevTemp<uint64_t> chunk_base;
  DevTemp<LayoutErrors> d_err;
  DevTemp<uint64_t> d_small;  // [0..1] special, [2..17] k2
  SBWT_HIP(chunk_base.alloc(n_chunks * 16));
  SBWT_HIP(d_err.alloc(1));
  SBWT_HIP(d_small.alloc(18));
  SBWT_HIP(hipMemset(d_err, 0, sizeof(LayoutErrors)));
  SBWT_HIP(hipMemset(d_small, 0xff, 2 * sizeof(uint64_t)));

  k_planes1<<<grid_for(nb, 256), 256>>>(blocks1, nb, raw.n, raw.jbwt, raw.n / 4 + 1, raw.fbwt, raw.loc, raw.n_loc, d_err);
  SBWT_LAUNCHED();
  if (raw.n_loc) k_loc_vals<<<grid_for(raw.n_loc, 256), 256>>>(raw.loc, raw.n_loc, raw.n, loc_vals, d_err);
  k_chunk_totals<4, Block1><<<(uint32_t)n_chunks, kChunk>>>(blocks1, nb, chunk_base);
  k_scan_chunks<<<1, 64>>>(chunk_base, n_chunks, 4);
  k_apply1<<<(uint32_t)n_chunks, kChunk>>>(blocks1, nb, chunk_base, d.c_function[0], d.c_function[1], d.c_function[2],
                                           d.c_function[3], raw.interval, raw.occ[0], raw.occ[1], raw.occ[2],
                                           raw.occ[3], raw.occ_len, d_err);
  SBWT_HIP(hipGetLastError());
  d.blocks1 = blocks1;
  d.loc_vals = loc_vals;

  LayoutErrors h_err;
  SBWT_HIP(hipMemcpy(&h_err, d_err, sizeof h_err, hipMemcpyDeviceToHost));
  if (h_err.loc_fbwt_mismatch || h_err.loc_not_sorted || h_err.occ_mismatch) {
    char msg[256];
    snprintf(msg, sizeof msg,
             "index tables are inconsistent (location_table/fbwt mismatches %llu, unsorted keys %llu, "
             "occ_function mismatches %llu): index files appear to be damaged",
             h_err.loc_fbwt_mismatch, h_err.loc_not_sorted, h_err.occ_mismatch);
    set_error(msg);
    return -2;  // SBWT_ERR_FORMAT
  }

  {
    DevTemp<SaCheck> d_chk;
    SaCheck h_chk;
    SBWT_HIP(d_chk.alloc(1));
    SBWT_HIP(hipMemset(d_chk, 0, sizeof(SaCheck)));
    SBWT_HIP(hipMemset(sa_full, 0xff, raw.n * sizeof(uint64_t)));
    if (raw.n_loc) k_sa_fast<<<grid_for(raw.n_loc, 256), 256>>>(d, raw.loc, raw.n_loc, sa_full, d_chk);
    k_sa_fill<<<grid_for(raw.n, 256), 256>>>(d, sa_full, d_chk);
    k_sa_check<<<grid_for(raw.n, 256), 256>>>(d, sa_full, d_chk);
    SBWT_LAUNCHED();
    SBWT_HIP(hipMemcpy(&h_chk, d_chk, sizeof h_chk, hipMemcpyDeviceToHost));
    if (h_chk.failed || getenv("SBWT_SLOW_SA")) {
      k_sa_full<<<grid_for(raw.n, 256), 256>>>(d, sa_full);
      SBWT_LAUNCHED();
    }
  }
  d.sa_full = sa_full;
  mark("blocks1, loc_vals, sa_full");
  if (raw.n >= (1ull << kPosBits)) {
    set_error("index has 2^40 suffixes or more");
    return -4;
  }
  {
    // text-jump tables; sa_full still holds plain positions here (rows never claimed keep run 0 = their
    // position with no run bits; rows holding kNoPosition keep it)
    DevTemp<unsigned long long> row_tab;  // it leaves as pos_lines
    uint32_t* textrc = nullptr;
    const uint64_t n_words = ((raw.n - 1 + 63) / 64) * 4 + 8;  // whole 16-byte pieces + 2 of padding: the comparison reads one piece ahead
    PosLine* pos_lines = nullptr;
    const uint64_t n_lines = (raw.n + kPosPerLine - 1) / kPosPerLine;
    if (int rc = dev_alloc(ix, &textrc, n_words)) return rc;
    if (int rc = dev_alloc(ix, &pos_lines, n_lines)) return rc;
    SBWT_HIP(row_tab.alloc(raw.n));
    DevTemp<uint8_t> badpos;
    DevTemp<uint64_t> chunk_last;
    const uint64_t n_run_chunks = (raw.n + kRunChunk - 1) / kRunChunk;
    SBWT_HIP(badpos.alloc(raw.n));
    SBWT_HIP(chunk_last.alloc(n_run_chunks));
    SBWT_HIP(hipMemset(badpos, 0, raw.n));
    SBWT_HIP(hipMemset(row_tab, 0xff, raw.n * sizeof(uint64_t)));
    k_row_scatter<<<grid_for(raw.n, 256), 256>>>(raw.n, sa_full, row_tab, badpos);
    SBWT_LAUNCHED();
    k_verify_rows<<<grid_for(raw.n, 256), 256>>>(d, sa_full, row_tab, badpos);
    SBWT_LAUNCHED();
    k_textrc<<<grid_for(n_words, 256), 256>>>(d, row_tab, textrc, n_words);
    SBWT_LAUNCHED();
    k_run_chunks<<<grid_for(n_run_chunks, 256), 256>>>(raw.n, badpos, row_tab, chunk_last);
    k_run_scan<<<1, 1024>>>(chunk_last, n_run_chunks);
    k_run_apply<<<grid_for(n_run_chunks, 256), 256>>>(raw.n, badpos, row_tab, chunk_last, sa_full);
    SBWT_LAUNCHED();
    k_pos_lines<<<grid_for(n_lines, 256), 256>>>(d, row_tab, textrc, pos_lines, n_lines);
    SBWT_LAUNCHED();
    SBWT_HIP(hipDeviceSynchronize());
    d.pos_lines = pos_lines;
    d.textrc = textrc;
  }
  mark("row checks, textrc, runs, pos_lines");

  k_planes2<<<grid_for(nb * 64, 256), 256>>>(d, blocks2, d_small);
  SBWT_LAUNCHED();
  k_k2<<<1, 64>>>(d, d_small + 2);
  k_chunk_totals<16, Block2><<<(uint32_t)n_chunks, kChunk>>>(blocks2, nb, chunk_base);
  k_scan_chunks<<<1, 64>>>(chunk_base, n_chunks, 16);
  k_apply2<<<(uint32_t)n_chunks, kChunk>>>(blocks2, nb, chunk_base, d_small + 2, sb_abs);
  SBWT_HIP(hipGetLastError());
  uint64_t h_small[2];
  SBWT_HIP(hipMemcpy(h_small, d_small, sizeof h_small, hipMemcpyDeviceToHost));
  d.special_pos[1] = h_small[0];
  d.special_code[1] = h_small[0] == ~0ull ? 0u : (uint32_t)h_small[1];
  d.blocks2 = blocks2;
  d.sb_abs = sb_abs;
  d.chain_tab = nullptr;
  d.chain_ev = nullptr;
  mark("blocks2");
  if (raw.interval <= 64 && raw.n + 64 < kPosMask && !getenv("SBWT_NO_CHAIN_TAB")) {
    // chain_tab (device_layout.h): entries for the rows sa_full + the text do not answer for.  A handful on a fully
    // sorted index (the ends of the text); every row within `interval` positions above an LF-inconsistent one on an index
    // whose suffix order stops at sort_len.  No table when they are more than a thirty-second of the rows of a large
    // index (a damaged index: the walk itself remains, k_multi_bfs / k_multi).
    DevTemp<uint64_t> chain_chunks;
    DevTemp<unsigned long long> d_extra;
    SBWT_HIP(chain_chunks.alloc(n_chunks + 1));
    SBWT_HIP(d_extra.alloc(2));
    SBWT_HIP(hipMemset(chain_chunks, 0, (n_chunks + 1) * sizeof(uint64_t)));
    SBWT_HIP(hipMemset(d_extra, 0, 2 * sizeof(unsigned long long)));
    k_chain_rows<<<grid_for(nb * 64, 256), 256>>>(d, blocks2);
    k_chain_totals<<<(uint32_t)n_chunks, kChunk>>>(blocks2, nb, chain_chunks);
    k_scan_chunks<<<1, 64>>>(chain_chunks, n_chunks + 1, 1);
    k_chain_bases<<<(uint32_t)n_chunks, kChunk>>>(blocks2, nb, chain_chunks);
    SBWT_LAUNCHED();
    uint64_t n_chain = 0;
    SBWT_HIP(hipMemcpy(&n_chain, chain_chunks + n_chunks, sizeof n_chain, hipMemcpyDeviceToHost));
    const uint64_t chain_cap = std::max<uint64_t>(raw.n / 32, 1ull << 23);
    ChainEntry* tab = nullptr;
    uint64_t* evs = nullptr;
    bool ok = n_chain > 0 && n_chain <= chain_cap;
    unsigned long long h_extra = 0;
    if (ok) {
      k_chain_walk<false><<<grid_for(nb * 64, 256), 256>>>(d, nullptr, nullptr, d_extra);
      SBWT_LAUNCHED();
      SBWT_HIP(hipMemcpy(&h_extra, d_extra, sizeof h_extra, hipMemcpyDeviceToHost));
      ok = h_extra <= 4 * chain_cap;
    }
    if (ok && (dev_alloc(ix, &tab, n_chain) || dev_alloc(ix, &evs, h_extra + 1))) {
      (void)hipGetLastError();  // (no room: same results through the walk)
      ok = false;
    }
    if (ok) {
      k_chain_walk<true><<<grid_for(nb * 64, 256), 256>>>(d, tab, evs, d_extra + 1);
      SBWT_LAUNCHED();
      SBWT_HIP(hipDeviceSynchronize());
      d.chain_tab = tab;
      d.chain_ev = evs;
    } else if (n_chain) {
      k_chain_clear<<<grid_for(nb, 256), 256>>>(blocks2, nb);
      SBWT_LAUNCHED();
    }
    if (getenv("SBWT_DEBUG"))
      fprintf(stderr, "[sbwt debug] chain_tab: %llu of %llu rows need an entry, %llu further events: %s\n", (unsigned long long)n_chain,
              (unsigned long long)raw.n, h_extra, d.chain_tab ? "built" : (n_chain ? "too many, none built" : "none needed"));
  }
  mark("chain_tab");
  {
    // widest order K <= 17 whose 4^K entries are at most 16 per suffix (beyond that nearly every K-mer that occurs
    // is a single row already) and whose table leaves kKmerHeadroom of this device's memory free (GRCh38: 17,
    // 137 GB; dm6: 16; E. coli: 13); SBWT_KMER_ORDER overrides (12 = none)
    uint32_t K = kKmer;
    size_t free_b = 0, total_b = 0;
    SBWT_HIP(hipMemGetInfo(&free_b, &total_b));
    // test knobs (tests/test_gpu_rare_paths.py): a smaller budget than the device reports, in bytes; and the
    // allocation of either prefix table made to fail as it would on a fragmented or shared device
    if (const char* e = getenv("SBWT_DEBUG_FREE_BYTES")) {
      const unsigned long long cap = strtoull(e, nullptr, 10);
      if (cap < free_b) free_b = (size_t)cap;
    }
    const char* fail_alloc = getenv("SBWT_DEBUG_FAIL_TABLE_ALLOC");  // "hash", "kmer" or "both"
    const bool fail_hash = fail_alloc && (!strcmp(fail_alloc, "hash") || !strcmp(fail_alloc, "both"));
    const bool fail_kmer = fail_alloc && (!strcmp(fail_alloc, "kmer") || !strcmp(fail_alloc, "both"));
    constexpr uint64_t kKmerHeadroom = 16ull << 30;  // batches, scratch; the raw tables still resident here (12 GB at GRCh38 size) are freed after the load
    // hash table of located order-20 prefixes (device_layout.h): 1.6 slots per suffix, on indexes of 2^22 suffixes
    // or more when it fits next to an order-16 direct table; SBWT_HASH_ORDER overrides (0 = none)
    uint32_t H = raw.n >= (1ull << 22) ? 20u : 0u;
    if (const char* e = getenv("SBWT_HASH_ORDER")) {
      const int want = atoi(e);
      if (want == 0 || (want > kKmer && want <= (int)kHashMaxOrder)) H = (uint32_t)want;
    }
    uint64_t n_buckets = 0;
    uint32_t tag_bits = 0;
    bool with_text = false;
    uint64_t slot_bytes = 8;
    if (H) {
      n_buckets = (raw.n * 8 / 5 + kHashSlots - 1) / kHashSlots;  // 1.6 slots per suffix
      n_buckets = (n_buckets + 7) & ~7ull;                        // whole groups of eight buckets, four 128-byte lines (hash_group)
      // a tag of at most 16 bits must tell the values of one bucket apart: 2^(tag_bits - 1) >= 4^H / n_buckets
      if (2 * H > 15 && n_buckets < (1ull << (2 * H - 15))) n_buckets = 1ull << (2 * H - 15);
      tag_bits = 1;
      while ((n_buckets << (tag_bits - 1)) < (1ull << (2 * H))) ++tag_bits;
      const bool big = raw.n >= (1ull << 28);
      const uint64_t direct16 = big ? (8ull << 32) : 0;  // what an order-16 direct table takes (34 GB), an order-15 one (9 GB)
      const uint64_t direct15 = big ? (8ull << 30) : 0;
      const uint64_t hash8 = n_buckets * (8 * kHashSlots), hash16 = n_buckets * (16 * kHashSlots);
      // 16-byte slots (with the 32 bases of text behind the prefix) when they fit next to an order-15 direct table,
      // else 8-byte slots next to an order-16 one, else no hash table; SBWT_HASH_TEXT=0/1 overrides the first choice
      with_text = hash16 + direct15 + kKmerHeadroom <= (uint64_t)free_b;
      if (const char* e = getenv("SBWT_HASH_TEXT")) with_text = atoi(e) != 0 && hash16 + kKmerHeadroom <= (uint64_t)free_b;
      if (!with_text && hash8 + direct16 + kKmerHeadroom > (uint64_t)free_b) H = 0;
      slot_bytes = with_text ? 16 : 8;
      if (raw.n + 256 >= kPosMask) H = 0;  // the high word of a slot must never be all ones (that marks an empty slot)
    }
    const uint32_t max_direct = H ? 16u : kMaxKmerOrder;
    for (uint32_t k = kKmer + 1; k <= max_direct; ++k) {
      const uint64_t entries = 1ull << (2 * k);
      const uint64_t other = H ? n_buckets * (slot_bytes * kHashSlots) : 0;
      if (entries > 16 * raw.n || entries * 8 + other + kKmerHeadroom > (uint64_t)free_b) break;
      K = k;
    }
    if (const char* e = getenv("SBWT_KMER_ORDER")) {
      const int want = atoi(e);
      if (want >= kKmer && want <= (int)kMaxKmerOrder) K = (uint32_t)want;
    }
    if (getenv("SBWT_DEBUG"))
      fprintf(stderr, "[sbwt debug] device memory: %.1f GB free of %.1f GB before the prefix tables; direct order %u (%.1f GB), hash order %u (%.1f GB, %u tag bits)\n",
              free_b / 1e9, total_b / 1e9, K, K > (uint32_t)kKmer ? 8.0 * (double)(1ull << (2 * K)) / 1e9 : 0.0, H,
              (double)n_buckets * (slot_bytes * kHashSlots) / 1e9, tag_bits);
    d.hash_tab = nullptr;
    d.hash_order = 0;
    d.hash_buckets = 0;
    d.hash_tag_bits = 0;
    d.hash_text = 0;
    d.hash_group = 1;
    d.hash_complete = 0;
    // both prefix tables are optional: an allocation that fails (fragmented or shared device memory) leaves the
    // index without that table, slower and with the same results
    uint64_t* htab = nullptr;
    if (H && (fail_hash || dev_alloc(ix, &htab, n_buckets * kHashSlots * (slot_bytes / 8)))) {
      (void)hipGetLastError();
      H = 0;
    }
    if (H) {
      DevTemp<unsigned long long> d_counts;  // [0] stored, [1] left out, [2] markers, [3] markers left out, [4] uncovered rows, [5] failed checks, [6] twin pairs
      DevTemp<uint64_t> d_uncovered;
      SBWT_HIP(d_counts.alloc(8));
      // rows k_hash_build cannot account for: a handful on a fully sorted index, every row of an LF-inconsistent region
      // on an index whose suffix order stops at sort_len (the reference's own builder): room for a sixteenth of the rows
      // where the device has it, else for 65,536 (more uncovered rows than the list holds: no proof, same results)
      uint64_t uncovered_cap = std::min<uint64_t>(raw.n / 16 + 65536, 1ull << 28);
      if (d_uncovered.alloc(uncovered_cap) != hipSuccess) {
        (void)hipGetLastError();
        uncovered_cap = 65536;
        SBWT_HIP(d_uncovered.alloc(uncovered_cap));
      }
      SBWT_HIP(hipMemset(d_counts, 0, 64));
      SBWT_HIP(hipMemset(htab, 0xff, n_buckets * (slot_bytes * kHashSlots)));
      // with text a prefix may sit in any bucket of its group (hash_insert): as many buckets as the tag can tell apart,
      // eight at most (2^tag_bits >= group * 4^H / n_buckets; a tag of at most 16 bits leaves 8 bits to `run`);
      // SBWT_HASH_GROUP (1, 2, 4, 8) lowers it
      uint32_t group = 1;
      if (with_text) {
        uint32_t want = 8;
        if (const char* e = getenv("SBWT_HASH_GROUP")) want = (uint32_t)atoi(e);
        for (uint32_t g = 2; g <= 8 && g <= want && n_buckets % g == 0; g *= 2) {
          uint32_t t = tag_bits;
          while ((n_buckets << t) < ((uint64_t)g << (2 * H))) ++t;
          if (t > 16) break;
          group = g;
          tag_bits = t;
        }
      }
      // twin slots flag themselves in the top bit of the position field: never set by a position (n < 2^39 - 2^8)
      const uint32_t twins = with_text && raw.n < (1ull << 39) - 256 && !getenv("SBWT_NO_TWINS") ? 1u : 0u;
      k_hash_build<<<grid_for(raw.n, 256), 256>>>(d, H, htab, n_buckets, tag_bits, with_text ? 1u : 0u, group, twins, d_counts, d_uncovered, uncovered_cap);
      SBWT_LAUNCHED();
      unsigned long long h_counts[8];
      SBWT_HIP(hipMemcpy(h_counts, d_counts, 64, hipMemcpyDeviceToHost));
      bool complete = with_text && h_counts[4] <= uncovered_cap && !getenv("SBWT_NO_HASH_COMPLETE");
      if (complete) {
        k_hash_checks<<<grid_for(kNumKmer, 256), 256>>>(d, d_counts);
        SBWT_LAUNCHED();
        if (h_counts[4]) {
          k_hash_cover<<<grid_for(h_counts[4], 64), 64>>>(d, H, htab, n_buckets, tag_bits, group, d_uncovered, (uint64_t)h_counts[4], d_counts);
          SBWT_LAUNCHED();
        }
        SBWT_HIP(hipMemcpy(h_counts, d_counts, 64, hipMemcpyDeviceToHost));
        complete = h_counts[5] == 0;
      }
      if (getenv("SBWT_DEBUG"))
        fprintf(stderr, "[sbwt debug] hash table: groups of %u buckets, %u tag bits; %llu prefixes stored, %llu left out (buckets full); %llu markers, %llu left out; %llu pairs of twin slots; %llu rows covered backwards, %llu failed checks: a miss %s\n",
                group, tag_bits, h_counts[0], h_counts[1], h_counts[2], h_counts[3], h_counts[6], h_counts[4], h_counts[5], complete ? "proves absence" : "proves nothing");
      d.hash_tab = htab;
      d.hash_order = H;
      d.hash_buckets = n_buckets;
      d.hash_tag_bits = tag_bits;
      d.hash_text = with_text ? 1u : 0u;
      d.hash_group = group;
      d.hash_complete = complete ? 1u : 0u;
    }
    mark("hash_tab");
    d.kmer_order = K;
    d.kmer_tab = nullptr;
    uint64_t* tab = nullptr;
    if (K > (uint32_t)kKmer && (fail_kmer || dev_alloc(ix, &tab, 1ull << (2 * K)))) {
      (void)hipGetLastError();
      K = kKmer;
      d.kmer_order = K;
    }
    if (K > (uint32_t)kKmer) {
      k_kmer_table<<<grid_for(1ull << (2 * K), 256), 256>>>(d, K, tab);
      SBWT_LAUNCHED();
      d.kmer_tab = tab;
    }
  }
  SBWT_HIP(hipDeviceSynchronize());
  mark("kmer_tab");
  return 0;
}

int upload_chr_tables(sbwt_index* ix) {
  const ChrTables& t = ix->chr;
  uint64_t *cs = nullptr, *nk = nullptr, *nv = nullptr;
  if (int rc = dev_alloc(ix, &cs, t.start.size())) return rc;
  if (int rc = dev_alloc(ix, &nk, t.npos_key.size())) return rc;
  if (int rc = dev_alloc(ix, &nv, t.npos_val.size())) return rc;
  SBWT_HIP(hipMemcpy(cs, t.start.data(), t.start.size() * 8, hipMemcpyHostToDevice));
  SBWT_HIP(hipMemcpy(nk, t.npos_key.data(), t.npos_key.size() * 8, hipMemcpyHostToDevice));
  SBWT_HIP(hipMemcpy(nv, t.npos_val.data(), t.npos_val.size() * 8, hipMemcpyHostToDevice));
  ix->d.chr_start = cs;
  ix->d.n_chr = (uint32_t)t.start.size();
  ix->d.npos_key = nk;
  ix->d.npos_val = nv;
  ix->d.n_npos = (uint32_t)t.npos_key.size();
  // upstream compares the chromosome NAMES of a hit's two ends (searcher.hpp:526): equal names are one class
  ix->d.chr_class = nullptr;
  std::vector<uint32_t> cls(t.names.size());
  bool repeated = false;
  std::unordered_map<std::string, uint32_t> first_of;
  for (size_t k = 0; k < t.names.size(); ++k) {
    auto it = first_of.emplace(t.names[k], (uint32_t)k);
    cls[k] = it.first->second;
    if (!it.second) repeated = true;
  }
  if (repeated) {
    uint32_t* cc = nullptr;
    if (int rc = dev_alloc(ix, &cc, cls.size())) return rc;
    SBWT_HIP(hipMemcpy(cc, cls.data(), cls.size() * 4, hipMemcpyHostToDevice));
    ix->d.chr_class = cc;
  }
  return 0;
}

}  // namespace sbwt

sbwt_index::~sbwt_index() {
  int cur = 0;
  if (hipGetDevice(&cur) == hipSuccess) {
    hipSetDevice(device);
    for (void* p : allocs) hipFree(p);
    hipSetDevice(cur);
  }
}

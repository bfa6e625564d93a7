// rank_device.h -- device-side rank primitives over the HBM layout of device_layout.h.
// gfx950 only (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sbwt_amd.h"
#include "device_layout.h"

namespace sbwt {

// bits [0, o) set; o in [0, 64]
__device__ __forceinline__ uint64_t below(uint32_t o) { return o >= 64 ? ~0ull : ((1ull << o) - 1ull); }

// positions of a 64-position block whose 2-bit code equals c
__device__ __forceinline__ uint64_t match1(uint64_t plane_lo, uint64_t plane_hi, uint32_t c) {
  return ((c & 1) ? plane_lo : ~plane_lo) & ((c & 2) ? plane_hi : ~plane_hi);
}

// value held by the other lane of the pair (lanes 2k, 2k+1): DPP quad_perm [1,0,3,2], full VALU rate
__device__ __forceinline__ uint32_t pair_swap(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);
}
__device__ __forceinline__ uint64_t pair_swap64(uint64_t v) {
  uint32_t lo = pair_swap((uint32_t)v), hi = pair_swap((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

// Aligner_table::get_occ_using_jbwt(i, c) + c_function[c]  (searcher.hpp:104), one lane,
// one 64-byte block: loads cnt[c] (8 B) and the two planes (16 B) of block i >> 6.
__device__ __forceinline__ uint64_t lf1(const DeviceIndex& ix, uint64_t i, uint32_t c) {
  const Block1* b = ix.blocks1 + (i >> 6);
  const uint64_t cnt = b->cnt[c];
  const uint4 pl = *reinterpret_cast<const uint4*>(&b->plane_lo);
  const uint64_t plo = ((uint64_t)pl.y << 32) | pl.x, phi = ((uint64_t)pl.w << 32) | pl.z;
  uint64_t r = cnt + (uint64_t)__popcll(match1(plo, phi, c) & below((uint32_t)(i & 63)));
  // '$' is stored as 'A': the upstream rank subtracts it back for i > first_location
  if (c == 0 && i > ix.first_location) r -= 1;
  return r;
}

// One Block2 as seen by one read: positions whose code matches, and the count before the block
// (K2[code] + superblock absolute + block relative).  rank2(code, i) = base + popcount(mask & below(i & 63)).
struct Step2 {
  uint64_t mask;
  uint64_t base;
};
__device__ __forceinline__ Step2 load_step2(const DeviceIndex& ix, uint64_t blk, uint32_t code) {
  const Block2* b = ix.blocks2 + blk;
  const uint4 p01 = *reinterpret_cast<const uint4*>(&b->plane[0]);  // bits of y
  const uint4 p23 = *reinterpret_cast<const uint4*>(&b->plane[2]);  // bits of x
  const uint32_t cn = b->cnt[code];
  const uint64_t sb = ix.sb_abs[(blk >> (kSuperShift - 6)) * 16 + code];
  const uint64_t q0 = ((uint64_t)p01.y << 32) | p01.x, q1 = ((uint64_t)p01.w << 32) | p01.z;
  const uint64_t q2 = ((uint64_t)p23.y << 32) | p23.x, q3 = ((uint64_t)p23.w << 32) | p23.z;
  Step2 s;
  s.mask = ((code & 1) ? q0 : ~q0) & ((code & 2) ? q1 : ~q1) & ((code & 4) ? q2 : ~q2) & ((code & 8) ? q3 : ~q3);
  s.base = sb + cn;
  return s;
}
__device__ __forceinline__ uint64_t rank2_correction(const DeviceIndex& ix, uint32_t code, uint64_t i) {
  return (uint64_t)(ix.special_pos[0] < i && code == ix.special_code[0]) +
         (uint64_t)(ix.special_pos[1] < i && code == ix.special_code[1]);
}

// exec_exact_match (searcher.hpp:102-105) for the two bases read[j], read[j+1] at once; `two` = their packed codes.
// Returns false when the interval died: lo/hi then hold what the upstream loop stops with (it stops after the
// FIRST base if that one already emptied the interval, searcher.hpp:406, so that base is replayed alone).
__device__ __forceinline__ bool step_two_bases(const DeviceIndex& ix, uint32_t two, uint64_t& lo, uint64_t& hi) {
  two ^= 15u;                                            // complement: the query is RC(read)
  const uint32_t code = ((two & 3u) << 2) | (two >> 2);  // x = first prepended base, y = second
  const uint64_t b_lo = lo >> 6;
  const bool same = ((hi - 1) >> 6) == b_lo;             // both ends in one block: one access serves both
  const Step2 s1 = load_step2(ix, b_lo, code);
  uint64_t nlo = s1.base + (uint64_t)__popcll(s1.mask & below((uint32_t)(lo & 63)));
  uint64_t nhi;
  if (same) {
    nhi = s1.base + (uint64_t)__popcll(s1.mask & below((uint32_t)(hi - (b_lo << 6))));
  } else {
    const Step2 s2 = load_step2(ix, hi >> 6, code);
    nhi = s2.base + (uint64_t)__popcll(s2.mask & below((uint32_t)(hi & 63)));
  }
  nlo -= rank2_correction(ix, code, lo);
  nhi -= rank2_correction(ix, code, hi);
  if (nlo >= nhi) {
    const uint64_t l1 = lf1(ix, lo, code >> 2), h1 = lf1(ix, hi, code >> 2);
    if (l1 >= h1) { lo = l1; hi = h1; } else { lo = nlo; hi = nhi; }
    return false;
  }
  lo = nlo;
  hi = nhi;
  return true;
}

// std::map::upper_bound(x) followed by advance(-1) (searcher.hpp:519-533) over a sorted key array
// whose first key is 0
__device__ __forceinline__ uint32_t upper_bound_minus1(const uint64_t* __restrict__ keys, uint32_t n, uint64_t x) {
  // binary lifting: floor(log2 n) + 1 steps whatever x is, so a wave runs them without diverging
  uint32_t cnt = 0;  // keys known to be <= x
  for (uint32_t step = n ? 1u << (31 - __clz((int)n)) : 0u; step; step >>= 1) {
    const uint32_t c = cnt + step;
    if (c <= n && keys[c - 1u] <= x) cnt = c;
  }
  return cnt - 1u;  // keys[0] = 0 <= x: cnt >= 1
}

// The same for four keys at once, as fixed-length binary lifting: the steps of the four searches interleave, so a
// step costs one table latency (LDS or cache), not four, and no lane diverges.
__device__ __forceinline__ void upper_bound_minus1_x4(const uint64_t* __restrict__ keys, uint32_t n, const uint64_t (&x)[4],
                                                      uint32_t (&out)[4]) {
  uint32_t cnt[4] = {0, 0, 0, 0};  // keys known to be <= x
  for (uint32_t step = n ? 1u << (31 - __clz((int)n)) : 0u; step; step >>= 1) {
    uint64_t v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = keys[(cnt[k] + step <= n ? cnt[k] + step : n) - 1u];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (cnt[k] + step <= n && v[k] <= x[k]) cnt[k] += step;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) out[k] = cnt[k] - 1u;  // keys[0] = 0 <= x: cnt >= 1
}

// searcher.hpp:504-518: orientation of a text position; false when the strand option drops it
__device__ __forceinline__ bool push_position_orient(const DeviceIndex& ix, uint64_t& position, uint32_t L, int strand, bool& is_rc) {
  is_rc = false;
  if (position >= ix.real_size) {
    is_rc = true;
    position = ix.real_size * 2 - position - L;
  }
  return !((strand == 0 && !is_rc) || (strand == 1 && is_rc));
}

// searcher.hpp:519-539 once upper_bound - 1 of the first base is known in both maps (c1, n1)
__device__ __forceinline__ bool push_position_finish(const DeviceIndex& ix, const uint64_t* chr, const uint64_t* npos_key,
                                                     const uint64_t* npos_val, uint64_t position, uint32_t L, bool is_rc,
                                                     uint32_t c1, uint32_t n1, sbwt_hit* out) {
  const uint64_t last = position + L - 1;
  if (last >= position) {
    if (c1 + 1 < ix.n_chr && chr[c1 + 1] <= last) {
      if (!ix.chr_class) return false;
      const uint32_t c3 = upper_bound_minus1(chr, ix.n_chr, last);
      if (ix.chr_class[c1] != ix.chr_class[c3]) return false;
    }
    if (n1 + 1 < ix.n_npos && npos_key[n1 + 1] <= last) return false;
  } else {
    const uint32_t c3 = upper_bound_minus1(chr, ix.n_chr, last);
    if (c1 != c3 && (!ix.chr_class || ix.chr_class[c1] != ix.chr_class[c3])) return false;
    if (n1 != upper_bound_minus1(npos_key, ix.n_npos, last)) return false;
  }
  out->pos0 = position - chr[c1] + npos_val[n1];
  out->chr_id = c1;
  out->is_rc = is_rc ? 1u : 0u;
  return true;
}

__device__ __forceinline__ bool push_position_keys(const DeviceIndex& ix, const uint64_t* chr, const uint64_t* npos_key,
                                                   const uint64_t* npos_val, uint64_t position, uint32_t L, int strand,
                                                   sbwt_hit* out) {
  bool is_rc;
  if (!push_position_orient(ix, position, L, strand, is_rc)) return false;
  const uint32_t c1 = upper_bound_minus1(chr, ix.n_chr, position);
  const uint32_t n1 = upper_bound_minus1(npos_key, ix.n_npos, position);
  return push_position_finish(ix, chr, npos_key, npos_val, position, L, is_rc, c1, n1, out);
}
// A ONE-row interval whose position the filters drop (searcher.hpp:516-533).  Upstream's walk goes on below that row
// when the row was sampled itself: the break of searcher.hpp:440 leaves only the loop, and find_possible runs while the
// result is empty (searcher.hpp:446-449, 464).  Where the positions below p are LF-consistent over the sampling distance
// (`run` of them: device_index.hip, k_verify_rows) the walk meets no second sampled row before it gives up
// (searcher.hpp:462); on tables with unordered ties it can, and then reports THAT position: such a read takes the
// depth-first locate like a multi-hit read (one row, one slot).
__device__ __forceinline__ bool walk_may_go_on(const DeviceIndex& ix, uint64_t p, uint64_t run) {
  const uint64_t need = p + 1 < ix.interval ? p + 1 : ix.interval;
  return run < need;
}
__device__ __forceinline__ bool push_position_lane(const DeviceIndex& ix, uint64_t position, uint32_t L, int strand,
                                                   sbwt_hit* out) {
  return push_position_keys(ix, ix.chr_start, ix.npos_key, ix.npos_val, position, L, strand, out);
}

// results[].flags of a read shorter than 12 or longer than its record (sbwt_pack_reads never produces one)
constexpr uint32_t kFlagInvalid = 0x80000000u;

}  // namespace sbwt

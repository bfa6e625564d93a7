// rank_device.h -- device-side rank primitives over the HBM layout of device_layout.h.
// gfx950 only (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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

// std::map::upper_bound(x) followed by advance(-1) (searcher.hpp:519-533) over a sorted key array
// whose first key is 0
__device__ __forceinline__ uint32_t upper_bound_minus1(const uint64_t* __restrict__ keys, uint32_t n, uint64_t x) {
  uint32_t l = 0, h = n;
  while (l < h) {
    uint32_t m = (l + h) >> 1;
    if (keys[m] <= x) l = m + 1; else h = m;
  }
  return l - 1;
}

}  // namespace sbwt

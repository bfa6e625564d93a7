// builder.hip -- index construction on the GPU.
//
// Stands in for Indexer::build (indexer.hpp:137-229): Genome_pre_handler::run_handler
// (genome_pre_handler.hpp:175-310), the depth-`sort_len` suffix ordering of Split_sort / Multikey_quicksort
// (split_sort.hpp:148-340, mkq_sort.hpp:278-358) and Aligner_table::createAllTable / saveTable
// (bodies binary-only; table definitions per SURVEY.md 8b, pinned by tests/golden/golden.json).
//
// Suffix ordering: text = fwd + revcomp(fwd) + '$' as 2-bit codes.  Every suffix gets a 63-bit key made of
// 27 base-5 digits ('$'/past-the-end = 0, A..T = 1..4), so '$' sorts first with no special case and a key
// that contains a 0 digit is unique.  Suffixes are partitioned by their leading digits, each partition is
// radix-sorted (rocPRIM; this is offline plumbing, not the hot path), and groups of equal keys are refined
// 27 bases at a time until they are resolved or 1080 >= sort_len bases deep -- a full suffix order
// wherever the text allows one, which is a valid sort_len-order for every sort_len <= 1080.  (Deeper than upstream's
// default 256 on purpose: suffixes left tied sit in position order, which is not LF-consistent where a tie ends, and
// reads whose rows fall there take the slow depth-first locate; 1080 leaves only long exact repeats tied.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <cstring>
using std::memset;
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/sbwt_amd.h"
#include "index_impl.hpp"

namespace sbwt {

constexpr int kDigits = 27;       // 5^27 < 2^63
constexpr int kMaxDepth = 1080;   // refinement stops here (>= default sort_len 256; deep enough to order everything but long exact repeats)
constexpr uint64_t kMaxPartition = 1ull << 29;

struct Text {
  const uint32_t* w;  // 16 bases per word, base j at bits 2*(j%16); zero padded by >= 4 words
  uint64_t m;         // number of real bases (2G); '$' sits at position m
};

__device__ __forceinline__ uint32_t base_at(const Text& T, uint64_t p) {
  return (T.w[p >> 4] >> (2 * (uint32_t)(p & 15))) & 3u;
}

// 27 base-5 digits of the suffix starting at q0 (most significant first)
__device__ __forceinline__ uint64_t key5(const Text& T, uint64_t q0) {
  if (q0 >= T.m) return 0;
  const uint64_t avail = T.m - q0;
  const int nd = avail >= (uint64_t)kDigits ? kDigits : (int)avail;
  const uint64_t wi = q0 >> 4;
  const uint32_t sh = 2 * (uint32_t)(q0 & 15);
  const uint64_t lo = (uint64_t)T.w[wi] | ((uint64_t)T.w[wi + 1] << 32);
  const uint64_t hi = T.w[wi + 2];
  const uint64_t bits = sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
  uint64_t key = 0;
#pragma unroll
  for (int t = 0; t < kDigits; ++t) {
    const uint64_t d = t < nd ? ((bits >> (2 * t)) & 3ull) + 1ull : 0ull;
    key = key * 5ull + d;
  }
  return key;
}

// leading D digits of suffix p as a number in [0, 5^D)
__device__ __forceinline__ uint32_t prefix_digits(const Text& T, uint64_t p, int D) {
  uint32_t v = 0;
  for (int t = 0; t < D; ++t) {
    const uint64_t q = p + (uint64_t)t;
    v = v * 5u + (q < T.m ? base_at(T, q) + 1u : 0u);
  }
  return v;
}

__global__ __launch_bounds__(256) void k_make_text(const uint32_t* __restrict__ fwd, uint64_t g, uint32_t* __restrict__ text,
                                                   uint64_t n_words) {
  const uint64_t w = global_thread();
  if (w >= n_words) return;
  const uint64_t m = 2 * g;
  uint32_t out = 0;
  for (int t = 0; t < 16; ++t) {
    const uint64_t j = w * 16 + t;
    uint32_t c = 0;
    if (j < g) c = (fwd[j >> 4] >> (2 * (uint32_t)(j & 15))) & 3u;
    else if (j < m) {  // genome_pre_handler.hpp:210-228: reverse complement appended
      const uint64_t s = m - 1 - j;
      c = 3u - ((fwd[s >> 4] >> (2 * (uint32_t)(s & 15))) & 3u);
    }
    out |= c << (2 * t);
  }
  text[w] = out;
}

// histogram of the leading D digits over the positions [begin, end)
__global__ __launch_bounds__(256) void k_prefix_hist(Text T, uint64_t begin, uint64_t end, int D, uint32_t n_part,
                                                     unsigned long long* hist) {
  __shared__ unsigned int h[15625];  // 5^6
  for (uint32_t k = threadIdx.x; k < n_part; k += blockDim.x) h[k] = 0;
  __syncthreads();
  // each workgroup sweeps a contiguous slab so that the global atomics stay few
  const uint64_t slab = 256ull * 64;
  const uint64_t a = begin + (uint64_t)blockIdx.x * slab;
  for (uint64_t p = a + threadIdx.x; p < a + slab && p < end; p += blockDim.x) atomicAdd(&h[prefix_digits(T, p, D)], 1u);
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < n_part; k += blockDim.x)
    if (h[k]) atomicAdd(&hist[k], (unsigned long long)h[k]);
}

struct PartitionOf {  // partition id (leading D digits, < 5^6 < 2^16) of the suffix at a position
  Text T;
  int D;
  __device__ uint16_t operator()(const uint64_t& p) const { return (uint16_t)prefix_digits(T, p, D); }
};

__global__ __launch_bounds__(256) void k_keys(Text T, const uint64_t* __restrict__ pos, uint64_t cnt, uint64_t off,
                                              uint64_t* __restrict__ keys) {
  const uint64_t t = global_thread();
  if (t < cnt) keys[t] = key5(T, pos[t] + off);
}

__global__ __launch_bounds__(256) void k_hash_keys(const uint64_t* __restrict__ pos, uint64_t cnt, uint64_t* __restrict__ keys) {
  const uint64_t t = global_thread();
  if (t >= cnt) return;
  uint64_t z = pos[t] + 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  keys[t] = z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void k_iota64(uint64_t* a, uint64_t cnt) {
  const uint64_t t = global_thread();
  if (t < cnt) a[t] = t;
}
__global__ __launch_bounds__(256) void k_iota(uint32_t* a, uint64_t cnt) {
  const uint64_t t = global_thread();
  if (t < cnt) a[t] = (uint32_t)t;
}

template <typename T>
__global__ __launch_bounds__(256) void k_gather(const T* __restrict__ src, const uint32_t* __restrict__ idx, uint64_t cnt,
                                                T* __restrict__ dst) {
  const uint64_t t = global_thread();
  if (t < cnt) dst[t] = src[idx[t]];
}

// mark[t] = t if element t starts a run of equal (gid, key) else 0; gid may be null (single group)
__global__ __launch_bounds__(256) void k_mark_heads(const uint32_t* __restrict__ gid, const uint64_t* __restrict__ key,
                                                    uint64_t cnt, uint32_t* __restrict__ gid_head_idx,
                                                    uint32_t* __restrict__ key_head_flag) {
  const uint64_t t = global_thread();
  if (t >= cnt) return;
  const bool gh = t == 0 || (gid && gid[t] != gid[t - 1]);
  const bool kh = gh || key[t] != key[t - 1];
  gid_head_idx[t] = gh ? (uint32_t)t : 0u;
  key_head_flag[t] = kh ? 1u : 0u;
}

// slot = gid + (t - start of its gid run); scatter the position; mark the slot of new-group heads
__global__ __launch_bounds__(256) void k_slots(const uint32_t* __restrict__ gid, const uint32_t* __restrict__ gid_start,
                                               const uint32_t* __restrict__ key_head_flag, const uint64_t* __restrict__ pos,
                                               uint64_t cnt, uint64_t* __restrict__ sa_part,
                                               uint32_t* __restrict__ head_slot) {
  const uint64_t t = global_thread();
  if (t >= cnt) return;
  const uint32_t slot = (gid ? gid[t] : 0u) + ((uint32_t)t - gid_start[t]);
  if (sa_part) sa_part[slot] = pos[t];
  head_slot[t] = key_head_flag[t] ? slot : 0u;
}

// tie[t] = element belongs to a (gid,key) run of length >= 2
__global__ __launch_bounds__(256) void k_tie_flags(const uint32_t* __restrict__ key_head_flag, uint64_t cnt,
                                                   uint8_t* __restrict__ tie) {
  const uint64_t t = global_thread();
  if (t >= cnt) return;
  const bool single = key_head_flag[t] && (t + 1 == cnt || key_head_flag[t + 1]);
  tie[t] = single ? 0 : 1;
}

// debug: order violations on the first 27 digits and the position checksum
__global__ __launch_bounds__(256) void k_check_sa(Text T, const uint64_t* __restrict__ sa, uint64_t n,
                                                  unsigned long long* out) {
  const uint64_t i = global_thread();
  unsigned long long v = 0;  // lanes past the end stay in the wave reduction with 0
  if (i < n) {
    const uint64_t s = sa[i];
    if (s > T.m) atomicAdd(&out[1], 1ull);
    if (i > 0 && key5(T, sa[i - 1]) > key5(T, s)) atomicAdd(&out[0], 1ull);
    v = s;
  }
  for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d);
  if ((threadIdx.x & 63) == 0) atomicAdd(&out[2], v);
}

// ---- table emission ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_emit_bwt(Text T, const uint64_t* __restrict__ sa, uint64_t n, uint64_t interval,
                                                  uint8_t* __restrict__ jbwt, uint8_t* __restrict__ fbwt,
                                                  uint64_t* first_location) {
  const uint64_t byte = global_thread();
  if (byte >= n / 4 + 1) return;
  uint32_t out = 0;
  for (int q = 0; q < 4; ++q) {
    const uint64_t i = byte * 4 + q;
    uint32_t c = 0;  // padding is 'A' (JBit::last_push_back, jbit.hpp:183-192)
    if (i < n) {
      const uint64_t s = sa[i];
      if (s == 0) *first_location = i;  // BWT char is '$', stored as 'A'
      else c = base_at(T, s - 1);
      fbwt[i] = (uint8_t)(s % interval);
    }
    out |= c << (6 - 2 * q);  // base 0 of a byte in bits 7..6 (jbit.hpp:193-206)
  }
  jbwt[byte] = (uint8_t)out;
}

// per checkpoint block k (positions [k*interval, (k+1)*interval)): counts of A,C,G,T ('$' as A) and of
// sampled positions
__global__ __launch_bounds__(256) void k_block_counts(const uint8_t* __restrict__ jbwt, const uint8_t* __restrict__ fbwt,
                                                      uint64_t n, uint64_t interval, uint64_t n_ckpt,
                                                      uint64_t* __restrict__ cA, uint64_t* __restrict__ cC,
                                                      uint64_t* __restrict__ cG, uint64_t* __restrict__ cT,
                                                      uint64_t* __restrict__ cS) {
  const uint64_t k = global_thread();
  if (k >= n_ckpt) return;
  uint64_t c[4] = {0, 0, 0, 0}, s = 0;
  const uint64_t a = k * interval, b = a + interval < n ? a + interval : n;
  for (uint64_t i = a; i < b; ++i) {
    c[(jbwt[i >> 2] >> (6 - 2 * (uint32_t)(i & 3))) & 3u]++;
    s += fbwt[i] == 0;
  }
  cA[k] = c[0]; cC[k] = c[1]; cG[k] = c[2]; cT[k] = c[3]; cS[k] = s;
}

__global__ __launch_bounds__(256) void k_emit_loc(const uint64_t* __restrict__ sa, const uint8_t* __restrict__ fbwt,
                                                  uint64_t n, uint64_t interval, uint64_t n_ckpt,
                                                  const uint64_t* __restrict__ base, uint64_t* __restrict__ loc) {
  const uint64_t k = global_thread();
  if (k >= n_ckpt) return;
  uint64_t o = base[k];
  const uint64_t a = k * interval, b = a + interval < n ? a + interval : n;
  for (uint64_t i = a; i < b; ++i)
    if (fbwt[i] == 0) {
      loc[2 * o] = i;
      loc[2 * o + 1] = sa[i];
      ++o;
    }
}

// c_functions, step 1: first SA index of every 12-mer that occurs with 12 real bases (SURVEY.md 8b rule)
__global__ __launch_bounds__(256) void k_kmer_starts(Text T, const uint64_t* __restrict__ sa, uint64_t n,
                                                     unsigned long long* __restrict__ starts) {
  const uint64_t i = global_thread();
  if (i >= n) return;
  const uint64_t s = sa[i];
  if (s + kKmer > T.m) return;
  uint64_t code = 0;
  for (int k = 0; k < kKmer; ++k) code = code * 4 + base_at(T, s + k);
  if (i > 0) {
    const uint64_t sp = sa[i - 1];
    if (sp + kKmer <= T.m) {
      uint64_t cp = 0;
      for (int k = 0; k < kKmer; ++k) cp = cp * 4 + base_at(T, sp + k);
      if (cp == code) return;  // not the head of its run
    }
  }
  atomicMin(&starts[code], (unsigned long long)i);
}

// step 2 (createAllTable, object code .text+0x40a0): last entry = n, every absent entry takes the next
// larger present one; entry 0 belongs to the '$' suffix (SA index 0).  One workgroup, suffix-min scan.
__global__ __launch_bounds__(1024) void k_kmer_fill(unsigned long long* __restrict__ starts, uint64_t n) {
  __shared__ unsigned long long chunk_min[1024];
  const uint64_t total = kNumKmer + 1;
  const uint64_t per = (total + 1023) / 1024;
  const uint64_t a = threadIdx.x * per, b = a + per < total ? a + per : total;
  if (threadIdx.x == 0) starts[0] = 0;
  if (b == total && a < b) starts[total - 1] = n;
  __syncthreads();
  unsigned long long mn = ~0ull;
  for (uint64_t k = b; k-- > a;) mn = starts[k] < mn ? starts[k] : mn;
  chunk_min[threadIdx.x] = mn;
  __syncthreads();
  unsigned long long carry = ~0ull;
  for (int t = 1023; t > (int)threadIdx.x; --t) carry = chunk_min[t] < carry ? chunk_min[t] : carry;
  for (uint64_t k = b; k-- > a;) {
    const unsigned long long v = starts[k];
    carry = v < carry ? v : carry;
    starts[k] = carry;
  }
}

__global__ __launch_bounds__(256) void k_base_hist(Text T, unsigned long long* hist) {
  __shared__ unsigned int h[4];
  if (threadIdx.x < 4) h[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t w = global_thread();
  const uint64_t n_words = (T.m + 15) / 16;
  if (w < n_words) {
    uint32_t c[4] = {0, 0, 0, 0};
    const uint32_t word = T.w[w];
    for (int t = 0; t < 16; ++t)
      if (w * 16 + t < T.m) c[(word >> (2 * t)) & 3u]++;
    for (int k = 0; k < 4; ++k)
      if (c[k]) atomicAdd(&h[k], c[k]);
  }
  __syncthreads();
  if (threadIdx.x < 4 && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

// ---- host side of the sorter ------------------------------------------------------------------------
static inline dim3 blocks_for(uint64_t items) { return grid_items(items, 256); }

struct Scratch {  // growing rocPRIM temporary storage
  void* p = nullptr;
  size_t bytes = 0;
  ~Scratch() { if (p) (void)hipFree(p); }
  int need(size_t b) {
    if (b <= bytes) return 0;
    if (p) (void)hipFree(p);
    p = nullptr; bytes = 0;
    SBWT_HIP(hipMalloc(&p, b));
    bytes = b;
    return 0;
  }
};

template <typename T>
struct DevBuf {
  T* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int alloc(uint64_t count) {
    if (p) (void)hipFree(p);
    p = nullptr;
    SBWT_HIP(hipMalloc(&p, (count ? count : 1) * sizeof(T)));
    return 0;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; }
};

template <typename K, typename V>
static int sort_pairs(Scratch& tmp, K* kin, K* kout, V* vin, V* vout, uint64_t cnt, unsigned end_bit) {
  size_t bytes = 0;
  SBWT_HIP(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, (size_t)cnt, 0u, end_bit));
  if (int rc = tmp.need(bytes)) return rc;
  SBWT_HIP(rocprim::radix_sort_pairs(tmp.p, bytes, kin, kout, vin, vout, (size_t)cnt, 0u, end_bit));
  return 0;
}
static int max_scan(Scratch& tmp, uint32_t* in, uint32_t* out, uint64_t cnt) {
  size_t bytes = 0;
  SBWT_HIP(rocprim::inclusive_scan(nullptr, bytes, in, out, (size_t)cnt, rocprim::maximum<uint32_t>()));
  if (int rc = tmp.need(bytes)) return rc;
  SBWT_HIP(rocprim::inclusive_scan(tmp.p, bytes, in, out, (size_t)cnt, rocprim::maximum<uint32_t>()));
  return 0;
}
static int excl_sum(Scratch& tmp, uint64_t* in, uint64_t* out, uint64_t cnt) {
  size_t bytes = 0;
  SBWT_HIP(rocprim::exclusive_scan(nullptr, bytes, in, out, (uint64_t)0, (size_t)cnt, rocprim::plus<uint64_t>()));
  if (int rc = tmp.need(bytes)) return rc;
  SBWT_HIP(rocprim::exclusive_scan(tmp.p, bytes, in, out, (uint64_t)0, (size_t)cnt, rocprim::plus<uint64_t>()));
  return 0;
}
template <typename T>
static int select_flagged(Scratch& tmp, T* in, uint8_t* flags, T* out, uint64_t* d_count, uint64_t cnt) {
  size_t bytes = 0;
  SBWT_HIP(rocprim::select(nullptr, bytes, in, flags, out, d_count, (size_t)cnt));
  if (int rc = tmp.need(bytes)) return rc;
  SBWT_HIP(rocprim::select(tmp.p, bytes, in, flags, out, d_count, (size_t)cnt));
  return 0;
}

// Sorts the suffixes pos[0..cnt) (all sharing their leading partition digits) into sa_part[0..cnt).
struct PartitionSorter {
  Scratch tmp;
  DevBuf<uint64_t> key_a, key_b, pos_a, pos_b, key_f;
  DevBuf<uint32_t> u0, u1, u2, u3, u4;
  DevBuf<uint8_t> tie;
  DevBuf<uint64_t> d_count;
  uint64_t cap = 0;

  int reserve(uint64_t c) {
    if (c <= cap) return 0;
    cap = c;
    if (int rc = key_a.alloc(c)) return rc;
    if (int rc = key_b.alloc(c)) return rc;
    if (int rc = pos_a.alloc(c)) return rc;
    if (int rc = pos_b.alloc(c)) return rc;
    if (int rc = key_f.alloc(c)) return rc;
    if (int rc = u0.alloc(c)) return rc;
    if (int rc = u1.alloc(c)) return rc;
    if (int rc = u2.alloc(c)) return rc;
    if (int rc = u3.alloc(c)) return rc;
    if (int rc = u4.alloc(c)) return rc;
    if (int rc = tie.alloc(c)) return rc;
    if (int rc = d_count.alloc(1)) return rc;
    return 0;
  }

  // Test knob SBWT_DEBUG_SORT_DEPTH=<bases>: stop the refinement that deep (rounded up to whole 27-base keys) and
  // leave the suffixes that still tie in a pseudo-random order, as upstream's depth-limited quicksort leaves its
  // ties beyond sort_len in no particular order (PSEUDO_DCS_FOR_SBWT::compare == true, difference_cover.hpp:30-41).
  uint64_t max_depth = (uint64_t)kMaxDepth;
  bool shuffle_ties = false;

  // pos_a holds the partition's positions on entry
  int run(const Text& T, uint64_t cnt, uint64_t* sa_part) {
    if (cnt == 0) return 0;
    const dim3 g = blocks_for(cnt);
    if (shuffle_ties) {
      // the first sort is stable: ties come out in the order the positions go in, so mix that order first
      k_hash_keys<<<g, 256>>>(pos_a.p, cnt, key_a.p);
      if (int rc = sort_pairs(tmp, key_a.p, key_b.p, pos_a.p, pos_b.p, cnt, 64)) return rc;
      SBWT_HIP(hipMemcpyAsync(pos_a.p, pos_b.p, cnt * 8, hipMemcpyDeviceToDevice, 0));
    }
    k_keys<<<g, 256>>>(T, pos_a.p, cnt, 0, key_a.p);
    if (int rc = sort_pairs(tmp, key_a.p, key_b.p, pos_a.p, pos_b.p, cnt, 63)) return rc;
    // sorted: key_b / pos_b.  Everything is one group (gid = null); find runs of equal keys.
    k_mark_heads<<<g, 256>>>(nullptr, key_b.p, cnt, u0.p, u1.p);   // u0 gid-head idx, u1 key-head flag
    if (int rc = max_scan(tmp, u0.p, u2.p, cnt)) return rc;          // u2 = start of gid run (0 everywhere)
    k_slots<<<g, 256>>>(nullptr, u2.p, u1.p, pos_b.p, cnt, sa_part, u3.p);  // u3 = slot of run heads
    if (int rc = max_scan(tmp, u3.p, u4.p, cnt)) return rc;          // u4 = new gid (slot of the run head)
    k_tie_flags<<<g, 256>>>(u1.p, cnt, tie.p);
    // subset of tied elements: (pos, gid)
    if (int rc = select_flagged(tmp, pos_b.p, tie.p, pos_a.p, d_count.p, cnt)) return rc;
    if (int rc = select_flagged(tmp, u4.p, tie.p, u0.p, d_count.p, cnt)) return rc;
    uint64_t ns = 0;
    SBWT_HIP(hipMemcpy(&ns, d_count.p, 8, hipMemcpyDeviceToHost));
    // subset: pos_a (positions), u0 (gid)
    for (uint64_t off = kDigits; ns > 0 && off < max_depth; off += kDigits) {
      const dim3 gs = blocks_for(ns);
      k_keys<<<gs, 256>>>(T, pos_a.p, ns, off, key_a.p);            // key_a = next 27 digits (subset order)
      k_iota<<<gs, 256>>>(u1.p, ns);
      if (int rc = sort_pairs(tmp, key_a.p, key_b.p, u1.p, u2.p, ns, 63)) return rc;   // u2 = order by key
      k_gather<uint32_t><<<gs, 256>>>(u0.p, u2.p, ns, u1.p);        // u1 = gid in key order
      if (int rc = sort_pairs(tmp, u1.p, u3.p, u2.p, u4.p, ns, 32)) return rc;         // stable by gid: u3 gid, u4 order
      k_gather<uint64_t><<<gs, 256>>>(key_a.p, u4.p, ns, key_f.p);  // keys in (gid, key) order
      k_gather<uint64_t><<<gs, 256>>>(pos_a.p, u4.p, ns, pos_b.p);  // positions in (gid, key) order
      k_mark_heads<<<gs, 256>>>(u3.p, key_f.p, ns, u1.p, u2.p);     // u1 gid-head idx, u2 key-head flag
      if (int rc = max_scan(tmp, u1.p, u0.p, ns)) return rc;         // u0 = start of gid run
      k_slots<<<gs, 256>>>(u3.p, u0.p, u2.p, pos_b.p, ns, sa_part, u1.p);  // scatter; u1 = slot of new heads
      if (int rc = max_scan(tmp, u1.p, u4.p, ns)) return rc;         // u4 = new gid
      k_tie_flags<<<gs, 256>>>(u2.p, ns, tie.p);
      if (int rc = select_flagged(tmp, pos_b.p, tie.p, pos_a.p, d_count.p, ns)) return rc;
      if (int rc = select_flagged(tmp, u4.p, tie.p, u0.p, d_count.p, ns)) return rc;
      SBWT_HIP(hipMemcpy(&ns, d_count.p, 8, hipMemcpyDeviceToHost));
    }
    SBWT_HIP(hipGetLastError());
    return 0;
  }
};

// Device-resident reference-format tables, owned.
struct OwnedRaw {
  RawDeviceTables raw;
  std::vector<void*> owned;
  ~OwnedRaw() { for (void* p : owned) if (p) (void)hipFree(p); }
  template <typename T>
  int alloc(T** p, uint64_t count) {
    void* q = nullptr;
    SBWT_HIP(hipMalloc(&q, (count ? count : 1) * sizeof(T)));
    owned.push_back(q);
    *p = static_cast<T*>(q);
    return 0;
  }
  void disown(const void* p) {
    for (auto& q : owned) if (q == p) q = nullptr;
  }
};

// fwd text on the device -> reference-format tables on the device
static int build_raw_tables(const uint32_t* d_fwd, uint64_t g, uint64_t interval, OwnedRaw& R) {
  const uint64_t m = 2 * g, n = m + 1;
  const uint64_t n_words = (m + 15) / 16;
  DevBuf<uint32_t> text;
  if (int rc = text.alloc(n_words + 8)) return rc;
  SBWT_HIP(hipMemset(text.p, 0, (n_words + 8) * 4));
  k_make_text<<<blocks_for(n_words), 256>>>(d_fwd, g, text.p, n_words);
  SBWT_LAUNCHED();
  Text T{text.p, m};

  // partition depth: smallest D with every partition <= kMaxPartition (expected n / 4^D)
  int D = 0;
  while (D < 6 && (n >> (2 * D)) > kMaxPartition / 2) ++D;
  uint32_t n_part = 1;
  for (int k = 0; k < D; ++k) n_part *= 5;
  // Positions grouped by partition.  The text is cut into slabs of 2^30 positions (rocPRIM indexes its input with
  // 32 bits); each slab is radix-sorted ONCE by partition id, stably, so inside a slab every partition's positions
  // are contiguous and ascending, and a partition is the concatenation of its stretch in every slab.
  constexpr uint64_t kSlab = 1ull << 30;
  const uint32_t n_slab = (uint32_t)((n + kSlab - 1) / kSlab);
  DevBuf<unsigned long long> d_hist;
  if (int rc = d_hist.alloc((uint64_t)n_slab * n_part)) return rc;
  SBWT_HIP(hipMemset(d_hist.p, 0, (uint64_t)n_slab * n_part * 8));
  std::vector<unsigned long long> slab_hist((uint64_t)n_slab * n_part, 0), hist(n_part, 0);
  DevBuf<uint64_t> pos_all;
  if (D == 0) hist[0] = slab_hist[0] = n;
  else {
    if (int rc = pos_all.alloc(n)) return rc;
    DevBuf<uint16_t> part_keys;
    if (int rc = part_keys.alloc(std::min<uint64_t>(kSlab, n))) return rc;
    Scratch tmp;
    unsigned bits = 1;
    while ((1u << bits) < n_part) ++bits;
    for (uint32_t s = 0; s < n_slab; ++s) {
      const uint64_t c0 = (uint64_t)s * kSlab, len = std::min<uint64_t>(kSlab, n - c0);
      k_prefix_hist<<<(uint32_t)((len + 256ull * 64 - 1) / (256ull * 64)), 256>>>(T, c0, c0 + len, D, n_part,
                                                                                 d_hist.p + (uint64_t)s * n_part);
      SBWT_LAUNCHED();
      auto keys_in = rocprim::make_transform_iterator(rocprim::counting_iterator<uint64_t>(c0), PartitionOf{T, D});
      rocprim::counting_iterator<uint64_t> vals_in(c0);
      size_t bytes = 0;
      SBWT_HIP(rocprim::radix_sort_pairs(nullptr, bytes, keys_in, part_keys.p, vals_in, pos_all.p + c0, (size_t)len, 0u, bits));
      if (int rc = tmp.need(bytes)) return rc;
      SBWT_HIP(rocprim::radix_sort_pairs(tmp.p, bytes, keys_in, part_keys.p, vals_in, pos_all.p + c0, (size_t)len, 0u, bits));
    }
    SBWT_HIP(hipMemcpy(slab_hist.data(), d_hist.p, slab_hist.size() * 8, hipMemcpyDeviceToHost));
    for (uint32_t s = 0; s < n_slab; ++s)
      for (uint32_t v = 0; v < n_part; ++v) hist[v] += slab_hist[(uint64_t)s * n_part + v];
  }
  uint64_t max_part = 0;
  for (auto h : hist) max_part = std::max<uint64_t>(max_part, h);

  DevBuf<uint64_t> sa;
  if (int rc = sa.alloc(n)) return rc;
  uint64_t base = 0;
  {
    PartitionSorter S;
    if (const char* e = getenv("SBWT_DEBUG_SORT_DEPTH")) {
      const uint64_t want = strtoull(e, nullptr, 10);
      if (want >= 1 && want < (uint64_t)kMaxDepth) {
        S.max_depth = (want + kDigits - 1) / kDigits * kDigits;
        S.shuffle_ties = true;
      }
    }
    if (int rc = S.reserve(max_part)) return rc;
    std::vector<uint64_t> slab_off(n_slab, 0);  // where the next partition starts inside each slab
    for (uint32_t v = 0; v < n_part; ++v) {
      const uint64_t cnt = hist[v];
      if (D == 0) {
        k_iota64<<<blocks_for(n), 256>>>(S.pos_a.p, n);
        SBWT_LAUNCHED();
      } else {
        uint64_t got = 0;
        for (uint32_t s = 0; s < n_slab; ++s) {
          const uint64_t c = slab_hist[(uint64_t)s * n_part + v];
          if (c) SBWT_HIP(hipMemcpyAsync(S.pos_a.p + got, pos_all.p + (uint64_t)s * kSlab + slab_off[s], c * 8,
                                         hipMemcpyDeviceToDevice, 0));
          slab_off[s] += c;
          got += c;
        }
      }
      if (cnt == 0) continue;
      if (int rc = S.run(T, cnt, sa.p + base)) return rc;
      base += cnt;
    }
    SBWT_HIP(hipDeviceSynchronize());
    if (getenv("SBWT_DEBUG")) {
      DevBuf<unsigned long long> chk;
      if (int rc = chk.alloc(3)) return rc;
      SBWT_HIP(hipMemset(chk.p, 0, 24));
      k_check_sa<<<blocks_for(n), 256>>>(T, sa.p, n, chk.p);
      SBWT_LAUNCHED();
      unsigned long long h[3];
      SBWT_HIP(hipMemcpy(h, chk.p, 24, hipMemcpyDeviceToHost));
      unsigned long long want = (unsigned long long)((n & 1) ? ((n - 1) / 2) * n : (n / 2) * (n - 1));
      fprintf(stderr, "[sbwt debug] n=%llu D=%d parts=%u max_part=%llu placed=%llu order_violations=%llu out_of_range=%llu checksum %s\n",
              (unsigned long long)n, D, n_part, (unsigned long long)max_part, (unsigned long long)base, h[0], h[1],
              h[2] == want ? "ok" : "BAD");
    }
  }
  pos_all.release();

  // ---- tables ----
  RawDeviceTables& raw = R.raw;
  raw.interval = interval;
  raw.n = n;
  raw.real_size = g;
  uint8_t *jbwt = nullptr, *fbwt = nullptr;
  if (int rc = R.alloc(&jbwt, n / 4 + 1)) return rc;
  if (int rc = R.alloc(&fbwt, n)) return rc;
  DevBuf<uint64_t> d_first;
  if (int rc = d_first.alloc(1)) return rc;
  k_emit_bwt<<<blocks_for(n / 4 + 1), 256>>>(T, sa.p, n, interval, jbwt, fbwt, d_first.p);
  SBWT_LAUNCHED();
  SBWT_HIP(hipMemcpy(&raw.first_location, d_first.p, 8, hipMemcpyDeviceToHost));
  raw.jbwt = jbwt;
  raw.fbwt = fbwt;

  const uint64_t n_ckpt = n / interval + 1;  // occ_len
  raw.occ_len = n_ckpt;
  uint64_t* occ[4];
  for (int c = 0; c < 4; ++c) if (int rc = R.alloc(&occ[c], n_ckpt)) return rc;
  {
    DevBuf<uint64_t> cnt[5], sbase;
    for (int c = 0; c < 5; ++c) if (int rc = cnt[c].alloc(n_ckpt)) return rc;
    if (int rc = sbase.alloc(n_ckpt)) return rc;
    k_block_counts<<<blocks_for(n_ckpt), 256>>>(jbwt, fbwt, n, interval, n_ckpt, cnt[0].p, cnt[1].p, cnt[2].p, cnt[3].p, cnt[4].p);
    Scratch tmp;
    for (int c = 0; c < 4; ++c) {
      if (int rc = excl_sum(tmp, cnt[c].p, occ[c], n_ckpt)) return rc;
      raw.occ[c] = occ[c];
    }
    if (int rc = excl_sum(tmp, cnt[4].p, sbase.p, n_ckpt)) return rc;
    raw.n_loc = (n - 1) / interval + 1;  // SA values 0, interval, 2*interval, ... <= n-1
    uint64_t* loc = nullptr;
    if (int rc = R.alloc(&loc, raw.n_loc * 2)) return rc;
    k_emit_loc<<<blocks_for(n_ckpt), 256>>>(sa.p, fbwt, n, interval, n_ckpt, sbase.p, loc);
    raw.loc = loc;
  }
  unsigned long long* starts = nullptr;
  if (int rc = R.alloc(&starts, kNumKmer + 1)) return rc;
  SBWT_HIP(hipMemset(starts, 0xff, (kNumKmer + 1) * 8));
  k_kmer_starts<<<blocks_for(n), 256>>>(T, sa.p, n, starts);
  SBWT_LAUNCHED();
  k_kmer_fill<<<1, 1024>>>(starts, n);
  raw.c_functions = reinterpret_cast<const uint64_t*>(starts);

  DevBuf<unsigned long long> bh;
  if (int rc = bh.alloc(4)) return rc;
  SBWT_HIP(hipMemset(bh.p, 0, 32));
  k_base_hist<<<blocks_for(n_words), 256>>>(T, bh.p);
  unsigned long long h4[4];
  SBWT_HIP(hipMemcpy(h4, bh.p, 32, hipMemcpyDeviceToHost));
  uint64_t acc = 1;  // the '$' suffix is SA index 0
  for (int c = 0; c < 4; ++c) {
    raw.c_function4[c] = acc;
    acc += h4[c];
  }
  SBWT_HIP(hipDeviceSynchronize());
  SBWT_HIP(hipGetLastError());
  return 0;
}

static int export_tables(const RawDeviceTables& raw, sbwt_host_tables** out) {
  sbwt_host_tables* t = static_cast<sbwt_host_tables*>(calloc(1, sizeof(sbwt_host_tables)));
  if (!t) return SBWT_ERR_NOMEM;
  t->interval = raw.interval; t->n = raw.n; t->first_location = raw.first_location; t->real_size = raw.real_size;
  t->occ_len = raw.occ_len; t->n_loc = raw.n_loc;
  for (int c = 0; c < 4; ++c) t->c_function4[c] = raw.c_function4[c];
  auto grab = [&](void** dst, const void* src, uint64_t bytes) -> int {
    *dst = malloc(bytes ? bytes : 1);
    if (!*dst) return SBWT_ERR_NOMEM;
    SBWT_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyDeviceToHost));
    return 0;
  };
  int rc = grab((void**)&t->c_functions, raw.c_functions, (kNumKmer + 1) * 8);
  for (int c = 0; c < 4 && !rc; ++c) rc = grab((void**)&t->occ[c], raw.occ[c], raw.occ_len * 8);
  if (!rc) rc = grab((void**)&t->location_table, raw.loc, raw.n_loc * 16);
  if (!rc) rc = grab((void**)&t->jbwt_seq, raw.jbwt, raw.n / 4 + 1);
  if (!rc) rc = grab((void**)&t->fbwt, raw.fbwt, raw.n);
  if (rc) { sbwt_host_tables_free(t); return rc; }
  *out = t;
  return 0;
}

// ---- FASTA (genome_pre_handler.hpp + fasta_reader_impl.hpp) ----------------------------------------------
struct ParsedGenome {
  std::vector<uint32_t> fwd;  // 2-bit packed
  uint64_t g = 0;
  ChrTables chr;
  std::vector<uint64_t> chr_len;
};

static void npos_insert(std::vector<uint64_t>& k, std::vector<uint64_t>& v, uint64_t key, uint64_t val, size_t* at) {
  size_t i = std::lower_bound(k.begin(), k.end(), key) - k.begin();
  if (i < k.size() && k[i] == key) { *at = i; return; }  // std::map::insert keeps the old value
  k.insert(k.begin() + i, key);
  v.insert(v.begin() + i, val);
  *at = i;
}

static bool parse_fasta(const std::string& path, ParsedGenome& G, std::string& err) {
  MappedFile f;
  if (!f.open(path, err)) return false;
  const char* d = reinterpret_cast<const char*>(f.data());
  const uint64_t fn = f.size();
  G.fwd.assign(fn / 16 + 2, 0);
  auto push_base = [&](uint32_t c) {
    G.fwd[G.g >> 4] |= c << (2 * (G.g & 15));
    ++G.g;
  };
  uint64_t pos = 0, accumulated = 0;
  std::string rec;
  while (pos < fn) {
    // fasta_reader_impl.hpp:65-110: first '>' line = name (whole trimmed line), other lines = sequence,
    // next '>' line ends the record
    std::string name;
    bool have_head = false;
    rec.clear();
    while (pos < fn) {
      const char* nl = static_cast<const char*>(memchr(d + pos, '\n', fn - pos));
      const uint64_t ll = nl ? (uint64_t)(nl - (d + pos)) : fn - pos;
      const char* a = d + pos;
      const char* b = a + ll;
      while (a < b && isspace((unsigned char)*a)) ++a;
      while (b > a && isspace((unsigned char)b[-1])) --b;
      if (a < b && *a == '>') {
        if (have_head) break;
        have_head = true;
        name.assign(a + 1, b);
      } else {
        rec.append(a, b);
      }
      pos += ll + (nl ? 1 : 0);
    }
    // genome_pre_handler.hpp:63-114 Segment(std::string&, idx): N* then [ACGTU]* ; :244-276 parser_combo
    struct Seg { uint64_t off, len; };
    std::vector<Seg> segs;
    uint64_t idx = 0, len_no_n = 0;
    const uint64_t rn = rec.size();
    while (idx < rn) {
      Seg s{0, 0};
      bool fin = false;
      while (!fin && idx < rn) {
        switch (rec[idx]) {
          case 'A': case 'a': case 'C': case 'c': case 'G': case 'g': case 'T': case 't': case 'U': case 'u': fin = true; break;
          case 'N': case 'n': ++s.off; ++idx; break;
          default: ++idx; break;
        }
      }
      fin = false;
      while (!fin && idx < rn) {
        switch (rec[idx]) {
          case 'A': case 'a': push_base(0); ++s.len; ++idx; break;
          case 'C': case 'c': push_base(1); ++s.len; ++idx; break;
          case 'G': case 'g': push_base(2); ++s.len; ++idx; break;
          case 'T': case 't': push_base(3); ++s.len; ++idx; break;
          case 'U': case 'u':
            err = "FASTA contains 'U': upstream keeps it in the text but cannot reverse-complement it";
            return false;
          case 'N': case 'n': fin = true; break;
          default: ++idx; break;
        }
      }
      segs.push_back(s);
      len_no_n += s.len;
    }
    if (segs.empty()) {
      err = "FASTA record '" + name + "' has no sequence (undefined behaviour upstream)";
      return false;
    }
    G.chr.names.push_back(name);
    G.chr.start.push_back(accumulated);  // genome_pre_handler.hpp:193
    G.chr_len.push_back(len_no_n);       // :197
    accumulated += len_no_n;
    // updateNpos, genome_pre_handler.hpp:279-310
    auto& K = G.chr.npos_key;
    auto& V = G.chr.npos_val;
    size_t last;
    if (K.empty()) npos_insert(K, V, 0, segs[0].off, &last);
    else { V.back() = segs[0].off; last = K.size() - 1; }
    size_t si = 0;
    while (si + 1 != segs.size()) {
      const uint64_t key = K[last] + segs[si].len, val = V[last] + segs[si + 1].off;
      ++si;
      npos_insert(K, V, key, val, &last);
    }
    if (segs[si].len != 0) {
      size_t dummy;
      npos_insert(K, V, K[last] + segs[si].len, V[last], &dummy);
    }
  }
  G.chr.real_size = G.g;
  if (G.g == 0) {
    err = "FASTA has no ACGT bases";
    return false;
  }
  return true;
}

static int save_tables(const sbwt_host_tables* t, const std::string& path) {
  std::string err;
  TableWriter w;
  if (!w.open(path, err)) { set_error(err); return SBWT_ERR_IO; }
  static const char kAcgt[4] = {'A', 'C', 'G', 'T'};
  w.header();
  w.u64(t->interval);
  // c_function (createAllTable, object code .text+0x4064): run starts, [255] = n, zero entries 255..1
  // take the next larger non-zero value, entry 0 untouched
  uint64_t cf[256];
  memset(cf, 0, sizeof cf);
  for (int c = 0; c < 4; ++c)
    if (c == 3 ? t->c_function4[3] < t->n : t->c_function4[c] < t->c_function4[c + 1]) cf[(unsigned char)kAcgt[c]] = t->c_function4[c];
  cf[255] = t->n;
  uint64_t fill = 0;
  for (int v = 255; v >= 1; --v) { if (cf[v]) fill = cf[v]; cf[v] = fill; }
  w.u64(256); w.bytes(cf, sizeof cf);
  w.u64(kNumKmer + 1); w.bytes(t->c_functions, (kNumKmer + 1) * 8);
  w.zeros(5); w.u64(256); w.u32(0);
  for (int v = 0; v < 256; ++v) {
    int c = v == 'A' ? 0 : v == 'C' ? 1 : v == 'G' ? 2 : v == 'T' ? 3 : -1;
    if (c < 0) { w.u64(0); continue; }
    w.u64(t->occ_len); w.bytes(t->occ[c], t->occ_len * 8);
  }
  w.zeros(5); w.u64(t->n_loc); w.bytes(t->location_table, t->n_loc * 16);
  w.u64(t->n); w.bytes(t->fbwt, t->n);
  w.u64(t->first_location);
  // occ_jbwt (using_jbwt, object code .text+0x6a0): rows 'A','C','G','T' only; [c][k][byte] = # of c among
  // the first k bases of the byte, [c][4] = whole byte
  w.zeros(5); w.u64(256); w.u32(0);
  std::vector<uint64_t> lut(256);
  for (int v = 0; v < 256; ++v) {
    int c = v == 'A' ? 0 : v == 'C' ? 1 : v == 'G' ? 2 : v == 'T' ? 3 : -1;
    w.u64(5); w.u32(0);
    for (int k = 0; k < 5; ++k) {
      for (int b = 0; b < 256; ++b) {
        uint64_t cnt = 0;
        if (c >= 0)
          for (int q = 0; q < k && q < 4; ++q) cnt += ((b >> (6 - 2 * q)) & 3) == c;
        lut[(size_t)b] = cnt;
      }
      w.u64(256); w.bytes(lut.data(), 256 * 8);
    }
  }
  w.u64(t->n / 4 + 1); w.bytes(t->jbwt_seq, t->n / 4 + 1);
  w.u64(1024);
  char idx_char[1024];
  for (int b = 0; b < 256; ++b)
    for (int q = 0; q < 4; ++q) idx_char[4 * b + q] = kAcgt[(b >> (6 - 2 * q)) & 3];
  w.bytes(idx_char, 1024);
  if (!w.close(err)) { set_error(err); return SBWT_ERR_IO; }
  return SBWT_OK;
}

}  // namespace sbwt

using namespace sbwt;

extern "C" void sbwt_host_tables_free(sbwt_host_tables* t) {
  if (!t) return;
  free(t->c_functions);
  for (int c = 0; c < 4; ++c) free(t->occ[c]);
  free(t->location_table); free(t->jbwt_seq); free(t->fbwt);
  free(t);
}

extern "C" int sbwt_host_tables_save(const sbwt_host_tables* t, const char* table_path) {
  if (!t || !table_path) return SBWT_ERR_ARG;
  return save_tables(t, table_path);
}

static int check_gfx950(int device) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    (void)hipGetLastError();
    set_error("no HIP device available: this library has no CPU fallback");
    return SBWT_ERR_HIP;
  }
  if (device < 0 || device >= count) { set_error("device ordinal out of range"); return SBWT_ERR_ARG; }
  SBWT_HIP(enter_device(device));
  return SBWT_OK;
}

extern "C" int sbwt_index_build_device(int device, const uint32_t* d_fwd_packed, uint64_t g, uint64_t interval,
                                       uint64_t n_chr, const uint64_t* chr_start, const char* const* chr_names,
                                       uint64_t n_npos, const uint64_t* npos_key, const uint64_t* npos_val,
                                       sbwt_index** out_index, sbwt_host_tables** out_tables) {
  if (!d_fwd_packed || g == 0 || !n_chr || !chr_start || !n_npos || !npos_key || !npos_val || (!out_index && !out_tables)) {
    set_error("sbwt_index_build_device: bad argument");
    return SBWT_ERR_ARG;
  }
  if (interval < 2 || interval > 1024 || (interval & (interval - 1))) {
    set_error("interval must be a power of two in 2..1024 (Aligner_table::set_interval knows no other)");
    return SBWT_ERR_ARG;
  }
  if (out_index) *out_index = nullptr;
  if (out_tables) *out_tables = nullptr;
  if (int rc = check_gfx950(device)) return rc;
  OwnedRaw R;
  if (int rc = build_raw_tables(d_fwd_packed, g, interval, R)) return rc;
  if (out_tables)
    if (int rc = export_tables(R.raw, out_tables)) return rc;
  if (out_index) {
    sbwt_index* ix = new (std::nothrow) sbwt_index;
    if (!ix) return SBWT_ERR_NOMEM;
    ix->device = device;
    for (uint64_t k = 0; k < n_chr; ++k) {
      ix->chr.start.push_back(chr_start[k]);
      ix->chr.names.push_back(chr_names && chr_names[k] ? std::string(chr_names[k]) : "chr" + std::to_string(k + 1));
    }
    ix->chr.real_size = g;
    ix->chr.npos_key.assign(npos_key, npos_key + n_npos);
    ix->chr.npos_val.assign(npos_val, npos_val + n_npos);
    // the index adopts c_functions; everything else of R is freed on return
    ix->allocs.push_back(const_cast<uint64_t*>(R.raw.c_functions));
    ix->device_bytes += (kNumKmer + 1) * 8;
    R.disown(R.raw.c_functions);
    int rc = build_device_layout(ix, R.raw);
    if (!rc) rc = upload_chr_tables(ix);
    if (!rc) rc = finish_index(ix);
    if (rc) {
      delete ix;
      if (out_tables && *out_tables) { sbwt_host_tables_free(*out_tables); *out_tables = nullptr; }
      return rc;
    }
    *out_index = ix;
  }
  return SBWT_OK;
}

extern "C" int sbwt_build_index(const char* fasta_path, const char* prefix_in, uint64_t sort_len, uint64_t interval,
                                int device) {
  if (!fasta_path || !prefix_in) return SBWT_ERR_ARG;
  if (sort_len > (uint64_t)kMaxDepth) {
    set_error("sort_len above 1080 is not supported");
    return SBWT_ERR_ARG;
  }
  std::string prefix(prefix_in);
  if (prefix.empty() || prefix.back() != '.') prefix += '.';  // indexer.hpp:140-142
  std::string err;
  ParsedGenome G;
  if (!parse_fasta(fasta_path, G, err)) { set_error(err); return SBWT_ERR_IO; }
  if (int rc = check_gfx950(device)) return rc;
  {
    FILE* fs = fopen((prefix + "chrStart").c_str(), "wb");
    FILE* fl = fopen((prefix + "chrLen").c_str(), "wb");
    if (!fs || !fl) { if (fs) fclose(fs); if (fl) fclose(fl); set_error("cannot write index files under " + prefix); return SBWT_ERR_IO; }
    bool ok = true;  // a full disk must not leave truncated side files behind a success status
    for (size_t k = 0; k < G.chr.names.size(); ++k) {
      ok = ok && fprintf(fs, "%s\t%llu\n", G.chr.names[k].c_str(), (unsigned long long)G.chr.start[k]) > 0;
      ok = ok && fprintf(fl, "%s\t%llu\n", G.chr.names[k].c_str(), (unsigned long long)G.chr_len[k]) > 0;
    }
    ok = (fclose(fs) == 0) && ok;
    ok = (fclose(fl) == 0) && ok;
    if (!ok) { set_error("write error on " + prefix + "chrStart / chrLen"); return SBWT_ERR_IO; }
  }
  if (!write_npos(prefix + "NposLen.z", G.chr.npos_key, G.chr.npos_val, err)) { set_error(err); return SBWT_ERR_IO; }
  uint32_t* d_fwd = nullptr;
  const uint64_t words = (G.g + 15) / 16 + 1;
  SBWT_HIP(hipMalloc(&d_fwd, words * 4));
  if (hipError_t e = hipMemcpy(d_fwd, G.fwd.data(), words * 4, hipMemcpyHostToDevice)) {
    (void)hipFree(d_fwd);
    return hip_fail(e, "hipMemcpy(genome)", __FILE__, __LINE__);
  }
  sbwt_host_tables* t = nullptr;
  int rc = sbwt_index_build_device(device, d_fwd, G.g, interval, G.chr.start.size(), G.chr.start.data(), nullptr,
                                   G.chr.npos_key.size(), G.chr.npos_key.data(), G.chr.npos_val.data(), nullptr, &t);
  (void)hipFree(d_fwd);
  if (rc) return rc;
  rc = save_tables(t, prefix + "t_table.bwt");
  sbwt_host_tables_free(t);
  return rc;
}

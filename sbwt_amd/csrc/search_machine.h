// search_machine.h -- k_search_sm: the per-read search as a lane-level state machine.
//
// The walk is latency-bound: a read is a chain of 4..8 dependent random HBM accesses (search prefix table, zero
// or more two-base steps, memoised locate, text comparison + final row), and reads differ in chain length.  A
// kernel that walks "one read per lane, phase by phase" leaves lanes idle whenever their neighbours need a step
// more.  Here every lane runs its own chain: each loop iteration a lane issues the loads of its CURRENT state
// through a fixed set of load slots (two 16-byte, one 16-byte read piece, four 8-byte, one 4-byte), the wave waits
// ONCE, every lane consumes its slots and moves to its next state, and a lane whose read is finished takes the
// next read of the wave's range.  All memory waits of all reads of the wave overlap, whatever their state.
//
// The arithmetic of every state is the upstream one (searcher.hpp:386-539), stated next to each state.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sbwt_amd.h"
#include "rank_device.h"

namespace sbwt {

enum SmState : uint32_t {
  S_DONE = 0,   // wants a new read
  S_IDLE,       // the wave's range is exhausted
  S_FETCH,      // read record: first 64 bases + length
  S_KMER,       // search prefix: kmer_tab entry, or the c_functions pair (init_exact_match, searcher.hpp:83-96)
  S_STEP_LO,    // two-base step, block of lo (and of hi when it is the same block)
  S_STEP_HI,    // two-base step, block of hi
  S_LF1,        // one base on both ends (last odd base / replay of a dying step / the base a text comparison died on)
  S_SA,         // sa_full[lo]: can the rest of the read be compared with the text?
  S_CMP,        // text comparison of up to 64 bases (+ the final row once the last chunk is in flight)
  S_MISROW,     // row of the text position a comparison died at
  S_REFETCH,    // next 64-base piece of the read (stepping across a piece boundary; reads longer than 64 only)
  S_LOCATE,     // sa_full[lo] of a finished size-1 interval
  S_FIN,        // nothing to load: the search of a prefetched read was over at once
  S_START,      // between reads: waits for room in the result ring, then starts its reserved read
};
enum SmMode : uint32_t { M_LAST = 0, M_REPLAY = 1, M_DEAD = 2, M_STEP = 3 };

struct SmTables {  // LDS copies (or the global originals when they do not fit)
  const uint64_t* chr;
  const uint64_t* npos_key;
  const uint64_t* npos_val;
  const uint64_t* sb_abs;
};

__device__ __forceinline__ uint64_t u64_of(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }

// DPP moves inside a quad (lanes 4q .. 4q+3): full-rate VALU, no LDS
template <int kSrc>
__device__ __forceinline__ uint32_t quad_bcast_t(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, kSrc * 0x55, 0xF, 0xF, true);  // quad_perm [k,k,k,k]
}
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v, int k) {
  switch (k) {
    case 0: return quad_bcast_t<0>(v);
    case 1: return quad_bcast_t<1>(v);
    case 2: return quad_bcast_t<2>(v);
    default: return quad_bcast_t<3>(v);
  }
}
__device__ __forceinline__ uint32_t quad_swap1(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true); }  // [1,0,3,2]
__device__ __forceinline__ uint32_t quad_swap2(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true); }  // [2,3,0,1]

// value of lane (pair base + kSrc) to both lanes of a pair (lanes 2q, 2q+1): quad_perm [h,h,2+h,2+h]
template <int kSrc>
__device__ __forceinline__ uint32_t pair_bcast(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, kSrc ? 0xF5 : 0xA0, 0xF, 0xF, true);
}

// bases [o, o + 64) of the 128 bases held in (q0..q3), o < 64: two 64-bit chunks
__device__ __forceinline__ void sm_funnel128(uint64_t q0, uint64_t q1, uint64_t q2, uint64_t q3, uint32_t o, uint64_t& c0,
                                             uint64_t& c1) {
  uint32_t sh = 2 * o;
  if (sh >= 64) { q0 = q1; q1 = q2; q2 = q3; sh -= 64; }
  c0 = sh ? ((q0 >> sh) | (q1 << (64 - sh))) : q0;
  c1 = sh ? ((q1 >> sh) | (q2 << (64 - sh))) : q1;
}

// Where the per-read records live: the 32-byte records, or (compact output) the 16-byte entries with wide[] behind
// them.  The multi-hit locate kernels read a queued read's interval and reserved slots and write back its list's
// place and NH through these.
struct ResultIO {
  sbwt_read_result* results;
  sbwt_compact_result* compact;  // null: results[]
  sbwt_wide_result* wide;
  uint64_t wide_cap;
  sbwt_hit* unique_hits = nullptr;  // results[]: where hits[read index] of the unique hits goes (null: the hits buffer of the call)
};
struct MultiRead {
  uint64_t lo, hi, first_hit;
};
__device__ __forceinline__ MultiRead rio_load(const ResultIO& io, uint64_t r) {
  MultiRead m;
  if (!io.compact) {
    const sbwt_read_result res = io.results[r];
    m.lo = res.sa_lo; m.hi = res.sa_hi; m.first_hit = res.first_hit;
    return m;
  }
  const sbwt_compact_result c = io.compact[r];
  const uint64_t payload = c.w1 & kPosMask;
  if ((uint32_t)(c.w1 >> 56) & SBWT_CR_WIDE) {
    m.lo = m.hi = m.first_hit = 0;  // a slot past wide_cap was never written: the call fails with SBWT_ERR_CAPACITY
    if (payload < io.wide_cap) {
      const sbwt_read_result res = io.wide[payload].r;
      m.lo = res.sa_lo; m.hi = res.sa_hi; m.first_hit = res.first_hit;
    }
    return m;
  }
  m.lo = c.w0 & kPosMask;
  m.hi = m.lo + (c.w0 >> kPosBits);
  m.first_hit = payload;
  return m;
}
// a queued multi-hit read's list starts at hits[first_hit] and holds nh hits; add_flags: SBWT_RF_* to set
__device__ __forceinline__ void rio_store_multi(const ResultIO& io, uint64_t r, uint64_t first_hit, bool set_nh, int32_t nh,
                                                uint32_t add_flags) {
  if (!io.compact) {
    io.results[r].first_hit = first_hit;
    if (set_nh) io.results[r].nh = nh;
    if (add_flags) io.results[r].flags |= add_flags;
    return;
  }
  const uint64_t w1 = io.compact[r].w1;
  const uint32_t meta = (uint32_t)(w1 >> 56);
  if (meta & SBWT_CR_WIDE) {
    const uint64_t slot = w1 & kPosMask;
    if (slot >= io.wide_cap) return;
    io.wide[slot].r.first_hit = first_hit;
    if (set_nh) io.wide[slot].r.nh = nh;
    if (add_flags) io.wide[slot].r.flags |= add_flags;
    return;
  }
  const uint64_t aux = set_nh ? (uint64_t)(uint32_t)nh & 0xFFFFull : (w1 >> kPosBits) & 0xFFFFull;
  io.compact[r].w1 = (first_hit & kPosMask) | (aux << kPosBits) | ((uint64_t)(meta | (add_flags << 3)) << 56);
}

struct WaveInfo {
  uint64_t slots;      // hit slots the wave's multi-hit reads need (k_search_sm) -> exclusive prefix over waves (k_multi_plan)
  uint32_t count;      // multi-hit reads of the wave
  uint32_t queue_base; // exclusive prefix of count over waves (k_multi_plan)
};

struct SmArgs {
  const uint32_t* packed;
  const uint32_t* lens;
  uint64_t n_reads;
  uint32_t wpr;
  int strand;
  uint64_t limit;
  sbwt_read_result* results;
  sbwt_hit* hits;
  uint64_t hits_cap;
  // multi-hit reads: no atomics.  Wave w keeps the queue entries of its own range of reads in
  // multi_queue[range_begin(w) ..) and numbers the hit slots it needs from 0; its totals go to wave_info[w], and
  // k_multi_plan turns them into global offsets for the locate kernels.
  uint32_t* multi_queue;
  struct WaveInfo* wave_info;
  unsigned long long* index_accesses;  // tally of random index accesses (for the roofline accounting)
  unsigned long long* hash_stats;      // [0] hash_tab lookups, [1] lookups that found their prefix (measurement only)
  // what a ring entry holds (kRingWidthMax, 16 bits of chr_id; a test knob lowers them to reach the direct path)
  uint64_t ring_width_max;
  uint32_t ring_chr_max;
  // compact output (sbwt_map_batch_device_compact): the 16-byte ring entries ARE the per-read records, results /
  // hits[read index] are not written; a result that does not fit an entry goes to wide[] (slot from wide_count)
  sbwt_compact_result* compact;  // null: the 32-byte records + hits[read index]
  sbwt_wide_result* wide;
  unsigned long long* wide_count;
  uint64_t wide_cap;
  uint32_t skip_hash;  // the batch holds only reads whose prefix is known not to be in hash_tab (k_settle looked them up)
  // the pending pass of a fast pass, launched before the host knows how many reads it left: the batch holds
  // min(*n_reads_dev, n_reads) reads, and skips hash_tab when *hash_skip_dev == 0 (null: n_reads / skip_hash as given)
  const unsigned long long* n_reads_dev;
  const unsigned long long* hash_skip_dev;
};

constexpr uint32_t kSmMaxKeys = 1536;   // chr + npos entries staged in LDS (chr 8 B, npos 16 B each)
constexpr uint32_t kSmMaxSuper = 16;    // superblocks of sb_abs staged in LDS (2^35 suffixes)
constexpr uint32_t kSmRing = 256;       // results staged per wave (LDS, 16 B each) before they are written as whole lines
// Block2 lines of the two-base steps arrive through LDS: ONE load instruction per line (settle.h, settle_lookup).  A lane
// that reads planes and count of its 128-byte line through three loads of its own makes two of them wait in the vector
// cache behind the first one's miss; instead the four lanes of a quad fetch, in round k, the line of the quad's lane k --
// lane 0 and 1 its planes, lane 2 the 16 bytes that hold cnt[code] -- as one LDS-DMA instruction (wave base + 16 * lane),
// and lane k reads its 48 bytes back from LDS.  Per wave: four rounds of 64 slots + one slot of padding a round.
constexpr uint32_t kSmStageRound = 65, kSmStage = 4 * kSmStageRound;
// ring entry: word 0 = sa_lo | (sa_hi - sa_lo) << 40 (24 bits), word 1 = payload (40 bits) | chr_id << 40 (16 bits) |
// meta << 56.  A result that does not fit (interval of 2^24 rows or more, chr_id above 65535) is written to global
// memory at once and its entry only says so.
constexpr uint32_t R_NH = 1u, R_RC = 2u, R_MULTI = 4u, R_FLAGS_SHIFT = 3u /* 2 bits */, R_INVALID = 32u, R_DIRECT = 64u, R_NOSA = 128u;
static_assert(R_NH == SBWT_CR_UNIQUE && R_RC == SBWT_CR_RC && R_MULTI == SBWT_CR_MULTI && (1u << R_FLAGS_SHIFT) == SBWT_CR_SUPPRESSED &&
              (2u << R_FLAGS_SHIFT) == SBWT_CR_TRUNCATED && R_INVALID == SBWT_CR_INVALID && R_DIRECT == SBWT_CR_WIDE && R_NOSA == SBWT_CR_NO_SA,
              "a ring entry is a sbwt_compact_result");
constexpr uint64_t kRingWidthMax = 0xFFFFFFull;

__device__ __forceinline__ void sm_emit(const SmArgs& a, uint64_t rd, uint64_t lo, uint64_t hi, uint64_t payload,
                                        uint32_t chr_id, uint32_t meta) {
  sbwt_read_result res;
  res.sa_lo = lo;
  res.sa_hi = hi;
  res.first_hit = (meta & R_NH) ? rd : ((meta & R_MULTI) ? payload : 0);
  res.nh = (meta & R_NH) ? 1 : 0;
  res.flags = ((meta >> R_FLAGS_SHIFT) & 3u) | ((meta & R_INVALID) ? kFlagInvalid : 0u);
  a.results[rd] = res;
  if (meta & R_NH) {  // unique hits live at the read's own index
    sbwt_hit h;
    h.pos0 = payload;
    h.chr_id = chr_id;
    h.is_rc = (meta & R_RC) ? 1u : 0u;
    a.hits[rd] = h;
  }
}

// compact output, a result no 16-byte entry holds: the whole record goes to wide[]; returns its slot (the entry's
// payload).  Past wide_cap nothing is written; the host sees the count and reports what a retry needs.
__device__ __forceinline__ uint64_t sm_emit_wide(const SmArgs& a, uint64_t lo, uint64_t hi, uint64_t payload, uint32_t chr_id,
                                                 uint32_t meta) {
  const uint64_t slot = atomicAdd(a.wide_count, 1ull);
  if (slot < a.wide_cap) {
    sbwt_wide_result w;
    w.r.sa_lo = lo;
    w.r.sa_hi = hi;
    w.r.first_hit = (meta & R_MULTI) ? payload : 0;
    w.r.nh = (meta & R_NH) ? 1 : 0;
    w.r.flags = ((meta >> R_FLAGS_SHIFT) & 3u) | ((meta & R_INVALID) ? kFlagInvalid : 0u);
    w.hit.pos0 = (meta & R_NH) ? payload : 0;
    w.hit.chr_id = chr_id;
    w.hit.is_rc = (meta & R_RC) ? 1u : 0u;
    a.wide[slot] = w;
  }
  return slot;
}

// one finished read leaves the ring: lane l writes read first + l, so a wave writes 2 KB of results and 1 KB of hits
// as whole lines
__device__ __forceinline__ void sm_flush(const SmArgs& a, const uint4* ring, uint64_t range_begin, uint64_t rd, bool valid) {
  if (!valid) return;
  const uint4 e = ring[(size_t)((rd - range_begin) & (kSmRing - 1))];
  if (a.compact) {  // the entry is the record: a wave writes 1 KB of whole lines
    reinterpret_cast<uint4*>(a.compact)[rd] = e;
    return;
  }
  const uint64_t e0 = ((uint64_t)e.y << 32) | e.x, e1 = ((uint64_t)e.w << 32) | e.z;
  const uint32_t meta = (uint32_t)(e1 >> 56);
  if (meta & R_DIRECT) return;
  const uint64_t lo = e0 & kPosMask;
  sm_emit(a, rd, lo, lo + (e0 >> kPosBits), e1 & kPosMask, (uint32_t)(e1 >> kPosBits) & 0xFFFFu, meta);
}

// aux bits above the state-private low 10
constexpr uint32_t A_TRIED = 1u << 16;    // sa_full was consulted for the text comparison
constexpr uint32_t A_LOCATED = 1u << 17;  // p holds the final text position
constexpr uint32_t A_NOPOS = 1u << 18;    // the upstream locate reports nothing for this row
constexpr uint32_t A_PIECE = 1u << 19;    // w holds the read piece j is in (set by S_REFETCH)
constexpr uint32_t A_CUR_W = 1u << 20;    // S_START: the read's record was prefetched (w_lo, w_hi, L)
constexpr uint32_t A_CUR_K = 1u << 21;    // ... and so was its search-prefix entry (ck)
constexpr uint32_t A_CUR_KH = 1u << 22;   // ... which came from hash_tab: a located prefix of hash_order bases
constexpr uint32_t A_CUR_P2 = 1u << 23;   // (long-read variant) w2 holds the 64-base piece after the one in w
constexpr uint32_t A_REQ_P2 = 1u << 25;   // (long-read variant) ... and is on its way with the next piece
constexpr uint32_t A_DIRECT = 1u << 24;   // the prefix entry was a located row: next state is the text comparison
constexpr uint32_t A_RUN_SEEN = 1u << 26; // S_LOCATE has seen sa_full of the final row: A_LOOSE is valid
constexpr uint32_t A_LOOSE = 1u << 27;    // ... and the positions below it are not LF-consistent over the sampling distance (walk_may_go_on)

// prefetch pipeline of a lane: its next read (1) and the one after (2)
constexpr uint32_t PF_W1 = 1u, PF_K1 = 2u, PF_W2 = 4u;        // record of 1, prefix entry of 1, record of 2 are here
constexpr uint32_t PF_RW1 = 8u, PF_RK1 = 16u, PF_RW2 = 32u;   // ... are in flight this iteration
constexpr uint32_t PF_H1 = 64u, PF_KH = 128u, PF_RH1 = 256u;  // hash_tab was asked about 1; it answered; in flight
constexpr uint32_t PF_P1 = 1024u, PF_RP1 = 2048u;             // (long-read variant) second 64-base piece of 1: here; in flight

// kLong: the variant launched for batches whose records hold more than 64 bases.  A lane also prefetches the second
// 64-base piece of its next read, and a text comparison that ends within the first 128 bases of the read runs in
// ONE iteration against the whole 128-base window of the PosLine (instead of one iteration and one access per piece).
template <bool kLds, bool kLong>
__global__ __launch_bounds__(256, 4) void k_search_sm(const DeviceIndex ix, const SmArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint64_t s_mem[];
  SmTables T{ix.chr_start, ix.npos_key, ix.npos_val, ix.sb_abs};
  const uint32_t n_super = (uint32_t)(((ix.n_blocks - 1) >> (kSuperShift - 6)) + 1);
  if (kLds) {
    uint64_t* s_chr = s_mem;
    uint64_t* s_nk = s_chr + ix.n_chr;
    uint64_t* s_nv = s_nk + ix.n_npos;
    uint64_t* s_sb = s_nv + ix.n_npos;
    for (uint32_t k = threadIdx.x; k < ix.n_chr; k += blockDim.x) s_chr[k] = ix.chr_start[k];
    for (uint32_t k = threadIdx.x; k < ix.n_npos; k += blockDim.x) { s_nk[k] = ix.npos_key[k]; s_nv[k] = ix.npos_val[k]; }
    for (uint32_t k = threadIdx.x; k < n_super * 16; k += blockDim.x) s_sb[k] = ix.sb_abs[k];
    __syncthreads();
    T.chr = s_chr; T.npos_key = s_nk; T.npos_val = s_nv; T.sb_abs = s_sb;
  }
  // per-wave ring of finished reads (after the tables; 16-byte aligned)
  const uint32_t table_words = kLds ? ((ix.n_chr + 2 * ix.n_npos + 16 * n_super + 1) & ~1u) : 0u;
  uint4* const ring = reinterpret_cast<uint4*>(s_mem + table_words) + (size_t)(threadIdx.x >> 6) * kSmRing;
  uint4* const s_blk = reinterpret_cast<uint4*>(s_mem + table_words) + 4u * kSmRing + (size_t)(threadIdx.x >> 6) * kSmStage;
  // the wave's contiguous range of reads; lane l owns reads begin + l, begin + l + 64, ...
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t n_waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
  const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  uint64_t n_reads = a.n_reads;
  if (a.n_reads_dev) {
    const uint64_t nd = *a.n_reads_dev;
    n_reads = nd < n_reads ? nd : n_reads;
  }
  uint64_t chunk = (n_reads + n_waves - 1) / n_waves;
  chunk = (chunk + 63) & ~63ull;
  const uint64_t range_begin = wave * chunk < n_reads ? wave * chunk : n_reads;
  const uint64_t range_end = range_begin + chunk < n_reads ? range_begin + chunk : n_reads;
  const uint32_t K = ix.kmer_order;
  const bool skip_hash = a.skip_hash || (a.hash_skip_dev && *a.hash_skip_dev == 0ull);
  const uint32_t H = skip_hash ? 0u : ix.hash_order;  // 0: no hash table
  const uint64_t hmask = (1ull << (2 * H)) - 1ull;
  const uint64_t kmask = (1ull << (2 * K)) - 1ull;  // K <= kMaxKmerOrder: the prefix lies in the first 32 bases, w_lo
  const uint32_t max_len = a.wpr * 16u;

  // lane state.  A lane holds its current read r and has RESERVED its next two, rn and rn2: lanes that start a read
  // in the same iteration reserve consecutive reads (ballot order), so the records they prefetch and, since such
  // lanes tend to finish together again, the results they write fall into the same lines.  A read is a chain of
  // three dependent loads at least (record, prefix entry, text + final row): with two reads reserved ahead, the
  // record of rn2 and the prefix entry of rn ride in slots the current state leaves idle, and a read whose
  // prefix entry is a located row takes ONE iteration.
  uint64_t r = range_begin + lane, rn = r + 64, rn2 = r + 128;
  uint64_t next = range_begin + 192;  // wave-uniform: first unreserved read
  uint32_t st = r < range_end ? S_FETCH : S_IDLE;
  bool reserve = false;
  uint64_t flushed = range_begin;  // wave-uniform: results of reads below it are in global memory
  uint32_t tally = 0;              // random index accesses this lane needed (128-byte lines)
  uint32_t tally_hq = 0, tally_hh = 0;  // hash_tab lookups and hits of this lane (a lane of a 2^32-read batch makes more than 2^16)
  uint64_t wave_slots = 0;  // wave-uniform: hit slots handed to the multi-hit reads of this wave so far
  uint32_t wave_multi = 0;  // wave-uniform: how many such reads
  uint32_t L = 0, j = 0, aux = 0, flags = 0, nL = 0, mL = 0, pf = 0;
  uint64_t lo = 0, hi = 0, w_lo = 0, w_hi = 0, p = 0, t_lo = 0, t_hi = 0;
  uint64_t ck = 0;                                     // prefix entry of the read about to start
  uint64_t nw_lo = 0, nw_hi = 0, nk = 0, mw_lo = 0, mw_hi = 0;  // records of rn (with its prefix entry) and rn2
  uint64_t w2_lo = 0, w2_hi = 0, nw2_lo = 0, nw2_hi = 0;        // kLong: bases 64..127 of the current read and of rn

  // interval after the search prefix, from a kmer_tab entry (device_layout.h)
  auto apply_prefix = [&](uint64_t e, uint32_t Kp) {  // Kp = bases the entry covers (kmer_order, or hash_order)
    uint64_t width = e >> kPosBits;
    if (e & kKmerLocated) {
      // one row, already located: straight to the text comparison when it can carry the rest of the read
      const uint64_t pos = e & kPosMask, run = (e >> kPosBits) & kKmerRunMax;
      const uint64_t rem = (uint64_t)L - Kp;
      if (rem >= 1 && run > rem && pos >= rem) {
        p = pos;
        j = Kp;
        lo = 0;  // the row itself is not needed: a comparison that dies reads it from pos_lines
        hi = 1;
        aux |= A_TRIED | A_DIRECT;
        return;
      }
      width = kKmerWide;  // otherwise as if not memoised
    }
    if (width == kKmerWide) {  // not memoised (interval wider than an entry holds): the 12-mer pair, read here
      const uint32_t kmer = (~(uint32_t)w_lo) & 0xFFFFFFu;
      lo = ix.c_functions[kmer];
      hi = ix.c_functions[kmer + 1];
    } else {
      lo = e & kPosMask;
      hi = lo + width;
      j = K;
    }
  };
  // what the read needs next (searcher.hpp:406-410: one exec_exact_match per remaining base while lo < hi);
  // returns true when the search is over
  auto choose = [&]() -> bool {
    if (aux & A_DIRECT) {
      aux &= ~A_DIRECT;
      st = S_CMP;
      return false;
    }
    if (lo >= hi || j >= L) return true;
    if ((j & 63u) == 0 && j >= 64u && !(aux & A_PIECE)) {
      aux |= A_PIECE;  // w must hold the piece j is in before anything looks at the next bases
      st = S_REFETCH;
    } else if (hi - lo == 1 && !(aux & A_TRIED)) {
      st = S_SA;
    } else {
      aux &= ~A_PIECE;
      const uint64_t wsel = (j & 32u) ? w_hi : w_lo;
      if (j + 1 == L || (j & 1u)) {
        // the last base, or the base after a prefix of odd order: two-base steps start at even j, so that the
        // pair never straddles the halves or the pieces of the read record
        const uint32_t c = ((uint32_t)(wsel >> (2 * (j & 31u))) & 3u) ^ 3u;
        aux = (aux & ~0x3ffu) | c | ((j + 1 == L ? M_LAST : M_STEP) << 8);
        st = S_LF1;
      } else {
        const uint32_t two = ((uint32_t)(wsel >> (2 * (j & 31u))) & 15u) ^ 15u;  // complement: the query is RC(read)
        aux = (aux & ~0x3ffu) | ((two & 3u) << 2) | (two >> 2);                  // x = first prepended base, y = second
        st = S_STEP_LO;
      }
    }
    return false;
  };

  const uint64_t lane_lt = (1ull << lane) - 1ull;
  for (;;) {
    if (!__any(st != S_IDLE)) break;
    // ---- flush: reads are reserved in increasing order, so every read below the lowest one a lane still works
    // on (or waits to start) is finished; 64 of them at a time leave the ring as whole lines
    for (;;) {
      const uint64_t cur = st == S_IDLE ? ~0ull : r;
      if (flushed + 64 > range_end || __any(cur < flushed + 64)) break;
      sm_flush(a, ring, range_begin, flushed + lane, true);
      flushed += 64;
    }
    // ---- start: a lane between reads takes its reserved one once the ring has room for it
    if (st == S_START && r < flushed + kSmRing) {
      reserve = true;
      const uint32_t had = aux;
      aux = had & A_CUR_P2;
      flags = 0;
      if (!(had & A_CUR_W)) {
        st = S_FETCH;
      } else {
        j = kKmer;
        if (L < (uint32_t)kKmer || L > max_len) {
          flags = kFlagInvalid;
          lo = hi = 0;
          st = S_FIN;
        } else if (had & A_CUR_K) {
          apply_prefix(ck, (had & A_CUR_KH) ? ix.hash_order : K);
          if (choose()) st = S_FIN;
        } else {
          st = S_KMER;
        }
      }
    }
    {
      const uint64_t wm = __ballot(reserve);
      if (wm) {
        if (reserve) rn2 = next + (uint64_t)__popcll(wm & lane_lt);
        next += (uint64_t)__popcll(wm);
        reserve = false;
      }
    }
    const uint32_t* rp = a.packed + r * (uint64_t)a.wpr;

    // ---- issue: every lane points the slots it needs at its own addresses
    // (a slot a lane does not need points at the start of the batch: one cached line for the whole wave; the
    // loads are unconditional so that all eight are in flight before the single wait)
    const uint4* const idle16 = reinterpret_cast<const uint4*>(a.packed);
    const uint64_t* const idle8 = reinterpret_cast<const uint64_t*>(a.packed);
    const uint4_a8* pa = reinterpret_cast<const uint4_a8*>(idle16);  // 16 B (8-byte aligned inside a PosLine)
    const uint4* pb = idle16;    // 16 B
    const uint4* pr = idle16;    // 16 B (read piece)
    const u64_a4* pc = idle8;    // 8 B (4-byte aligned when it points into a PosLine)
    const uint64_t* pe = idle8;  // 8 B
    const uint64_t* pk = idle8;  // 8 B (prefetch: prefix entry)
    const uint64_t* pl = idle8;  // 8 B (prefetch: the aligned 8 bytes holding a length)
    const uint4* ph = idle16;    // 2 x 16 B (prefetch: one hash_tab bucket)
    const uint4* pr2 = idle16;   // 16 B (kLong; prefetch: second piece of the next read)
    const uint32_t* pd = a.packed;  // 4 B
    bool pair = false;           // S_KMER through the c_functions pair
    bool free_r = true, free_r2 = true;
    uintptr_t step_line = 0;     // S_STEP_*: the Block2 line of the step
    switch (st) {
      case S_FETCH:
        pr = reinterpret_cast<const uint4*>(rp);
        pd = a.lens + r;
        free_r = false;
        break;
      case S_KMER: {
        const uint64_t cw = ~w_lo;  // complemented codes of bases 0..31
        tally += 1;
        if (ix.kmer_tab && L >= K) {
          pc = ix.kmer_tab + (cw & kmask);
        } else {
          // c_functions[kmer], c_functions[kmer + 1]: one 16-byte load (two 8-byte loads of one line would queue behind each other)
          pa = reinterpret_cast<const uint4_a8*>(ix.c_functions + ((uint32_t)cw & 0xFFFFFFu));
          pair = true;
        }
        break;
      }
      case S_STEP_LO:
      case S_STEP_HI:
        step_line = reinterpret_cast<uintptr_t>(ix.blocks2 + ((st == S_STEP_LO ? lo : hi) >> 6));  // fetched below, a quad at a time
        tally += 1;
        break;
      case S_LF1: {
        const Block1* bl = ix.blocks1 + (lo >> 6);
        const Block1* bh = ix.blocks1 + (hi >> 6);
        pa = reinterpret_cast<const uint4_a8*>(&bl->plane_lo);
        pc = &bl->cnt[aux & 3u];
        if (bl != bh) {  // both ends in one block: what was loaded for lo serves hi (a load request less each)
          pb = reinterpret_cast<const uint4*>(&bh->plane_lo);
          pe = &bh->cnt[aux & 3u];
        }
        tally += bl == bh ? 1u : 2u;
        break;
      }
      case S_SA:
      case S_LOCATE:
        pc = ix.sa_full + lo;
        tally += 1;
        break;
      case S_CMP: {
        // read[j + t] continues the match iff it equals textrc[x0 + t].  The PosLine of the position the read
        // ends on holds the 128 bases in front of it: when the rest of the read lies inside them, every 64-base
        // piece is compared against that line, and the last piece finds the final row in it too.  One 16-byte
        // load from the 32-base word the piece starts in when it ends inside those 64 bases, else two.
        const uint64_t x0 = ix.n - 1 - p;
        const uint32_t piece_end = (j | 63u) + 1u;
        const bool last = piece_end >= L;
        const uint32_t nb = (last ? L : piece_end) - j;  // bases compared now
        const uint32_t rem = L - j;                      // bases to the end of the read
        const uint64_t pfin = p - (uint64_t)rem;
        const uint64_t line = pfin / kPosPerLine;
        const uint32_t slot = (uint32_t)(pfin - line * kPosPerLine);
        const char* lb = reinterpret_cast<const char*>(ix.pos_lines + line);
        uint32_t o, sh = 0;
        const bool full = kLong && (aux & A_CUR_P2) && !last && (j & 63u) + rem <= 128u && slot + rem <= 128u;
        if (full) {
          // the rest of the read against the whole window, and the final row: one access
          const uint32_t ob = 32u + 5u * slot;
          pc = reinterpret_cast<const u64_a4*>(lb + (ob & ~3u));
          pa = reinterpret_cast<const uint4_a8*>(lb);
          pb = reinterpret_cast<const uint4*>(lb) + 1;
          tally += 1;
          aux = (aux & ~0x3ffu) | (128u - (slot + rem)) | ((ob & 3u) << 7) | 0x200u;
          break;
        }
        if (last) {
          const uint32_t ob = 32u + 5u * slot;
          pc = reinterpret_cast<const u64_a4*>(lb + (ob & ~3u));
          sh = ob & 3u;
          tally += 1;
        } else {
          pr = reinterpret_cast<const uint4*>(rp) + (piece_end >> 6);  // next piece of the read
          free_r = false;
          if (kLong && ((piece_end >> 6) + 2u) * 4u <= a.wpr) {  // ... and the one after it, when the record has one
            pr2 = pr + 1;
            free_r2 = false;
            aux |= A_REQ_P2;
          }
        }
        if (slot + rem <= 128u) {
          const uint32_t off = 128u - (slot + rem);
          if ((off & 31u) + nb <= 64u) {
            pa = reinterpret_cast<const uint4_a8*>(lb + 8u * (off >> 5));
            o = off & 31u;
          } else {
            pa = reinterpret_cast<const uint4_a8*>(lb + 16u * (off >> 6));
            pb = reinterpret_cast<const uint4*>(lb) + (off >> 6) + 1;
            o = off & 63u;
          }
          if (!last) tally += 1;
        } else {
          const char* tb = reinterpret_cast<const char*>(ix.textrc);
          if ((uint32_t)(x0 & 31) + nb <= 64u) {
            pa = reinterpret_cast<const uint4_a8*>(tb + 8u * (x0 >> 5));
            o = (uint32_t)(x0 & 31);
          } else {
            pa = reinterpret_cast<const uint4_a8*>(tb + 16u * (x0 >> 6));
            pb = reinterpret_cast<const uint4*>(tb) + (x0 >> 6) + 1;
            o = (uint32_t)(x0 & 63);
          }
          const uint64_t x1 = x0 + nb - 1;                      // last base compared
          tally += (uint32_t)((x1 >> 9) - (x0 >> 9)) + 1u;      // 512 bases per 128-byte line
        }
        aux = (aux & ~0x3ffu) | o | (sh << 7);
        break;
      }
      case S_MISROW: {
        const uint64_t line = p / kPosPerLine;
        const uint32_t ob = 32u + 5u * (uint32_t)(p - line * kPosPerLine);
        pc = reinterpret_cast<const u64_a4*>(reinterpret_cast<const char*>(ix.pos_lines + line) + (ob & ~3u));
        aux = (aux & ~0x3ffu) | ((ob & 3u) << 7);
        tally += 1;
        break;
      }
      case S_REFETCH:
        pr = reinterpret_cast<const uint4*>(rp) + (j >> 6);
        free_r = false;
        break;
      default:  // S_FIN, S_IDLE
        break;
    }
    // the lane's next reads ride in slots this state leaves idle: the record of rn, or of rn2 once that of rn is
    // here (16 B + the aligned 8 bytes holding its length), and the search-prefix entry of rn
    if (st != S_IDLE) {
      if (free_r) {
        const uint64_t q = !(pf & PF_W1) ? rn : rn2;
        if (!(pf & PF_W2) && q + 1 < n_reads && q < range_end) {
          pr = reinterpret_cast<const uint4*>(a.packed + q * (uint64_t)a.wpr);
          pl = reinterpret_cast<const uint64_t*>(reinterpret_cast<uintptr_t>(a.lens + q) & ~(uintptr_t)7);
          pf |= !(pf & PF_W1) ? PF_RW1 : PF_RW2;
        }
      }
      if (kLong && free_r2 && (pf & (PF_W1 | PF_P1)) == PF_W1 && nL > 64u && nL <= max_len) {
        pr2 = reinterpret_cast<const uint4*>(a.packed + rn * (uint64_t)a.wpr) + 1;
        pf |= PF_RP1;
      }
      if ((pf & (PF_W1 | PF_K1)) == PF_W1 && nL <= max_len) {
        if (H && !(pf & PF_H1) && nL > H) {
          // the hash table first: a located prefix of H bases, or nothing
          const uint64_t vv = hash_mix((~nw_lo) & hmask, H);
          ph = reinterpret_cast<const uint4*>(ix.hash_tab) + __umul64hi(vv << (64u - 2u * H), ix.hash_buckets) * (ix.hash_text ? 4u : 2u);
          pf |= PF_RH1;
          tally += 1;
        } else if (ix.kmer_tab && nL >= K) {
          pk = ix.kmer_tab + ((~nw_lo) & kmask);
          pf |= PF_RK1;
          tally += 1;
        }
      }
    }
    // The wave barriers are no instructions; being convergent they keep the compiler from duplicating the loads
    // into the per-state branches on either side, which would put a memory wait inside every branch.
    __builtin_amdgcn_wave_barrier();
    const uint4_a8 va = *pa;
    const uint4 vb = *pb, vr = *pr;
    const uint64_t vc = *pc, ve = *pe, vk = *pk, vl = *pl;
    const uint32_t vd = *pd;
    uint4 vh0 = make_uint4(0, 0, 0, 0), vh1 = vh0;
    if (H && !ix.hash_text) { vh0 = ph[0]; vh1 = ph[1]; }  // wave-uniform conditions; 8-byte slots: the bucket in two loads
    uint64_t hq[4] = {~0ull, ~0ull, ~0ull, ~0ull};
    uint32_t hwant[4] = {0, 0, 0, 0};
    if (H && ix.hash_text) {
      // 16-byte slots (position word + the text k_settle compares).  ONE load instruction per line (settle.h): in
      // round k the four lanes of a quad read the four position words of the bucket its lane k asked for.
      const uintptr_t pha = reinterpret_cast<uintptr_t>(ph);
      const uint32_t rq = (pf & PF_RH1) ? 1u : 0u;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t a_lo = quad_bcast(pha & 0xFFFFFFFFu, k), a_hi = quad_bcast((uint32_t)(pha >> 32), k);
        hwant[k] = quad_bcast(rq, k);
        if (hwant[k]) hq[k] = reinterpret_cast<const uint64_t*>(u64_of(a_lo, a_hi))[2u * (lane & 3u)];
      }
    }
    uint4 vr2 = make_uint4(0, 0, 0, 0);
    if (kLong) vr2 = *pr2;
    const bool stepping = st == S_STEP_LO || st == S_STEP_HI;
    if (__any(stepping)) {  // wave-uniform
      const uint32_t sub = lane & 3u;
      const uint32_t sl_lo = (uint32_t)step_line, sl_hi = (uint32_t)(step_line >> 32);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t want = quad_bcast(stepping ? 1u : 0u, k);
        if (!__any(want)) continue;  // wave-uniform
        const uint32_t code_k = quad_bcast(aux & 15u, k);
        const uintptr_t q = u64_of(quad_bcast(sl_lo, k), quad_bcast(sl_hi, k)) + (sub < 2u ? 16u * sub : 32u + 16u * (code_k >> 2));
        if (want && sub < 3u)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)q,
                                           (__attribute__((address_space(3))) void*)(s_blk + kSmStageRound * k), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      __builtin_amdgcn_wave_barrier();
    }
    __builtin_amdgcn_wave_barrier();

    if (pf & PF_RW1) {
      nw_lo = u64_of(vr.x, vr.y);
      nw_hi = u64_of(vr.z, vr.w);
      nL = (reinterpret_cast<uintptr_t>(a.lens + rn) & 4) ? (uint32_t)(vl >> 32) : (uint32_t)vl;
      pf = (pf | PF_W1) & ~PF_RW1;
    } else if (pf & PF_RW2) {
      mw_lo = u64_of(vr.x, vr.y);
      mw_hi = u64_of(vr.z, vr.w);
      mL = (reinterpret_cast<uintptr_t>(a.lens + rn2) & 4) ? (uint32_t)(vl >> 32) : (uint32_t)vl;
      pf = (pf | PF_W2) & ~PF_RW2;
    }
    if (pf & PF_RK1) {
      nk = vk;
      pf = (pf | PF_K1) & ~PF_RK1;
    }
    if (kLong && (pf & PF_RP1)) {
      nw2_lo = u64_of(vr2.x, vr2.y);
      nw2_hi = u64_of(vr2.z, vr2.w);
      pf = (pf | PF_P1) & ~PF_RP1;
    }
    uint64_t hfound = ~0ull;
    if (H && ix.hash_text) {  // wave-uniform: every lane takes part in the exchange
      const uint32_t tmask = (1u << ix.hash_tag_bits) - 1u;
      const uint32_t tag = (uint32_t)hash_mix((~nw_lo) & hmask, H) & tmask;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t tag_k = quad_bcast(tag, k);
        const uint32_t q_lo = (uint32_t)hq[k], q_hi = (uint32_t)(hq[k] >> 32);
        // (twin slots, device_layout.h, are the fast passes' business: here their prefix reads as not in the table)
        const bool hit = hwant[k] && !(q_hi >> 31) && ((q_lo ^ tag_k) & tmask) == 0;
        // at most one such slot of a bucket carries the tag: OR the masked slot over the quad
        uint32_t v0 = hit ? q_lo : 0u, v1 = hit ? q_hi : 0u, any = hit ? 1u : 0u;
        v0 |= quad_swap1(v0); v1 |= quad_swap1(v1); any |= quad_swap1(any);
        v0 |= quad_swap2(v0); v1 |= quad_swap2(v1); any |= quad_swap2(any);
        if ((lane & 3u) == (uint32_t)k && any) hfound = u64_of(v0, v1);
      }
    }
    if (pf & PF_RH1) {
      // four slots: position << 24 | run << tag_bits | tag; an empty slot is all ones (no position is)
      const uint64_t vv = hash_mix((~nw_lo) & hmask, H);
      // (tag in the low word; the high word of an empty slot is all ones, that of a position never is)
      const uint32_t tmask = (1u << ix.hash_tag_bits) - 1u, tag = (uint32_t)vv & tmask;
      uint64_t found = ~0ull;
      if (!ix.hash_text) {
        const uint32_t sl_lo[4] = {vh0.x, vh0.z, vh1.x, vh1.z};
        const uint32_t sl_hi[4] = {vh0.y, vh0.w, vh1.y, vh1.w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (((sl_lo[q] ^ tag) & tmask) == 0 && sl_hi[q] != 0xFFFFFFFFu) found = u64_of(sl_lo[q], sl_hi[q]);
      } else {
        found = hfound;
      }
      pf = (pf | PF_H1) & ~PF_RH1;
      const uint64_t run = (found & 0xFFFFFFull) >> ix.hash_tag_bits;
      if (run == 0) found = ~0ull;  // a marker (device_layout.h): the prefix occurs, but not as one located row
      tally_hq += 1u;
      tally_hh += found != ~0ull ? 1u : 0u;
      if (found != ~0ull) {
        nk = kKmerLocated | (found >> 24) | (run << kPosBits);
        pf |= PF_K1 | PF_KH;
      }
    }

    // ---- consume
    bool finish = false;  // the search part of the read is over: lo/hi final
    switch (st) {
      case S_FETCH: {
        L = vd;
        w_lo = u64_of(vr.x, vr.y);
        w_hi = u64_of(vr.z, vr.w);
        j = kKmer;
        if (L < (uint32_t)kKmer || L > max_len) {
          flags = kFlagInvalid;
          lo = hi = 0;
          finish = true;
        } else {
          st = S_KMER;
        }
        break;
      }
      case S_KMER: {
        if (pair) {  // init_exact_match, searcher.hpp:83-96
          lo = u64_of(va.x, va.y);
          hi = u64_of(va.z, va.w);
        } else {
          apply_prefix(vc, K);
        }
        finish = choose();
        break;
      }
      case S_STEP_LO:
      case S_STEP_HI: {
        // exec_exact_match (searcher.hpp:102-105) for two bases; aux & 15 = code.  One block serves the end it
        // belongs to, and the other end too when that one lies in the same block.
        const uint32_t code = aux & 15u;
        const bool second = st == S_STEP_HI;
        const uint64_t x = second ? hi : lo;
        const uint64_t blk = x >> 6;
        // (the stage is read here, in the state that needs it: the values are consumed below, long before the next
        // iteration's loads can land in it)
        const uint4* im = s_blk + kSmStageRound * (lane & 3u) + (lane & ~3u);  // round `lane & 3` served this lane
        const uint4 sb0 = im[0], sb1 = im[1], sb2 = im[2];
        const uint64_t q0 = u64_of(sb0.x, sb0.y), q1 = u64_of(sb0.z, sb0.w), q2 = u64_of(sb1.x, sb1.y), q3 = u64_of(sb1.z, sb1.w);
        const uint64_t mask = ((code & 1) ? q0 : ~q0) & ((code & 2) ? q1 : ~q1) & ((code & 4) ? q2 : ~q2) & ((code & 8) ? q3 : ~q3);
        const uint32_t cn = (code & 2u) ? ((code & 1u) ? sb2.w : sb2.z) : ((code & 1u) ? sb2.y : sb2.x);  // cnt[code] of the 16 bytes fetched
        const uint64_t base = T.sb_abs[(blk >> (kSuperShift - 6)) * 16 + code] + cn;
        const uint64_t vx = base + (uint64_t)__popcll(mask & below((uint32_t)(x & 63))) - rank2_correction(ix, code, x);
        bool both = second;
        if (second) {
          t_hi = vx;
        } else {
          t_lo = vx;
          if (((hi - 1) >> 6) == blk) {
            t_hi = base + (uint64_t)__popcll(mask & below((uint32_t)(hi - (blk << 6)))) - rank2_correction(ix, code, hi);
            both = true;
          } else {
            st = S_STEP_HI;
          }
        }
        if (both) {
          if (t_lo >= t_hi) {
            aux = (aux & ~0x3ffu) | (code >> 2) | (M_REPLAY << 8);  // replay the first base alone
            st = S_LF1;
          } else {
            lo = t_lo;
            hi = t_hi;
            j += 2;
            finish = choose();
          }
        }
        break;
      }
      case S_LF1: {
        // get_occ_using_jbwt + c_function for base c on both ends
        const uint32_t c = aux & 3u;
        const uint32_t mode = (aux >> 8) & 3u;
        uint64_t l1 = vc + (uint64_t)__popcll(match1(u64_of(va.x, va.y), u64_of(va.z, va.w), c) & below((uint32_t)(lo & 63)));
        const bool one_block = (lo >> 6) == (hi >> 6);
        const uint64_t hp_lo = one_block ? u64_of(va.x, va.y) : u64_of(vb.x, vb.y), hp_hi = one_block ? u64_of(va.z, va.w) : u64_of(vb.z, vb.w);
        uint64_t h1 = (one_block ? vc : ve) + (uint64_t)__popcll(match1(hp_lo, hp_hi, c) & below((uint32_t)(hi & 63)));
        if (c == 0 && lo > ix.first_location) l1 -= 1;
        if (c == 0 && hi > ix.first_location) h1 -= 1;
        if (mode == M_REPLAY) {
          // the interval died inside a two-base step: upstream stops after the FIRST base if that one already
          // emptied it (loop condition, searcher.hpp:406), otherwise with the two-base result
          if (l1 >= h1) { lo = l1; hi = h1; } else { lo = t_lo; hi = t_hi; }
        } else {
          lo = l1;
          hi = h1;
          if (mode == M_LAST || mode == M_STEP) j += 1;
        }
        finish = mode == M_STEP ? choose() : true;
        break;
      }
      case S_SA: {
        const uint64_t e = vc;
        const uint64_t pos = e & kPosMask, run = e >> kPosBits;
        const uint32_t rem = L - j;
        aux |= A_TRIED;
        if (e != kNoPosition && run > (uint64_t)rem && pos >= (uint64_t)rem) {
          p = pos;
          st = S_CMP;
        } else {
          finish = choose();
        }
        break;
      }
      case S_CMP: {
        if (kLong && (aux & 0x200u)) {
          // bases [j, L), L <= 128, against the 128-base window of the PosLine from offset o on
          const uint32_t o = aux & 127u, nb = L - j;
          const uint64_t q0 = u64_of(va.x, va.y), q1 = u64_of(va.z, va.w), q2 = u64_of(vb.x, vb.y), q3 = u64_of(vb.z, vb.w);
          uint64_t t0, t1, t2 = 0, t3 = 0, r0, r1, r2, r3;
          if (o < 64u) {
            sm_funnel128(q0, q1, q2, q3, o, t0, t1);
            sm_funnel128(q2, q3, 0, 0, o, t2, t3);
          } else {
            sm_funnel128(q2, q3, 0, 0, o - 64u, t0, t1);
          }
          const uint32_t jo = j & 63u;  // w holds the piece j is in, w2 the next
          sm_funnel128(w_lo, w_hi, w2_lo, w2_hi, jo, r0, r1);
          sm_funnel128(w2_lo, w2_hi, 0, 0, jo, r2, r3);
          uint64_t d[4] = {r0 ^ t0, r1 ^ t1, r2 ^ t2, r3 ^ t3};
          uint32_t t = nb;
#pragma unroll
          for (int q = 3; q >= 0; --q) {
            const uint32_t first = 32u * (uint32_t)q;  // this word covers bases [first, first + 32) of the nb
            uint64_t dq = d[q];
            if (nb <= first) dq = 0;
            else if (nb < first + 32u) dq &= (1ull << (2 * (nb - first))) - 1ull;
            if (dq) t = first + ((uint32_t)(__ffsll((unsigned long long)dq) - 1) >> 1);
          }
          j += t;
          p -= t;
          if (jo + t >= 64u) { w_lo = w2_lo; w_hi = w2_hi; aux &= ~A_CUR_P2; }  // w holds the piece j is in
          if (t < nb) {
            st = S_MISROW;
          } else {
            lo = (vc >> (8u * ((aux >> 7) & 3u))) & kPosMask;
            hi = lo + 1;
            aux |= A_LOCATED;
            finish = true;
          }
          break;
        }
        // bases [j, min(L, end of the read's 64-base piece)) against the text at x0.. (device_index.hip: text-jump)
        uint64_t t0, t1, r0, r1;
        sm_funnel128(u64_of(va.x, va.y), u64_of(va.z, va.w), u64_of(vb.x, vb.y), u64_of(vb.z, vb.w), aux & 127u, t0, t1);
        sm_funnel128(w_lo, w_hi, 0, 0, j & 63u, r0, r1);
        const uint32_t piece_end = (j | 63u) + 1u;
        const uint32_t stop = piece_end < L ? piece_end : L;
        const uint32_t nb = stop - j;  // 1..64
        uint64_t d0 = r0 ^ t0, d1 = r1 ^ t1;
        if (nb < 32u) d0 &= (1ull << (2 * nb)) - 1ull;
        if (nb <= 32u) d1 = 0;
        else if (nb < 64u) d1 &= (1ull << (2 * (nb - 32u))) - 1ull;
        uint32_t t = nb;
        if (d0) t = (uint32_t)(__ffsll((unsigned long long)d0) - 1) >> 1;
        else if (d1) t = 32u + ((uint32_t)(__ffsll((unsigned long long)d1) - 1) >> 1);
        j += t;
        p -= t;
        if (t < nb) {
          st = S_MISROW;  // the chain dies at base j: exec_exact_match on the row of text position p
        } else if (j == L) {
          lo = (vc >> (8u * ((aux >> 7) & 3u))) & kPosMask;  // the row the chain ends on (row of position p)
          hi = lo + 1;
          aux |= A_LOCATED;  // the position is p
          finish = true;
        } else {
          w_lo = u64_of(vr.x, vr.y);  // next piece of the read
          w_hi = u64_of(vr.z, vr.w);
          if (kLong) {
            aux &= ~A_CUR_P2;
            if (aux & A_REQ_P2) {
              w2_lo = u64_of(vr2.x, vr2.y);
              w2_hi = u64_of(vr2.z, vr2.w);
              aux |= A_CUR_P2;
            }
          }
        }
        if (kLong) aux &= ~A_REQ_P2;
        break;
      }
      case S_MISROW: {
        lo = (vc >> (8u * ((aux >> 7) & 3u))) & kPosMask;
        hi = lo + 1;
        const uint32_t c = ((uint32_t)((j & 32u ? w_hi : w_lo) >> (2 * (j & 31u))) & 3u) ^ 3u;
        aux = (aux & ~0x3ffu) | c | (M_DEAD << 8);
        st = S_LF1;
        break;
      }
      case S_REFETCH: {
        w_lo = u64_of(vr.x, vr.y);
        w_hi = u64_of(vr.z, vr.w);
        aux &= ~A_CUR_P2;  // w2 is no longer the piece after w
        finish = choose();
        break;
      }
      case S_LOCATE: {
        p = vc & kPosMask;
        if (vc == kNoPosition) aux |= A_NOPOS;  // the upstream locate finds nothing
        else if (walk_may_go_on(ix, p, vc >> kPosBits)) aux |= A_LOOSE;
        aux |= A_LOCATED | A_RUN_SEEN;
        finish = true;
        break;
      }
      case S_FIN:
        finish = true;
        break;
      default:
        break;
    }

    // ---- a finished search: suppression (searcher.hpp:415), locate of a size-1 interval, filters, output
    const uint64_t m = finish && hi > lo ? hi - lo : 0;
    if (finish && m == 1 && m <= a.limit && !(aux & A_LOCATED)) {
      st = S_LOCATE;
      finish = false;
    }
    // one row: its position through the filters (searcher.hpp:504-539).  A position they drop, on a row below which the
    // tables are not LF-consistent over the sampling distance: upstream's walk may surface again (rank_device.h,
    // walk_may_go_on) -- the read is located like a multi-hit read.  (A position that came out of a text comparison: S_LOCATE
    // looks at the final row first.)
    sbwt_hit h1;
    bool passes = false;
    if (finish && m == 1 && m <= a.limit && !(aux & A_NOPOS)) {
      passes = push_position_keys(ix, T.chr, T.npos_key, T.npos_val, p, L, a.strand, &h1);
      if (!passes && !(aux & A_RUN_SEEN)) {
        st = S_LOCATE;
        finish = false;
      }
    }
    const bool walk1 = finish && m == 1 && m <= a.limit && !passes && (aux & A_LOOSE);
    // multi-hit reads take hit slots (numbered within the wave; k_multi_plan makes them global) and a queue entry
    // in the wave's own stretch of the queue
    const uint64_t want = (finish && (m > 1 || walk1) && m <= a.limit) ? m : 0;
    uint64_t multi_off = 0;
    const uint64_t mm = __ballot(want != 0);
    if (mm) {
      uint64_t inc = want;  // inclusive scan of `want` over the wave
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint64_t up = ((uint64_t)__shfl_up((int)(inc >> 32), d) << 32) | (uint32_t)__shfl_up((int)(uint32_t)inc, d);
        if ((int)lane >= d) inc += up;
      }
      const uint64_t total = ((uint64_t)__shfl((int)(inc >> 32), 63) << 32) | (uint32_t)__shfl((int)(uint32_t)inc, 63);
      if (want) {
        multi_off = wave_slots + (inc - want);
        a.multi_queue[range_begin + wave_multi + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull))] = (uint32_t)r;
      }
      wave_slots += total;
      wave_multi += (uint32_t)__popcll(mm);
    }
    if (finish) {
      uint32_t meta = 0;
      uint64_t payload = 0;  // unique hit: position in its chromosome; multi-hit read: first reserved slot
      uint32_t chr_id = 0;
      if (m > a.limit) {
        flags |= SBWT_RF_SUPPRESSED;
      } else if (want) {
        payload = multi_off;  // first slot, wave-relative until k_multi_sort adds the wave's base
        meta = R_MULTI;
      } else if (passes) {
        payload = h1.pos0;
        chr_id = h1.chr_id;
        meta = R_NH | (h1.is_rc ? R_RC : 0u);
      }
      // flags: bits 0..1 and bit 31 are in use
      meta |= ((flags & 3u) << R_FLAGS_SHIFT) | ((flags & kFlagInvalid) ? R_INVALID : 0u);
      uint64_t width = hi - lo;
      if (hi < lo || width >= a.ring_width_max || chr_id > a.ring_chr_max || payload > kPosMask ||
          (a.compact && want > 0xFFFFull)) {  // (compact: the entry of a multi-hit read carries NH in 16 bits)
        // does not fit a ring entry
        if (a.compact) {
          payload = sm_emit_wide(a, lo, hi, payload, chr_id, meta);
          meta |= R_DIRECT;
          lo = 0;
          width = 0;
          chr_id = 0;
          if (payload > kPosMask) payload = kPosMask;  // (2^40 wide results cannot happen; keeps the entry well-formed)
        } else {
          sm_emit(a, r, lo, hi, payload, chr_id, meta);
          meta = R_DIRECT;
        }
      }
      const uint64_t e0 = lo | (width << kPosBits), e1 = payload | ((uint64_t)chr_id << kPosBits) | ((uint64_t)meta << 56);
      ring[(size_t)((r - range_begin) & (kSmRing - 1))] =
          make_uint4((uint32_t)e0, (uint32_t)(e0 >> 32), (uint32_t)e1, (uint32_t)(e1 >> 32));

      // the lane moves on to the read it reserved first; what was prefetched for it becomes current, and the
      // read after it moves up (its prefix entry is fetched from now on); the new rn2 is reserved at the start
      r = rn;
      rn = rn2;
      rn2 = ~0ull;
      aux = ((pf & PF_W1) ? A_CUR_W : 0u) | ((pf & PF_K1) ? A_CUR_K : 0u) | ((pf & PF_KH) ? A_CUR_KH : 0u);
      if (kLong) {
        if (pf & PF_P1) aux |= A_CUR_P2;
        w2_lo = nw2_lo;
        w2_hi = nw2_hi;
      }
      w_lo = nw_lo;
      w_hi = nw_hi;
      L = nL;
      ck = nk;
      nw_lo = mw_lo;
      nw_hi = mw_hi;
      nL = mL;
      pf = (pf & PF_W2) ? PF_W1 : 0u;
      st = r < range_end ? S_START : S_IDLE;
    }
  }
  if (lane == 0) {
    WaveInfo wi;
    wi.slots = wave_slots;
    wi.count = wave_multi;
    wi.queue_base = 0;
    a.wave_info[wave] = wi;
  }
  {
    uint32_t t = tally;
    for (int d = 32; d > 0; d >>= 1) t += __shfl_xor(t, d);
    if (lane == 0) atomicAdd(a.index_accesses, (unsigned long long)t);
    if (H) {  // wave-uniform
      uint32_t q = tally_hq, h = tally_hh;
      for (int d = 32; d > 0; d >>= 1) { q += __shfl_xor(q, d); h += __shfl_xor(h, d); }
      if (lane == 0) { atomicAdd(a.hash_stats, (unsigned long long)q); atomicAdd(a.hash_stats + 1, (unsigned long long)h); }
    }
  }
  // what is left in the ring
  for (; flushed < range_end; flushed += 64) sm_flush(a, ring, range_begin, flushed + lane, flushed + lane < range_end);
}

}  // namespace sbwt

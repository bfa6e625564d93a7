// rank_pass.h -- k_rank: the fast pass of an index WITHOUT a text-carrying hash table (no memory for it, SBWT_HASH_TEXT=0,
// SBWT_HASH_ORDER=0: the path the reference itself walks -- 12-mer table, then rank queries).
//
// k_search_sm (search_machine.h) maps such a batch as a lane-level state machine: every loop iteration a wave executes
// the code of EVERY state one of its lanes is in, some 800 VALU instructions for one index line per lane -- on the rank
// path the machine is bound by its own instruction stream (VALU 61 % busy under a kernel that moves 32 G lines/s of the
// 48 the device gathers).  Reads of one batch are alike, though: prefix entry, two or three two-base steps until the
// interval is one row, sa_full, the text comparison.  This kernel runs exactly that, in LOCKSTEP: one read per lane, a
// round of the loop = one index line per lane that still needs one, the whole wave in the same few instructions:
//
//   prefix   the interval after the memoised search prefix: kmer_tab entry, or the c_functions pair (init_exact_match,
//            searcher.hpp:83-96) -- one load
//   steps    exec_exact_match (searcher.hpp:102-105) two bases at a time on the Block2 lines of lo and -- when it lies in
//            another block -- of hi, while the interval holds more than one row: the lines of a round arrive through
//            quad-cooperative LDS-DMA loads, one instruction per line (search_machine.h, S_STEP)
//   sa_full  the one row's text position and its run of LF-consistent positions -- one load
//   text     the rest of the read against the window of the PosLine it ends on, and the final row from the same line
//            (direct.h: eight lanes a line, pieces swizzled against the banks) -- one line
//
// and writes the 16-byte record.  Whatever leaves that pattern -- an interval that dies, several rows left at the end,
// a row the text does not answer for, a window that does not reach, a result no record holds -- is
// left PENDING for the general kernels exactly as k_settle / k_direct leave theirs.  The arithmetic is that of the
// states it replaces.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sbwt_amd.h"
#include "direct.h"
#include "rank_device.h"
#include "search_machine.h"
#include "settle.h"

namespace sbwt {

// a.packed: records of 4 * kPiecesT words (64 or 128 bases); a.out: compact records (SA interval + hit), or -- kPlain --
// a.results / a.hits: the 32-byte records of sbwt_map_batch_device with the hit of a read at hits[its own index]
// kLds: the filter tables and sb_abs are staged in LDS (the launch carries (n_chr + 2 n_npos + 16 n_super) * 8 bytes)
template <int kPiecesT, bool kLds, bool kPlain = false>
__global__ __launch_bounds__(256) void k_rank(const DeviceIndex ix, const SettleArgs a) {
  constexpr uint32_t kPieces = (uint32_t)kPiecesT;
  constexpr bool kLong = kPiecesT >= 2;
  constexpr uint32_t kMaxLen = 64u * kPieces;
  extern __shared__ __attribute__((aligned(16))) uint64_t s_tab[];
  __shared__ __attribute__((aligned(16))) uint4 s_img_all[4][512];  // per wave: the Block2 stage of a round, then the PosLines (8 KB)
  __shared__ uint32_t s_pend_all[4][kDirectBuf];
  static_assert(sizeof(s_img_all[0]) / sizeof(uint4) >= kSmStage, "the stage of a step round fits the wave's image buffer");
  const uint64_t *t_chr = ix.chr_start, *t_nk = ix.npos_key, *t_nv = ix.npos_val, *t_sb = ix.sb_abs;
  if (kLds) {
    const uint32_t n_super = (uint32_t)(((ix.n_blocks - 1) >> (kSuperShift - 6)) + 1);
    uint64_t* s_chr = s_tab;
    uint64_t* s_nk = s_chr + ix.n_chr;
    uint64_t* s_nv = s_nk + ix.n_npos;
    uint64_t* s_sb = s_nv + ix.n_npos;
    for (uint32_t k = threadIdx.x; k < ix.n_chr; k += blockDim.x) s_chr[k] = ix.chr_start[k];
    for (uint32_t k = threadIdx.x; k < ix.n_npos; k += blockDim.x) { s_nk[k] = ix.npos_key[k]; s_nv[k] = ix.npos_val[k]; }
    for (uint32_t k = threadIdx.x; k < n_super * 16; k += blockDim.x) s_sb[k] = ix.sb_abs[k];
    __syncthreads();
    t_chr = s_chr; t_nk = s_nk; t_nv = s_nv; t_sb = s_sb;
  }
  const uint32_t lane = threadIdx.x & 63, sub = lane & 3u;
  const uint32_t K = ix.kmer_order;
  const uint64_t kmask = (1ull << (2 * K)) - 1ull;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint4* const recs = reinterpret_cast<const uint4*>(a.packed);
  uint4* const s_img = s_img_all[threadIdx.x >> 6];
  uint32_t* const s_pend = s_pend_all[threadIdx.x >> 6];
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
  uint32_t lines_n = 0, other_n = 0;

  const uint64_t n_round = (a.n_reads + 63) & ~63ull;  // whole waves stay in the loop together (ballots, DPP, LDS-DMA)
  uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool valid = r < a.n_reads;
  uint4 rec = make_uint4(0, 0, 0, 0), rec2 = rec;
  uint32_t L = 0;
  if (valid) {
    rec = recs[r * kPieces];
    if (kLong) rec2 = recs[r * kPieces + 1];
    L = a.lens[r];
  }
  for (; r < n_round; r += stride) {
    // the record of the read after this one is on its way while this one's lines are awaited
    const uint64_t rn = r + stride;
    const bool nvalid = rn < a.n_reads;
    uint4 nrec = make_uint4(0, 0, 0, 0), nrec2 = nrec;
    uint32_t nL = 0;
    if (nvalid) {
      nrec = recs[rn * kPieces];
      if (kLong) nrec2 = recs[rn * kPieces + 1];
      nL = a.lens[rn];
    }

    const uint64_t w_lo = u64_of(rec.x, rec.y), w_hi = u64_of(rec.z, rec.w);
    const uint64_t w2 = u64_of(rec2.x, rec2.y), w3 = u64_of(rec2.z, rec2.w);  // bases 64..127 (kLong)
    // the 32-base word base j lies in
    auto word_of = [&](uint32_t jj) -> uint64_t {
      if (kLong && (jj & 64u)) return (jj & 32u) ? w3 : w2;
      return (jj & 32u) ? w_hi : w_lo;
    };
    bool pending = valid;
    uint64_t e0 = 0, e1 = 0;  // the record
    if (valid && (L < (uint32_t)kKmer || L > kMaxLen)) {
      e1 = (uint64_t)R_INVALID << 56;  // as k_search_sm reports a read its record cannot hold: nothing mapped
      pending = false;
    }
    // ---- prefix: one load (a read shorter than the memoised prefix takes the 12-mer pair)
    const bool use_tab = ix.kmer_tab != nullptr && L >= K;
    const uint32_t j0 = use_tab ? K : (uint32_t)kKmer;
    bool go = pending;
    uint64_t lo = 0, hi = 0, p = 0;
    uint32_t j = j0;
    bool cmp = false;  // p holds the position the one row left reports; bases [j, L) remain
    uint64_t run_fin = 0;  // LF-consistent positions from the read's last position down (as far as the entry's field counts)
    {
      const uint64_t cw = ~w_lo;  // complemented codes of bases 0..31
      uint64_t ent = 0;
      uint4_a8 pair = {0, 0, 0, 0};
      if (go && use_tab) ent = ix.kmer_tab[cw & kmask];
      if (go && !use_tab) pair = *reinterpret_cast<const uint4_a8*>(ix.c_functions + ((uint32_t)cw & 0xFFFFFFu));
      if (go) ++lines_n;
      if (go && use_tab) {
        const uint64_t width = ent >> kPosBits;
        if (ent & kKmerLocated) {
          const uint64_t pos = ent & kPosMask, run = (ent >> kPosBits) & kKmerRunMax;
          const uint64_t rem = (uint64_t)L - K;
          if (rem >= 1 && run > rem && pos >= rem) { p = pos; run_fin = run - rem; cmp = true; lo = 0; hi = 1; }
          else go = false;
        } else if (width == kKmerWide) {
          go = false;  // (not memoised: the pair; rare -- left to the general kernels)
        } else {
          lo = ent & kPosMask;
          hi = lo + width;
        }
      } else if (go) {
        lo = u64_of(pair.x, pair.y);
        hi = u64_of(pair.z, pair.w);
      }
      if (go && !cmp && lo >= hi) go = false;  // (an empty interval: its record is the general kernels')
    }
    // ---- a prefix of odd order: two-base steps start at even j (the pair never straddles the halves of the record), so
    // one base goes alone first (search_machine.h, S_LF1) -- for the few entries of such a table that are not a located row
    {
      const bool odd = go && !cmp && (j & 1u) && hi - lo > 1 && j < L;
      if (__any(odd)) {
        if (odd) {
          const uint32_t c = ((uint32_t)(word_of(j) >> (2 * (j & 31u))) & 3u) ^ 3u;
          const uint64_t l1 = lf1(ix, lo, c), h1 = lf1(ix, hi, c);
          lines_n += (lo >> 6) == (hi >> 6) ? 1u : 2u;
          lo = l1;
          hi = h1;
          j += 1;
          if (lo >= hi) go = false;  // (the interval died: its record is the general kernels')
        }
      }
    }
    // ---- steps: while the interval holds several rows and two bases are left
    bool second = false;   // this lane's next line is that of hi (lo's block is done: t_lo)
    uint64_t t_lo = 0;
    for (;;) {
      const bool stepping = go && !cmp && hi - lo > 1 && j + 2u <= L;
      if (!__any(stepping)) break;
      const uint64_t wsel = word_of(j);
      const uint32_t two = ((uint32_t)(wsel >> (2 * (j & 31u))) & 15u) ^ 15u;  // complement: the query is RC(read)
      const uint32_t code = ((two & 3u) << 2) | (two >> 2);                    // x = first prepended base, y = second
      const uint64_t x = second ? hi : lo;
      const uint64_t blk = x >> 6;
      const uintptr_t line = reinterpret_cast<uintptr_t>(ix.blocks2 + blk);
      const uint32_t sl_lo = (uint32_t)line, sl_hi = (uint32_t)(line >> 32);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t want = quad_bcast(stepping ? 1u : 0u, k);
        if (!__any(want)) continue;  // wave-uniform
        const uint32_t code_k = quad_bcast(code, k);
        const uintptr_t q = u64_of(quad_bcast(sl_lo, k), quad_bcast(sl_hi, k)) + (sub < 2u ? 16u * sub : 32u + 16u * (code_k >> 2));
        if (want && sub < 3u)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)q,
                                           (__attribute__((address_space(3))) void*)(s_img + kSmStageRound * k), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      if (stepping) {
        ++lines_n;
        const uint4* im = s_img + kSmStageRound * sub + (lane & ~3u);  // round `sub` served this lane
        const uint4 sb0 = im[0], sb1 = im[1], sb2 = im[2];
        const uint64_t q0 = u64_of(sb0.x, sb0.y), q1 = u64_of(sb0.z, sb0.w), q2 = u64_of(sb1.x, sb1.y), q3 = u64_of(sb1.z, sb1.w);
        const uint64_t mask = ((code & 1) ? q0 : ~q0) & ((code & 2) ? q1 : ~q1) & ((code & 4) ? q2 : ~q2) & ((code & 8) ? q3 : ~q3);
        const uint32_t cn = (code & 2u) ? ((code & 1u) ? sb2.w : sb2.z) : ((code & 1u) ? sb2.y : sb2.x);
        const uint64_t base = t_sb[(blk >> (kSuperShift - 6)) * 16 + code] + cn;
        const uint64_t vx = base + (uint64_t)__popcll(mask & below((uint32_t)(x & 63))) - rank2_correction(ix, code, x);
        uint64_t t_hi = 0;
        bool both = second;
        if (second) {
          t_hi = vx;
        } else {
          t_lo = vx;
          if (((hi - 1) >> 6) == blk) {
            t_hi = base + (uint64_t)__popcll(mask & below((uint32_t)(hi - (blk << 6)))) - rank2_correction(ix, code, hi);
            both = true;
          } else {
            second = true;
          }
        }
        if (both) {
          second = false;
          if (t_lo >= t_hi) {
            go = false;  // the interval died inside the step: the first base is replayed alone (S_LF1) by the general kernels
          } else {
            lo = t_lo;
            hi = t_hi;
            j += 2;
          }
        }
      }
      // the next round loads into the stage again: not before these reads have returned
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
    // ---- one row left and bases to go: its position, if the text answers for as many as remain
    if (go && !cmp) {
      if (hi - lo == 1 && j < L) {
        const uint64_t e = ix.sa_full[lo];
        ++lines_n;
        const uint64_t pos = e & kPosMask, run = e >> kPosBits;
        const uint32_t rem = L - j;
        if (e != kNoPosition && run > (uint64_t)rem && pos >= (uint64_t)rem) { p = pos; run_fin = run - rem; cmp = true; }
        else go = false;
      } else {
        go = false;  // several rows at the end of the read, one base left on a wide interval, a row at the very end: general kernels
      }
    }
    // ---- the text comparison: bases [j, L) against the window of the PosLine the read ends on; the final row
    const uint32_t rem = L - j;
    const uint64_t pfin = p - (uint64_t)rem;
    const uint64_t pline = pfin / kPosPerLine;
    const uint32_t slot19 = (uint32_t)(pfin - pline * kPosPerLine);
    const bool need_line = go && cmp && slot19 + rem <= 128u;
    const uint64_t need_mask = __ballot(need_line);
    if (need_mask) {
      const uintptr_t lp = reinterpret_cast<uintptr_t>(ix.pos_lines + pline);
      const uint32_t lp_lo = (uint32_t)lp, lp_hi = (uint32_t)(lp >> 32);
#pragma unroll
      for (int k = 0; k < 8; ++k) {  // (direct.h: eight lanes a line, piece q of read j at 8 j + (q ^ f(j)))
        const int src = 8 * k + (int)(lane >> 3);
        const uint32_t fk = 2u * ((uint32_t)k & 3u) | ((lane >> 4) & 1u);
        const uintptr_t q = u64_of((uint32_t)__shfl((int)lp_lo, src), (uint32_t)__shfl((int)lp_hi, src)) + 16u * ((lane & 7u) ^ fk);
        if ((need_mask >> src) & 1ull)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)q,
                                           (__attribute__((address_space(3))) void*)(s_img + 64 * k), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      __builtin_amdgcn_wave_barrier();
    }
    if (need_line) {
      ++lines_n;
      const uint4* im = s_img + 8u * lane;
      const uint32_t f = ((lane >> 2) & 6u) | ((lane >> 1) & 1u);
      const uint4 ta = im[f], tb = im[1u ^ f];
      const uint32_t ob = 32u + 5u * slot19;
      const uint32_t* iw = reinterpret_cast<const uint32_t*>(im);
      const uint32_t d0 = ob >> 2, d1 = d0 + 1u;
      const uint64_t rw = u64_of(iw[(((d0 >> 2) ^ f) << 2) | (d0 & 3u)], iw[(((d1 >> 2) ^ f) << 2) | (d1 & 3u)]);
      const uint64_t row = (rw >> (8u * (ob & 3u))) & kPosMask;
      // the rem bases of the read from j on against the 128-base window from offset o on (search_machine.h, S_CMP)
      const uint32_t o = 128u - slot19 - rem;
      const uint64_t q0 = u64_of(ta.x, ta.y), q1 = u64_of(ta.z, ta.w), q2 = u64_of(tb.x, tb.y), q3 = u64_of(tb.z, tb.w);
      uint64_t t0, t1, t2 = 0, t3 = 0, r0, r1, r2 = 0, r3 = 0;
      if (o < 64u) {
        sm_funnel128(q0, q1, q2, q3, o, t0, t1);
        if (kLong) sm_funnel128(q2, q3, 0, 0, o, t2, t3);
      } else {
        sm_funnel128(q2, q3, 0, 0, o - 64u, t0, t1);
      }
      // bases [j, j + 128) of the record (j < 64 when kLong is off; a long read that stepped past base 63 keeps no window)
      const uint32_t jo = j & 63u;
      if (kLong && (j & 64u)) {
        sm_funnel128(w2, w3, 0, 0, jo, r0, r1);
      } else {
        sm_funnel128(w_lo, w_hi, w2, w3, jo, r0, r1);
        if (kLong) sm_funnel128(w2, w3, 0, 0, jo, r2, r3);
      }
      const uint64_t d[4] = {r0 ^ t0, r1 ^ t1, r2 ^ t2, r3 ^ t3};
      bool differs = false;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t first = 32u * (uint32_t)q;  // this word covers bases [first, first + 32) of the rem
        uint64_t dq = d[q];
        if (rem <= first) dq = 0;
        else if (rem < first + 32u) dq &= (1ull << (2 * (rem - first))) - 1ull;
        differs = differs || dq != 0;
      }
      if (!differs) {
        // the interval is the one row the chain ends on; its position is pfin
        uint32_t meta = 0;
        uint64_t payload = 0;
        uint32_t chr_id = 0;
        bool fits = true;
        if (1ull > a.limit) {
          meta |= 1u << R_FLAGS_SHIFT;  // SBWT_RF_SUPPRESSED (searcher.hpp:415)
        } else {
          sbwt_hit h;
          if (push_position_keys(ix, t_chr, t_nk, t_nv, pfin, L, a.strand, &h)) {  // searcher.hpp:504-539
            payload = h.pos0;
            chr_id = h.chr_id;
            meta |= R_NH | (h.is_rc ? R_RC : 0u);
            fits = chr_id <= a.ring_chr_max && payload <= kPosMask;
          } else if (walk_may_go_on(ix, pfin, run_fin)) {
            fits = false;  // (rank_device.h: the general kernels)
          }
        }
        if (fits) {  // (a result no record holds stays pending: the general kernels own the wide path)
          e0 = row | (1ull << kPosBits);
          e1 = payload | ((uint64_t)chr_id << kPosBits) | ((uint64_t)meta << 56);
          pending = false;
        }
      }
      // (a read that differs from the text: its dead interval is the general kernels' business)
    }
    if (pending) ++other_n;
    // records leave as whole lines: a pending read's slot is written too (zeros) and overwritten by k_scatter_pending
    if (kPlain) {
      // 32-byte records (sm_emit's): a settled read is an interval of one row, or a read no record can hold
      if (valid && !pending) {
        const uint32_t meta = (uint32_t)(e1 >> 56);
        sbwt_read_result res;
        res.sa_lo = e0 & kPosMask;
        res.sa_hi = res.sa_lo + (e0 >> kPosBits);
        res.first_hit = (meta & R_NH) ? r : 0;
        res.nh = (meta & R_NH) ? 1 : 0;
        res.flags = ((meta >> R_FLAGS_SHIFT) & 3u) | ((meta & R_INVALID) ? kFlagInvalid : 0u);
        a.results[r] = res;
        if (meta & R_NH) {
          sbwt_hit h;
          h.pos0 = e1 & kPosMask;
          h.chr_id = (uint32_t)(e1 >> kPosBits) & 0xFFFFu;
          h.is_rc = (meta & R_RC) ? 1u : 0u;
          a.hits[r] = h;
        }
      }
    } else if (valid) {
      reinterpret_cast<uint4*>(a.out)[r] = make_uint4((uint32_t)e0, (uint32_t)(e0 >> 32), (uint32_t)e1, (uint32_t)(e1 >> 32));
    }
    const uint64_t pm = __ballot(pending);
    if (pm) {
      if (pending) s_pend[n_buf + (uint32_t)__popcll(pm & ((1ull << lane) - 1ull))] = (uint32_t)r;
      n_buf += (uint32_t)__popcll(pm);
      if (n_buf > kDirectBuf - 64) flush_pending();
    }
    valid = nvalid;
    rec = nrec;
    rec2 = nrec2;
    L = nL;
  }
  // (every pending read still wants its hash lookup, where the index has an 8-byte-slot table: none was made here)
  const uint32_t tallies[5] = {0u, 0u, other_n, 0u, lines_n};
  settle_epilogue(a, s_pend, n_buf, tallies, lane);  // (settle.h: one set of atomics per workgroup)
}

}  // namespace sbwt

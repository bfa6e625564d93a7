// search_rounds.h -- k_search_rounds: the general per-read search (searcher.hpp:386-458) in LOCKSTEP rounds.
//
// k_search_sm (search_machine.h) runs the search as a lane-level state machine: no lane ever waits for another, but every
// loop iteration the wave executes the code of every state one of its lanes is in -- some 800 VALU instructions for one
// index line per lane: the machine is bound by its own instruction stream (VALU 61 % busy on the rank path, 18 G lines/s
// on the pending reads of the small-RNA workload where the device gathers 48).  This kernel is the other trade: 64
// consecutive reads per wave at a time, one read per lane, and the loop is a sequence of ROUNDS, each the load and the
// arithmetic of ONE kind of step for the lanes that need it now, skipped when no lane does:
//
//   step     exec_exact_match (searcher.hpp:102-105) for two bases on a Block2 line (quad-cooperative LDS-DMA loads,
//            one instruction per line; lo's block, then hi's when it lies in another)
//   one base the last base of a read, the base behind a prefix of odd order, the replay of a two-base step the interval
//            died in (searcher.hpp:406: the loop stops after the FIRST base if that one emptied it), the base a text
//            comparison failed on -- Block1 lines
//   sa_full  a one-row interval's position: to go on by text comparison, or to report it (the locate of a unique hit)
//   text     the rest of the read against the window of the PosLine it ends on + the final row (eight lanes a line)
//   row      the row of the position a failed comparison stopped at (its dead interval is one base further)
//
// A lane that is through waits for the others of its batch -- reads of one batch are alike, and the instruction stream of
// a round is a tenth of the machine's iteration.  Everything the machine reports is reported here, bit for bit: SA
// interval of every read (dead ones included), NH, flags, unique hits, multi-hit reads queued for the locate kernels with
// their slots (same protocol: search_machine.h, SmArgs), wide results.  Records of up to 128 bases; k_search_sm keeps
// the longer ones, and the hash table (whose located prefixes the fast passes already used) is not consulted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sbwt_amd.h"
#include "rank_device.h"
#include "search_machine.h"

namespace sbwt {

enum RoundAct : uint32_t { RA_NONE = 0, RA_STEP, RA_ONE, RA_SA, RA_LOCATE, RA_CMP, RA_ROW, RA_PAIR };

template <int kPiecesT, bool kLds>  // records of 4 * kPiecesT words; kLds: filter tables and sb_abs staged in LDS (as k_search_sm)
__global__ __launch_bounds__(256) void k_search_rounds(const DeviceIndex ix, const SmArgs a) {
  constexpr uint32_t kPieces = (uint32_t)kPiecesT;
  constexpr bool kLong = kPiecesT >= 2;
  constexpr uint32_t kMaxLen = 64u * kPieces;
  extern __shared__ __attribute__((aligned(16))) uint64_t s_mem[];
  __shared__ __attribute__((aligned(16))) uint4 s_img_all[4][512];  // per wave: the Block2 stage of a round / the PosLines (8 KB)
  static_assert(sizeof(s_img_all[0]) / sizeof(uint4) >= kSmStage, "the stage of a step round fits the wave's image buffer");
  SmTables T{ix.chr_start, ix.npos_key, ix.npos_val, ix.sb_abs};
  if (kLds) {
    const uint32_t n_super = (uint32_t)(((ix.n_blocks - 1) >> (kSuperShift - 6)) + 1);
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
  uint4* const s_img = s_img_all[threadIdx.x >> 6];
  const uint32_t lane = threadIdx.x & 63, sub = lane & 3u;
  const uint64_t lane_lt = (1ull << lane) - 1ull;
  // the wave's contiguous range of reads (as k_search_sm: the locate kernels find a wave's queue at range_begin)
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
  const uint64_t kmask = (1ull << (2 * K)) - 1ull;
  const uint4* const recs = reinterpret_cast<const uint4*>(a.packed);
  uint64_t wave_slots = 0;  // wave-uniform: hit slots handed to the multi-hit reads of this wave so far
  uint32_t wave_multi = 0;  // wave-uniform: how many such reads
  uint32_t tally = 0;       // index lines this lane read

  // A batch runs as many rounds as its longest search needs, and a lane that is through idles (tried: refilling it at once
  // -- every round type in every iteration cost more than it saved).  So the reads of a batch should need about the same
  // number of rounds, and what says most about that is the length: the wave orders every 256 consecutive reads of its range
  // by length (a counting sort over the lengths in steps of two, in LDS) and takes its batches from that order.  Batches of
  // one length, as in a batch of fixed-length reads, stay what they were.
  __shared__ uint32_t s_len_hist_all[4][64];
  __shared__ uint16_t s_perm_all[4][256];
  uint32_t* const s_len_hist = s_len_hist_all[threadIdx.x >> 6];
  uint16_t* const s_perm = s_perm_all[threadIdx.x >> 6];
  uint4 rec = make_uint4(0, 0, 0, 0), rec2 = rec;
  uint32_t L = 0;
  for (uint64_t base = range_begin; base < range_end; base += 64) {
    const uint32_t in_group = (uint32_t)((base - range_begin) & 255u);     // 0, 64, 128, 192
    const uint64_t group = base - in_group;
    const uint32_t n_group = (uint32_t)(range_end - group < 256u ? range_end - group : 256u);
    if (in_group == 0) {  // (wave-uniform) a new group of up to 256 reads: its order
      s_len_hist[lane] = 0;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_wave_barrier();
      uint32_t bkt[4], rnk[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t x = lane + 64u * (uint32_t)k;
        bkt[k] = 0; rnk[k] = 0;
        if (x < n_group) {
          const uint32_t len = a.lens[group + x];
          bkt[k] = len >> 1 < 63u ? len >> 1 : 63u;
          rnk[k] = atomicAdd(&s_len_hist[bkt[k]], 1u);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_wave_barrier();
      const uint32_t mine = s_len_hist[lane];
      uint32_t inc = mine;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)inc, d);
        if ((int)lane >= d) inc += up;
      }
      __builtin_amdgcn_wave_barrier();
      s_len_hist[lane] = inc - mine;  // first place of the bucket
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t x = lane + 64u * (uint32_t)k;
        if (x < n_group) s_perm[s_len_hist[bkt[k]] + rnk[k]] = (uint16_t)x;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_wave_barrier();
      rec = make_uint4(0, 0, 0, 0); rec2 = rec; L = 0;
      if (lane < n_group) {
        const uint64_t r0 = group + s_perm[lane];
        rec = recs[r0 * kPieces];
        if (kLong) rec2 = recs[r0 * kPieces + 1];
        L = a.lens[r0];
      }
    }
    const bool valid = in_group + lane < n_group;
    const uint64_t r = valid ? group + s_perm[in_group + lane] : range_end;
    // the records of the group's next batch are on their way while this one is searched
    uint4 nrec = make_uint4(0, 0, 0, 0), nrec2 = nrec;
    uint32_t nL = 0;
    if (in_group + 64u + lane < n_group) {
      const uint64_t rn = group + s_perm[in_group + 64u + lane];
      nrec = recs[rn * kPieces];
      if (kLong) nrec2 = recs[rn * kPieces + 1];
      nL = a.lens[rn];
    }
    const uint64_t w0 = u64_of(rec.x, rec.y), w1 = u64_of(rec.z, rec.w), w2 = u64_of(rec2.x, rec2.y), w3 = u64_of(rec2.z, rec2.w);
    auto word_of = [&](uint32_t jj) -> uint64_t {  // the 32-base word base jj lies in
      if (kLong && (jj & 64u)) return (jj & 32u) ? w3 : w2;
      return (jj & 32u) ? w1 : w0;
    };
    auto base_at = [&](uint32_t jj) -> uint32_t { return ((uint32_t)(word_of(jj) >> (2 * (jj & 31u))) & 3u) ^ 3u; };  // complemented: the query is RC(read)

    // ---- lane state
    uint64_t lo = 0, hi = 0, p = 0, t_lo = 0, t_hi = 0;
    uint32_t j = (uint32_t)kKmer, flags = 0, act = RA_NONE, code = 0, mode = 0;
    bool done = !valid, located = false, nopos = false, tried = false, second = false;
    bool over = false;  // the search part is over whatever the interval (a replayed or failed base, the last base)
    if (valid && (L < (uint32_t)kKmer || L > kMaxLen)) {
      flags = kFlagInvalid;
      done = true;
    }
    // can the comparison of the bases [jj, L) end in the window of the PosLine of position pos - (L - jj)?
    auto window_reaches = [&](uint64_t pos, uint32_t jj) -> bool {
      const uint64_t pf = pos - (uint64_t)(L - jj);
      return (uint32_t)(pf % kPosPerLine) + (L - jj) <= 128u;
    };
    // ---- the search prefix: kmer_tab entry, else (short read, entry not memoised or not usable) the 12-mer pair
    {
      const uint64_t cw = ~w0;  // complemented codes of bases 0..31
      const bool use_tab = !done && ix.kmer_tab != nullptr && L >= K;
      uint64_t ent = 0;
      if (use_tab) { ent = ix.kmer_tab[cw & kmask]; ++tally; }
      bool want_pair = !done && !use_tab;
      if (use_tab) {
        const uint64_t width = ent >> kPosBits;
        if (ent & kKmerLocated) {
          // one row, already located: straight to the text comparison when it can carry the rest of the read
          const uint64_t pos = ent & kPosMask, run = (ent >> kPosBits) & kKmerRunMax;
          const uint64_t rem = (uint64_t)L - K;
          if (rem >= 1 && run > rem && pos >= rem && window_reaches(pos, K)) {
            p = pos;
            j = K;
            lo = 0;  // the row itself is not needed: a comparison that dies reads it from pos_lines
            hi = 1;
            tried = true;
            act = RA_CMP;
          } else {
            want_pair = true;  // otherwise as if not memoised
          }
        } else if (width == kKmerWide) {
          want_pair = true;
        } else {
          lo = ent & kPosMask;
          hi = lo + width;
          j = K;
        }
      }
      if (__any(want_pair)) {
        if (want_pair) {  // init_exact_match, searcher.hpp:83-96
          const uint4_a8 pr = *reinterpret_cast<const uint4_a8*>(ix.c_functions + ((uint32_t)cw & 0xFFFFFFu));
          ++tally;
          lo = u64_of(pr.x, pr.y);
          hi = u64_of(pr.z, pr.w);
          j = (uint32_t)kKmer;
        }
      }
    }
    // ---- rounds, until every read of the batch is through (searcher.hpp:406-410: one exec_exact_match per remaining base
    // while lo < hi; then the locate of a one-row interval)
    for (;;) {
      if (!done && act == RA_NONE) {
        if (over || lo >= hi || j >= L) {
          const uint64_t m = hi > lo ? hi - lo : 0;
          if (m == 1 && m <= a.limit && !located) act = RA_LOCATE;
          else done = true;
        } else if (hi - lo == 1 && !tried) {
          act = RA_SA;
        } else if (j + 1 == L || (j & 1u)) {
          // the last base, or the base after a prefix of odd order: two-base steps start at even j
          code = base_at(j);
          mode = j + 1 == L ? M_LAST : M_STEP;
          act = RA_ONE;
        } else {
          const uint32_t two = ((uint32_t)(word_of(j) >> (2 * (j & 31u))) & 15u) ^ 15u;
          code = ((two & 3u) << 2) | (two >> 2);  // x = first prepended base, y = second
          second = false;
          act = RA_STEP;
        }
      }
      if (!__any(!done)) break;

      // ---- step: two bases on a Block2 line
      const bool stepping = !done && act == RA_STEP;
      if (__any(stepping)) {
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
          ++tally;
          const uint4* im = s_img + kSmStageRound * sub + (lane & ~3u);  // round `sub` served this lane
          const uint4 sb0 = im[0], sb1 = im[1], sb2 = im[2];
          const uint64_t q0 = u64_of(sb0.x, sb0.y), q1 = u64_of(sb0.z, sb0.w), q2 = u64_of(sb1.x, sb1.y), q3 = u64_of(sb1.z, sb1.w);
          const uint64_t mask = ((code & 1) ? q0 : ~q0) & ((code & 2) ? q1 : ~q1) & ((code & 4) ? q2 : ~q2) & ((code & 8) ? q3 : ~q3);
          const uint32_t cn = (code & 2u) ? ((code & 1u) ? sb2.w : sb2.z) : ((code & 1u) ? sb2.y : sb2.x);
          const uint64_t cbase = T.sb_abs[(blk >> (kSuperShift - 6)) * 16 + code] + cn;
          const uint64_t vx = cbase + (uint64_t)__popcll(mask & below((uint32_t)(x & 63))) - rank2_correction(ix, code, x);
          bool both = second;
          if (second) {
            t_hi = vx;
          } else {
            t_lo = vx;
            if (((hi - 1) >> 6) == blk) {
              t_hi = cbase + (uint64_t)__popcll(mask & below((uint32_t)(hi - (blk << 6)))) - rank2_correction(ix, code, hi);
              both = true;
            } else {
              second = true;
            }
          }
          if (both) {
            second = false;
            if (t_lo >= t_hi) {
              code >>= 2;  // replay the first base alone
              mode = M_REPLAY;
              act = RA_ONE;
            } else {
              lo = t_lo;
              hi = t_hi;
              j += 2;
              act = RA_NONE;
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the stage is loaded again: not before these reads have returned
        __builtin_amdgcn_wave_barrier();
      }

      // ---- one base: get_occ_using_jbwt + c_function on both ends
      const bool one = !done && act == RA_ONE;
      if (__any(one)) {
        if (one) {
          const uint32_t c = code & 3u;
          const uint64_t l1 = lf1(ix, lo, c), h1 = lf1(ix, hi, c);
          tally += (lo >> 6) == (hi >> 6) ? 1u : 2u;
          if (mode == M_REPLAY) {
            // the interval died inside a two-base step: upstream stops after the FIRST base if that one already emptied
            // it (loop condition, searcher.hpp:406), otherwise with the two-base result
            if (l1 >= h1) { lo = l1; hi = h1; } else { lo = t_lo; hi = t_hi; }
          } else {
            lo = l1;
            hi = h1;
            if (mode == M_LAST || mode == M_STEP) j += 1;
          }
          over = mode != M_STEP;  // (M_REPLAY, M_DEAD, M_LAST end the search: searcher.hpp:406)
          act = RA_NONE;
        }
      }

      // ---- sa_full of a one-row interval: to go on by text comparison (RA_SA) or to report its position (RA_LOCATE)
      const bool sa = !done && (act == RA_SA || act == RA_LOCATE);
      if (__any(sa)) {
        if (sa) {
          const uint64_t e = ix.sa_full[lo];
          ++tally;
          if (act == RA_LOCATE) {
            p = e & kPosMask;
            nopos = e == kNoPosition;  // the upstream locate finds nothing
            located = true;
            done = true;
            act = RA_NONE;
          } else {
            const uint64_t pos = e & kPosMask, run = e >> kPosBits;
            const uint32_t rem = L - j;
            tried = true;
            if (e != kNoPosition && run > (uint64_t)rem && pos >= (uint64_t)rem && window_reaches(pos, j)) {
              p = pos;
              act = RA_CMP;
            } else {
              act = RA_NONE;  // the rank queries go on
            }
          }
        }
      }

      // ---- text: bases [j, L) against the window of the PosLine the read ends on; the final row
      const bool cmp = !done && act == RA_CMP;
      const uint32_t rem = L - j;
      const uint64_t pfin = p - (uint64_t)rem;
      const uint64_t pline = pfin / kPosPerLine;
      const uint32_t slot19 = (uint32_t)(pfin - pline * kPosPerLine);
      const uint64_t cmp_mask = __ballot(cmp);
      if (cmp_mask) {
        const uintptr_t lp = reinterpret_cast<uintptr_t>(ix.pos_lines + pline);
        const uint32_t lp_lo = (uint32_t)lp, lp_hi = (uint32_t)(lp >> 32);
#pragma unroll
        for (int k = 0; k < 8; ++k) {  // (direct.h: eight lanes a line, piece q of read j at 8 j + (q ^ f(j)))
          const int src = 8 * k + (int)(lane >> 3);
          const uint32_t fk = 2u * ((uint32_t)k & 3u) | ((lane >> 4) & 1u);
          const uintptr_t q = u64_of((uint32_t)__shfl((int)lp_lo, src), (uint32_t)__shfl((int)lp_hi, src)) + 16u * ((lane & 7u) ^ fk);
          if ((cmp_mask >> src) & 1ull)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)q,
                                             (__attribute__((address_space(3))) void*)(s_img + 64 * k), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (cmp) {
          ++tally;
          const uint4* im = s_img + 8u * lane;
          const uint32_t f = ((lane >> 2) & 6u) | ((lane >> 1) & 1u);
          const uint4 ta = im[f], tb = im[1u ^ f];
          const uint32_t ob = 32u + 5u * slot19;
          const uint32_t* iw = reinterpret_cast<const uint32_t*>(im);
          const uint32_t d0 = ob >> 2, d1 = d0 + 1u;
          const uint64_t rw = u64_of(iw[(((d0 >> 2) ^ f) << 2) | (d0 & 3u)], iw[(((d1 >> 2) ^ f) << 2) | (d1 & 3u)]);
          const uint64_t row = (rw >> (8u * (ob & 3u))) & kPosMask;
          const uint32_t o = 128u - slot19 - rem;  // the window from offset o on
          const uint64_t q0 = u64_of(ta.x, ta.y), q1 = u64_of(ta.z, ta.w), q2 = u64_of(tb.x, tb.y), q3 = u64_of(tb.z, tb.w);
          uint64_t t0, t1, t2 = 0, t3 = 0, r0, r1, r2 = 0, r3 = 0;
          if (o < 64u) {
            sm_funnel128(q0, q1, q2, q3, o, t0, t1);
            if (kLong) sm_funnel128(q2, q3, 0, 0, o, t2, t3);
          } else {
            sm_funnel128(q2, q3, 0, 0, o - 64u, t0, t1);
          }
          const uint32_t jo = j & 63u;
          if (kLong && (j & 64u)) {
            sm_funnel128(w2, w3, 0, 0, jo, r0, r1);
          } else {
            sm_funnel128(w0, w1, w2, w3, jo, r0, r1);
            if (kLong) sm_funnel128(w2, w3, 0, 0, jo, r2, r3);
          }
          const uint64_t d[4] = {r0 ^ t0, r1 ^ t1, r2 ^ t2, r3 ^ t3};
          uint32_t t = rem;  // bases that agree
#pragma unroll
          for (int q = 3; q >= 0; --q) {
            const uint32_t first = 32u * (uint32_t)q;  // this word covers bases [first, first + 32) of the rem
            uint64_t dq = d[q];
            if (rem <= first) dq = 0;
            else if (rem < first + 32u) dq &= (1ull << (2 * (rem - first))) - 1ull;
            if (dq) t = first + ((uint32_t)(__ffsll((unsigned long long)dq) - 1) >> 1);
          }
          j += t;
          p -= t;
          if (t < rem) {
            act = RA_ROW;  // the chain dies at base j: exec_exact_match on the row of text position p
          } else {
            lo = row;  // the row the chain ends on; its position is p
            hi = lo + 1;
            located = true;
            act = RA_NONE;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
      }

      // ---- the row of position p (a comparison stopped there): its interval after the base that differs is the dead one
      const bool rowq = !done && act == RA_ROW;
      if (__any(rowq)) {
        if (rowq) {
          const uint64_t line = p / kPosPerLine;
          const uint32_t ob = 32u + 5u * (uint32_t)(p - line * kPosPerLine);
          const uint64_t rw = *reinterpret_cast<const u64_a4*>(reinterpret_cast<const char*>(ix.pos_lines + line) + (ob & ~3u));
          ++tally;
          lo = (rw >> (8u * (ob & 3u))) & kPosMask;
          hi = lo + 1;
          code = base_at(j);
          mode = M_DEAD;
          act = RA_ONE;
        }
      }
    }

    // ---- the batch is through: suppression (searcher.hpp:415), filters, multi-hit slots, records (as k_search_sm)
    const uint64_t m = valid && hi > lo ? hi - lo : 0;
    // one row: its position through the filters (searcher.hpp:504-539).  A position they drop, on a row below which the
    // tables are not LF-consistent over the sampling distance: upstream's walk may surface again (rank_device.h,
    // walk_may_go_on) -- the read is located like a multi-hit read
    sbwt_hit h1;
    const bool one_row = valid && m == 1 && m <= a.limit && !nopos;
    const bool passes = one_row && push_position_keys(ix, T.chr, T.npos_key, T.npos_val, p, L, a.strand, &h1);
    bool walk1 = false;
    const bool dropped = one_row && !passes;
    if (__any(dropped)) {
      if (dropped) {
        const uint64_t e = ix.sa_full[lo];
        ++tally;
        walk1 = e != kNoPosition && walk_may_go_on(ix, e & kPosMask, e >> kPosBits);
      }
    }
    const uint64_t want = (valid && (m > 1 || walk1) && m <= a.limit) ? m : 0;
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
        a.multi_queue[range_begin + wave_multi + (uint32_t)__popcll(mm & lane_lt)] = (uint32_t)r;
      }
      wave_slots += total;
      wave_multi += (uint32_t)__popcll(mm);
    }
    if (valid) {
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
      meta |= ((flags & 3u) << R_FLAGS_SHIFT) | ((flags & kFlagInvalid) ? R_INVALID : 0u);
      uint64_t width = hi - lo;
      bool direct = false;
      if (hi < lo || width >= a.ring_width_max || chr_id > a.ring_chr_max || payload > kPosMask || (a.compact && want > 0xFFFFull)) {
        // does not fit a 16-byte entry
        if (a.compact) {
          payload = sm_emit_wide(a, lo, hi, payload, chr_id, meta);
          meta |= R_DIRECT;
          lo = 0;
          width = 0;
          chr_id = 0;
          if (payload > kPosMask) payload = kPosMask;
        } else {
          sm_emit(a, r, lo, hi, payload, chr_id, meta);
          direct = true;
        }
      }
      if (a.compact) {
        const uint64_t e0 = lo | (width << kPosBits), e1 = payload | ((uint64_t)chr_id << kPosBits) | ((uint64_t)meta << 56);
        reinterpret_cast<uint4*>(a.compact)[r] = make_uint4((uint32_t)e0, (uint32_t)(e0 >> 32), (uint32_t)e1, (uint32_t)(e1 >> 32));
      } else if (!direct) {
        sm_emit(a, r, lo, lo + width, payload, chr_id, meta);
      }
    }
    rec = nrec;
    rec2 = nrec2;
    L = nL;
  }
  if (lane == 0) {
    WaveInfo wi;
    wi.slots = wave_slots;
    wi.count = wave_multi;
    wi.queue_base = 0;
    a.wave_info[wave] = wi;
  }
  // the tally: one atomic per workgroup (all waves of a launch end together, and atomics on one line serialise: settle.h)
  __shared__ uint32_t s_tally[4];
  for (int d = 32; d > 0; d >>= 1) tally += __shfl_xor(tally, d);
  if (lane == 0) s_tally[threadIdx.x >> 6] = tally;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long t4 = (unsigned long long)s_tally[0] + s_tally[1] + s_tally[2] + s_tally[3];
    if (t4) atomicAdd(a.index_accesses, t4);
  }
}

}  // namespace sbwt

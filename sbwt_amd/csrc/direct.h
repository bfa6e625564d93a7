// direct.h -- k_direct: the two-access fast pass (compact records with SA intervals; records of up to 128 bases).
//
// k_settle (settle.h) answers a read from its hash bucket alone, which takes two things: the caller wants no SA
// interval, and the bases behind the located prefix fit the 32 the slot carries.  This kernel drops both conditions at
// the price of one more access: a read of hash_order + 1 .. hash_order + 142 bases whose first hash_order bases are
// a located row of the table compares ALL its remaining bases with the text in front of that row's position -- the
// first 32 of them in the bucket's slot, the last 110 .. 128 in the window of the PosLine of the position it ends on
// (device_layout.h), which overlap or meet for such a read -- and finds the row the chain ends on in that same line: the
// same two lines the state machine reads for such a read (search_machine.h, S_CMP), but as a streaming kernel: one
// read per lane, no state, and every line touched by exactly one load instruction.  The PosLines of a wave's 64 reads
// come in through eight LDS-DMA loads (global_load_lds, 16 bytes per lane): in load k lane l fetches piece l % 8 of
// the line of the wave's read 8 k + l / 8, so the 128 bytes of a line are one request of one instruction and land in
// LDS as one image the read's own lane then reads its text window and row from.
//
// Everything else stays PENDING exactly as in k_settle (prefix not in the table, run of LF-consistent positions too
// short, a mismatch whose dead interval the caller wants, a record no 16 bytes hold) and goes through the general
// kernels as a dense batch.  The arithmetic is that of the states it replaces (searcher.hpp:102-105 for a one-row
// interval, :415, :504-539).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sbwt_amd.h"
#include "rank_device.h"
#include "search_machine.h"
#include "settle.h"

namespace sbwt {

constexpr uint32_t kDirectBuf = 256;  // pending indices a wave holds in LDS before it reserves list slots for them

// a.packed: records of 4 * kPieces words (64, 128 or 192 bases); a.hits_only as SBWT_MAP_HITS_ONLY
template <int kPiecesT, bool kLds, bool kPlain = false>  // kLds: the filter tables are staged in LDS (as in k_settle); kPlain: 32-byte records + hits[]
__global__ __launch_bounds__(256) void k_direct(const DeviceIndex ix, const SettleArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint64_t s_tab[];
  __shared__ __attribute__((aligned(16))) uint4 s_img_all[4][512];  // per wave: the PosLines of its 64 reads (8 KB)
  __shared__ uint32_t s_pend_all[4][kDirectBuf];
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
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t H = ix.hash_order;
  const uint64_t hmask = (1ull << (2 * H)) - 1ull;
  const uint32_t tmask = (1u << ix.hash_tag_bits) - 1u;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  constexpr uint32_t kPieces = (uint32_t)kPiecesT;  // 16-byte pieces of a record
  constexpr bool kLong = kPiecesT >= 2, kLong3 = kPiecesT >= 3;
  constexpr uint32_t kMaxLen = 64u * kPieces;
  const uint4* const recs = reinterpret_cast<const uint4*>(a.packed);
  const uint4* const tab = reinterpret_cast<const uint4*>(ix.hash_tab);
  static_assert(sizeof(s_img_all[0]) / sizeof(uint4) >= kBktLds, "settle_lookup stages its buckets in the wave's image buffer");
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
  uint32_t looked = 0, found_n = 0, other_n = 0, lines_n = 0, absent_n = 0;

  const uint64_t n_round = (a.n_reads + 63) & ~63ull;  // whole waves stay in the loop together (ballots, DPP, LDS-DMA)
  uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool valid = r < a.n_reads;
  uint4 rec = make_uint4(0, 0, 0, 0), rec2 = rec, rec3 = rec;
  uint32_t L = 0;
  if (valid) {
    rec = recs[r * kPieces];
    if (kLong) rec2 = recs[r * kPieces + 1];
    if (kLong3) rec3 = recs[r * kPieces + 2];
    L = a.lens[r];
  }
  for (; r < n_round; r += stride) {
    // the record of the read after this one is on its way while this one's lines are awaited
    const uint64_t rn = r + stride;
    const bool nvalid = rn < a.n_reads;
    uint4 nrec = make_uint4(0, 0, 0, 0), nrec2 = nrec, nrec3 = nrec;
    uint32_t nL = 0;
    if (nvalid) {
      nrec = recs[rn * kPieces];
      if (kLong) nrec2 = recs[rn * kPieces + 1];
      if (kLong3) nrec3 = recs[rn * kPieces + 2];
      nL = a.lens[rn];
    }

    const uint64_t w0 = u64_of(rec.x, rec.y), w1 = u64_of(rec.z, rec.w), w2 = u64_of(rec2.x, rec2.y), w3 = u64_of(rec2.z, rec2.w);
    const uint64_t w4 = u64_of(rec3.x, rec3.y), w5 = u64_of(rec3.z, rec3.w);
    bool pending = valid;
    uint64_t e0 = 0, e1 = 0;  // the record
    if (valid && (L < (uint32_t)kKmer || L > kMaxLen)) {
      e1 = (uint64_t)R_INVALID << 56;  // as k_search_sm reports a read its record cannot hold: nothing mapped
      pending = false;
    }
    const uint32_t rem = L - H;
    // (the slot's text covers the first 32 bases behind the prefix, the window of the last position's PosLine the last
    // 128 - slot, slot <= 18: whether they meet is known once the position is)
    // (on a hash_complete index a longer read is looked up too when only hits are wanted: a prefix that occurs nowhere
    // settles it)
    const bool look = pending && L >= H && (rem <= 32u + 128u || (a.hits_only && ix.hash_complete));
    const uint64_t vv = hash_mix((~w0) & hmask, H);
    const uint64_t bucket = look ? __umul64hi(vv << (64u - 2u * H), ix.hash_buckets) : 0ull;
    const uint32_t tag = (uint32_t)vv & tmask;
    if (look) ++looked;
    uint64_t slot = ~0ull, text = 0, slot2 = ~0ull, text2 = 0;
    bool absent = false;
    settle_lookup(ix, tab, bucket, tag, tmask, look, lane, s_img, slot, text, slot2, text2, absent, lines_n);  // (s_img: free until the lines arrive)

    bool missed = false;    // looked up and not in the table: the general kernels need not look again
    bool need_line = false; // second access: the PosLine of the position the read ends on
    uint64_t pfin = 0, run_fin = 0;  // run_fin: LF-consistent positions from pfin down (as far as the slot's field counts)
    if (look) {
      uint64_t pos = (slot >> 24) & kHashPosMask, run = (slot & 0xFFFFFFull) >> ix.hash_tag_bits;
      const bool twin = slot != ~0ull && (slot & kHashTwin) != 0;
      // (run 0: a marker -- the prefix occurs, on several rows or on one not located; so does a twin without its partner)
      missed = slot == ~0ull || run == 0 || (twin && (slot2 == ~0ull || !(slot2 & kHashTwin)));
      if (slot == ~0ull && absent && ix.hash_complete && a.hits_only) {
        // the first H bases occur nowhere in the text: the upstream search ends on an empty interval (searcher.hpp:406)
        e1 = (uint64_t)R_NOSA << 56;
        pending = false;
        ++absent_n;
      } else if (!missed) {
        ++found_n;
        // the first min(rem, 32) bases behind the prefix against the slot's text: a read that differs there needs no
        // second access (no hit; its dead interval, when wanted, is the general kernels' business)
        uint64_t r0, r1;
        sm_funnel128(w0, w1, w2, w3, H, r0, r1);
        const uint64_t cmask = rem < 32u ? (1ull << (2 * rem)) - 1ull : ~0ull;
        uint64_t diff = (r0 ^ text) & cmask;
        bool usable = run > (uint64_t)rem && pos >= (uint64_t)rem;
        if (twin) {
          // two rows hold the prefix (settle.h): the search goes on with the one whose text the read goes on with; when
          // both do over these bases the general kernels take the read
          const uint64_t pos2 = (slot2 >> 24) & kHashPosMask, run2 = (slot2 & 0xFFFFFFull) >> ix.hash_tag_bits;
          const uint64_t diff2 = (r0 ^ text2) & cmask;
          usable = usable && run2 > (uint64_t)rem && pos2 >= (uint64_t)rem && (diff != 0 || diff2 != 0);
          if (diff != 0 && diff2 == 0) { pos = pos2; run = run2; diff = 0; }
        }
        if (usable) {
          const uint32_t s19 = (uint32_t)((pos - rem) % kPosPerLine);
          if (rem > 32u + 128u - s19) {
            // the slot's text and the window leave a gap: the general kernels
          } else if (diff == 0) {
            need_line = true;
            pfin = pos - rem;
            run_fin = run - rem;
          } else if (a.hits_only) {
            e1 = (uint64_t)R_NOSA << 56;
            pending = false;
          }
        }
      }
    }
    // ---- the PosLines of the wave's reads, one LDS-DMA request per line
    const uint64_t need_mask = __ballot(need_line);
    const uint64_t line = pfin / kPosPerLine;
    const uint32_t slot19 = (uint32_t)(pfin - line * kPosPerLine);
    if (need_mask) {
      const uintptr_t lp = reinterpret_cast<uintptr_t>(ix.pos_lines + line);
      const uint32_t lp_lo = (uint32_t)lp, lp_hi = (uint32_t)(lp >> 32);
      // Load k brings the lines of reads 8k .. 8k+7, eight lanes a line, to s_img[64k ..): read j finds its line at
      // s_img[8j .. 8j+8).  WHICH piece a lane fetches is swizzled by its read: piece q of read j lies at 8j + (q ^ f(j)),
      // f(j) = 2 (j/8 mod 4) + (j/2 mod 2), so that the sixteen lanes ds_read_b128 serves in one LDS cycle (lanes 0-3,
      // 12-15, 20-27 ...) read sixteen different 16-byte slots of the 256-byte bank row when every lane reads piece q of
      // its own line (unswizzled they hit two slots, eight lanes each: SQ_LDS_BANK_CONFLICT was half of the LDS cycles)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int src = 8 * k + (int)(lane >> 3);
        const uint32_t fk = 2u * ((uint32_t)k & 3u) | ((lane >> 4) & 1u);  // f(src)
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
      const uint4* im = s_img + 8u * lane;  // this read's line, its pieces swizzled (above)
      const uint32_t f = ((lane >> 2) & 6u) | ((lane >> 1) & 1u);
      const uint4 ta = im[f], tb = im[1u ^ f];
      const uint32_t ob = 32u + 5u * slot19;
      const uint32_t* iw = reinterpret_cast<const uint32_t*>(im);
      const uint32_t d0 = ob >> 2, d1 = d0 + 1u;  // the two words that hold the 40 bits (d1 <= 31)
      const uint64_t rw = u64_of(iw[(((d0 >> 2) ^ f) << 2) | (d0 & 3u)], iw[(((d1 >> 2) ^ f) << 2) | (d1 & 3u)]);
      const uint64_t row = (rw >> (8u * (ob & 3u))) & kPosMask;
      // the last nb bases of the read, [j0, L), against the 128-base window from offset o on (search_machine.h, S_CMP);
      // bases [H, j0) -- at most 32 -- were compared with the slot's text above
      const uint32_t nb = rem < 128u - slot19 ? rem : 128u - slot19;
      const uint32_t o = 128u - slot19 - nb, j0 = L - nb;
      const uint64_t q0 = u64_of(ta.x, ta.y), q1 = u64_of(ta.z, ta.w), q2 = u64_of(tb.x, tb.y), q3 = u64_of(tb.z, tb.w);
      uint64_t t0, t1, t2 = 0, t3 = 0, r0, r1, r2, r3;
      if (o < 64u) {
        sm_funnel128(q0, q1, q2, q3, o, t0, t1);
        sm_funnel128(q2, q3, 0, 0, o, t2, t3);
      } else {
        sm_funnel128(q2, q3, 0, 0, o - 64u, t0, t1);
      }
      sm_funnel128(w0, w1, w2, w3, j0, r0, r1);  // j0 <= H + 32 < 64
      sm_funnel128(w2, w3, w4, w5, j0, r2, r3);
      const uint64_t d[4] = {r0 ^ t0, r1 ^ t1, r2 ^ t2, r3 ^ t3};
      bool differs = false;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t first = 32u * (uint32_t)q;  // this word covers bases [first, first + 32) of the nb
        uint64_t dq = d[q];
        if (nb <= first) dq = 0;
        else if (nb < first + 32u) dq &= (1ull << (2 * (nb - first))) - 1ull;
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
      } else if (a.hits_only) {
        e1 = (uint64_t)R_NOSA << 56;
        pending = false;
      }
    }
    if (pending && !missed) ++other_n;
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
    rec3 = nrec3;
    L = nL;
  }
  const uint32_t tallies[5] = {looked, found_n, other_n, absent_n, lines_n};
  settle_epilogue(a, s_pend, n_buf, tallies, lane);  // (settle.h: one set of atomics per workgroup)
}

}  // namespace sbwt

// device_layout.h -- HBM layout of the FM-index used by the gfx950 kernels.
//
// The on-disk format (prefix.t_table.bwt) is untouched; at load time the tables of
// Aligner_table<BWT_Aligner> (aligner_table.hpp:98-161) are re-laid so that ONE random HBM access
// answers one (or two) rank steps.  Measured on MI355X (tools/gather_bench.hip): dependent random
// gathers run at ~48 G accesses/s for any granule of 16..128 bytes as long as one wave instruction
// touches few distinct lines, i.e. the walk is bound by the random-access (row-activate) rate, not
// by bytes, up to 128 B per access.  So each access is made to do as much as 128 bytes allow.
#pragma once
#include <stdint.h>

namespace sbwt {

constexpr int kKmer = 12;                           // searcher.hpp:299 c_functions_interval
constexpr uint64_t kNumKmer = 1ull << (2 * kKmer);  // 4^12

// 64-byte block: 64 BWT positions, single-base rank (get_occ_using_jbwt) in one access.
// Used for the odd step of the search, the multi-hit DFS (find_possible) and the load-time walks.
struct alignas(64) Block1 {
  uint64_t cnt[4];   // C[c] + #c in bwt[0 .. 64b), '$' counted as 'A' exactly as occ_function does
  uint64_t plane_lo; // bit j = low bit of the 2-bit code of bwt[64b + j]
  uint64_t plane_hi; // bit j = high bit
  uint64_t sampled;  // bit j = fbwt[64b+j] == 0 and (64b+j) is a key of location_table
  uint64_t sampled_base; // number of sampled positions before this block = index into loc_vals
};
static_assert(sizeof(Block1) == 64, "Block1 is one 64-byte granule");

// 128-byte block: 64 BWT positions, TWO backward-search steps in one access.
// code(j) = 4 * x + y,  x = bwt[j] (first prepended base), y = bwt[LF(j)] (second prepended base).
// rank2(code, i) = sb_abs[i >> kSuperShift][code] + cnt[code] + popcount(match(code) & below(i & 63))
// and  C[y] + Occ(y, C[x] + Occ(x, i)) == rank2(4x+y, i)  holds arithmetically for any table content
// (LF restricted to the x-positions is an order-preserving bijection onto [C[x], C[x+1]) by definition).
struct alignas(128) Block2 {
  uint64_t plane[4];  // plane[k] bit j = bit k of code(64b + j)   (k = 0,1: y ; k = 2,3: x)
  uint32_t cnt[16];   // #code in bwt2[superblock start .. 64b)
  uint64_t chain_rows;  // bit j: row 64b + j has an entry in chain_tab (below); rides in what the line had to spare
  uint64_t chain_base;  // index in chain_tab of the first such row of this block
  uint64_t reserved[2];
};
static_assert(sizeof(Block2) == 128, "Block2 is one 128-byte granule");
constexpr int kSuperShift = 31;  // superblock = 2^31 positions, so relative counts fit 32 bits

// 128-byte line per 19 consecutive text positions q0 .. q0+18: what the END of a text comparison needs, in one
// access.  rows: the row whose suffix starts at each position (inverse of sa_full, 40 bits each, little endian, at
// byte 5 s).  text: the 128 bases textrc[xs .. xs + 128), xs = n - 2 - q0 - 127 (zero where x < 0), i.e. the text in
// front of these positions as a read meets it: a read that ends its comparison on position q0 + s after `rem`
// more bases compares them with the bases from offset 128 - (s + rem) on, whenever s + rem <= 128.
constexpr uint32_t kPosPerLine = 19;
struct alignas(128) PosLine {
  uint32_t text[8];
  uint8_t rows[96];
};
static_assert(sizeof(PosLine) == 128, "PosLine is one 128-byte granule");
typedef uint64_t u64_a4 __attribute__((aligned(4)));  // 8-byte load from a 4-byte aligned address
struct __attribute__((aligned(8))) uint4_a8 { uint32_t x, y, z, w; };  // 16-byte load from an 8-byte aligned address

constexpr uint64_t kNoPosition = ~0ull;  // sa_full sentinel: the upstream unique-hit walk reports nothing
// sa_full entry = text position (low 40 bits) | run (high 24 bits, saturating): the number of consecutive text
// positions p, p-1, ..., p-run+1 whose rows are LF-consistent (see device_index.hip, k_verify_rows), i.e. how far
// a size-1 interval can be carried by text comparison instead of rank queries.
constexpr uint64_t kKmerWide = (1ull << 24) - 1;
constexpr uint32_t kMaxKmerOrder = 17;  // widest memoised search prefix (4^17 entries of 8 B = 137 GB)
// kmer_tab entry of a prefix whose interval is ONE row that sa_full locates: bit 63, text position (low 40 bits),
// run (23 bits, saturating) -- the search goes straight to the text comparison, without the sa_full access
constexpr uint64_t kKmerLocated = 1ull << 63;
constexpr uint64_t kKmerRunMax = (1ull << 23) - 1;
// hash_tab: located search prefixes of a LONGER order than a direct table can hold (default 20 bases).  Buckets of
// four 8-byte slots (32 B: two 16-byte loads; the search kernel is bound by load requests as much as by lines); a slot = text position << 24 | run (saturating) << tag_bits | tag, all ones = empty.
// With hash_text the slots are 16 bytes (a bucket = 64 B, four loads): the second word holds the 32 bases of text
// behind the prefix, so that a read of up to hash_order + 32 bases is settled by the bucket alone when only its hits
// are wanted (SBWT_MAP_HITS_ONLY): equal -> the one hit, different -> the interval dies; no further access.
// A K-mer v (complemented read codes, as for kmer_tab) is mixed by an odd multiplier modulo 4^K (a bijection), the
// bucket is the mixed value scaled to the bucket count (so the values of one bucket form an interval shorter than
// 2^(tag_bits - 1)) and the tag its low tag_bits bits: bucket and tag identify v exactly.  K-mers whose upstream
// search interval is ONE located row are in the table with their position (of those, the few that found their
// buckets full are left out): a miss falls back to kmer_tab / c_functions, so the table never changes a result.
// With hash_text every OTHER K-mer whose interval is not empty has a MARKER slot (run = 0: "occurs, not settled
// here"), which is what lets a miss prove that a read maps nowhere (hash_complete).
// TWINS: a K-mer whose interval is exactly TWO rows, both located, has two slots with the same tag, each with its own
// position, run and text, flagged kHashTwin (the top bit of the position field; no position reaches it: n < 2^39).  Under
// exec_exact_match a row of the interval survives a base iff its bwt character is that base, i.e. -- where LF-consistent --
// iff the text in front of its position continues with it: a read whose further bases equal the text of ONE twin ends on
// that twin's chain alone (one hit), of neither on an empty interval, of both stays with the general kernels.  A twin
// whose partner is not in the same bucket, and a twin to any lookup that does not know twins, is a marker.
constexpr uint64_t kHashTwin = 1ull << 63;
constexpr uint64_t kHashPosMask = (1ull << 39) - 1ull;  // the position of a slot, twin or not: (slot >> 24) & kHashPosMask
constexpr uint64_t kHashMul = 0x9E3779B97F4A7C15ull;
constexpr uint32_t kHashSlots = 4;
constexpr uint32_t kHashMaxOrder = 28;
__host__ __device__ inline uint64_t hash_mix(uint64_t v, uint32_t order) { return (v * kHashMul) & ((1ull << (2 * order)) - 1ull); }

// chain_tab: the depth-first locate of ONE row, memoised, for the rows sa_full + the text do not answer for (rows of
// LF-inconsistent regions: an index whose ties beyond sort_len are unordered, as upstream's own builder leaves them).
// find_possible (searcher.hpp:460-503) reaches the rows of a node from the rows of its parent through LF, base by base;
// seen from ONE row i of the read's interval that is the chain r_0 = i, r_{d+1} = LF(r_d): the bases it prepends
// (c_d = bwt[r_d]) say which node holds the chain at every depth, and every sampled r_d is a hit (depth d, position
// d + SA[r_d]).  An entry = 32 bytes: the up to 63 bases of the chain (two words, first base most significant), meta =
// number of events | chain length << 8 | index of the further events in chain_ev << 16, and the first event
// (depth << 58 | position).  Where the tables are LF-consistent the chain is the text and has ONE event: those rows have
// no entry.  multi_locate.h (k_multi_chain) puts a read's chains back together into upstream's emission order.
struct ChainEntry {
  uint64_t k0, k1, meta, ev0;
};
static_assert(sizeof(ChainEntry) == 32, "two 16-byte loads");
constexpr uint64_t kChainEvOffShift = 16;

constexpr int kPosBits = 40;
constexpr uint64_t kPosMask = (1ull << kPosBits) - 1;
constexpr uint64_t kRunMax = (1ull << (64 - kPosBits)) - 1;

// Everything the kernels need, passed by value.
struct DeviceIndex {
  uint64_t n;               // number of suffixes
  uint64_t first_location;  // SA index whose BWT char is '$' (stored as 'A')
  uint64_t real_size;       // Aligner_table::_realSize
  uint64_t interval;        // DFS depth limit and SA sampling distance
  uint64_t n_blocks;        // blocks cover positions 0 .. n (inclusive), = n/64 + 1
  // the two positions of the 2-mer BWT whose code is undefined because one base is '$'
  uint64_t special_pos[2];  // [0] = first_location (x = '$'), [1] = position whose LF is first_location (y = '$'), or ~0
  uint32_t special_code[2]; // code stored there; rank2 subtracts it back out
  uint64_t c_function[4];   // C['A'], C['C'], C['G'], C['T']
  const uint64_t* c_functions;  // 4^12 + 1 entries, verbatim from the file
  const Block1* blocks1;
  const Block2* blocks2;
  const uint64_t* sb_abs;   // [n_super][16]: K2[code] + #code before the superblock
  const uint64_t* loc_vals; // SA value of the k-th sampled position (location_table[k].second)
  const uint64_t* sa_full;  // memoised unique-hit locate: result of the upstream walk started at i (+ run, see above)
  const PosLine* pos_lines; // inverse of sa_full (row whose walk reports text position p) + the text in front of it
  // memoised search prefix: entry v = interval after init_exact_match + exec_exact_match over the first kmer_order
  // bases whose complemented 2-bit codes are v (low 40 bits lo, high 24 bits width; kKmerWide = not memoised; kKmerLocated form below)
  const uint64_t* kmer_tab; uint32_t kmer_order;  // kmer_order = 12: no table
  const uint64_t* hash_tab; uint64_t hash_buckets; uint32_t hash_order; uint32_t hash_tag_bits;  // hash_order = 0: none
  uint32_t hash_group;      // 1, 2, 4 or 8 (above 1 with hash_text only): a prefix whose bucket b was full may sit in b ^ 1 (the other half of
                            // its 128-byte line), b ^ 2, ... b ^ (hash_group - 1): the first of them that had a free slot
  uint32_t hash_text;       // 1: 16-byte slots, the second word = the 32 bases textrc[n-1-pos .. n-1-pos+32) a read's next bases must equal
  uint32_t hash_complete;   // 1: EVERY prefix of hash_order bases whose upstream search interval is not empty has a slot (a located
                            // one, or a marker: run = 0) unless its two buckets are full -- a prefix that is in neither bucket while
                            // one of them has a free slot occurs nowhere: the read maps nowhere (device_index.hip, k_hash_cover)
  const ChainEntry* chain_tab;  // null: none (device_layout.h above); indexed through Block2::chain_rows / chain_base
  const uint64_t* chain_ev;     // events of the chains beyond their first
  const uint32_t* textrc;   // 2-bit text implied by the tables, reverse-complemented: textrc[x] = 3 - bwt[row of position n-1-x]
  // push_position tables (searcher.hpp:504-539)
  const uint64_t* chr_start; uint32_t n_chr;
  const uint32_t* chr_class;  // null unless chromosome names repeat: index of the first chromosome with the same name (upstream compares names, :526)
  const uint64_t* npos_key; const uint64_t* npos_val; uint32_t n_npos;
};

}  // namespace sbwt

// pack.hpp -- ASCII read -> 2-bit record (encoding documented in include/sbwt_amd.h)
#pragma once
#include <stdint.h>
#include <string.h>

#include "../../include/sbwt_amd.h"
#include "device_layout.h"

namespace sbwt {

namespace packdetail {
constexpr uint64_t kOnes = 0x0101010101010101ull;
// 0x80 in every byte of v that is not zero
inline uint64_t nonzero_bytes(uint64_t v) {
  return (((v & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | v) & 0x8080808080808080ull;
}
// eight ASCII bases (little endian, first base in the low byte) -> 16 bits of 2-bit codes, first base lowest;
// *bad collects a non-zero value when a byte is none of A, C, G, T
inline uint32_t pack8(uint64_t x, uint64_t* bad) {
  *bad |= nonzero_bytes(x ^ (kOnes * 'A')) & nonzero_bytes(x ^ (kOnes * 'C')) & nonzero_bytes(x ^ (kOnes * 'G')) &
          nonzero_bytes(x ^ (kOnes * 'T'));
  // A 0x41, C 0x43, G 0x47, T 0x54: bits 2..1 read 0, 1, 3, 2; xor with bit 2 gives 0, 1, 2, 3
  uint64_t y = ((x >> 1) & (kOnes * 3)) ^ ((x >> 2) & kOnes);
  y = (y | (y >> 6)) & 0x000f000f000f000full;
  y = (y | (y >> 12)) & 0x000000ff000000ffull;
  return (uint32_t)((y | (y >> 24)) & 0xffffull);
}
}  // namespace packdetail

// returns SBWT_OK, SBWT_ERR_ILLEGAL_CHAR (searcher.hpp:399), SBWT_ERR_TOO_SHORT (searcher.hpp:404)
// or SBWT_ERR_ARG when the read does not fit the record
inline int pack_read(const char* s, uint64_t len, uint32_t wpr, uint32_t* w) {
  if (len > (uint64_t)wpr * 16) {
    memset(w, 0, (size_t)wpr * 4);
    return SBWT_ERR_ARG;
  }
  uint64_t bad = 0, j = 0;
  uint32_t k = 0;
  for (; j + 16 <= len; j += 16, ++k) {
    uint64_t a, b;
    memcpy(&a, s + j, 8);
    memcpy(&b, s + j + 8, 8);
    w[k] = packdetail::pack8(a, &bad) | (packdetail::pack8(b, &bad) << 16);
  }
  if (j < len) {  // last, partial word: pad with 'A' (code 0, as the record's zero fill)
    char t[16];
    memset(t, 'A', 16);
    memcpy(t, s + j, (size_t)(len - j));
    uint64_t a, b;
    memcpy(&a, t, 8);
    memcpy(&b, t + 8, 8);
    w[k++] = packdetail::pack8(a, &bad) | (packdetail::pack8(b, &bad) << 16);
  }
  for (; k < wpr; ++k) w[k] = 0;
  if (bad) return SBWT_ERR_ILLEGAL_CHAR;
  return len < (uint64_t)kKmer ? SBWT_ERR_TOO_SHORT : SBWT_OK;
}

}  // namespace sbwt

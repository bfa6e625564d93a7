// pack.hpp -- ASCII read -> 2-bit record (encoding documented in include/sbwt_amd.h)
#pragma once
#include <stdint.h>
#include <string.h>

#include "../../include/sbwt_amd.h"
#include "device_layout.h"

namespace sbwt {

// returns SBWT_OK, SBWT_ERR_ILLEGAL_CHAR (searcher.hpp:399), SBWT_ERR_TOO_SHORT (searcher.hpp:404)
// or SBWT_ERR_ARG when the read does not fit the record
inline int pack_read(const char* s, uint64_t len, uint32_t wpr, uint32_t* w) {
  memset(w, 0, (size_t)wpr * 4);
  if (len > (uint64_t)wpr * 16) return SBWT_ERR_ARG;
  int bad = SBWT_OK;
  for (uint64_t j = 0; j < len; ++j) {
    uint32_t code;
    switch (s[j]) {
      case 'A': code = 0; break;
      case 'C': code = 1; break;
      case 'G': code = 2; break;
      case 'T': code = 3; break;
      default: code = 0; bad = SBWT_ERR_ILLEGAL_CHAR; break;
    }
    w[j >> 4] |= code << (2 * (j & 15));
  }
  if (bad == SBWT_OK && len < (uint64_t)kKmer) bad = SBWT_ERR_TOO_SHORT;
  return bad;
}

}  // namespace sbwt

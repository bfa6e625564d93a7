// builder.hip -- placeholder until the GPU suffix sorter lands (next commit).
#include "../../include/sbwt_amd.h"
#include "index_impl.hpp"
extern "C" int sbwt_build_index(const char*, const char*, uint64_t, uint64_t, int) {
  sbwt::set_error("sbwt_build_index: not built yet");
  return SBWT_ERR_ARG;
}

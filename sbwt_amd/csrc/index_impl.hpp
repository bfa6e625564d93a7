// index_impl.hpp -- the object behind the opaque `sbwt_index` handle, plus internal helpers.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "device_layout.h"
#include "index_file.hpp"

namespace sbwt {

void set_error(const std::string& msg);
int hip_fail(hipError_t e, const char* what, const char* file, int line);
#define SBWT_HIP(x)                                                     \
  do {                                                                  \
    hipError_t e__ = (x);                                               \
    if (e__ != hipSuccess) return ::sbwt::hip_fail(e__, #x, __FILE__, __LINE__); \
  } while (0)

// HIP refuses a launch whose gridDim.x * blockDim.x does not fit 32 bits (hipErrorInvalidConfiguration), and
// GRCh38 has more than 2^32 suffixes: every per-item kernel is launched on a 2-D grid and linearises it.
inline dim3 grid2d(uint64_t blocks) {
  const uint64_t kMaxX = 1ull << 20;
  if (blocks == 0) blocks = 1;
  if (blocks <= kMaxX) return dim3((uint32_t)blocks, 1, 1);
  return dim3((uint32_t)kMaxX, (uint32_t)((blocks + kMaxX - 1) / kMaxX), 1);
}
inline dim3 grid_items(uint64_t items, uint32_t per_block = 256) { return grid2d((items + per_block - 1) / per_block); }
#if defined(__HIPCC__)
__device__ __forceinline__ uint64_t global_block() { return (uint64_t)blockIdx.y * gridDim.x + blockIdx.x; }
__device__ __forceinline__ uint64_t global_thread() { return global_block() * blockDim.x + threadIdx.x; }
#endif
// a failed launch must not go unnoticed until some later call happens to report it
#define SBWT_LAUNCHED() SBWT_HIP(hipGetLastError())

// Every entry point starts here: the device of its index, and a clean error slate -- hipGetLastError() reports the last
// failure of ANY earlier runtime call on this thread, the caller's own included, and SBWT_LAUNCHED must not mistake one
// for a failed launch of ours.
inline hipError_t enter_device(int device) {
  const hipError_t e = hipSetDevice(device);
  (void)hipGetLastError();
  return e;
}

// device temporary, freed on every way out of its scope (release() frees it early)
template <typename T>
struct DevTemp {
  T* p = nullptr;
  DevTemp() = default;
  DevTemp(const DevTemp&) = delete;
  DevTemp& operator=(const DevTemp&) = delete;
  ~DevTemp() { release(); }
  hipError_t alloc(uint64_t count) { return hipMalloc(reinterpret_cast<void**>(&p), (count ? count : 1) * sizeof(T)); }
  void release() { if (p) (void)hipFree(p); p = nullptr; }
  operator T*() const { return p; }
  T* operator->() const { return p; }
};

// Reference-format tables resident on the device (temporary inputs of the re-layout).
struct RawDeviceTables {
  uint64_t interval = 0, n = 0, first_location = 0, real_size = 0;
  uint64_t c_function4[4] = {0, 0, 0, 0};
  const uint64_t* c_functions = nullptr;  // 4^12 + 1; ownership moves to the index when own_c_functions
  const uint8_t* jbwt = nullptr;          // n/4 + 1 bytes
  const uint8_t* fbwt = nullptr;          // n bytes
  const uint64_t* occ[4] = {nullptr, nullptr, nullptr, nullptr};  // may be null: no consistency check
  uint64_t occ_len = 0;
  const uint64_t* loc = nullptr;  // n_loc pairs
  uint64_t n_loc = 0;
};

}  // namespace sbwt

// launch shapes of the mapping kernels for an index on its device (map_kernels.hip: plan_launches)
struct sbwt_launch_plan {
  uint32_t grid_sm_lds = 0, grid_sm_glb = 0, grid_sort = 0, grid_bfs = 0;
  size_t sm_lds_bytes = 0, sm_ring_bytes = 0, sort_lds_bytes = 0;
  // test knobs of the entry point without a ctx, read when the index is made (map_kernels.hip: read_pass_knobs)
  bool no_bfs = false, no_chain = false, no_rounds = false;
  uint32_t grid_rounds_lds = 0, grid_rounds_glb = 0;
  uint32_t no_segments = 0;
  uint32_t ring_width_max = 0xFFFFFFu, ring_chr_max = 0xFFFFu;
};

struct sbwt_index {
  int device = 0;
  sbwt_launch_plan plan;
  sbwt::DeviceIndex d{};
  std::vector<void*> allocs;  // every hipMalloc the index owns
  uint64_t device_bytes = 0;
  sbwt::ChrTables chr;
  uint64_t n_loc = 0;
  ~sbwt_index();
};

namespace sbwt {
// Builds blocks1 / loc_vals / sa_full / blocks2 / sb_abs from raw device tables.  `c_functions` is
// adopted (not copied).  Runs on the current device, default stream, synchronous.
int build_device_layout(sbwt_index* ix, const RawDeviceTables& raw);
int upload_chr_tables(sbwt_index* ix);
// map_kernels.hip: per-device set-up of the static scratch (sbwt_map_batch_device_static) and the launch plan of
// the index; the last step of every call that creates an index
int static_scratch_setup(int device);
int finish_index(sbwt_index* ix);
template <typename T>
int dev_alloc(sbwt_index* ix, T** p, uint64_t count) {
  void* q = nullptr;
  uint64_t bytes = count * sizeof(T);
  if (bytes == 0) bytes = sizeof(T);
  hipError_t e = hipMalloc(&q, bytes);
  if (e != hipSuccess) return hip_fail(e, "hipMalloc", __FILE__, __LINE__);
  ix->allocs.push_back(q);
  ix->device_bytes += bytes;
  *p = static_cast<T*>(q);
  return 0;
}
}  // namespace sbwt

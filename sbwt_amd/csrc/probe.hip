// probe.hip -- the random-access roofline of the device, measured over the index's own tables.
//
// SURVEY.md 8(d) defines the roofline of this path as "the measured throughput of a random-granule gather
// microbenchmark over a table of the same footprint on the same GPU".  The search kernel is a chain of dependent
// random accesses, each one 128-byte line of HBM whatever part of it is used, so the number
// that bounds it is dependent random line accesses per second.  This kernel measures exactly that on the index the
// caller holds: every lane walks its own chain of `chain_len` dependent 16-byte loads at pseudo-random 128-byte
// lines of the largest table of the index (hash_tab when there is one, else sa_full), the next address depending
// on the loaded bytes.  No upstream counterpart: measurement only (bench.py reports it beside the kernel's rate).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/sbwt_amd.h"
#include "index_impl.hpp"

namespace sbwt {

__device__ __forceinline__ uint64_t probe_mix(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

// kPer independent chains per lane (memory-level parallelism of a lane of the search kernel: its current state
// plus the prefetches of its next two reads)
template <int kPer>
__global__ __launch_bounds__(256) void k_probe(const uint4* __restrict__ table, uint64_t n_lines, uint32_t chain_len,
                                               unsigned long long* __restrict__ sink) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t state[kPer];
#pragma unroll
  for (int k = 0; k < kPer; ++k) state[k] = probe_mix(tid * kPer + k + 12345);
  uint32_t acc = 0;
  for (uint32_t it = 0; it < chain_len; ++it) {
    uint4 v[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) v[k] = table[__umul64hi(state[k], n_lines) * 8];  // first 16 bytes of a 128-byte line
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const uint32_t s = v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
      acc += s;
      state[k] = probe_mix(state[k] ^ s);
    }
  }
  if (acc == 0x9e3779b9u) atomicAdd(sink, 1ull);  // keeps the loads alive; practically never true
}

// The same gather as the mapping kernels meet it: independent (not chained) random lines whose addresses come from a
// STREAMED 16-byte input record, with a 16-byte result written per access -- records in, one table line, records out,
// nothing else.  What k_settle's line rate is to be held against: the random reads share the memory pipeline with the
// two streams.
template <int kVariant>  // 0: as described; 1: no result written; 2: no record read; 3: neither; 4: 8-byte results; 5: non-temporal record loads and result stores (experiments)
__global__ __launch_bounds__(256) void k_probe_stream(const uint4* __restrict__ table, uint64_t n_lines, const uint4* __restrict__ in,
                                                      uint4* __restrict__ out, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint4 rec = make_uint4(0, 0, 0, 0);
  constexpr bool kRead = kVariant == 0 || kVariant == 1 || kVariant >= 4, kWrite = kVariant == 0 || kVariant == 2 || kVariant >= 4;
  auto load_rec = [&](uint64_t k) -> uint4 {
    if (kVariant == 5) {
      const uint32_t* q = reinterpret_cast<const uint32_t*>(in + k);
      return make_uint4(__builtin_nontemporal_load(q), __builtin_nontemporal_load(q + 1), __builtin_nontemporal_load(q + 2),
                        __builtin_nontemporal_load(q + 3));
    }
    return in[k];
  };
  if (kRead && i < n) rec = load_rec(i);
  uint32_t acc = 0;
  for (; i < n; i += stride) {
    uint4 nrec = make_uint4(0, 0, 0, 0);
    if (kRead && i + stride < n) nrec = load_rec(i + stride);  // the next record is on its way while this one's line is awaited
    const uint64_t h = probe_mix((((uint64_t)rec.y << 32) | rec.x) ^ i);
    const uint4 v = table[__umul64hi(h, n_lines) * 8];
    if (kVariant == 4) reinterpret_cast<uint2*>(out)[i] = make_uint2(v.x ^ rec.x, v.y ^ rec.y);
    else if (kVariant == 5) {
      uint32_t* q = reinterpret_cast<uint32_t*>(out + i);
      __builtin_nontemporal_store(v.x ^ rec.x, q); __builtin_nontemporal_store(v.y ^ rec.y, q + 1);
      __builtin_nontemporal_store(v.z ^ rec.z, q + 2); __builtin_nontemporal_store(v.w ^ rec.w, q + 3);
    }
    else if (kWrite) out[i] = make_uint4(v.x ^ rec.x, v.y ^ rec.y, v.z ^ rec.z, v.w ^ rec.w);
    else acc += v.x ^ v.y ^ v.z ^ v.w;
    rec = nrec;
  }
  if (!kWrite && acc == 0x9e3779b9u) out[0] = make_uint4(acc, 0, 0, 0);
}

}  // namespace sbwt

using namespace sbwt;

extern "C" int sbwt_probe_gather_stream(const sbwt_index* ix, uint64_t n_accesses, uint32_t workgroups_per_cu, double* accesses_per_s) {
  if (!ix || !accesses_per_s || n_accesses == 0) {
    set_error("sbwt_probe_gather_stream: bad argument");
    return SBWT_ERR_ARG;
  }
  SBWT_HIP(enter_device(ix->device));
  const void* table = ix->d.hash_tab ? (const void*)ix->d.hash_tab : (const void*)ix->d.sa_full;
  const uint64_t bytes = ix->d.hash_tab ? ix->d.hash_buckets * ((ix->d.hash_text ? 16ull : 8ull) * kHashSlots) : ix->d.n * 8ull;
  const uint64_t n_lines = bytes / 128;
  if (n_lines < 2) {
    set_error("sbwt_probe_gather_stream: the index is too small to probe");
    return SBWT_ERR_ARG;
  }
  hipDeviceProp_t prop;
  SBWT_HIP(hipGetDeviceProperties(&prop, ix->device));
  if (workgroups_per_cu == 0) workgroups_per_cu = 6;
  const uint32_t grid = (uint32_t)prop.multiProcessorCount * workgroups_per_cu;
  DevTemp<uint4> in, out;
  SBWT_HIP(in.alloc(n_accesses));
  SBWT_HIP(out.alloc(n_accesses));
  SBWT_HIP(hipMemset(in, 0x5A, n_accesses * sizeof(uint4)));
  hipEvent_t a = nullptr, b = nullptr;
  SBWT_HIP(hipEventCreate(&a));
  hipError_t e = hipEventCreate(&b);
  float best = 0.f;
  if (e == hipSuccess) {
    k_probe_stream<0><<<grid, 256>>>(static_cast<const uint4*>(table), n_lines, in, out, n_accesses < 4096 ? n_accesses : 4096);  // warm-up
    e = hipDeviceSynchronize();
  }
  const char* ve = getenv("SBWT_PROBE_VARIANT");  // (tools/probe_variants.py: which streams run beside the gather; read before anything is timed)
  const int variant = ve ? atoi(ve) : 0;
  for (int rep = 0; rep < 3 && e == hipSuccess; ++rep) {
    e = hipEventRecord(a, nullptr);
    {
      const uint4* t = static_cast<const uint4*>(table);
      if (variant == 1) k_probe_stream<1><<<grid, 256>>>(t, n_lines, in, out, n_accesses);
      else if (variant == 2) k_probe_stream<2><<<grid, 256>>>(t, n_lines, in, out, n_accesses);
      else if (variant == 3) k_probe_stream<3><<<grid, 256>>>(t, n_lines, in, out, n_accesses);
      else if (variant == 4) k_probe_stream<4><<<grid, 256>>>(t, n_lines, in, out, n_accesses);
      else if (variant == 5) k_probe_stream<5><<<grid, 256>>>(t, n_lines, in, out, n_accesses);
      else k_probe_stream<0><<<grid, 256>>>(t, n_lines, in, out, n_accesses);
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipEventRecord(b, nullptr);
    if (e == hipSuccess) e = hipEventSynchronize(b);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
    if (e == hipSuccess && (best == 0.f || ms < best)) best = ms;
  }
  if (a) (void)hipEventDestroy(a);
  if (b) (void)hipEventDestroy(b);
  if (e != hipSuccess) return hip_fail(e, "sbwt_probe_gather_stream", __FILE__, __LINE__);
  *accesses_per_s = best > 0.f ? (double)n_accesses / (best * 1e-3) : 0.0;
  return SBWT_OK;
}

extern "C" int sbwt_probe_random_access(const sbwt_index* ix, uint32_t chains_per_lane, uint32_t chain_len,
                                        uint32_t workgroups_per_cu, double* accesses_per_s, uint64_t* table_bytes) {
  if (!ix || !accesses_per_s || chain_len == 0 || (chains_per_lane != 1 && chains_per_lane != 2 && chains_per_lane != 4)) {
    set_error("sbwt_probe_random_access: bad argument (chains_per_lane is 1, 2 or 4)");
    return SBWT_ERR_ARG;
  }
  SBWT_HIP(enter_device(ix->device));
  const void* table = ix->d.hash_tab ? (const void*)ix->d.hash_tab : (const void*)ix->d.sa_full;
  const uint64_t bytes = ix->d.hash_tab ? ix->d.hash_buckets * ((ix->d.hash_text ? 16ull : 8ull) * kHashSlots) : ix->d.n * 8ull;
  const uint64_t n_lines = bytes / 128;
  if (n_lines < 2) {
    set_error("sbwt_probe_random_access: the index is too small to probe");
    return SBWT_ERR_ARG;
  }
  if (table_bytes) *table_bytes = n_lines * 128;
  hipDeviceProp_t prop;
  SBWT_HIP(hipGetDeviceProperties(&prop, ix->device));
  if (workgroups_per_cu == 0) workgroups_per_cu = 8;
  const uint32_t grid = (uint32_t)prop.multiProcessorCount * workgroups_per_cu;
  DevTemp<unsigned long long> sink;
  SBWT_HIP(sink.alloc(1));
  SBWT_HIP(hipMemset(sink, 0, 8));
  hipEvent_t a = nullptr, b = nullptr;
  SBWT_HIP(hipEventCreate(&a));
  hipError_t e = hipEventCreate(&b);
  float best = 0.f;
  auto launch = [&](uint32_t len) {
    const uint4* t = static_cast<const uint4*>(table);
    if (chains_per_lane == 1) k_probe<1><<<grid, 256>>>(t, n_lines, len, sink);
    else if (chains_per_lane == 2) k_probe<2><<<grid, 256>>>(t, n_lines, len, sink);
    else k_probe<4><<<grid, 256>>>(t, n_lines, len, sink);
  };
  if (e == hipSuccess) {
    launch(4);  // warm-up: code object, TLB
    e = hipDeviceSynchronize();
  }
  for (int rep = 0; rep < 3 && e == hipSuccess; ++rep) {
    e = hipEventRecord(a, nullptr);
    launch(chain_len);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipEventRecord(b, nullptr);
    if (e == hipSuccess) e = hipEventSynchronize(b);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
    if (e == hipSuccess && (best == 0.f || ms < best)) best = ms;
  }
  if (a) (void)hipEventDestroy(a);
  if (b) (void)hipEventDestroy(b);
  if (e != hipSuccess) return hip_fail(e, "sbwt_probe_random_access", __FILE__, __LINE__);
  const double accesses = (double)grid * 256.0 * chains_per_lane * chain_len;
  *accesses_per_s = best > 0.f ? accesses / (best * 1e-3) : 0.0;
  return SBWT_OK;
}

// Random-granule gather microbenchmark: defines the roofline for the FM-index walk
// (SURVEY.md 8d).  Every lane (or lane group) runs a chain of dependent gathers of one
// G-byte granule at a pseudo-random granule index of a table of `bytes` bytes.
//   usage: gather_bench [table_GiB=6] [chain=64] [blocks_per_cu=8]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

// G = granule bytes, LPG = lanes cooperating on one granule (each loads G/LPG bytes as 16-B pieces)
template <int G, int LPG>
__global__ __launch_bounds__(256) void gather_kernel(const uint4* __restrict__ table, uint64_t ngran, int chain,
                                                     uint64_t* __restrict__ out) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t gid = tid / LPG;
  const int sub = (int)(tid % LPG);
  constexpr int PIECES = G / 16 / LPG;  // 16-B loads per lane
  uint64_t state = mix(gid * 0x51ed27ull + 12345);
  uint64_t acc = 0;
  for (int it = 0; it < chain; ++it) {
    uint64_t idx = __umul64hi(state, ngran);
    const uint4* p = table + idx * (G / 16) + sub * PIECES;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < PIECES; ++k) {
      uint4 v = p[k];
      s += v.x ^ v.y ^ v.z ^ v.w;
    }
#pragma unroll
    for (int m = 1; m < LPG; m <<= 1) s += __shfl_xor(s, m);
    acc += s;
    state = mix(state ^ s);
  }
  if (sub == 0) out[gid] = acc ^ state;
}

// pair shape of the 2-mer step: lane 0 loads PL 16-B pieces of planes at +0, lane 1 loads one dword counter
template <int G, int PL>
__global__ __launch_bounds__(256) void pair_kernel(const uint4* __restrict__ table, uint64_t ngran, int chain,
                                                   uint64_t* __restrict__ out) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t gid = tid >> 1;
  const int sub = (int)(tid & 1);
  uint64_t state = mix(gid * 0x51ed27ull + 12345);
  uint64_t acc = 0;
  for (int it = 0; it < chain; ++it) {
    uint64_t idx = __umul64hi(state, ngran);
    const char* base = (const char*)table + idx * G;
    uint32_t s = 0;
    if (sub == 0) {
      const uint4* p = (const uint4*)base;
#pragma unroll
      for (int k = 0; k < PL; ++k) { uint4 v = p[k]; s += v.x ^ v.y ^ v.z ^ v.w; }
    } else {
      s = *(const uint32_t*)(base + 16 * PL + 4 * (state & 15));
    }
    s += __shfl_xor(s, 1);
    acc += s;
    state = mix(state ^ s);
  }
  if (sub == 0) out[gid] = acc ^ state;
}

template <int G, int PL>
static void run_pair(const uint4* table, uint64_t bytes, int chain, int blocks, uint64_t* out, const char* tag) {
  uint64_t ngran = bytes / G;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  pair_kernel<G, PL><<<blocks, 256>>>(table, ngran, 4, out);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(a));
    pair_kernel<G, PL><<<blocks, 256>>>(table, ngran, chain, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  double gathers = (double)blocks * 256 / 2 * chain;
  printf("%-22s G=%3d PL=%d  blocks=%6d  %.3f ms  %.2f Ggather/s  %.1f GB/s(line)\n", tag, G, PL, blocks, best,
         gathers / best / 1e6, gathers * G / best / 1e6);
  fflush(stdout);
}

template <int G, int LPG>
static void run(const uint4* table, uint64_t bytes, int chain, int blocks, uint64_t* out, const char* tag) {
  uint64_t ngran = bytes / G;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  gather_kernel<G, LPG><<<blocks, 256>>>(table, ngran, 4, out);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(a));
    gather_kernel<G, LPG><<<blocks, 256>>>(table, ngran, chain, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  double gathers = (double)blocks * 256 / LPG * chain;
  printf("%-22s G=%3d LPG=%d blocks=%6d  %.3f ms  %.2f Ggather/s  %.1f GB/s\n", tag, G, LPG, blocks, best,
         gathers / best / 1e6, gathers * G / best / 1e6);
  fflush(stdout);
}

int main(int argc, char** argv) {
  double gib = argc > 1 ? atof(argv[1]) : 6.0;
  int chain = argc > 2 ? atoi(argv[2]) : 64;
  int bpc = argc > 3 ? atoi(argv[3]) : 8;
  uint64_t bytes = (uint64_t)(gib * (1ull << 30)) & ~4095ull;
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs, table %.2f GiB, chain %d, %d blocks/CU of 256 threads\n", prop.name, cus, gib, chain, bpc);
  uint4* table; CK(hipMalloc(&table, bytes));
  CK(hipMemset(table, 0x5a, bytes));
  uint64_t* out; CK(hipMalloc(&out, (uint64_t)cus * 64 * 256 * 8));
  for (int mult = 1; mult <= 2; mult *= 2) {
    int blocks = cus * bpc * mult;  // mult>1 oversubscribes: more workgroups than resident slots
    run_pair<128, 2>(table, bytes, chain, blocks, out, "pair 32B+4B");
    run_pair<64, 1>(table, bytes, chain, blocks, out, "pair 16B+4B");
    run<128, 4>(table, bytes, chain, blocks, out, "4 lanes/granule");
    run<128, 2>(table, bytes, chain, blocks, out, "2 lanes/granule");
    run<64, 1>(table, bytes, chain, blocks, out, "lane/granule");
    run<64, 4>(table, bytes, chain, blocks, out, "4 lanes/granule");
    run<64, 2>(table, bytes, chain, blocks, out, "2 lanes/granule");
    run<32, 1>(table, bytes, chain, blocks, out, "lane/granule");
    run<32, 2>(table, bytes, chain, blocks, out, "2 lanes/granule");
    run<128, 1>(table, bytes, chain, blocks, out, "lane/granule");
    run<128, 8>(table, bytes, chain, blocks, out, "8 lanes/granule");
    run<16, 1>(table, bytes, chain, blocks, out, "lane/granule");
  }
  CK(hipFree(table)); CK(hipFree(out));
  return 0;
}

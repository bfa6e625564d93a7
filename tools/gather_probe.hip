// gather_probe: what bounds the dependent random gather of the FM-index walk?  Variants of the
// tools/gather_bench.hip chain that separate (a) table footprint, (b) address-translation reach (all lanes
// of a wave confined to one window of W bytes per step), (c) granule size beyond one 128-B line.
//   usage: gather_probe [max_table_GiB=48] [chain=64] [blocks_per_cu=8]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

// G = granule bytes, LPG lanes per granule (each loads G/LPG bytes as 16-B pieces).
// window_gran > 0: the granule index of a step is  wave-uniform window base + lane-private offset < window_gran
template <int G, int LPG>
__global__ __launch_bounds__(256) void k_chain(const uint4* __restrict__ table, uint64_t ngran, uint64_t window_gran,
                                               int chain, uint64_t* __restrict__ out) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t gid = tid / LPG;
  const int sub = (int)(tid % LPG);
  constexpr int PIECES = G / 16 / LPG;
  uint64_t state = mix(gid * 0x51ed27ull + 12345);
  uint64_t wstate = mix((tid >> 6) * 0x7f4a7c15ull + 99);
  uint64_t acc = 0;
  for (int it = 0; it < chain; ++it) {
    uint64_t idx;
    if (window_gran) {
      const uint64_t nwin = ngran / window_gran;
      const uint64_t win = __umul64hi(wstate, nwin);
      idx = win * window_gran + __umul64hi(state, window_gran);
      wstate = mix(wstate);
    } else {
      idx = __umul64hi(state, ngran);
    }
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

template <int G, int LPG>
static void run(const uint4* table, uint64_t bytes, uint64_t window_bytes, int chain, int blocks, uint64_t* out) {
  const uint64_t ngran = bytes / G;
  const uint64_t wg = window_bytes ? window_bytes / G : 0;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  k_chain<G, LPG><<<blocks, 256>>>(table, ngran, wg, 4, out);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(a));
    k_chain<G, LPG><<<blocks, 256>>>(table, ngran, wg, chain, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  const double gathers = (double)blocks * 256 / LPG * chain;
  printf("table %8.3f GiB  G=%3d LPG=%2d window %10llu B  %8.3f ms  %6.2f Ggather/s  %7.1f GB/s\n",
         bytes / 1073741824.0, G, LPG, (unsigned long long)window_bytes, best, gathers / best / 1e6,
         gathers * G / best / 1e6);
  fflush(stdout);
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
}

int main(int argc, char** argv) {
  const double max_gib = argc > 1 ? atof(argv[1]) : 48.0;
  const int chain = argc > 2 ? atoi(argv[2]) : 64;
  const int bpc = argc > 3 ? atoi(argv[3]) : 8;
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const uint64_t max_bytes = (uint64_t)(max_gib * (1ull << 30)) & ~((1ull << 21) - 1);
  printf("device %s, %d CUs, chain %d, %d blocks/CU of 256 threads\n", prop.name, cus, chain, bpc);
  uint4* table; CK(hipMalloc(&table, max_bytes));
  CK(hipMemset(table, 0x5a, max_bytes));
  uint64_t* out; CK(hipMalloc(&out, (uint64_t)cus * 64 * 256 * 8));
  const int blocks = cus * bpc;
  printf("-- (a) footprint, 2 lanes x 64 B of one 128-B line\n");
  for (double gib : {1.0 / 64, 1.0 / 16, 0.125, 0.25, 0.5, 1.0, 4.0, 12.0, 48.0, 96.0}) {
    const uint64_t bytes = (uint64_t)(gib * (1ull << 30));
    if (bytes > max_bytes) break;
    run<128, 2>(table, bytes, 0, chain, blocks, out);
  }
  printf("-- (b) translation reach: all 64 lanes of a wave inside one window per step (table = max)\n");
  for (uint64_t w : {4096ull, 65536ull, 1ull << 21, 1ull << 25, 1ull << 30})
    if (w <= max_bytes) run<128, 2>(table, max_bytes, w, chain, blocks, out);
  printf("-- (c) granule size (table = max)\n");
  run<64, 4>(table, max_bytes, 0, chain, blocks, out);
  run<128, 8>(table, max_bytes, 0, chain, blocks, out);
  run<256, 16>(table, max_bytes, 0, chain, blocks, out);
  run<256, 8>(table, max_bytes, 0, chain, blocks, out);
  run<512, 32>(table, max_bytes, 0, chain, blocks, out);
  run<512, 16>(table, max_bytes, 0, chain, blocks, out);
  CK(hipFree(table)); CK(hipFree(out));
  return 0;
}

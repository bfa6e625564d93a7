// gather_probe2: the access pattern of k_search's unique-read path as a pure memory microbenchmark.
// Per item (one lane): 8 B from T0 (kmer_tab) -> one 128-B block of T1 (2 x 16 B + 4 B) -> 8 B from T2 (sa_full)
// -> {2 x 16 B from T3 (textrc), 8 B from T4 (row_tab)} issued together.  Each index depends on the previous load.
//   usage: gather_probe2 [scale=1.0] [items_M=64] [one_alloc=0|1]
// Sizes (GiB, x scale): T0 32, T1 11, T2 44, T3 1.4, T4 44.  one_alloc=1 carves all five from one hipMalloc.
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

struct Tabs {
  const uint64_t* t0; uint64_t n0;   // 8-B entries
  const uint4* t1; uint64_t n1;      // 128-B blocks
  const uint64_t* t2; uint64_t n2;
  const uint4* t3; uint64_t n3;      // 16-B pieces
  const uint64_t* t4; uint64_t n4;
};

template <int STAGES>
__global__ __launch_bounds__(256) void k_chain(Tabs T, uint64_t items, uint64_t* __restrict__ out) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < items; i += stride) {
    uint64_t s = mix(i * 0x51ed27ull + 12345);
    s = mix(s ^ T.t0[__umul64hi(s, T.n0)]);
    if (STAGES >= 2) {
      const uint4* b = T.t1 + __umul64hi(s, T.n1) * 8;
      const uint4 a = b[0], c = b[1];
      const uint32_t cn = reinterpret_cast<const uint32_t*>(b + 2)[s & 15];
      s = mix(s ^ a.x ^ a.w ^ c.y ^ c.z ^ cn);
    }
    if (STAGES >= 3) s = mix(s ^ T.t2[__umul64hi(s, T.n2)]);
    if (STAGES >= 4) {
      const uint4* t = T.t3 + __umul64hi(s, T.n3 - 1);
      const uint64_t r = T.t4[__umul64hi(s ^ 0x1234567, T.n4)];
      const uint4 a = t[0], c = t[1];
      s = mix(s ^ a.x ^ a.w ^ c.y ^ c.z ^ r);
    }
    out[i] = s;
  }
}

template <int STAGES>
static void run(const Tabs& T, uint64_t items, int blocks, uint64_t* out, double accesses_per_item) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  k_chain<STAGES><<<blocks, 256>>>(T, items / 8, out);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(a));
    k_chain<STAGES><<<blocks, 256>>>(T, items, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  printf("stages %d  blocks %5d  %8.3f ms  %6.2f G items/s  %6.2f G line accesses/s\n", STAGES, blocks, best,
         items / best / 1e6, items * accesses_per_item / best / 1e6);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double scale = argc > 1 ? atof(argv[1]) : 1.0;
  const uint64_t items = (uint64_t)((argc > 2 ? atof(argv[2]) : 64.0) * 1e6);
  const int one = argc > 3 ? atoi(argv[3]) : 0;
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const double gib[5] = {32 * scale, 11 * scale, 44 * scale, 1.4 * scale, 44 * scale};
  uint64_t bytes[5], total = 0;
  for (int k = 0; k < 5; ++k) { bytes[k] = ((uint64_t)(gib[k] * (1ull << 30)) + 4095) & ~4095ull; total += bytes[k]; }
  char* base[5];
  if (one) {
    char* p; CK(hipMalloc(&p, total)); CK(hipMemset(p, 0x5a, total));
    for (int k = 0; k < 5; ++k) { base[k] = p; p += bytes[k]; }
  } else {
    for (int k = 0; k < 5; ++k) { CK(hipMalloc(&base[k], bytes[k])); CK(hipMemset(base[k], 0x5a, bytes[k])); }
  }
  Tabs T{(const uint64_t*)base[0], bytes[0] / 8, (const uint4*)base[1], bytes[1] / 128, (const uint64_t*)base[2], bytes[2] / 8,
         (const uint4*)base[3], bytes[3] / 16, (const uint64_t*)base[4], bytes[4] / 8};
  uint64_t* out; CK(hipMalloc(&out, items * 8));
  printf("device %s, %d CUs, %.1f GiB in %s, %llu M items\n", prop.name, prop.multiProcessorCount, total / 1073741824.0,
         one ? "one allocation" : "five allocations", (unsigned long long)(items / 1000000));
  for (int bpc : {4, 8}) {
    const int blocks = prop.multiProcessorCount * bpc;
    run<1>(T, items, blocks, out, 1);
    run<2>(T, items, blocks, out, 2);
    run<3>(T, items, blocks, out, 3);
    run<4>(T, items, blocks, out, 5);
  }
  return 0;
}

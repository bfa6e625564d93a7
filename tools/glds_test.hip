// Layout check of the LDS-DMA load (global_load_lds_dwordx4) with per-lane source addresses:
// lane l of a wave loads the 16 bytes at src[perm[l]] ; expected LDS image: base + l * 16.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

__global__ __launch_bounds__(256) void k_test(const uint4* __restrict__ src, const uint32_t* __restrict__ perm, uint4* __restrict__ out,
                                              int masked) {
  __shared__ __attribute__((aligned(16))) uint4 s_img[4][2][64];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = 0; k < 2; ++k) s_img[wave][k][lane] = make_uint4(0xdeadbeefu, 0, 0, 0);
  __syncthreads();
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  for (int k = 0; k < 2; ++k) {
    const uint4* p = src + perm[(g * 2 + k) % (gridDim.x * 512)];
    if (!masked || (lane & 8) == 0)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                       (__attribute__((address_space(3))) void*)&s_img[wave][k][0], 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  for (int k = 0; k < 2; ++k) out[(size_t)g * 2 + k] = s_img[wave][k][lane];
}

int main() {
  const int blocks = 64, n = blocks * 512;
  std::vector<uint4> h(n);
  std::vector<uint32_t> perm(n);
  for (int i = 0; i < n; ++i) { h[i] = make_uint4(i, i * 3 + 1, ~i, 0x1234); perm[i] = (uint32_t)((i * 7919u + 13u) % n); }
  uint4 *d_src, *d_out; uint32_t* d_perm;
  hipMalloc(&d_src, n * 16); hipMalloc(&d_out, n * 16); hipMalloc(&d_perm, n * 4);
  hipMemcpy(d_src, h.data(), n * 16, hipMemcpyHostToDevice);
  hipMemcpy(d_perm, perm.data(), n * 4, hipMemcpyHostToDevice);
  for (int masked = 0; masked < 2; ++masked) {
    k_test<<<blocks, 256>>>(d_src, d_perm, d_out, masked);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    std::vector<uint4> o(n);
    hipMemcpy(o.data(), d_out, n * 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int g = 0; g < blocks * 256; ++g)
      for (int k = 0; k < 2; ++k) {
        const uint4 got = o[(size_t)g * 2 + k];
        const bool skipped = masked && ((g & 63) & 8);
        const uint4 want = skipped ? make_uint4(0xdeadbeefu, 0, 0, 0) : h[perm[(g * 2 + k) % n]];
        if (got.x != want.x || got.y != want.y || got.z != want.z || got.w != want.w) {
          if (bad < 5) printf("masked %d thread %d k %d: got %08x want %08x\n", masked, g, k, got.x, want.x);
          ++bad;
        }
      }
    printf("masked=%d: %d mismatches of %d\n", masked, bad, n);
  }
  return 0;
}

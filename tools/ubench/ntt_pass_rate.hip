// Micro-benchmark: the register part of one radix-4 pass of the transform (ntt_core.h: two butterfly stages on four elements, lazy
// sums, one normalisation per element) in a loop, operands in registers -- the ceiling of k_ntt_tile without LDS, barriers and HBM.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../kogarashi_amd/csrc -I../../include ntt_pass_rate.hip -o ntt_pass_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include "ntt_tile.h"
using namespace kg;
using Fr = Fp<FrParams>;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int KIND> struct TwSrc {
  const uint32_t* tab; uint32_t j;
  __device__ Fr mul(const Fr& x, int t, int k0) const {
    if constexpr (KIND == 0) return mulc(x, NttIO<Fr>::twc(tab, (size_t)((j + 17u * (uint32_t)(t + k0)) & 1023u)));      // per-lane table entries (L1 / L2)
    else return mulc(x, NttIO<Fr>::twc(tab, (size_t)(t + k0)));                                                         // uniform entries
  }
};
template <int KIND, bool NORM>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) k_pass(const uint32_t* tab, uint32_t* out, int iters) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  Fr x[4];
  for (int k = 0; k < 4; ++k) { x[k] = Fr::one(); x[k].l[0] += tid + k; }
  for (int it = 0; it < iters; ++it) {
    TwSrc<KIND> tw{tab, tid * 7u + (uint32_t)it * 13u};
    dit_network<2>(x, false, tw);
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = NORM ? vred(norm(x[k])) : norm(x[k]);     // (a pass stores normalised elements; the value reduction stands in for the stores' bound)
  }
  uint32_t s = 0;
  for (int k = 0; k < 4; ++k) for (int i = 0; i < 9; ++i) s += x[k].l[i];
  out[tid] = s;
}
template <class K> float time_it(K launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  uint32_t *d, *tab;
  CHECK(hipMalloc(&d, 1 << 24)); CHECK(hipMalloc(&tab, 1024 * 72));
  CHECK(hipMemset(tab, 0x11, 1024 * 72));
  const int iters = 400;
  for (int wps : {1, 2, 4}) {
    const int blocks = 256 * wps;      // 256 threads = one wave per SIMD per block and CU
    float ms;
    ms = time_it([&] { k_pass<0, false><<<blocks, 256>>>(tab, d, iters); });
    printf("pass, table twiddles           w/SIMD=%d %8.3f ms  %8.2f G element-passes/s\n", wps, ms, (double)blocks * 256 * 4 * iters / ms * 1e-6);
    ms = time_it([&] { k_pass<1, false><<<blocks, 256>>>(tab, d, iters); });
    printf("pass, uniform twiddles         w/SIMD=%d %8.3f ms  %8.2f G element-passes/s\n", wps, ms, (double)blocks * 256 * 4 * iters / ms * 1e-6);
    ms = time_it([&] { k_pass<0, true><<<blocks, 256>>>(tab, d, iters); });
    printf("pass + value reduction, table  w/SIMD=%d %8.3f ms  %8.2f G element-passes/s\n", wps, ms, (double)blocks * 256 * 4 * iters / ms * 1e-6);
  }
  return 0;
}

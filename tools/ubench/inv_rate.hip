// inv_rate.hip -- cost of one modular inversion on gfx950, in Montgomery-product equivalents: the Fermat ladder a^(p-2)
// (fp29.h: inv) against the constant-time binary GCD (fp_inv.h: inv_bingcd), both on 9 x 29-bit limbs, next to a chain
// of plain products at the same occupancy.  Every lane inverts its own value (no cross-lane work).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o inv_rate inv_rate.hip && ./inv_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../kogarashi_amd/csrc/fp29.h"
#include "../../kogarashi_amd/csrc/fp_inv.h"
using namespace kg;

template <int MODE>
__global__ void __launch_bounds__(256) k_rate(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int reps) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  Fq x;
#pragma unroll
  for (int k = 0; k < 9; ++k) x.l[k] = in[i * 9 + k];
  Fq acc = x;
  for (int r = 0; r < reps; ++r) {
    if (MODE == 0) acc = mul(acc, x);                         // one product
    else if (MODE == 1) acc = norm(add(inv(acc), x));         // Fermat
    else acc = norm(add(inv_bingcd(acc), x));                 // binary GCD
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) out[i * 9 + k] = acc.l[k];
}

template <int MODE>
static double run(const uint32_t* d_in, uint32_t* d_out, size_t n, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_rate<MODE>, dim3((unsigned)(n / 256)), dim3(256), 0, 0, d_in, d_out, reps);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_rate<MODE>, dim3((unsigned)(n / 256)), dim3(256), 0, 0, d_in, d_out, reps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return (double)n * reps / (ms * 1e-3);
}

int main() {
  const size_t n = (size_t)256 * 4 * 256 * 4;                // four rounds of 4 waves per SIMD
  std::vector<uint32_t> h(n * 9);
  unsigned long long s = 88172645463325252ull;
  for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)s & 0x0fffffffu; }
  uint32_t *d_in, *d_out;
  hipMalloc(&d_in, n * 36); hipMalloc(&d_out, n * 36);
  hipMemcpy(d_in, h.data(), n * 36, hipMemcpyHostToDevice);
  // agreement of the two inversions on the device
  hipLaunchKernelGGL(k_rate<1>, dim3((unsigned)(n / 256)), dim3(256), 0, 0, d_in, d_out, 1);
  std::vector<uint32_t> a(n * 9), b(n * 9);
  hipMemcpy(a.data(), d_out, n * 36, hipMemcpyDeviceToHost);
  hipLaunchKernelGGL(k_rate<2>, dim3((unsigned)(n / 256)), dim3(256), 0, 0, d_in, d_out, 1);
  hipMemcpy(b.data(), d_out, n * 36, hipMemcpyDeviceToHost);
  size_t diff = 0;
  for (size_t i = 0; i < n; ++i) {
    Fq x, y;
    for (int k = 0; k < 9; ++k) { x.l[k] = a[i * 9 + k]; y.l[k] = b[i * 9 + k]; }
    diff += !same_limbs(reduce(x), reduce(y));
  }
  const double mul_rate = run<0>(d_in, d_out, n, 2000), fermat = run<1>(d_in, d_out, n, 8), gcd = run<2>(d_in, d_out, n, 40);
  printf("products      : %8.1f G/s\n", mul_rate / 1e9);
  printf("inv, Fermat   : %8.2f G/s = %6.1f product-equivalents\n", fermat / 1e9, mul_rate / fermat);
  printf("inv, bin. GCD : %8.2f G/s = %6.1f product-equivalents   (%zu of %zu lanes differ from Fermat)\n", gcd / 1e9, mul_rate / gcd, diff, n);
  return diff != 0;
}

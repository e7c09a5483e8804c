// Experiment: compiler-scheduled mul (fp29.h) vs a single-chain inline-asm mad sequence.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "fp29.h"
using namespace kg;

__device__ __forceinline__ void mad(uint64_t& acc, uint32_t a, uint32_t b) {
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
}
template <class P>
__device__ __forceinline__ Fp<P> mul_asm(const Fp<P>& a, const Fp<P>& b) {
  uint32_t m[9];
  Fp<P> r;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) mad(acc, a.l[i], b.l[k - i]);
#pragma unroll
    for (int i = 0; i < k; ++i) mad(acc, m[i], P::P[k - i]);
    m[k] = ((uint32_t)acc * P::INV) & M29;
    mad(acc, m[k], P::P[0]);
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int i = k - 8; i <= 8; ++i) mad(acc, a.l[i], b.l[k - i]);
#pragma unroll
    for (int i = k - 8; i <= 8; ++i) mad(acc, m[i], P::P[k - i]);
    r.l[k - 9] = (uint32_t)acc & M29;
    acc >>= 29;
  }
  r.l[8] = (uint32_t)acc;
  return r;
}
// two interleaved chains per column (a*b terms / m*p terms), merged once per column
template <class P>
__device__ __forceinline__ Fp<P> mul_asm2(const Fp<P>& a, const Fp<P>& b) {
  uint32_t m[9];
  Fp<P> r;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    uint64_t acc2 = 0;
#pragma unroll
    for (int i = 0; i <= k; ++i) mad(acc, a.l[i], b.l[k - i]);
#pragma unroll
    for (int i = 0; i < k; ++i) mad(acc2, m[i], P::P[k - i]);
    acc += acc2;
    m[k] = ((uint32_t)acc * P::INV) & M29;
    mad(acc, m[k], P::P[0]);
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
    uint64_t acc2 = 0;
#pragma unroll
    for (int i = k - 8; i <= 8; ++i) mad(acc, a.l[i], b.l[k - i]);
#pragma unroll
    for (int i = k - 8; i <= 8; ++i) mad(acc2, m[i], P::P[k - i]);
    acc += acc2;
    r.l[k - 9] = (uint32_t)acc & M29;
    acc >>= 29;
  }
  r.l[8] = (uint32_t)acc;
  return r;
}

template <int KIND>
__global__ void __launch_bounds__(256) k_field(uint32_t* out, int iters) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  Fq a = Fq::one(), b = Fq::from_const(FqParams::G1_B3);
  a.l[0] += tid & 0xffff; b.l[1] ^= tid & 0xfff;
  for (int i = 0; i < iters; ++i) {
    if constexpr (KIND == 0) a = mul(a, b);
    else if constexpr (KIND == 1) a = mul_asm(a, b);
    else a = mul_asm2(a, b);
  }
  uint32_t s = 0;
  for (int k = 0; k < 9; ++k) s += a.l[k];
  out[tid] = s;
}
template <class K> float time_it(K launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  uint32_t* d; hipMalloc(&d, 1 << 24);
  // correctness: all three agree
  uint32_t h[3][256];
  k_field<0><<<1, 256>>>(d, 50); hipMemcpy(h[0], d, 1024, hipMemcpyDeviceToHost);
  k_field<1><<<1, 256>>>(d, 50); hipMemcpy(h[1], d, 1024, hipMemcpyDeviceToHost);
  k_field<2><<<1, 256>>>(d, 50); hipMemcpy(h[2], d, 1024, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 256; ++i) bad += (h[0][i] != h[1][i]) + (h[0][i] != h[2][i]);
  printf("mismatches: %d\n", bad);
  const char* nm[] = {"compiler", "asm 1 chain", "asm 2 chains"};
  for (int wps : {1, 2, 4}) {
    int blocks = 256 * wps, iters = 2000;
    float ms;
    ms = time_it([&] { k_field<0><<<blocks, 256>>>(d, iters); }); printf("%-14s w/SIMD=%d %8.3f ms %8.2f Gmul/s\n", nm[0], wps, ms, (double)blocks * 256 * iters / ms * 1e-6);
    ms = time_it([&] { k_field<1><<<blocks, 256>>>(d, iters); }); printf("%-14s w/SIMD=%d %8.3f ms %8.2f Gmul/s\n", nm[1], wps, ms, (double)blocks * 256 * iters / ms * 1e-6);
    ms = time_it([&] { k_field<2><<<blocks, 256>>>(d, iters); }); printf("%-14s w/SIMD=%d %8.3f ms %8.2f Gmul/s\n", nm[2], wps, ms, (double)blocks * 256 * iters / ms * 1e-6);
  }
  return 0;
}

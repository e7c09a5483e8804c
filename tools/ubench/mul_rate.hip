// Micro-benchmark: throughput of the field multiplier / squarer / point addition of fp29.h + curve.h on gfx950,
// and issue rates of the auxiliary ops (64-bit shifts etc.).  Build: hipcc --offload-arch=gfx950 -O3 -I../../kogarashi_amd/csrc
#include <hip/hip_runtime.h>
#include <cstdio>
#include "curve.h"
using namespace kg;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int KIND>
__global__ void __launch_bounds__(64) k_field(uint32_t* out, int iters) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  Fq a = Fq::one(), b = Fq::from_const(FqParams::G1_B3);
  a.l[0] += tid & 0xffff; b.l[1] ^= tid & 0xfff;
  if constexpr (KIND == 0) { for (int i = 0; i < iters; ++i) a = mul(a, b); }
  else if constexpr (KIND == 1) { for (int i = 0; i < iters; ++i) a = sqr(a); }
  else if constexpr (KIND == 2) { for (int i = 0; i < iters; ++i) { a = mul(a, b); b = mul(b, a); } }   // 2 per iter, some ILP
  else if constexpr (KIND == 3) {
    XYZZ<Fq> p = from_affine(Affine<Fq>{a, b});
    Affine<Fq> q{b, a};
    // the bucket kernel's routine (sign folded, X not value-reduced), operands in registers
    for (int i = 0; i < iters; ++i) { p = add_mixed_signed(p, q, (i & 1) != 0); q.x.l[0] ^= 1; }
    a = norm(add(p.x, p.zzz));
  } else if constexpr (KIND == 4) { for (int i = 0; i < iters; ++i) a = vred(norm(sub<4, 1>(a, b))); }
  uint32_t s = 0;
  for (int k = 0; k < 9; ++k) s += a.l[k];
  out[tid] = s;
}

template <int KIND>
__global__ void __launch_bounds__(256) k_op(uint32_t* out, int iters) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t acc[8]; uint32_t x[8];
  for (int u = 0; u < 8; ++u) { acc[u] = tid * 77u + u; x[u] = tid + u; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if constexpr (KIND == 0) asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(acc[u]));
      else if constexpr (KIND == 1) asm volatile("v_ashrrev_i64 %0, 29, %0" : "+v"(acc[u]));
      else if constexpr (KIND == 2) asm volatile("v_and_b32 %0, 0x1fffffff, %0" : "+v"(x[u]));
      else if constexpr (KIND == 3) asm volatile("v_alignbit_b32 %0, %0, %1, 29" : "+v"(x[u]) : "v"(x[(u + 1) & 7]));
      else if constexpr (KIND == 4) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[u]) : "v"(x[(u + 1) & 7]), "v"(x[(u + 2) & 7]));
      else if constexpr (KIND == 5) asm volatile("v_mov_b32 %0, %1" : "+v"(x[u]) : "v"(x[(u + 1) & 7]));
      else if constexpr (KIND == 6) { uint32_t lo = (uint32_t)acc[u]; asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[u]) : "v"(lo), "v"(x[u]) : "vcc"); }
      else if constexpr (KIND == 7) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[u]) : "v"(acc[(u + 1) & 7]));
      else if constexpr (KIND == 8) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[u]) : "v"(x[(u + 1) & 7]));
      else if constexpr (KIND == 9) asm volatile("v_lshrrev_b32 %0, 29, %0" : "+v"(x[u]));
      else if constexpr (KIND == 10) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x[u]) : "v"(x[(u + 1) & 7]), "v"(x[(u + 2) & 7]));
    }
  }
  uint64_t s = 0;
  for (int u = 0; u < 8; ++u) s += acc[u] + x[u];
  out[tid] = (uint32_t)s;
}

template <class K> float time_it(K launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  uint32_t* d; CHECK(hipMalloc(&d, 1 << 24));
  const char* fn[] = {"mul (dependent chain)", "sqr (dependent chain)", "mul x2 interleaved", "add_mixed", "sub+norm+vred"};
  for (int wps : {1, 2, 3, 4, 5, 6, 8}) {
    int blocks = 256 * wps, iters = 2000;
    float ms;
    ms = time_it([&] { k_field<0><<<blocks * 4, 64>>>(d, iters); }); printf("%-24s w/SIMD=%d %8.3f ms  %8.2f Gop/s\n", fn[0], wps, ms, (double)blocks * 256 * iters / ms * 1e-6);
    ms = time_it([&] { k_field<1><<<blocks * 4, 64>>>(d, iters); }); printf("%-24s w/SIMD=%d %8.3f ms  %8.2f Gop/s\n", fn[1], wps, ms, (double)blocks * 256 * iters / ms * 1e-6);
    ms = time_it([&] { k_field<2><<<blocks * 4, 64>>>(d, iters); }); printf("%-24s w/SIMD=%d %8.3f ms  %8.2f Gop/s\n", fn[2], wps, ms, (double)blocks * 256 * iters * 2 / ms * 1e-6);
    ms = time_it([&] { k_field<3><<<blocks * 4, 64>>>(d, 200); }); printf("%-24s w/SIMD=%d %8.3f ms  %8.2f Gop/s\n", fn[3], wps, ms, (double)blocks * 256 * 200 / ms * 1e-6);
    ms = time_it([&] { k_field<4><<<blocks * 4, 64>>>(d, iters); }); printf("%-24s w/SIMD=%d %8.3f ms  %8.2f Gop/s\n", fn[4], wps, ms, (double)blocks * 256 * iters / ms * 1e-6);
  }
  const char* on[] = {"v_lshrrev_b64", "v_ashrrev_i64", "v_and_b32", "v_alignbit_b32", "v_add3_u32", "v_mov_b32", "v_mad_i64_i32", "v_lshl_add_u64", "v_add_u32", "v_lshrrev_b32", "v_and_or_b32"};
  int blocks = 1024, iters = 4096;
#define RUN(K) { float ms = time_it([&] { k_op<K><<<blocks, 256>>>(d, iters); }); printf("%-16s %8.3f ms %9.2f Gop/s\n", on[K], ms, (double)blocks * 256 * iters * 8 / ms * 1e-6); }
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10)
  return 0;
}

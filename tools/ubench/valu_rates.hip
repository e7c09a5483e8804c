// Micro-benchmark: issue rates of the integer / fp64 VALU ops a 256-bit Montgomery
// multiplier can be built from, on gfx950.  Prints ops/s per instruction kind.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 16;   // independent chains per thread

template <int KIND>
__global__ void __launch_bounds__(256) k_rate(uint32_t* out, uint32_t seed) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t acc[UNROLL];
  uint32_t a = tid * 2654435761u + seed, b = (tid ^ seed) * 40503u + 12345u;
  double da = (double)(a & 0xfffff) + 0.5, db = (double)(b & 0xfffff) + 0.25;
  double dacc[UNROLL];
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) { acc[u] = a + u; dacc[u] = (double)u; }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if constexpr (KIND == 0) {          // v_mad_u64_u32
        uint32_t lo = (uint32_t)acc[u];
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[u]) : "v"(lo), "v"(b) : "vcc");
      } else if constexpr (KIND == 1) {   // v_mul_lo_u32
        uint32_t x = (uint32_t)acc[u];
        asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(b));
        acc[u] = x;
      } else if constexpr (KIND == 2) {   // v_mul_hi_u32
        uint32_t x = (uint32_t)acc[u];
        asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(b));
        acc[u] = x | 0x80000001u;
      } else if constexpr (KIND == 3) {   // v_mad_u32_u24
        uint32_t x = (uint32_t)acc[u];
        asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(a));
        acc[u] = x;
      } else if constexpr (KIND == 4) {   // v_fma_f64
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(dacc[u]) : "v"(da), "v"(db));
      } else if constexpr (KIND == 5) {   // v_add_co_u32 + v_addc_co_u32 (64-bit add)
        uint32_t lo = (uint32_t)acc[u], hi = (uint32_t)(acc[u] >> 32);
        asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc");
        acc[u] = ((uint64_t)hi << 32) | lo;
      } else if constexpr (KIND == 6) {   // v_mul_u32_u24
        uint32_t x = (uint32_t)acc[u];
        asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(b));
        acc[u] = x | 1u;
      } else if constexpr (KIND == 7) {   // v_mul_hi_u32_u24
        uint32_t x = (uint32_t)acc[u];
        asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x) : "v"(b));
        acc[u] = x | 0x800001u;
      } else if constexpr (KIND == 8) {   // v_dot4_u32_u8
        uint32_t x = (uint32_t)acc[u];
        asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
        acc[u] = x;
      } else if constexpr (KIND == 9) {   // v_add_u32 (baseline full-rate)
        uint32_t x = (uint32_t)acc[u];
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));
        acc[u] = x;
      } else if constexpr (KIND == 10) {  // v_mad_i32_i24
        uint32_t x = (uint32_t)acc[u];
        asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(a));
        acc[u] = x;
      } else if constexpr (KIND == 11) {  // v_lshl_add_u64 (gfx940+)
        asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[u]) : "v"(acc[(u + 1) % UNROLL]));
      } else if constexpr (KIND == 12) {  // v_dot2_u32_u16
        uint32_t x = (uint32_t)acc[u];
        asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
        acc[u] = x;
      } else if constexpr (KIND == 13) {  // v_pk_mad_u16
        uint32_t x = (uint32_t)acc[u];
        asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(a));
        acc[u] = x;
      } else if constexpr (KIND == 14) {  // v_mul_f64
        asm volatile("v_mul_f64 %0, %0, %1" : "+v"(dacc[u]) : "v"(da));
      } else if constexpr (KIND == 15) {  // v_add_f64
        asm volatile("v_add_f64 %0, %0, %1" : "+v"(dacc[u]) : "v"(da));
      }
    }
  }
  uint64_t s = 0; double ds = 0;
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) { s += acc[u]; ds += dacc[u]; }
  if (s == 0x1234567 && ds == 1.5) out[tid] = 1;   // keep live
  if (tid == 0) out[0] = (uint32_t)s + (uint32_t)ds;
}

template <int KIND>
int run(const char* name, uint32_t* d_out, int waves_per_simd) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  int blocks = 256 * waves_per_simd;   // 256 threads = 4 waves = 1 wave/SIMD per block per CU
  k_rate<KIND><<<blocks, 256>>>(d_out, 1); CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  k_rate<KIND><<<blocks, 256>>>(d_out, 2);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double ops = (double)blocks * 256 * ITERS * UNROLL;
  double per_simd_cycle = ops / (ms * 1e-3) / (256.0 * 4) / 2.4e9;   // lanes per clk per SIMD at nominal 2.4 GHz
  printf("%-22s waves/SIMD=%d  %8.3f ms  %9.2f Gop/s  %6.2f lanes/clk/SIMD(@2.4GHz)\n", name, waves_per_simd, ms, ops / ms * 1e-6, per_simd_cycle);
  return 0;
}

int main() {
  uint32_t* d_out; CHECK(hipMalloc(&d_out, 256 * 256 * 8 * 4 * 4));
  for (int w : {1, 2, 4}) {
    run<9>("v_add_u32", d_out, w);
    run<0>("v_mad_u64_u32", d_out, w);
    run<1>("v_mul_lo_u32", d_out, w);
    run<2>("v_mul_hi_u32", d_out, w);
    run<3>("v_mad_u32_u24", d_out, w);
    run<10>("v_mad_i32_i24", d_out, w);
    run<6>("v_mul_u32_u24", d_out, w);
    run<7>("v_mul_hi_u32_u24", d_out, w);
    run<4>("v_fma_f64", d_out, w);
    run<14>("v_mul_f64", d_out, w);
    run<15>("v_add_f64", d_out, w);
    run<5>("v_add_co+addc (64b)", d_out, w);
    run<11>("v_lshl_add_u64", d_out, w);
    run<8>("v_dot4_u32_u8", d_out, w);
    run<12>("v_dot2_u32_u16", d_out, w);
    run<13>("v_pk_mad_u16", d_out, w);
  }
  return 0;
}

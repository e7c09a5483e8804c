// Micro-benchmark: can the matrix pipe take the CONSTANT multiplications of the field arithmetic?
//
// The m * p half of a Montgomery product (fp29.h mul) and the q * p product of a Shoup constant multiplication (mulc) multiply a
// per-lane value by the SAME constant p: 64 lanes x 38 seven-bit digits times the Toeplitz matrix of p's digits is an integer GEMM,
// [96 x 48] x [48 x 64] per wave, which v_mfma_i32_32x32x16_i8 issues beside the VALU.  This file measures the whole thing a kernel
// would have to do -- digit split of the 9 x 29-bit limbs, the lane exchange that builds the B operand, 14 MFMAs per wave, the lane
// exchange that brings a product's 75 column sums back into its lane, recombination into 29-bit limbs -- against the same half
// product on the VALU (81 v_mad_u64_u32 + carries), checks both against each other and against 128-bit host arithmetic, and prints
// products per second.  Result (profiles/r04_mfma_const_mul.txt): the marshalling costs several times the 81 multiply-accumulates it
// saves; DESIGN.md section 2 records the decision.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../kogarashi_amd/csrc -o mfma_const_mul mfma_const_mul.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include "fp29.h"
using namespace kg;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef int v16i __attribute__((ext_vector_type(16)));
constexpr int ND = 38;          // seven-bit digits of a 261-bit value (9 x 29 bits)
constexpr int NPD = 37;         // seven-bit digits of p (254 bits)
constexpr int NROW = 75;        // digits of the product

__constant__ uint64_t c_atab[3][3][64];       // A operand (Toeplitz(p)^T, 96 x 48) per (row block, k block) and lane: 8 bytes

// ---- VALU reference: the 18 limbs (29 bits each) of m * p ---------------------------------------------------------------------
template <class P>
__device__ __forceinline__ void mp_valu(const uint32_t (&m)[9], uint32_t (&out)[18]) {
  uint64_t acc = 0;
#pragma unroll
  for (int col = 0; col < 17; ++col) {
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int j = col - i;
      if (j >= 0 && j < 9) acc += (uint64_t)m[i] * P::P[j];
    }
    out[col] = (uint32_t)acc & M29;
    acc >>= 29;
  }
  out[17] = (uint32_t)acc;
}

// ---- MFMA path -------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void swap32(uint32_t& lo_keeps_low_lanes, uint32_t& hi_keeps_high_lanes) {
  // v_permlane32_swap: lanes 32..63 of the first operand <-> lanes 0..31 of the second
  auto r = __builtin_amdgcn_permlane32_swap(lo_keeps_low_lanes, hi_keeps_high_lanes, false, false);
  lo_keeps_low_lanes = r[0];
  hi_keeps_high_lanes = r[1];
}

// seven-bit digits of the 261-bit value, eight per 64-bit word: dig[w] holds digits 8w .. 8w+7 (one byte each), w < 6
__device__ __forceinline__ void split_digits(const uint32_t (&m)[9], uint64_t (&dig)[6]) {
  uint32_t w[10];                                   // the value as a contiguous bit string, 32-bit words
#pragma unroll
  for (int k = 0; k < 10; ++k) w[k] = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int bit = 29 * i, k = bit >> 5, sh = bit & 31;
    w[k] |= m[i] << sh;
    if (sh > 3) w[k + 1] |= m[i] >> (32 - sh);
  }
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const int d = 8 * q + b;
      uint32_t v = 0;
      if (d < ND) {
        const int bit = 7 * d, k = bit >> 5, sh = bit & 31;
        v = (sh <= 25 ? (w[k] >> sh) : ((w[k] >> sh) | (w[k + 1] << (32 - sh)))) & 127u;
      }
      if (b < 4) lo |= v << (8 * b); else hi |= v << (8 * (b - 4));
    }
    dig[q] = ((uint64_t)hi << 32) | lo;
  }
}

__device__ __forceinline__ void mp_mfma(const uint32_t (&m)[9], const uint64_t (&A)[3][3], uint32_t (&out)[18]) {
  uint64_t dig[6];
  split_digits(m, dig);
  // B operand: column n = lane % 32 is a product, k = 8 * (lane / 32) + byte.  lo block: products of lanes 0..31, hi block: 32..63.
  uint64_t blo[3], bhi[3];
#pragma unroll
  for (int kb = 0; kb < 3; ++kb) {
    uint32_t d0l = (uint32_t)dig[2 * kb], d0h = (uint32_t)(dig[2 * kb] >> 32), d1l = (uint32_t)dig[2 * kb + 1], d1h = (uint32_t)(dig[2 * kb + 1] >> 32);
    swap32(d0l, d1l);                               // d0: lanes < 32 own digits 16kb..+7 | lanes >= 32 the low lanes' digits 16kb+8..+15  = B of the lo block
    swap32(d0h, d1h);                               // d1: lanes < 32 the high lanes' digits 16kb..+7 | lanes >= 32 own 16kb+8..+15       = B of the hi block
    blo[kb] = ((uint64_t)d0h << 32) | d0l;
    bhi[kb] = ((uint64_t)d1h << 32) | d1l;
  }
  v16i clo[3], chi[3];
#pragma unroll
  for (int mb = 0; mb < 3; ++mb) {
    clo[mb] = (v16i)(0);
    chi[mb] = (v16i)(0);
#pragma unroll
    for (int kb = 0; kb < 3; ++kb) {
      if ((mb == 0 && kb == 2) || (mb == 2 && kb == 0)) continue;            // all-zero blocks of the Toeplitz matrix
      clo[mb] = __builtin_amdgcn_mfma_i32_32x32x16_i8((long)A[mb][kb], (long)blo[kb], clo[mb], 0, 0, 0);
      chi[mb] = __builtin_amdgcn_mfma_i32_32x32x16_i8((long)A[mb][kb], (long)bhi[kb], chi[mb], 0, 0, 0);
    }
  }
  // bring every product's column sums into its own lane: after the swap `x` holds rows 8q + r, `y` rows 8q + 4 + r (q = v / 4, r = v % 4)
  uint64_t L[18];
#pragma unroll
  for (int j = 0; j < 18; ++j) L[j] = 0;
#pragma unroll
  for (int mb = 0; mb < 3; ++mb) {
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      uint32_t x = (uint32_t)clo[mb][v], y = (uint32_t)chi[mb][v];
      swap32(x, y);
      const int ra = 32 * mb + 8 * (v / 4) + (v % 4), rb = ra + 4;
      if (ra < NROW) { const int bit = 7 * ra; L[bit / 29] += (uint64_t)x << (bit % 29); }
      if (rb < NROW) { const int bit = 7 * rb; L[bit / 29] += (uint64_t)y << (bit % 29); }
    }
  }
  uint64_t carry = 0;
#pragma unroll
  for (int j = 0; j < 18; ++j) {
    const uint64_t t = L[j] + carry;
    out[j] = (uint32_t)t & M29;
    carry = t >> 29;
  }
}

template <int KIND>
__global__ void __launch_bounds__(64) k_rate(uint32_t* out, int iters) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t m[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) m[i] = (0x1234567u * (tid + 1) + 0x9e3779bu * (i + 1)) & M29;
  uint64_t A[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) A[a][b] = c_atab[a][b][threadIdx.x];
  uint32_t r[18];
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == 0) mp_valu<FqParams>(m, r); else mp_mfma(m, A, r);
#pragma unroll
    for (int i = 0; i < 9; ++i) m[i] = (r[i] ^ r[9 + i]) & M29;                 // the next operand depends on the whole result
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) s += m[i];
  out[tid] = s;
}

__global__ void __launch_bounds__(64) k_check(const uint32_t* __restrict__ in, uint32_t* __restrict__ out_valu, uint32_t* __restrict__ out_mfma) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t m[9];
  for (int i = 0; i < 9; ++i) m[i] = in[tid * 9 + i];
  uint64_t A[3][3];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) A[a][b] = c_atab[a][b][threadIdx.x];
  uint32_t r0[18], r1[18];
  mp_valu<FqParams>(m, r0);
  mp_mfma(m, A, r1);
  for (int i = 0; i < 18; ++i) { out_valu[tid * 18 + i] = r0[i]; out_mfma[tid * 18 + i] = r1[i]; }
}

template <class K> float time_it(K launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  // the modulus as seven-bit digits, and the A operand: A[row][k] = pd[row - k]; lane l of block (mb, kb): row 32 mb + l % 32, k = 16 kb + 8 (l / 32) + byte
  uint32_t pl[9];
  for (int i = 0; i < 9; ++i) pl[i] = FqParams::P[i];
  auto bit_of = [&](int b) { return (pl[b / 29] >> (b % 29)) & 1u; };
  int pd[NPD + 1];
  for (int d = 0; d <= NPD; ++d) { pd[d] = 0; for (int b = 0; b < 7; ++b) if (7 * d + b < 261) pd[d] |= bit_of(7 * d + b) << b; }
  static uint64_t atab[3][3][64];
  for (int mb = 0; mb < 3; ++mb)
    for (int kb = 0; kb < 3; ++kb)
      for (int l = 0; l < 64; ++l) {
        uint64_t v = 0;
        for (int b = 0; b < 8; ++b) {
          const int row = 32 * mb + l % 32, k = 16 * kb + 8 * (l / 32) + b, d = row - k;
          const uint64_t e = (d >= 0 && d <= NPD && k < ND) ? (uint64_t)pd[d] : 0;
          v |= e << (8 * b);
        }
        atab[mb][kb][l] = v;
      }
  CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_atab), atab, sizeof(atab)));

  // correctness: 4096 values through both paths and through 128-bit host arithmetic
  const int NC = 4096;
  std::vector<uint32_t> hin(NC * 9), hv(NC * 18), hm(NC * 18);
  uint64_t st = 0x9e3779b97f4a7c15ull;
  for (auto& x : hin) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; x = (uint32_t)st & M29; }
  for (int i = 0; i < 9; ++i) { hin[i] = M29; hin[9 + i] = 0; hin[18 + i] = i == 0; }          // all ones, zero, one
  uint32_t *din, *dv, *dm;
  CHECK(hipMalloc(&din, hin.size() * 4)); CHECK(hipMalloc(&dv, hv.size() * 4)); CHECK(hipMalloc(&dm, hm.size() * 4));
  CHECK(hipMemcpy(din, hin.data(), hin.size() * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_check, dim3(NC / 64), dim3(64), 0, 0, din, dv, dm);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(hv.data(), dv, hv.size() * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(hm.data(), dm, hm.size() * 4, hipMemcpyDeviceToHost));
  int bad_v = 0, bad_m = 0;
  for (int t = 0; t < NC; ++t) {
    unsigned __int128 acc = 0;
    for (int col = 0; col < 18; ++col) {
      for (int i = 0; i < 9; ++i) { const int j = col - i; if (j >= 0 && j < 9) acc += (unsigned __int128)hin[t * 9 + i] * pl[j]; }
      const uint32_t want = col < 17 ? (uint32_t)acc & M29 : (uint32_t)acc;
      if (col < 17) acc >>= 29;
      bad_v += hv[t * 18 + col] != want;
      bad_m += hm[t * 18 + col] != want;
    }
  }
  printf("check over %d values: VALU path %s, MFMA path %s\n", NC, bad_v ? "WRONG" : "ok", bad_m ? "WRONG" : "ok");

  uint32_t* out;
  const int blocks = 256 * 4 * 8, iters = 200;          // eight one-wave workgroups per SIMD
  CHECK(hipMalloc(&out, (size_t)blocks * 64 * 4));
  for (int waves = 1; waves <= 8; waves *= 2) {
    const int nb = 256 * 4 * waves;
    const float t0 = time_it([&] { hipLaunchKernelGGL(k_rate<0>, dim3(nb), dim3(64), 0, 0, out, iters); });
    const float t1 = time_it([&] { hipLaunchKernelGGL(k_rate<1>, dim3(nb), dim3(64), 0, 0, out, iters); });
    const double n = (double)nb * 64 * iters;
    printf("waves/SIMD=%d  m*p on the VALU: %.1f G half-products/s   through the MFMA: %.1f G half-products/s   ratio %.2f\n", waves, n / t0 / 1e6, n / t1 / 1e6,
           t1 / t0);
  }
  CHECK(hipDeviceSynchronize());
  return (bad_v || bad_m) ? 2 : 0;
}

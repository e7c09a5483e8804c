// dpfp_mul.hip -- a double-precision-FMA Montgomery multiplier for the BN254 base field, measured against the
// integer 9 x 29-bit multiplier of fp29.h (v_fma_f64 issues at the same ~34 T/s as v_mad_u64_u32 and
// covers 52 x 52 bits per instruction pair).
//
// Scheme (Emmart's DPFP trick): 5 limbs of 52 bits held as doubles, Montgomery radix 2^260.  With round-toward-zero,
//   ph = fma(a, b, 2^104)                 -> 2^104 + floor(ab / 2^52) * 2^52    (exact: one binade, ulp 2^52)
//   pl = fma(a, b, (2^104 + 2^52) - ph)   -> 2^52 + (ab mod 2^52)               (exact: below 2^53)
// and the BIT PATTERNS of ph / pl are exponent | integer, so partial products are accumulated with 64-bit integer adds
// into columns whose start values cancel the exponents.  Per 52 x 52 partial product: 2 FMA + 1 FP add + 2 int64 adds.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../kogarashi_amd/csrc -o dpfp_mul dpfp_mul.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include "fp29.h"
using namespace kg;

struct Dp { double l[5]; };

// BN254 q in 52-bit limbs and -q^-1 mod 2^52 (filled by the host from the 29-bit constants)
struct DpConsts { double p[5]; double np0; };

__device__ __forceinline__ void split(double a, double b, uint64_t& hi, uint64_t& lo) {
  const double C1 = 0x1p104, C2 = 0x1p104 + 0x1p52;
  // asm: the compiler does not model the rounding mode as a dependency, so builtin FMAs may be scheduled across the mode
  // switch (and folded under round-to-nearest); volatile asm keeps them in order with the s_setreg
  double ph, pl;
  asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(ph) : "v"(a), "v"(b), "v"(C1));
  const double sub = C2 - ph;
  asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(pl) : "v"(a), "v"(b), "v"(sub));
  hi = (uint64_t)__double_as_longlong(ph);
  lo = (uint64_t)__double_as_longlong(pl);
}
__device__ __forceinline__ double to_double52(uint64_t v) {          // integer < 2^52 -> double, exactly
  return __longlong_as_double((long long)(v | 0x4330000000000000ull)) - 0x1p52;
}

__device__ __forceinline__ Dp dp_mont(const Dp& a, const Dp& b, const DpConsts& K) {
  constexpr uint64_t EH = 0x4670000000000000ull, EL = 0x4330000000000000ull, M52 = (1ull << 52) - 1;
  // column k receives lo terms from (i + j == k) and hi terms from (i + j + 1 == k), twice (a*b and m*p)
  uint64_t col[11];
#pragma unroll
  for (int k = 0; k < 11; ++k) {
    const int nlo = k < 5 ? k + 1 : (k < 9 ? 9 - k : 0);
    const int nhi = (k >= 1 && k <= 5) ? k : (k >= 6 && k <= 9 ? 10 - k : 0);
    col[k] = 0ull - 2ull * ((uint64_t)nlo * EL + (uint64_t)nhi * EH);
  }
#pragma unroll
  for (int i = 0; i < 5; ++i) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      uint64_t hi, lo;
      split(a.l[i], b.l[j], hi, lo);
      col[i + j] += lo;
      col[i + j + 1] += hi;
    }
    // m = (col[i] * np0) mod 2^52
    uint64_t mh, ml;
    split(to_double52(col[i] & M52), K.np0, mh, ml);
    const double m = __longlong_as_double((long long)ml) - 0x1p52;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      uint64_t hi, lo;
      split(m, K.p[j], hi, lo);
      col[i + j] += lo;
      col[i + j + 1] += hi;
    }
    col[i + 1] += col[i] >> 52;                          // the low 52 bits of col[i] are zero now
  }
  Dp r;
  uint64_t cy = 0;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const uint64_t v = col[5 + k] + cy;
    r.l[k] = to_double52(k < 4 ? (v & M52) : v);
    cy = v >> 52;
  }
  return r;
}

__global__ void __launch_bounds__(64) k_dp(DpConsts K, const double* in, double* out, int iters) {
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3");   // MODE.fp_round[3:2] (f64 / f16) = toward zero
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  Dp a, b;
#pragma unroll
  for (int k = 0; k < 5; ++k) { a.l[k] = in[k]; b.l[k] = in[5 + k]; }
  a.l[0] += (double)(tid & 0xffff);
  for (int i = 0; i < iters; ++i) a = dp_mont(a, b, K);
#pragma unroll
  for (int k = 0; k < 5; ++k) out[(size_t)tid * 5 + k] = a.l[k];
}
__global__ void k_split_test(const double* in, uint64_t* out) {
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3");
  uint64_t hi, lo;
  split(in[threadIdx.x], in[5 + threadIdx.x], hi, lo);
  out[2 * threadIdx.x] = hi; out[2 * threadIdx.x + 1] = lo;
}
__global__ void __launch_bounds__(64) k_int(uint32_t* out, int iters) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  Fq a = Fq::one(), b = Fq::from_const(FqParams::G1_B3);
  a.l[0] += tid & 0xffff; b.l[1] ^= tid & 0xfff;
  for (int i = 0; i < iters; ++i) a = mul(a, b);
  uint32_t s = 0;
  for (int k = 0; k < 9; ++k) s += a.l[k];
  out[tid] = s;
}

// ---- host big-integer check: r * 2^260 == a * b (mod p), 5 x 52-bit limbs ------------------------------
typedef unsigned __int128 u128;
struct Big { uint64_t w[12]; };                          // little endian 64-bit words
static Big from52(const double* l) {
  Big r; memset(&r, 0, sizeof r);
  for (int k = 0; k < 5; ++k) {
    const uint64_t v = (uint64_t)l[k];
    const int bit = 52 * k, wd = bit / 64, sh = bit % 64;
    r.w[wd] |= v << sh;
    if (sh > 12) r.w[wd + 1] |= v >> (64 - sh);
  }
  return r;
}
static int cmp(const Big& a, const Big& b) { for (int i = 11; i >= 0; --i) if (a.w[i] != b.w[i]) return a.w[i] < b.w[i] ? -1 : 1; return 0; }
static void sub(Big& a, const Big& b) { u128 br = 0; for (int i = 0; i < 12; ++i) { u128 d = (u128)a.w[i] - b.w[i] - (uint64_t)br; a.w[i] = (uint64_t)d; br = (d >> 64) & 1; } }
static void dbl_mod(Big& a, const Big& p) { uint64_t c = 0; for (int i = 0; i < 12; ++i) { uint64_t n = a.w[i] >> 63; a.w[i] = (a.w[i] << 1) | c; c = n; } if (cmp(a, p) >= 0) sub(a, p); }
static void add_mod(Big& a, const Big& b, const Big& p) { u128 c = 0; for (int i = 0; i < 12; ++i) { c += (u128)a.w[i] + b.w[i]; a.w[i] = (uint64_t)c; c >>= 64; } if (cmp(a, p) >= 0) sub(a, p); }
static Big reduce(Big a, const Big& p) { while (cmp(a, p) >= 0) sub(a, p); return a; }
static Big mul_mod(const Big& a, const Big& b, const Big& p) {      // double-and-add over the bits of b
  Big r; memset(&r, 0, sizeof r);
  for (int bit = 319; bit >= 0; --bit) { dbl_mod(r, p); if ((b.w[bit / 64] >> (bit % 64)) & 1) add_mod(r, a, p); }
  return r;
}

template <class K> float time_it(K launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  // q as one integer from the 29-bit limbs, then 52-bit limbs
  Big P; memset(&P, 0, sizeof P);
  for (int k = 0; k < 9; ++k) {
    const uint64_t v = FqParams::P[k];
    const int bit = 29 * k, wd = bit / 64, sh = bit % 64;
    P.w[wd] |= v << sh;
    if (sh > 35) P.w[wd + 1] |= v >> (64 - sh);
  }
  DpConsts K;
  for (int k = 0; k < 5; ++k) {
    const int bit = 52 * k, wd = bit / 64, sh = bit % 64;
    uint64_t v = P.w[wd] >> sh;
    if (sh > 12) v |= P.w[wd + 1] << (64 - sh);
    K.p[k] = (double)(v & ((1ull << 52) - 1));
  }
  uint64_t p0 = (uint64_t)K.p[0], inv = 1;               // -p^-1 mod 2^52 by Newton iteration
  for (int i = 0; i < 6; ++i) inv *= 2 - p0 * inv;
  K.np0 = (double)((0 - inv) & ((1ull << 52) - 1));
  double h_in[10];
  uint64_t s = 0x9E3779B97F4A7C15ull;
  for (int k = 0; k < 10; ++k) { s = s * 6364136223846793005ull + 1442695040888963407ull; h_in[k] = (double)((s >> 12) & ((k % 5 == 4) ? ((1ull << 45) - 1) : ((1ull << 52) - 1))); }
  double *d_in, *d_out; uint32_t* d_u;
  hipMalloc(&d_in, sizeof h_in); hipMalloc(&d_out, (size_t)(1 << 21) * 5 * 8); hipMalloc(&d_u, 1 << 24);
  hipMemcpy(d_in, h_in, sizeof h_in, hipMemcpyHostToDevice);
  {
    uint64_t* d_s; hipMalloc(&d_s, 80);
    k_split_test<<<1, 5>>>(d_in, d_s);
    uint64_t hs[10]; hipMemcpy(hs, d_s, 80, hipMemcpyDeviceToHost);
    for (int k = 0; k < 5; ++k) {
      const u128 pr = (u128)(uint64_t)h_in[k] * (uint64_t)h_in[5 + k];
      const uint64_t whi = (uint64_t)(pr >> 52), wlo = (uint64_t)pr & ((1ull << 52) - 1);
      printf("split %d: hi %s (%016llx vs %016llx)  lo %s (%016llx vs %016llx)\n", k, (hs[2 * k] - 0x4670000000000000ull) == whi ? "ok" : "BAD",
             (unsigned long long)(hs[2 * k] - 0x4670000000000000ull), (unsigned long long)whi, (hs[2 * k + 1] - 0x4330000000000000ull) == wlo ? "ok" : "BAD",
             (unsigned long long)(hs[2 * k + 1] - 0x4330000000000000ull), (unsigned long long)wlo);
    }
  }
  // correctness: one product per lane, lane t has a.l[0] += t
  k_dp<<<1, 64>>>(K, d_in, d_out, 1);
  double h_out[64 * 5];
  hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost);
  Big R; memset(&R, 0, sizeof R); R.w[4] = 1ull << 4;    // 2^260
  R = reduce(R, P);
  int bad = 0;
  for (int t = 0; t < 64; ++t) {
    double a[5]; memcpy(a, h_in, sizeof a); a[0] += (double)t;
    const Big want = mul_mod(reduce(from52(a), P), reduce(from52(h_in + 5), P), P);
    const Big got = mul_mod(reduce(from52(h_out + 5 * t), P), R, P);
    bad += cmp(want, got) != 0;
  }
  printf("a:"); for (int k = 0; k < 5; ++k) printf(" %.0f", h_in[k]); printf("\nb:"); for (int k = 0; k < 5; ++k) printf(" %.0f", h_in[5 + k]);
  printf("\nr:"); for (int k = 0; k < 5; ++k) printf(" %.0f", h_out[k]); printf("\np:"); for (int k = 0; k < 5; ++k) printf(" %.0f", K.p[k]); printf("\nnp0: %.0f\n", K.np0);
  printf("DPFP Montgomery product vs big-integer check: %d of 64 lanes wrong\n", bad);
  for (int wps : {1, 2, 3, 4, 6, 8}) {
    const int blocks = 256 * 4 * wps, iters = 2000;
    float ms = time_it([&] { k_dp<<<blocks, 64>>>(K, d_in, d_out, iters); });
    float mi = time_it([&] { k_int<<<blocks, 64>>>(d_u, iters); });
    printf("waves/SIMD=%d   f64 5x52: %8.3f ms %8.2f Gmul/s   |   int 9x29: %8.3f ms %8.2f Gmul/s\n", wps, ms, (double)blocks * 64 * iters / ms * 1e-6, mi,
           (double)blocks * 64 * iters / mi * 1e-6);
  }
  hipFuncAttributes fa;
  hipFuncGetAttributes(&fa, (const void*)k_dp);
  printf("k_dp: %d VGPRs\n", fa.numRegs);
  return 0;
}

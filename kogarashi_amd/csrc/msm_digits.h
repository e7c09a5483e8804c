// msm_digits.h -- the signed window digits of a scalar, as plain host / device functions (the short-input MSM of msm_small_kernels.h
// extracts them per window inside its one kernel; tests/host/hosttest.cpp checks sum_w d_w 2^(wc) == k for every width).
//
// k is the canonical integer of the scalar (< 2^254: both scalar fields).  With W = ceil(255 / c) windows and the bias
// H = sum_{w < W-1} 2^(wc + c - 1), window w < W - 1 of k + H holds e_w in [0, 2^c) and its signed digit is d_w = e_w - 2^(c-1) in
// [-2^(c-1), 2^(c-1)); the top window is the unsigned rest of k + H, at most 2^(254 - (W-1)c) <= 2^(c-1) (W c >= 255).  So every
// |d_w| is a bucket number in 1 .. 2^(c-1) (0: no addition) and k = sum_w d_w 2^(wc).  The same digits as msm_sort_kernels.h
// (window_digit) -- replaces the unsigned windows of groth16/src/msm.rs:75-91 (get_at).
#pragma once
#include <cstdint>
#include "fp29.h"
#include "glv_consts.h"

namespace kg {

// H as eight 32-bit words
KG_HD void small_bias(int c, int W, uint32_t H[8]) {
  for (int j = 0; j < 8; ++j) H[j] = 0;
  for (int w = 0; w < W - 1; ++w) {
    const int bit = w * c + c - 1;
    H[bit >> 5] |= 1u << (bit & 31);
  }
}

// kb = k + H (eight words, little endian); returns |d_w| (0 = skip) and its sign.  c <= 16.
KG_HD uint32_t small_window_digit(const uint32_t kb[8], int w, int c, int W, bool& negative) {
  const int o = w * c;
  const int j = o >> 5, sh = o & 31;
  uint64_t v = kb[j];
  if (j + 1 < 8) v |= (uint64_t)kb[j + 1] << 32;
  const uint32_t e = (uint32_t)(v >> sh);           // bits o .. o + 31 (the top window starts at >= 255 - c: nothing above is lost)
  if (w == W - 1) {
    negative = false;
    return e;
  }
  const int32_t d = (int32_t)(e & ((1u << c) - 1u)) - (int32_t)(1u << (c - 1));
  negative = d < 0;
  return (uint32_t)(d < 0 ? -d : d);
}

// ---- GLV: k = k1 + k2 lambda (mod n), |k1|, |k2| < 2^127 (glv_consts.h; tools/gen/glv_consts.py states the bounds it measured) ------------
// All three curves of the path have j = 0, so (x, y) -> (beta x, y) is multiplication by lambda on the prime-order group: a pair (k, P)
// becomes two pairs (|k1|, +-P), (|k2|, +-(beta x, y)) with scalars of half the length -- half the windows on the device and half the
// doublings of the host chain.  c_i = (G_i k + 2^255) >> 256 rounds k (b2, -b1) / n to the nearest integer up to the truncation of G_i (an
// error in [0, 1/4) for k < 2^254): k1 = e1 a1 + e2 a2, k2 = e1 b1 + e2 b2 with e_i in (-1/2, 3/4), so |k1|, |k2| <= 3/4 (|a1| + |a2|) < 2^126.4
// -- every window width keeps its top digit within 2^(c-1) buckets (the top window starts at bit 120 .. 126 and |k_i| + H < 2^127).  k1 = k - c1 a1
// - c2 a2 and k2 = -c1 b1 - c2 b2 are computed modulo 2^160 in two's complement.
// words [lo, lo + nr) of a (na words) * b (nb words)
template <int NA, int NB, int LO, int NR>
KG_HD void glv_mul_words(const uint32_t* a, const uint32_t* b, uint32_t* r) {
  uint64_t carry = 0;                                  // column sums of up to NB products of 2^64: hi parts are carried as a second word
  uint64_t carry_hi = 0;
#pragma unroll
  for (int col = 0; col < LO + NR; ++col) {
    uint64_t lo = carry, hi = carry_hi;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int j = col - i;
      if (j < 0 || j >= NB) continue;
      const uint64_t pr = (uint64_t)a[i] * b[j];
      lo += pr & 0xffffffffu;
      hi += pr >> 32;
    }
    if (col >= LO) r[col - LO] = (uint32_t)lo;
    carry = (lo >> 32) + (hi & 0xffffffffu);
    carry_hi = hi >> 32;
  }
}
template <class L>
KG_HD void glv_decompose_with(const uint32_t k[8], uint32_t k1[4], bool& neg1, uint32_t k2[4], bool& neg2) {
  uint32_t c1[3], c2[5];
  {                                                    // c_i = (G_i k + 2^255) >> 256: ROUNDED -- with a plain floor the halves reach 1.25 (a1 + a2) = 2^127.1
    uint32_t t[6];                                     // and the top window's digit can leave its bucket range (c = 4, 8)
    glv_mul_words<8, 3, 7, 4>(k, L::G1, t);
    uint64_t cy = ((uint64_t)t[0] + 0x80000000u) >> 32;
#pragma unroll
    for (int i = 0; i < 3; ++i) { const uint64_t v = (uint64_t)t[1 + i] + cy; c1[i] = (uint32_t)v; cy = v >> 32; }
    glv_mul_words<8, 5, 7, 6>(k, L::G2, t);
    cy = ((uint64_t)t[0] + 0x80000000u) >> 32;
#pragma unroll
    for (int i = 0; i < 5; ++i) { const uint64_t v = (uint64_t)t[1 + i] + cy; c2[i] = (uint32_t)v; cy = v >> 32; }
  }
  uint32_t t1[5], t2[5], r1[5], r2[5];
  glv_mul_words<3, 5, 0, 5>(c1, L::A1, t1);
  glv_mul_words<5, 5, 0, 5>(c2, L::A2, t2);
  {
    int64_t br = 0;                                    // r1 = k - t1 - t2 (mod 2^160)
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int64_t d = (int64_t)k[i] - (int64_t)t1[i] - (int64_t)t2[i] + br;
      r1[i] = (uint32_t)d;
      br = d >> 32;                                    // arithmetic shift: -2 .. 0
    }
  }
  glv_mul_words<3, 5, 0, 5>(c1, L::B1, t1);
  glv_mul_words<5, 5, 0, 5>(c2, L::B2, t2);
  {
    int64_t br = 0;                                    // r2 = -t1 - t2
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int64_t d = -(int64_t)t1[i] - (int64_t)t2[i] + br;
      r2[i] = (uint32_t)d;
      br = d >> 32;
    }
  }
  neg1 = (r1[4] >> 31) != 0;
  neg2 = (r2[4] >> 31) != 0;
  uint64_t cy1 = neg1 ? 1 : 0, cy2 = neg2 ? 1 : 0;     // |x| = neg ? ~x + 1 : x; the fifth word is the sign extension
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint64_t s1 = (uint64_t)(neg1 ? ~r1[i] : r1[i]) + cy1;
    k1[i] = (uint32_t)s1; cy1 = s1 >> 32;
    const uint64_t s2 = (uint64_t)(neg2 ? ~r2[i] : r2[i]) + cy2;
    k2[i] = (uint32_t)s2; cy2 = s2 >> 32;
  }
}

}  // namespace kg

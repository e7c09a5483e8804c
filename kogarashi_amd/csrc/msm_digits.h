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

}  // namespace kg

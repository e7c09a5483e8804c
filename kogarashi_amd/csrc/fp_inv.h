// fp_inv.h -- modular inversion by a constant-time binary GCD on the 9 x 29-bit limbs (device + host).
//
// Replaces the Fermat ladder a^(p-2) of zkstd/src/arithmetic/limbs/bits_256/normal.rs:256-270 (254 squarings + ~127
// products, ~270 product-equivalents on this machine) where an inversion sits on a per-element path: the window-table
// build (one inversion per point and window), to_affine, the vector op KG_OP_INVERT.  The result is the same field element.
//
// Algorithm: Pornin, "Optimized Binary GCD for Modular Inversion" (2020), with 29 inner steps per round so that the exact division of
// every outer round is a limb shift: 18 rounds of 29 inner steps (>= 2 * 254 - 1 iterations).  A round runs the plain
// binary-GCD steps on 60-bit approximations of (a, b) (their 29 low bits and the 31 bits below the common top), collecting
// the update as four factors |f|, |g| <= 2^29, then applies them to the full values: (a, b) exactly, (u, v) modulo p with
// one Montgomery step (+ t * p, / 2^29), which keeps a = u * y and b = v * y (mod p) without a scale drift.  No branch
// depends on data: every lane of a wave runs the same instruction stream.
#pragma once
#include "fp29.h"

#if defined(KG_HOST_TEST) && !defined(__HIP_DEVICE_COMPILE__)
#include <cstdio>
#include <cstdlib>
// host builds of the tests (tests/host/, -DKG_HOST_TEST): the bounds the comments below claim are asserted on every call -- the
// 64-bit accumulators of lincomb, the limb ranges after each round, |f|, |g| <= 2^29, the lifted result below 2^261.  (The
// bound-tracking FpChecked type routes inv_fast to the Fermat ladder: this is the binary GCD's own bound discipline.)
#define KG_INV_CHECK(cond) do { if (!(cond)) { std::fprintf(stderr, "fp_inv.h bound violated: %s (line %d)\n", #cond, __LINE__); std::abort(); } } while (0)
#else
#define KG_INV_CHECK(cond) ((void)0)
#endif

namespace kg {

namespace invd {
KG_HD bool acc_ok(int64_t acc) { return acc < ((int64_t)1 << 62) && acc > -((int64_t)1 << 62); }
// x * f + y * g (+ t * p) over nine signed-top limbs, divided by 2^29 (the low limb of the sum is zero by construction)
template <class P, bool MOD>
KG_HD void lincomb(const int32_t* x, const int32_t* y, int32_t f, int32_t g, int32_t* out) {
  int64_t acc = (int64_t)x[0] * f + (int64_t)y[0] * g;
  uint32_t t = 0;
  if (MOD) {
    t = (((uint32_t)acc & M29) * P::INV) & M29;
    acc += (int64_t)((uint64_t)t * P::P[0]);
  }
  KG_INV_CHECK(acc_ok(acc) && ((uint32_t)acc & M29) == 0);          // the low limb of the sum is zero by construction
  acc >>= 29;
#pragma unroll
  for (int i = 1; i < 9; ++i) {
    acc += (int64_t)x[i] * f + (int64_t)y[i] * g;
    if (MOD) acc += (int64_t)((uint64_t)t * P::P[i]);
    KG_INV_CHECK(acc_ok(acc));
    out[i - 1] = (int32_t)((uint32_t)acc & M29);
    acc >>= 29;
  }
  KG_INV_CHECK(acc < ((int64_t)1 << 30) && acc >= -((int64_t)1 << 30));      // the signed top limb fits 32 bits with room for the next round's factors
  out[8] = (int32_t)acc;
}
// x <- -x where neg (all-ones mask) says so; limbs 0..7 stay in [0, 2^29), the top limb carries the sign
KG_HD void cond_negate(int32_t* x, int32_t neg) {
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int32_t t = 0 - x[i] - c;
    c = (t >> 31) & 1;
    x[i] = (neg & (t & (int32_t)M29)) | (~neg & x[i]);
  }
  x[8] = (neg & (0 - x[8] - c)) | (~neg & x[8]);
}
}  // namespace invd

// a^-1 in the internal Montgomery form (0 for a == 0, like the Fermat ladder); a: any loose value mul() accepts
template <class P>
KG_HD Fp<P> inv_bingcd(const Fp<P>& x) {
  const Fp<P> y = reduce(x);                      // the canonical integer A = a * 2^261 mod p
  int32_t a[9], b[9], u[9], v[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) { a[i] = (int32_t)y.l[i]; b[i] = (int32_t)P::P[i]; u[i] = 0; v[i] = 0; }
  u[0] = 1;
  for (int round = 0; round < 18; ++round) {
    // 60-bit approximations (k = 30: k - 1 = 29 low bits, k + 1 = 31 bits under the common top n = max bit length; exact
    // values when n <= 60): scan for the top non-zero limb pair, keeping the two limbs below it
    uint32_t ta = (uint32_t)a[2], tb = (uint32_t)b[2], na = (uint32_t)a[1], nb = (uint32_t)b[1], la = (uint32_t)a[0], lb = (uint32_t)b[0];
    bool above = false;                           // a limb above index 2 is set
#pragma unroll
    for (int i = 3; i < 9; ++i) {
      const bool nz = ((uint32_t)a[i] | (uint32_t)b[i]) != 0;
      ta = nz ? (uint32_t)a[i] : ta; tb = nz ? (uint32_t)b[i] : tb;
      na = nz ? (uint32_t)a[i - 1] : na; nb = nz ? (uint32_t)b[i - 1] : nb;
      la = nz ? (uint32_t)a[i - 2] : la; lb = nz ? (uint32_t)b[i - 2] : lb;
      above = above || nz;
    }
    uint32_t top = ta | tb;
    int s = 0;                                    // bit length of the top limb pair
#pragma unroll
    for (int sh = 16; sh >= 1; sh >>= 1) { const bool big = (top >> sh) != 0; s += big ? sh : 0; top = big ? (top >> sh) : top; }
    s += (int)top;                                // top is now 0 or 1
    const bool wide = above || s > 2;             // n = 29 t + s > 60
    // the three limbs from the top hold s + 58 bits; their top 31: ((t:n) * 4 + (l >> 27)) >> s
    const uint64_t ha = ((((uint64_t)ta << 29) | na) << 2) | (la >> 27), hb = ((((uint64_t)tb << 29) | nb) << 2) | (lb >> 27);
    const uint64_t ea = ((uint64_t)(uint32_t)a[2] << 58) | ((uint64_t)(uint32_t)a[1] << 29) | (uint32_t)a[0];
    const uint64_t eb = ((uint64_t)(uint32_t)b[2] << 58) | ((uint64_t)(uint32_t)b[1] << 29) | (uint32_t)b[0];
    uint64_t xa = wide ? (((ha >> s) << 29) | (uint32_t)a[0]) : ea;
    uint64_t xb = wide ? (((hb >> s) << 29) | (uint32_t)b[0]) : eb;
    int32_t f0 = 1, g0 = 0, f1 = 0, g1 = 1;
    for (int i = 0; i < 29; ++i) {
      const uint64_t odd = 0 - (xa & 1);
      const uint64_t swp = (xa < xb ? ~(uint64_t)0 : 0) & odd;
      const uint64_t tx = (xa ^ xb) & swp;
      xa ^= tx; xb ^= tx;
      const int32_t sw32 = (int32_t)swp, od32 = (int32_t)odd;
      const int32_t tf = (f0 ^ f1) & sw32, tg = (g0 ^ g1) & sw32;
      f0 ^= tf; f1 ^= tf; g0 ^= tg; g1 ^= tg;
      xa -= xb & odd;
      f0 -= f1 & od32; g0 -= g1 & od32;
      xa >>= 1;
      f1 <<= 1; g1 <<= 1;
    }
    KG_INV_CHECK(f0 <= (1 << 29) && f0 >= -(1 << 29) && g0 <= (1 << 29) && g0 >= -(1 << 29) && f1 <= (1 << 29) && f1 >= -(1 << 29) &&
                 g1 <= (1 << 29) && g1 >= -(1 << 29));
    int32_t an[9], bn[9], un[9], vn[9];
    invd::lincomb<P, false>(a, b, f0, g0, an);
    invd::lincomb<P, false>(a, b, f1, g1, bn);
    const int32_t nga = an[8] >> 31, ngb = bn[8] >> 31;
    invd::cond_negate(an, nga);
    invd::cond_negate(bn, ngb);
    f0 = (f0 ^ nga) - nga; g0 = (g0 ^ nga) - nga;
    f1 = (f1 ^ ngb) - ngb; g1 = (g1 ^ ngb) - ngb;
    invd::lincomb<P, true>(u, v, f0, g0, un);
    invd::lincomb<P, true>(u, v, f1, g1, vn);
#pragma unroll
    for (int i = 0; i < 9; ++i) { a[i] = an[i]; b[i] = bn[i]; u[i] = un[i]; v[i] = vn[i]; }
#if defined(KG_HOST_TEST) && !defined(__HIP_DEVICE_COMPILE__)
    for (int i = 0; i < 8; ++i) KG_INV_CHECK(a[i] >= 0 && a[i] < (1 << 29) && b[i] >= 0 && b[i] < (1 << 29) && u[i] >= 0 && u[i] < (1 << 29) && v[i] >= 0 && v[i] < (1 << 29));
    KG_INV_CHECK(a[8] >= 0 && b[8] >= 0);                          // (a, b) stay non-negative after the conditional negation
    KG_INV_CHECK(u[8] > -(1 << 27) && u[8] < (1 << 27) && v[8] > -(1 << 27) && v[8] < (1 << 27));      // |u|, |v| < 19 p: top limb |.| < 19 * 2^22 < 2^27
#endif
  }
  // b = gcd = 1 (or p when a was 0, with v = 0); v = A^-1 mod p as a signed value in (-19p, 19p): lift by 32p and move to the
  // internal form of the inverse: mont(A^-1, 2^783) = A^-1 * 2^522 = a^-1 * 2^261
  Fp<P> r;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint32_t t = (uint32_t)v[i] + P::P32[i] + c;
    r.l[i] = t & M29;
    c = t >> 29;
  }
  r.l[8] = (uint32_t)(v[8] + (int32_t)P::P32[8] + (int32_t)c);
  KG_INV_CHECK(v[8] + (int32_t)P::P32[8] + (int32_t)c >= 0 && r.l[8] < (1u << 29));      // 0 <= v + 32 p < 51 p < 2^261: a loose value mul() accepts
  return mul(r, Fp<P>::from_const(P::C_R3));
}

// the inversion the kernels call (the bound-checking shadow type and the host's 4 x 64-bit type route to their own)
template <class P> KG_HD Fp<P> inv_fast(const Fp<P>& a) { return inv_bingcd(a); }
template <class F>
KG_HD Fp2<F> inv_fast(const Fp2<F>& a) {            // bn254/src/fqn.rs:348-357 over the base field's inversion
  F t = inv_fast(norm(add(sqr(a.c0), sqr(a.c1))));
  return {mul(t, a.c0), mul(t, norm(sub<16, 1>(F::zero(), a.c1)))};
}

}  // namespace kg

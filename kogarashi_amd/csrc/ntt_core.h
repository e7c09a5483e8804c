// ntt_core.h -- register-resident radix-2^G decimation-in-time butterfly network (G consecutive radix-2 stages
// on 2^G elements held by one lane).  Shared by the NTT kernel (F = Fr) and the host bound checker (F = FrC).
//
// Follows butterfly_arithmetic (groth16/src/fft.rs:195-218): t = right * w; right = left - t; left = left + t.
// Sums and differences stay lazy (no carry propagation) for the whole network: after g stages a limb is below
// 2^29 + g * 2^30 (< 2^32 for g <= 3) and the worst product column 9 * (2^29 + 2 * 2^30) * 2^29 < 2^64, which
// FpChecked verifies in tests/host.  The caller normalises once when it stores the elements.
#pragma once
#include "fp29.h"

namespace kg {

// x[k], k in [0, 2^G): element whose G "middle" index bits equal k.  Stage t (1..G) pairs k0 / k0 + 2^(t-1).
// tw.mul(x, t, k0) returns x times the twiddle of that butterfly (the kernel: a Shoup-form constant product, fp29.h mulc);
// trivial_first: this pass runs global stages 1..G.
// (compile-time recursion instead of loops: every x[] index is a constant expression, so the array stays in registers)
template <int G, int T, int PI, class F, class TwFn>
KG_HD void dit_step(F (&x)[1 << G], bool trivial_first, const TwFn& tw) {
  if constexpr (T <= G) {
    constexpr int half = 1 << (T - 1);
    constexpr int k0 = ((PI >> (T - 1)) << T) | (PI & (half - 1));
    constexpr int k1 = k0 + half;
    // first pass of a tile (global stages 1..G): the twiddle of stage T is w^(k0 mod 2^(T-1)), i.e. 1 for 1, 2, 4 of the
    // butterflies of stages 1, 2, 3 -- 7 of the 12 products of a radix-8 pass are skipped.  The operands of a skipped
    // product are lazy sums: stage 1 sees raw loads (value < 2^256: fat constant 8p), later stages reduce the value first.
    if (trivial_first && (k0 & (half - 1)) == 0) {
      if constexpr (T == 1) {
        F a = x[k0], b = norm(x[k1]);
        x[k0] = add(a, b);
        x[k1] = sub<8, 1>(a, b);
      } else {                                       // value reduction (a tenth of a product) keeps the growth of a product's path
        F a = x[k0], b = vred(norm(x[k1]));
        x[k0] = add(a, b);
        x[k1] = sub<4, 1>(a, b);
      }
    } else {
      F tt = tw.mul(x[k1], T, k0);                   // < 2.4p, normalised (mulc, or mul: < 2p): the 3p fat constant dominates it,
      F a = x[k0];                                   // so the value bound of the never-multiplied path grows by 3p per stage
      x[k0] = add(a, tt);
      x[k1] = sub<3, 1>(a, tt);
    }
    if constexpr (PI + 1 < (1 << (G - 1))) dit_step<G, T, PI + 1>(x, trivial_first, tw);
    else dit_step<G, T + 1, 0>(x, trivial_first, tw);
  }
}
template <int G, class F, class TwFn>
KG_HD void dit_network(F (&x)[1 << G], bool trivial_first, const TwFn& tw) {
#ifdef KG_NTT_EXP_NOMATH      // phase-off experiment (tools/dbg): data movement only
  (void)trivial_first; (void)tw;
  return;
#endif
  dit_step<G, 1, 0>(x, trivial_first, tw);
}

}  // namespace kg

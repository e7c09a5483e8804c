// vecops.h -- the lazy arithmetic of the vector kernels (vec.hip) as templates over the field type, so that the host
// bound checker (tests/host/hosttest.cpp, F = FpChecked) proves the same instruction sequences the device runs.
//
// Values cross memory in the ABI's Montgomery domain (x * 2^256).  A raw load times a constant in internal form (c * 2^261)
// Montgomery-multiplies to x * c * 2^256 (fp29.h: mul divides by 2^261); a raw load times a raw load gives x y 2^251, and the
// sparse products below stay in that domain until a constant the next multiplication needs anyway takes them out of it.
#pragma once
#include "fp29.h"

namespace kg {

// one term of a sparse row product (zkstd/src/matrix/row.rs:43-51): sum + z * v with BOTH factors raw (the ABI words as
// limbs, any 256-bit value): the term is z * v * 2^512 / 2^261, i.e. the sums live in the "raw product" domain x * 2^251
template <class F>
KG_HD F dot_step(const F& sum, const F& z_raw, const F& v_raw) { return vred(norm(add(sum, mul(z_raw, v_raw)))); }
// two partial sums of one row
template <class F>
KG_HD F dot_merge(const F& a, const F& b) { return vred(norm(add(a, b))); }

// Nova's cross term for one constraint row (nova/src/prover.rs:81-89):
//   AZ1 * BZ2 + AZ2 * BZ1 - u1 * CZ2 - u2 * CZ1
// az*, bz*, cz*: row sums as dot_step / dot_merge leave them (x * 2^251, below 1.06p).  u1s, u2s: u * 2^266 (the ABI form
// times 2^10; C_XT_U does it on the device, ten doublings on the host), so that (c * 2^251)(u * 2^266) / 2^261 = c u 2^256.
// had_const: 2^276 (P::C_XT_HAD): (a * 2^251)(b * 2^251) / 2^261 = a b 2^241, times 2^276 / 2^261 = a b 2^256.
// Result: ABI domain, normalised, below 2p (the caller canonicalises).
template <class F>
KG_HD F cross_term_row(const F& az1, const F& az2, const F& bz1, const F& bz2, const F& cz1, const F& cz2, const F& u1s, const F& u2s,
                       const F& had_const) {
  const F had = mul(mul2add(az1, bz2, az2, bz1), had_const);
  const F c1 = mul(cz2, u1s);
  const F c2 = mul(cz1, u2s);
  return vred(norm(sub<8, 1>(norm(sub<4, 1>(had, c1)), c2)));
}

}  // namespace kg

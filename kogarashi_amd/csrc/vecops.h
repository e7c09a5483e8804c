// vecops.h -- the lazy arithmetic of the vector kernels (vec.hip) as templates over the field type, so that the host
// bound checker (tests/host/hosttest.cpp, F = FpChecked) proves the same instruction sequences the device runs.
//
// Values stay in the ABI's Montgomery domain (x * 2^256): a raw load times a constant in internal form (c * 2^261)
// Montgomery-multiplies to x * c * 2^256 (fp29.h: mul divides by 2^261).
#pragma once
#include "fp29.h"

namespace kg {

// one term of a sparse row product (zkstd/src/matrix/row.rs:43-51): sum + z * v, z raw (any 256-bit value), v internal
template <class F>
KG_HD F dot_step(const F& sum, const F& z_raw, const F& v) { return vred(norm(add(sum, mul(z_raw, v)))); }
// two partial sums of one row
template <class F>
KG_HD F dot_merge(const F& a, const F& b) { return vred(norm(add(a, b))); }

// Nova's cross term for one constraint row (nova/src/prover.rs:81-89):
//   AZ1 * BZ2 + AZ2 * BZ1 - u1 * CZ2 - u2 * CZ1
// az*, bz*, cz*: row products as dot_step / dot_merge leave them (ABI domain, below 1.06p); u1, u2: internal form;
// from_ref_const: 2^522 / 2^256 (P::C_FROM_REF), the constant that brings a product of two ABI-domain values back to the
// ABI domain (the one KG_OP_MUL uses).  Result: ABI domain, normalised, below 2p (the caller canonicalises).
template <class F>
KG_HD F cross_term_row(const F& az1, const F& az2, const F& bz1, const F& bz2, const F& cz1, const F& cz2, const F& u1, const F& u2,
                       const F& from_ref_const) {
  const F had = mul(mul2add(az1, bz2, az2, bz1), from_ref_const);
  const F c1 = mul(cz2, u1);
  const F c2 = mul(cz1, u2);
  return vred(norm(sub<8, 1>(norm(sub<4, 1>(had, c1)), c2)));
}

}  // namespace kg

// curve.h -- short-Weierstrass a = 0 point arithmetic (BN254 G1 / G2, Grumpkin) for the MSM kernels.
//
// The reference accumulates buckets in homogeneous projective coordinates with the branchy formulas of
// zkstd/src/arithmetic/points/weierstrass.rs:6-163 (add_affine_point / add_mixed_point /
// add_projective_point / double_*).  MSM and commitment outputs are compared as AFFINE points (the unique
// canonical value), so the device is free to use a cheaper system: XYZZ coordinates
// (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; EFD "xyzz": madd-2008-s 8M+2S, add-2008-s 12M+2S, dbl-2008-s-1).
// The exceptional cases the reference branches on (identity operands, equal x => doubling or identity;
// weierstrass.rs:7-24, 64-86, 102-123) are all handled here, because CRS bases do contain identities and
// repeated points (groth16/src/zksnark.rs:62-64,177-185).
//
// Everything is a template over the field type F (Fq, Fr, Fq2, or their FpChecked shadows); the lazy
// add/sub/norm discipline of fp29.h is written out explicitly and machine-checked in tests/host/.
#pragma once
#include "fp29.h"
#include "fp_inv.h"

namespace kg {

template <class F>
struct Affine {   // never the identity: identity bases are filtered by the caller via their flag
  F x, y;
};

// identity <=> ZZ == 0 (exactly zero limbs; every path that produces the identity writes zeros)
template <class F>
struct XYZZ {
  F x, y, zz, zzz;
  static KG_HD XYZZ identity() { return {F::zero(), F::zero(), F::zero(), F::zero()}; }
};

template <class F>
KG_HD bool is_identity(const XYZZ<F>& p) { return is_zero_2p(p.zz); }

template <class F>
KG_HD XYZZ<F> from_affine(const Affine<F>& a) { return {a.x, a.y, F::one(), F::one()}; }

// 2 * (affine point)  (dbl-mdbl-2008-s-1).  In a prime-order group y != 0, result is never the identity
// for G1/Grumpkin; for G2 inputs in the r-torsion the same holds.
template <class F>
KG_HD XYZZ<F> double_affine(const Affine<F>& a) {
  F u = norm(dbl(a.y));                 // 2y
  F v = sqr(u);                         // 4y^2
  F w = mul(u, v);                      // 8y^3
  F s = mul(a.x, v);                    // 4xy^2
  F xx = sqr(a.x);
  F m = norm(add(dbl(xx), xx));         // 3x^2
  F x3 = vred(norm(sub<4, 1>(sqr(m), norm(dbl(s)))));
  F y3 = vred(norm(sub<4, 1>(mul(m, norm(sub<4, 1>(s, x3))), mul(w, a.y))));
  return {x3, y3, v, w};
}

// 2 * P  (dbl-2008-s-1)
template <class F>
KG_HD XYZZ<F> double_xyzz(const XYZZ<F>& p) {
  if (is_identity(p)) return p;
  F u = norm(dbl(p.y));
  F v = sqr(u);
  F w = mul(u, v);
  F s = mul(p.x, v);
  F xx = sqr(p.x);
  F m = norm(add(dbl(xx), xx));
  F x3 = vred(norm(sub<4, 1>(sqr(m), norm(dbl(s)))));
  F y3 = vred(norm(sub<4, 1>(mul(m, norm(sub<4, 1>(s, x3))), mul(w, p.y))));
  return {x3, y3, mul(v, p.zz), mul(w, p.zzz)};
}

// P + (affine a)  (madd-2008-s); a is not the identity.
// Invariant on stored XYZZ coordinates: normalised limbs, every coordinate < 2p (X, Y < 1.06p via vred).
template <class F>
KG_HD XYZZ<F> add_mixed(const XYZZ<F>& p, const Affine<F>& a) {
  if (is_identity(p)) return from_affine(a);
  F u2 = mul(a.x, p.zz);
  F s2 = mul(a.y, p.zzz);
  F pp_ = norm(sub<4, 1>(u2, p.x));     // P = U2 - X1
  F r = norm(sub<4, 1>(s2, p.y));       // R = S2 - Y1
  F pp = sqr(pp_);
  if (is_zero_2p(pp)) {                 // same x: doubling or inverse (weierstrass.rs:75-81)
    if (is_zero(r)) return double_affine(a);
    return XYZZ<F>::identity();
  }
  F ppp = mul(pp_, pp);
  F q = mul(p.x, pp);
  F x3 = vred(norm(sub<8, 3>(sqr(r), add(ppp, dbl(q)))));   // PPP + 2Q stays lazy: the 8p constant with 3 x 2^29 limbs dominates it
  F y3 = mul2sub(r, norm(sub<4, 1>(q, x3)), p.y, ppp);          // R*(Q - X3) - Y1*PPP with ONE reduction
  return {x3, y3, mul(p.zz, pp), mul(p.zzz, ppp)};
}

// P + (+-a): the bucket kernel's form.  The sign of a signed digit is folded into R = +-S2 - Y1 (a fat subtraction
// from zero: 8p - S2) instead of negating the point's y first (subtract, normalise, value-reduce: ~80 instructions that
// every lane of a wave pays as soon as one lane's digit is negative).
// (Base fields only: R reaches K = 12 in the negated case, which an Fq2 square's inner subtraction does not admit --
// the Fq2 overload below negates the point as before.)
template <class F>
KG_HD XYZZ<F> add_mixed_signed(const XYZZ<F>& p, const Affine<F>& a, bool negate) {
  if (is_identity(p)) return from_affine(negate ? neg_affine(a) : a);
  F u2 = mul(a.x, p.zz);
  F s2 = mul(a.y, p.zzz);
  F pp_ = norm(sub<8, 1>(u2, p.x));     // P = U2 - X1   (X1 < 6p: see X3 below)
  F t = negate ? sub<8, 1>(F::zero(), s2) : s2;   // S2 < 2.1p: the 4p constant does not dominate its top limb
  F r = norm(sub<4, 1>(t, p.y));        // R = +-S2 - Y1
  F pp = sqr(pp_);
  if (is_zero_2p(pp)) {                 // same x: doubling or inverse (weierstrass.rs:75-81)
    if (is_zero(r)) return double_affine(negate ? neg_affine(a) : a);
    return XYZZ<F>::identity();
  }
  F ppp = mul(pp_, pp);
  F q = mul(p.x, pp);
  // X3 = R^2 + 4p - (PPP + 2Q) < 6p is kept WITHOUT a value reduction: the accumulator's X only ever enters products and
  // the two fat subtractions of the next addition (8p constants), and the bucket array is value-reduced on export
  F x3 = norm(sub<4, 3>(sqr(r), add(ppp, dbl(q))));
  F y3 = mul2sub(r, norm(sub<8, 1>(q, x3)), p.y, ppp);          // R*(Q - X3) - Y1*PPP with ONE reduction
  return {x3, y3, mul(p.zz, pp), mul(p.zzz, ppp)};
}

template <class G>
KG_HD XYZZ<Fp2<G>> add_mixed_signed(const XYZZ<Fp2<G>>& p, const Affine<Fp2<G>>& a, bool negate) {
  return add_mixed(p, negate ? neg_affine(a) : a);
}

// P + Q  (add-2008-s)
template <class F>
KG_HD XYZZ<F> add_xyzz(const XYZZ<F>& p, const XYZZ<F>& q_) {
  if (is_identity(p)) return q_;
  if (is_identity(q_)) return p;
  F u1 = mul(p.x, q_.zz);
  F u2 = mul(q_.x, p.zz);
  F s1 = mul(p.y, q_.zzz);
  F s2 = mul(q_.y, p.zzz);
  F pp_ = norm(sub<4, 1>(u2, u1));
  F r = norm(sub<4, 1>(s2, s1));
  F pp = sqr(pp_);
  if (is_zero_2p(pp)) {                 // weierstrass.rs:114-120
    if (is_zero(r)) return double_xyzz(p);
    return XYZZ<F>::identity();
  }
  F ppp = mul(pp_, pp);
  F q = mul(u1, pp);
  F x3 = vred(norm(sub<8, 3>(sqr(r), add(ppp, dbl(q)))));   // PPP + 2Q stays lazy: the 8p constant with 3 x 2^29 limbs dominates it
  F y3 = mul2sub(r, norm(sub<4, 1>(q, x3)), s1, ppp);
  return {x3, y3, mul(mul(p.zz, q_.zz), pp), mul(mul(p.zzz, q_.zzz), ppp)};
}

// P + Q and 2P with the operands read coordinate by coordinate through accessors (x(), y(), zz(), zzz() return a field
// element) and the result written the same way (x(v), ...): the operations of add_xyzz / double_xyzz in an order that keeps at
// most six field elements live, so that a kernel built on it fits the registers a resident accumulation leaves free
// (k_halve: 96 VGPRs instead of 160, no scratch).  A coordinate that is needed twice is read twice -- the second read comes out of
// the cache.  ALIASING CONTRACT: `out` may be the same point as the FIRST operand p (the running sum living in its output slot:
// k_gather_sum, k_sum_tasks, k_hot_sum, k_hot_fold) and must not alias q.  It holds because every coordinate of p is read before
// the same coordinate of out is written -- the write order is zz, zzz, x, y, and y of p is read last, inside the expression that
// writes y; a reordering of these bodies must keep that (tests/host/hosttest.cpp mode 9 / 10 run both formulas in place, every
// branch, with the bound checker on).
#if defined(__HIP_DEVICE_COMPILE__)
#define KG_STREAM_FENCE() __asm__ volatile("" ::: "memory")     // keeps a later read of a coordinate from being merged with an earlier one
#else
#define KG_STREAM_FENCE() ((void)0)
#endif
template <class F, class A, class D>
KG_HD void double_xyzz_stream(const A& p, D& out) {      // p is not the identity
  F v, w;
  {
    const F u = norm(dbl(p.y()));
    v = sqr(u);
    w = mul(u, v);
  }
  KG_STREAM_FENCE();
  out.zz(mul(v, p.zz()));
  KG_STREAM_FENCE();
  out.zzz(mul(w, p.zzz()));
  KG_STREAM_FENCE();
  F s, m;
  {
    const F x = p.x();
    s = mul(x, v);
    const F xx = sqr(x);
    m = norm(add(dbl(xx), xx));
  }
  const F x3 = vred(norm(sub<4, 1>(sqr(m), norm(dbl(s)))));
  out.x(x3);
  KG_STREAM_FENCE();
  out.y(vred(norm(sub<4, 1>(mul(m, norm(sub<4, 1>(s, x3))), mul(w, p.y())))));
}
template <class F, class A, class D>
KG_HD void copy_xyzz_stream(const A& p, D& out) { out.x(p.x()); out.y(p.y()); out.zz(p.zz()); out.zzz(p.zzz()); }

template <class F, class A, class B, class D>
KG_HD void add_xyzz_stream(const A& p, const B& q_, D& out) {
  F u1, pp_, zz12;
  {
    const F zz1 = p.zz(), zz2 = q_.zz();
    if (is_zero_2p(zz1)) { copy_xyzz_stream<F>(q_, out); return; }
    if (is_zero_2p(zz2)) { copy_xyzz_stream<F>(p, out); return; }
    u1 = mul(p.x(), zz2);
    KG_STREAM_FENCE();
    pp_ = norm(sub<4, 1>(mul(q_.x(), zz1), u1));
    zz12 = mul(zz1, zz2);
  }
  KG_STREAM_FENCE();
  F s1, r;
  {
    s1 = mul(p.y(), q_.zzz());
    KG_STREAM_FENCE();
    r = norm(sub<4, 1>(mul(q_.y(), p.zzz()), s1));
  }
  KG_STREAM_FENCE();
  const F pp = sqr(pp_);
  if (is_zero_2p(pp)) {                 // weierstrass.rs:114-120
    if (is_zero(r)) { double_xyzz_stream<F>(p, out); return; }
    const F z = F::zero();
    out.x(z); out.y(z); out.zz(z); out.zzz(z);
    return;
  }
  const F ppp = mul(pp_, pp);
  const F q = mul(u1, pp);
  out.zz(mul(zz12, pp));
  KG_STREAM_FENCE();
  out.zzz(mul(mul(p.zzz(), q_.zzz()), ppp));
  KG_STREAM_FENCE();
  const F x3 = vred(norm(sub<8, 3>(sqr(r), add(ppp, dbl(q)))));   // PPP + 2Q stays lazy, as in add_xyzz
  out.x(x3);
  out.y(mul2sub(r, norm(sub<4, 1>(q, x3)), s1, ppp));
}

// -P
template <class F>
KG_HD XYZZ<F> neg_xyzz(const XYZZ<F>& p) { return {p.x, vred(norm(sub<4, 1>(F::zero(), p.y))), p.zz, p.zzz}; }
template <class F>
KG_HD Affine<F> neg_affine(const Affine<F>& a) { return {a.x, vred(norm(sub<4, 1>(F::zero(), a.y)))}; }

// XYZZ -> affine (x = X/ZZ, y = Y/ZZZ); returns false for the identity.  One field inversion
// (the reference's to_affine, macros/curve/weierstrass.rs:57-66, does the same with z^-1).
template <class F>
KG_HD bool to_affine(const XYZZ<F>& p, Affine<F>& out) {
  if (is_identity(p)) return false;
  F zi = inv_fast(p.zzz);               // ZZZ^-1 (binary GCD, fp_inv.h)
  F zzi = mul(mul(zi, zi), sqr(p.zz));  // ZZ^-1 = ZZZ^-2 * ZZ^2   (ZZ^3 = ZZZ^2)
  out.x = mul(p.x, zzi);
  out.y = mul(p.y, zi);
  return true;
}

}  // namespace kg

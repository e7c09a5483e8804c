// fp29_checked.h -- HOST-ONLY shadow of Fp<P> that carries worst-case bounds with every value and aborts
// when a precondition of fp29.h is violated.  The curve formulas and kernel bodies are templates over the
// field type, so instantiating them with FpChecked proves (by interval arithmetic over the executed path,
// independent of the actual operand values) that no 64-bit column accumulator, 32-bit limb or Montgomery
// value bound can overflow on the device.  Used only by tests/host/.
#pragma once
#include <cstdio>
#include <cstdlib>
#include "fp29.h"

namespace kg {

struct BoundFail {
  static void fail(const char* what, double a, double b) {
    std::fprintf(stderr, "FpChecked: bound violated: %s (%.6g vs %.6g)\n", what, a, b);
    std::abort();
  }
};

template <class P>
struct FpChecked {
  using Params = P;
  Fp<P> v;
  double lb;  // bound on limbs 0..7 (inclusive)
  double tb;  // bound on the top limb (inclusive)
  double kb;  // value < kb * p

  static constexpr double RHO() { return 169.0; }  // floor(2^261 / p) for both BN254 primes is 169
  static double ptop1() { return (double)P::PTOP + 1.0; }

  static FpChecked zero() { return {Fp<P>::zero(), 0, 0, 0}; }
  static FpChecked one() { return {Fp<P>::one(), (double)M29, (double)P::ONE[8], 1.0}; }
  template <class A>
  static FpChecked from_const(const A& c) { return {Fp<P>::from_const(c), (double)M29, (double)c[8], 1.0}; }
  // wrap a raw value the caller vouches for: normalised limbs, value < k*p
  static FpChecked wrap(const Fp<P>& x, double k) { return {x, (double)M29, k * ptop1(), k}; }

  void check_actual() const {
    for (int i = 0; i < 8; ++i)
      if ((double)v.l[i] > lb) BoundFail::fail("actual limb above tracked bound", v.l[i], lb);
    if ((double)v.l[8] > tb) BoundFail::fail("actual top limb above tracked bound", v.l[8], tb);
  }
};

namespace chk {
constexpr double TWO64 = 18446744073709551616.0;
template <class P>
inline void col(double la, double ta, double lb, double tb, double extra = 0) {
  // worst column: 9 products of the largest limbs + 9 m*p products + incoming carry
  double A = la > ta ? la : ta, B = lb > tb ? lb : tb;
  double s = 9.0 * A * B + extra + 9.0 * (double)M29 * (double)M29 + 68719476736.0;
  if (!(s < TWO64)) BoundFail::fail("64-bit column accumulator", s, TWO64);
}
}  // namespace chk

template <class P>
inline FpChecked<P> mul(const FpChecked<P>& a, const FpChecked<P>& b) {
  a.check_actual(); b.check_actual();
  chk::col<P>(a.lb, a.tb, b.lb, b.tb);
  double kk = a.kb * b.kb;
  if (!(kk < FpChecked<P>::RHO())) BoundFail::fail("mul value bound Ka*Kb", kk, FpChecked<P>::RHO());
  double ko = kk / FpChecked<P>::RHO() + 1.0;
  FpChecked<P> r{mul(a.v, b.v), (double)M29, ko * FpChecked<P>::ptop1(), ko};
  r.check_actual();
  return r;
}
template <class P>
inline FpChecked<P> mulc(const FpChecked<P>& a, const FpConst<P>& c) {
  a.check_actual();
  const double A = a.lb > a.tb ? a.lb : a.tb;
  const double s = 9.0 * A * (double)M29 + 9.0 * (double)M29 * (double)M29 + 68719476736.0;     // worst column of either phase
  if (!(s < chk::TWO64)) BoundFail::fail("mulc 64-bit column accumulator", s, chk::TWO64);
  if (!(a.kb < FpChecked<P>::RHO())) BoundFail::fail("mulc operand above 2^261", a.kb, FpChecked<P>::RHO());
  for (int i = 0; i < 9; ++i)
    if (c.w[i] > M29 || (i < 8 && c.q[i] > M29)) BoundFail::fail("mulc constant limbs not normalised", c.w[i], (double)M29);
  const double ko = 2.0 + a.kb / FpChecked<P>::RHO();
  FpChecked<P> r{mulc(a.v, c), (double)M29, ko * FpChecked<P>::ptop1(), ko};
  r.check_actual();
  return r;
}
template <class P>
inline FpChecked<P> sqr(const FpChecked<P>& a) {
  a.check_actual();
  if (!(2.0 * a.lb < 4294967296.0 && 2.0 * a.tb < 4294967296.0)) BoundFail::fail("sqr doubled limb", 2 * a.lb, 4294967296.0);
  chk::col<P>(a.lb, a.tb, a.lb, a.tb);
  double kk = a.kb * a.kb;
  if (!(kk < FpChecked<P>::RHO())) BoundFail::fail("sqr value bound Ka^2", kk, FpChecked<P>::RHO());
  double ko = kk / FpChecked<P>::RHO() + 1.0;
  FpChecked<P> r{sqr(a.v), (double)M29, ko * FpChecked<P>::ptop1(), ko};
  r.check_actual();
  return r;
}
template <class P>
inline FpChecked<P> mul2add(const FpChecked<P>& a, const FpChecked<P>& b, const FpChecked<P>& c, const FpChecked<P>& d) {
  a.check_actual(); b.check_actual(); c.check_actual(); d.check_actual();
  double C = c.lb > c.tb ? c.lb : c.tb, D = d.lb > d.tb ? d.lb : d.tb;
  chk::col<P>(a.lb, a.tb, b.lb, b.tb, 9.0 * C * D);
  double kk = a.kb * b.kb + c.kb * d.kb;
  if (!(kk < FpChecked<P>::RHO())) BoundFail::fail("mul2add value bound", kk, FpChecked<P>::RHO());
  double ko = kk / FpChecked<P>::RHO() + 1.0;
  FpChecked<P> r{mul2add(a.v, b.v, c.v, d.v), (double)M29, ko * FpChecked<P>::ptop1(), ko};
  r.check_actual();
  return r;
}
template <class P>
inline FpChecked<P> mul2sub(const FpChecked<P>& a, const FpChecked<P>& b, const FpChecked<P>& c, const FpChecked<P>& d) {
  a.check_actual(); b.check_actual(); c.check_actual(); d.check_actual();
  double A = a.lb > a.tb ? a.lb : a.tb, B = b.lb > b.tb ? b.lb : b.tb;
  double C = c.lb > c.tb ? c.lb : c.tb, D = d.lb > d.tb ? d.lb : d.tb;
  const double TWO63 = 9223372036854775808.0, TWO31 = 2147483648.0;
  if (!(A < TWO31 && B < TWO31 && C < TWO31 && D < TWO31)) BoundFail::fail("mul2sub limb above 2^31", A, TWO31);
  double pos = 9.0 * A * B + 9.0 * (double)M29 * (double)M29 + 68719476736.0, negs = 9.0 * C * D + 68719476736.0;
  if (!(pos < TWO63 && negs < TWO63)) BoundFail::fail("mul2sub signed column", pos > negs ? pos : negs, TWO63);
  if (!(a.kb * b.kb < FpChecked<P>::RHO() && c.kb * d.kb < FpChecked<P>::RHO())) BoundFail::fail("mul2sub value bound", a.kb * b.kb, c.kb * d.kb);
  double ko = a.kb * b.kb / FpChecked<P>::RHO() + 1.0;
  // a negative intermediate gets +p and lands in (0, p); a non-negative one is below ko*p
  FpChecked<P> r{mul2sub(a.v, b.v, c.v, d.v), (double)M29, ko * FpChecked<P>::ptop1(), ko};
  r.check_actual();
  return r;
}
template <class P>
inline FpChecked<P> mul2pm(const FpChecked<P>& a, const FpChecked<P>& b, const FpChecked<P>& c, const FpChecked<P>& d, bool negate) {
  a.check_actual(); b.check_actual(); c.check_actual(); d.check_actual();
  double A = a.lb > a.tb ? a.lb : a.tb, B = b.lb > b.tb ? b.lb : b.tb;
  double C = c.lb > c.tb ? c.lb : c.tb, D = d.lb > d.tb ? d.lb : d.tb;
  const double TWO63 = 9223372036854775808.0, TWO31 = 2147483648.0;
  if (!(A < TWO31 && B < TWO31 && C < TWO31 && D < TWO31)) BoundFail::fail("mul2pm limb above 2^31", A, TWO31);
  // either sign must be safe: the sign is data on the device (both products may add up in a column)
  double pos = 9.0 * (A * B + C * D) + 9.0 * (double)M29 * (double)M29 + 68719476736.0;
  if (!(pos < TWO63)) BoundFail::fail("mul2pm signed column", pos, TWO63);
  double kk = a.kb * b.kb + c.kb * d.kb;
  if (!(kk < FpChecked<P>::RHO())) BoundFail::fail("mul2pm value bound", kk, FpChecked<P>::RHO());
  double ko = kk / FpChecked<P>::RHO() + 1.0;
  FpChecked<P> r{mul2pm(a.v, b.v, c.v, d.v, negate), (double)M29, ko * FpChecked<P>::ptop1(), ko};
  r.check_actual();
  return r;
}
template <class P>
inline FpChecked<P> vred(const FpChecked<P>& a) {
  a.check_actual();
  if (!(a.lb <= (double)M29)) BoundFail::fail("vred input limbs not normalised", a.lb, (double)M29);
  if (!(a.tb < 134217728.0)) BoundFail::fail("vred top limb above 2^27", a.tb, 134217728.0);
  FpChecked<P> r{vred(a.v), (double)M29, 1.06 * FpChecked<P>::ptop1(), 1.06};
  r.check_actual();
  return r;
}
template <class P>
inline FpChecked<P> add(const FpChecked<P>& a, const FpChecked<P>& b) {
  FpChecked<P> r{add(a.v, b.v), a.lb + b.lb, a.tb + b.tb, a.kb + b.kb};
  if (!(r.lb < 4294967296.0 && r.tb < 4294967296.0)) BoundFail::fail("add limb overflow", r.lb, 4294967296.0);
  return r;
}
template <class P>
inline FpChecked<P> dbl(const FpChecked<P>& a) { return add(a, a); }
template <int C, int T, class P>
inline FpChecked<P> sub(const FpChecked<P>& a, const FpChecked<P>& b) {
  b.check_actual();
  double dom = (double)T * 536870912.0 - (double)T;
  if (!(b.lb <= dom)) BoundFail::fail("sub: limbs of b not dominated by the fat constant", b.lb, dom);
  double ztop = (double)FatZ<P, C, T>::at(8);
  if (!(b.tb <= ztop)) BoundFail::fail("sub: top limb of b not dominated", b.tb, ztop);
  double zmax = 0;
  for (int i = 0; i < 8; ++i) zmax = zmax > (double)FatZ<P, C, T>::at(i) ? zmax : (double)FatZ<P, C, T>::at(i);
  FpChecked<P> r{sub<C, T>(a.v, b.v), a.lb + zmax, a.tb + ztop, a.kb + (double)C};
  if (!(r.lb < 4294967296.0 && r.tb < 4294967296.0)) BoundFail::fail("sub limb overflow", r.lb, 4294967296.0);
  return r;
}
template <class P>
inline FpChecked<P> norm(const FpChecked<P>& a) {
  if (!(a.lb + a.lb / 536870912.0 + 1 < 4294967296.0)) BoundFail::fail("norm carry overflow", a.lb, 4294967296.0);
  double t = a.kb * FpChecked<P>::ptop1();
  double t2 = a.tb + a.lb / 536870912.0 + 1;
  FpChecked<P> r{norm(a.v), (double)M29, t < t2 ? t : t2, a.kb};
  r.check_actual();
  return r;
}
template <class P>
inline FpChecked<P> reduce_2p(const FpChecked<P>& a) {
  if (!(a.kb <= 2.0 + 1e-9 && a.lb <= (double)M29)) BoundFail::fail("reduce_2p input not in [0,2p) normalised", a.kb, 2.0);
  return {reduce_2p(a.v), (double)M29, FpChecked<P>::ptop1(), 1.0};
}
template <class P>
inline FpChecked<P> reduce(const FpChecked<P>& a) { return reduce_2p(mul(a, FpChecked<P>::one())); }
template <class P>
inline bool is_zero_2p(const FpChecked<P>& a) {
  if (!(a.kb <= 2.0 + 1e-9 && a.lb <= (double)M29)) BoundFail::fail("is_zero_2p input not in [0,2p) normalised", a.kb, 2.0);
  return is_zero_2p(a.v);
}
template <class P>
inline bool is_zero(const FpChecked<P>& a) { return is_zero_2p(mul(a, FpChecked<P>::one())); }
template <class P>
inline FpChecked<P> inv(const FpChecked<P>& a) {
  (void)mul(a, a);  // inv() squares and multiplies by a: same precondition
  return {inv(a.v), (double)M29, 2.0 * FpChecked<P>::ptop1(), 2.0};
}

template <class P>
inline FpChecked<P> inv_fast(const FpChecked<P>& a) { return inv(a); }      // same value; the binary GCD has no column bounds to track

using FqC = FpChecked<FqParams>;
using FrC = FpChecked<FrParams>;
using Fq2C = Fp2<FqC>;

}  // namespace kg

// host_fp.h -- host-side field type for the short sequential tails that stay on the CPU (the final
// double-and-add over <= 272 window/bit partial sums of an MSM, and the one field inversion of to_affine,
// exactly where the reference also runs sequential code: groth16/src/msm.rs:41-47, macros/curve/weierstrass.rs:57-66).
// A GPU lane needs ~0.5 us per dependent Montgomery product; a host core needs ~25 ns, so a 500-step
// dependent chain belongs here.
//
// HostFp<P> is 4 x u64 Montgomery (R = 2^256), always fully reduced -- the ABI representation itself -- and
// implements the same free-function interface as kg::Fp (fp29.h) so that curve.h's templates are reused
// unchanged on the host.  It is part of the product (not of oracle/).
#pragma once
#include <cstdint>
#include <cstring>
#include "fp29.h"

namespace kg {

struct HostFqP {
  static constexpr uint64_t P[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
  static constexpr uint64_t INV = 0x87d20782e4866389ULL;
  static constexpr uint64_t ONE[4] = {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL};
};
struct HostFrP {
  static constexpr uint64_t P[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
  static constexpr uint64_t INV = 0xc2e1f593efffffffULL;
  static constexpr uint64_t ONE[4] = {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL};
};

template <class P>
struct HostFp {
  using Params = P;
  uint64_t v[4];
  static HostFp zero() { return {{0, 0, 0, 0}}; }
  static HostFp one() { return {{P::ONE[0], P::ONE[1], P::ONE[2], P::ONE[3]}}; }
  static HostFp from_words(const uint64_t* w) { return {{w[0], w[1], w[2], w[3]}}; }
  void to_words(uint64_t* w) const { std::memcpy(w, v, 32); }
};

namespace hostfp {
typedef unsigned __int128 u128;
template <class P>
inline bool geq_p(const uint64_t a[4]) {
  for (int i = 3; i >= 0; --i) {
    if (a[i] != P::P[i]) return a[i] > P::P[i];
  }
  return true;
}
template <class P>
inline void sub_p(uint64_t a[4]) {
  u128 b = 0;
  for (int i = 0; i < 4; ++i) {
    u128 d = (u128)a[i] - P::P[i] - (uint64_t)b;
    a[i] = (uint64_t)d;
    b = (d >> 64) & 1;
  }
}
}  // namespace hostfp

template <class P>
inline HostFp<P> mul(const HostFp<P>& a, const HostFp<P>& b) {
  using hostfp::u128;
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) {
    uint64_t c = 0;
    for (int j = 0; j < 4; ++j) {
      u128 s = (u128)a.v[i] * b.v[j] + t[j] + c;
      t[j] = (uint64_t)s;
      c = (uint64_t)(s >> 64);
    }
    u128 s = (u128)t[4] + c;
    t[4] = (uint64_t)s;
    t[5] = (uint64_t)(s >> 64);
    uint64_t m = t[0] * P::INV;
    s = (u128)m * P::P[0] + t[0];
    c = (uint64_t)(s >> 64);
    for (int j = 1; j < 4; ++j) {
      s = (u128)m * P::P[j] + t[j] + c;
      t[j - 1] = (uint64_t)s;
      c = (uint64_t)(s >> 64);
    }
    s = (u128)t[4] + c;
    t[3] = (uint64_t)s;
    t[4] = t[5] + (uint64_t)(s >> 64);
  }
  HostFp<P> r{{t[0], t[1], t[2], t[3]}};
  if (t[4] || hostfp::geq_p<P>(r.v)) hostfp::sub_p<P>(r.v);
  return r;
}
template <class P>
inline HostFp<P> sqr(const HostFp<P>& a) { return mul(a, a); }
template <class P>
inline HostFp<P> add(const HostFp<P>& a, const HostFp<P>& b) {
  using hostfp::u128;
  HostFp<P> r;
  uint64_t c = 0;
  for (int i = 0; i < 4; ++i) {
    u128 s = (u128)a.v[i] + b.v[i] + c;
    r.v[i] = (uint64_t)s;
    c = (uint64_t)(s >> 64);
  }
  if (c || hostfp::geq_p<P>(r.v)) hostfp::sub_p<P>(r.v);
  return r;
}
template <class P>
inline HostFp<P> dbl(const HostFp<P>& a) { return add(a, a); }
template <int C, int T, class P>
inline HostFp<P> sub(const HostFp<P>& a, const HostFp<P>& b) {
  using hostfp::u128;
  HostFp<P> r;
  uint64_t br = 0;
  for (int i = 0; i < 4; ++i) {
    u128 d = (u128)a.v[i] - b.v[i] - br;
    r.v[i] = (uint64_t)d;
    br = (uint64_t)(d >> 64) & 1;
  }
  if (br) {
    uint64_t c = 0;
    for (int i = 0; i < 4; ++i) {
      u128 s = (u128)r.v[i] + P::P[i] + c;
      r.v[i] = (uint64_t)s;
      c = (uint64_t)(s >> 64);
    }
  }
  return r;
}
template <class P>
inline HostFp<P> norm(const HostFp<P>& a) { return a; }
template <class P>
inline HostFp<P> vred(const HostFp<P>& a) { return a; }
template <class P>
inline HostFp<P> reduce(const HostFp<P>& a) { return a; }
template <class P>
inline bool is_zero_2p(const HostFp<P>& a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) == 0; }
template <class P>
inline bool is_zero(const HostFp<P>& a) { return is_zero_2p(a); }
template <class P>
inline HostFp<P> mul2add(const HostFp<P>& a, const HostFp<P>& b, const HostFp<P>& c, const HostFp<P>& d) {
  return add(mul(a, b), mul(c, d));
}
template <class P>
inline HostFp<P> mul2sub(const HostFp<P>& a, const HostFp<P>& b, const HostFp<P>& c, const HostFp<P>& d) {
  return sub<4, 1>(mul(a, b), mul(c, d));
}
template <class P>
inline HostFp<P> mul2pm(const HostFp<P>& a, const HostFp<P>& b, const HostFp<P>& c, const HostFp<P>& d, bool negate) {
  return negate ? mul2sub(a, b, c, d) : mul2add(a, b, c, d);
}
// a^(p-2) (zkstd normal.rs:256-270); 0 for a == 0
template <class P>
inline HostFp<P> inv(const HostFp<P>& a) {
  uint64_t e[4] = {P::P[0] - 2, P::P[1], P::P[2], P::P[3]};
  HostFp<P> r = HostFp<P>::one();
  for (int i = 255; i >= 0; --i) {
    r = sqr(r);
    if ((e[i >> 6] >> (i & 63)) & 1) r = mul(r, a);
  }
  return r;
}
template <class P>
inline HostFp<P> inv_fast(const HostFp<P>& a) { return inv(a); }          // host finish: one inversion per MSM, the 4 x 64-bit ladder is fine

using HostFq = HostFp<HostFqP>;
using HostFr = HostFp<HostFrP>;
using HostFq2 = Fp2<HostFq>;

}  // namespace kg

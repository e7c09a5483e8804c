// coop_add.h -- one XYZZ point addition (or doubling) computed by FOUR lanes (a quad): operands and results in an LDS image, intermediate values in registers.
//
// Why: the reduction trees of an MSM (partial sums of a bucket, the halving levels, the window sum) are chains of DEPENDENT point
// additions run by few lanes on an otherwise idle CU.  A wave issues ~one VALU instruction every 5 cycles whatever the number of active
// lanes, so a lane that computes the 14 products of add-2008-s one after the other needs 8.5 us per level (measured, one wave per SIMD:
// profiles/r06_small_stamps.txt) while 60 lanes of its wave idle.  The products of one addition are mostly independent:
//
//   step 1   U1 = X1 ZZ2        U2 = X2 ZZ1        S1 = Y1 ZZZ2       S2 = Y2 ZZZ1
//   step 2   PP = (U2 - U1)^2   RR = (S2 - S1)^2   ZZ12 = ZZ1 ZZ2     ZZZ12 = ZZZ1 ZZZ2
//   step 3   PPP = P PP         Q = U1 PP          ZZ3 = ZZ12 PP      --
//   step 4   X3 = RR - PPP - 2Q, Y3 = R (Q - X3) - S1 PPP             ZZZ3 = ZZZ12 PPP
//
// so lane j of a quad computes column j: four steps of one product each (the last one a double product) instead of fourteen.  A doubling
// (dbl-2008-s-1) is three steps.  SIMT: the lanes of a wave share one instruction stream, so a step is written as ONE product whose
// operands every lane picks by ADDRESS (per-lane element references into the image), with the cheap linear pre-operations (a lazy
// subtraction, a doubling) computed by all lanes and selected -- a `switch (lane)` around four different products would run them one after
// the other.  Between the steps only the quad's own lanes exchange values: each lane keeps what it computed in registers and the next step's
// operands arrive by DPP quad_perm (one v_mov per word; coop_a1 .. a4, coop_d1 .. d3 and the exchange schedule further down) -- a level ends
// with ONE workgroup barrier, for the readers of its results.  (The first form kept ten temporaries per quad in LDS and ended every step with a
// barrier: coop_add_s1 .. s4, still what KG_COOP_LDS builds and two of the host test modes run; the register form took the short-input kernel
// from 217 to 149 VGPRs.)
//
// The exceptional cases of the reference's formulas (zkstd/src/arithmetic/points/weierstrass.rs:102-123: identity operands, equal x =>
// doubling or the identity) are kept: identity operands turn the addition into a copy, PP = 0 sends lane 0 of the quad through the one-lane
// formulas (rare: equal or inverse points).
//
// The arithmetic is add_xyzz / double_xyzz of curve.h product for product, with the same lazy subtractions, so the value bounds are the
// ones tests/host/ checks there -- and this file's own sequences run under the bound-tracking type as well (hosttest.cpp modes 12, 13: the
// step functions are plain host / device code over an abstract quad).
//
// Image layout: structure of arrays -- word k of coordinate cd (x, y, zz, zzz) of item q at img[(cd * E + k) * cap + q], E words per field
// element; temporaries: slot s of quad t at tmp[(s * E + k) * tcap + t], a flag word per quad at flg[t].
#pragma once
#include "curve.h"

namespace kg {

template <class F> struct CoopEl;                  // one field element of an image
template <class P> struct CoopEl<Fp<P>> {
  static constexpr uint32_t E = 9;
  static KG_HD Fp<P> ld(const uint32_t* base, uint32_t stride) {
    Fp<P> r;
#pragma unroll
    for (uint32_t k = 0; k < 9; ++k) r.l[k] = base[k * stride];
    return r;
  }
  static KG_HD void st(uint32_t* base, uint32_t stride, const Fp<P>& v) {
#pragma unroll
    for (uint32_t k = 0; k < 9; ++k) base[k * stride] = v.l[k];
  }
  static KG_HD Fp<P> pick(bool first, const Fp<P>& a, const Fp<P>& b) {
    Fp<P> r;
#pragma unroll
    for (int k = 0; k < 9; ++k) r.l[k] = first ? a.l[k] : b.l[k];
    return r;
  }
};
#if defined(__HIP_DEVICE_COMPILE__)
// lane i of every quad of the wave receives the element held by lane PERM[i] of its quad (DPP quad_perm: one v_mov per word, no LDS)
template <int PERM, class P>
__device__ __forceinline__ Fp<P> coop_xchg(const Fp<P>& v) {
  Fp<P> r;
#pragma unroll
  for (int k = 0; k < 9; ++k) r.l[k] = (uint32_t)__builtin_amdgcn_mov_dpp((int)v.l[k], PERM, 0xf, 0xf, true);
  return r;
}
template <int PERM, class G>
__device__ __forceinline__ Fp2<G> coop_xchg(const Fp2<G>& v) { return {coop_xchg<PERM>(v.c0), coop_xchg<PERM>(v.c1)}; }
#endif
constexpr int coop_perm(int p0, int p1, int p2, int p3) { return p0 | (p1 << 2) | (p2 << 4) | (p3 << 6); }

template <class G> struct CoopEl<Fp2<G>> {
  static constexpr uint32_t E = 18;
  static KG_HD Fp2<G> ld(const uint32_t* base, uint32_t stride) { return {CoopEl<G>::ld(base, stride), CoopEl<G>::ld(base + 9 * stride, stride)}; }
  static KG_HD void st(uint32_t* base, uint32_t stride, const Fp2<G>& v) {
    CoopEl<G>::st(base, stride, v.c0);
    CoopEl<G>::st(base + 9 * stride, stride, v.c1);
  }
  static KG_HD Fp2<G> pick(bool first, const Fp2<G>& a, const Fp2<G>& b) { return {CoopEl<G>::pick(first, a.c0, b.c0), CoopEl<G>::pick(first, a.c1, b.c1)}; }
};

constexpr uint32_t COOP_TMP_SLOTS = 10;            // temporaries per quad (the LDS form of the steps: KG_COOP_LDS builds, tests/host)
// LDS words a kernel sets aside for the temporaries and flags of nquads quads: none in the register form
template <class F> constexpr uint32_t coop_lds_words(uint32_t nquads) {
#ifdef KG_COOP_LDS
  return (COOP_TMP_SLOTS * CoopEl<F>::E + 1) * nquads;
#else
  return 0 * nquads;
#endif
}

// What one lane of a quad works on.  ia, ib: operand items; io: result item (may be ia or ib: every read of an operand coordinate comes
// before the step that writes the same coordinate of the result).  Ref: where an element lives (first word, stride between its words).
template <class F>
struct CoopQuad {
  using Field = F;
  struct Ref { uint32_t* p; uint32_t stride; };
  uint32_t* img; uint32_t cap;
  uint32_t* tmp; uint32_t tcap, tq;
  uint32_t* flg;
  uint32_t ia, ib, io;
  int lane;                                        // 0 .. 3
  static constexpr uint32_t E = CoopEl<F>::E;
  KG_HD Ref coord(uint32_t item, uint32_t cd) const { return {img + (size_t)cd * E * cap + item, cap}; }
  KG_HD Ref t(uint32_t s) const { return {tmp + (size_t)s * E * tcap + tq, tcap}; }
  KG_HD F ld(const Ref& r) const { return CoopEl<F>::ld(r.p, r.stride); }
  KG_HD void st(const Ref& r, const F& v) const { CoopEl<F>::st(r.p, r.stride, v); }
  KG_HD uint32_t flag() const { return flg[tq]; }
  KG_HD void set_flag(uint32_t v) const { flg[tq] = v; }
  static KG_HD F pick(bool first, const F& a, const F& b) { return CoopEl<F>::pick(first, a, b); }
};
template <class Q>
KG_HD XYZZ<typename Q::Field> coop_point(const Q& q, uint32_t item) {
  return {q.ld(q.coord(item, 0)), q.ld(q.coord(item, 1)), q.ld(q.coord(item, 2)), q.ld(q.coord(item, 3))};
}
template <class Q>
KG_HD void coop_set_point(const Q& q, uint32_t item, const XYZZ<typename Q::Field>& p) {
  q.st(q.coord(item, 0), p.x); q.st(q.coord(item, 1), p.y); q.st(q.coord(item, 2), p.zz); q.st(q.coord(item, 3), p.zzz);
}

enum { COOP_ADD = 0, COOP_TAKE_B = 1, COOP_TAKE_A = 2, COOP_IDLE = 3 };

// which case the quad's addition is (every lane of the quad computes the same answer from the operands' ZZ)
template <class Q>
KG_HD int coop_add_mode(const Q& q) {
  if (is_zero_2p(q.ld(q.coord(q.ia, 2)))) return COOP_TAKE_B;
  if (is_zero_2p(q.ld(q.coord(q.ib, 2)))) return COOP_TAKE_A;
  return COOP_ADD;
}
// temporaries: 0 U1, 1 U2 (then PPP), 2 S1, 3 S2 (then Q), 4 PP, 5 P, 6 RR, 7 R, 8 ZZ12, 9 ZZZ12
template <class Q>
KG_HD void coop_add_s1(const Q& q) {
  const int l = q.lane;
  // lane 0: X1 ZZ2 -> U1   lane 1: X2 ZZ1 -> U2   lane 2: Y1 ZZZ2 -> S1   lane 3: Y2 ZZZ1 -> S2
  const uint32_t first = (l & 1) ? q.ib : q.ia, second = (l & 1) ? q.ia : q.ib;
  const uint32_t cd = (uint32_t)(l >> 1);            // x for lanes 0, 1; y for lanes 2, 3
  q.st(q.t((uint32_t)l), mul(q.ld(q.coord(first, cd)), q.ld(q.coord(second, 2 + cd))));
}
template <class Q>
KG_HD void coop_add_s2(const Q& q) {
  using F = typename Q::Field;
  const int l = q.lane;
  // lanes 0, 1: d = T[2l+1] - T[2l] (P, R), kept in T5 / T7, its square -> T4 / T6; lanes 2, 3: ZZ1 ZZ2 -> T8, ZZZ1 ZZZ2 -> T9
  const bool diff = l < 2;
  const F a = q.ld(diff ? q.t((uint32_t)(2 * l + 1)) : q.coord(q.ia, (uint32_t)l));
  const F b = q.ld(diff ? q.t((uint32_t)(2 * l)) : q.coord(q.ib, (uint32_t)l));
  const F d = norm(sub<4, 1>(a, b));
  if (diff) q.st(q.t((uint32_t)(5 + 2 * l)), d);
  const F r = mul(Q::pick(diff, d, a), Q::pick(diff, d, b));
  q.st(q.t(diff ? (uint32_t)(4 + 2 * l) : (uint32_t)(6 + l)), r);
  if (l == 0) q.set_flag(is_zero_2p(r) ? 1u : 0u);   // PP = 0: equal x -- doubling or the identity (step 3, lane 0)
}
// step 3; also the copies of an addition with an identity operand (mode TAKE_A / TAKE_B: lane j moves coordinate j)
template <class Q>
KG_HD void coop_add_s3(const Q& q, int mode) {
  using F = typename Q::Field;
  if (mode == COOP_TAKE_A || mode == COOP_TAKE_B) {
    const uint32_t src = mode == COOP_TAKE_A ? q.ia : q.ib;
    if (src != q.io) q.st(q.coord(q.io, (uint32_t)q.lane), q.ld(q.coord(src, (uint32_t)q.lane)));
    return;
  }
  if (mode != COOP_ADD) return;
  if (q.flag()) {                                    // weierstrass.rs:114-120 -- the one-lane formulas
    if (q.lane == 0) {
      if (is_zero(q.ld(q.t(7)))) coop_set_point(q, q.io, double_xyzz(coop_point(q, q.ia)));
      else coop_set_point(q, q.io, XYZZ<F>::identity());
    }
    return;
  }
  const int l = q.lane;
  if (l == 3) return;
  // lane 0: P PP -> T1 (PPP; U2 is dead)   lane 1: U1 PP -> T3 (Q; S2 is dead)   lane 2: ZZ12 PP -> ZZ3
  const F a = q.ld(q.t(l == 0 ? 5u : (l == 1 ? 0u : 8u)));
  const F r = mul(a, q.ld(q.t(4)));
  q.st(l == 2 ? q.coord(q.io, 2) : q.t((uint32_t)(1 + 2 * l)), r);
}
template <class Q>
KG_HD void coop_add_s4(const Q& q, int mode) {
  using F = typename Q::Field;
  if (mode != COOP_ADD || q.flag() || q.lane >= 2) return;
  const bool l0 = q.lane == 0;
  // lane 0: X3 = RR - PPP - 2Q, Y3 = R (Q - X3) - S1 PPP;   lane 1: ZZZ3 = ZZZ12 PPP, as the same double product with a zero subtrahend
  const F ppp = q.ld(q.t(1)), qq = q.ld(q.t(3));
  const F x3 = vred(norm(sub<8, 3>(q.ld(q.t(6)), add(ppp, dbl(qq)))));          // PPP + 2Q stays lazy, as in add_xyzz
  const F a = q.ld(q.t(l0 ? 7u : 9u));
  const F b = Q::pick(l0, norm(sub<4, 1>(qq, x3)), ppp);
  const F c = Q::pick(l0, q.ld(q.t(2)), F::zero());
  const F r = mul2sub(a, b, c, ppp);
  if (l0) q.st(q.coord(q.io, 0), x3);
  q.st(q.coord(q.io, l0 ? 1u : 3u), r);
}

// ---- doubling in place (item ia = io), three steps; a quad whose point is the identity idles.  Temporaries: 0 V, 1 XX, 2 W, 3 S, 4 MM
template <class Q>
KG_HD bool coop_dbl_active(const Q& q) { return !is_zero_2p(q.ld(q.coord(q.ia, 2))); }
template <class Q>
KG_HD void coop_dbl_s1(const Q& q) {
  using F = typename Q::Field;
  if (q.lane >= 2) return;
  const bool l0 = q.lane == 0;
  const F a = q.ld(q.coord(q.ia, l0 ? 1u : 0u));     // lane 0: V = (2Y)^2   lane 1: XX = X^2
  const F u = Q::pick(l0, norm(dbl(a)), a);
  q.st(q.t(l0 ? 0u : 1u), mul(u, u));
}
template <class Q>
KG_HD void coop_dbl_s2(const Q& q) {
  using F = typename Q::Field;
  const int l = q.lane;
  // lane 0: (2Y) V -> W (T2)   lane 1: X V -> S (T3)   lane 2: V ZZ -> ZZ3   lane 3: (3 XX)^2 -> MM (T4)
  const F a = q.ld(l == 0 ? q.coord(q.ia, 1) : (l == 1 ? q.coord(q.ia, 0) : (l == 2 ? q.coord(q.ia, 2) : q.t(1))));
  const F v = q.ld(q.t(0));
  const F two = norm(dbl(a)), three = norm(add(dbl(a), a));
  const F u = Q::pick(l == 0, two, Q::pick(l == 3, three, a));
  const F r = mul(u, Q::pick(l == 3, three, v));
  q.st(l == 2 ? q.coord(q.io, 2) : q.t(l == 3 ? 4u : (uint32_t)(2 + l)), r);
}
template <class Q>
KG_HD void coop_dbl_s3(const Q& q) {
  using F = typename Q::Field;
  if (q.lane >= 2) return;
  const bool l0 = q.lane == 0;
  // lane 0: X3 = MM - 2S, Y3 = M (S - X3) - W Y;   lane 1: ZZZ3 = W ZZZ (the same double product with a zero subtrahend)
  const F xx = q.ld(q.t(1)), s = q.ld(q.t(3)), w = q.ld(q.t(2));
  const F m = norm(add(dbl(xx), xx));
  const F x3 = vred(norm(sub<4, 1>(q.ld(q.t(4)), norm(dbl(s)))));
  const F a = Q::pick(l0, m, w);
  const F b = Q::pick(l0, norm(sub<4, 1>(s, x3)), q.ld(q.coord(q.ia, 3)));
  const F c = Q::pick(l0, w, F::zero());
  const F r = mul2sub(a, b, c, q.ld(q.coord(q.ia, 1)));
  if (l0) q.st(q.coord(q.io, 0), x3);
  q.st(q.coord(q.io, l0 ? 1u : 3u), r);
}

// ---- the same steps with the values BETWEEN them in registers: pure per-lane functions of (lane, the operands' coordinates in the image,
// the values handed in).  The driver moves values between the lanes of a quad -- DPP on the device (coop_add_level below: no LDS
// temporaries, one workgroup barrier per level instead of four), array indexing in tests/host/hosttest.cpp (host_coop_add_regs).
// Exchange schedule of an addition (r1, d, r2, r3: what a lane returned from steps 1, 2, 2, 3):
//   before step 2:  a = r1 of lanes [1, 3, -, -],  b = r1 of lanes [0, 2, -, -]          (U2, S2 | U1, S1; lanes 2, 3 read the image)
//   before step 3:  u1 = r1 of lane 0, pp = r2 of lane 0 (all lanes); lane 0 multiplies its own d, lane 2 its own r2
//   before step 4:  e0 = r3 of [1, -, -, -] (Q), e1 = (lane 0: d of lane 1 = R | lane 1: r3 of lane 0 = PPP), e2 = r2 of [1, 3, -, -] (RR | ZZZ12),
//                   e3 = r1 of [2, -, -, -] (S1)
template <class Q>
KG_HD typename Q::Field coop_a1(const Q& q) {
  const int l = q.lane;
  // lane 0: X1 ZZ2 (U1)   lane 1: X2 ZZ1 (U2)   lane 2: Y1 ZZZ2 (S1)   lane 3: Y2 ZZZ1 (S2)
  const uint32_t first = (l & 1) ? q.ib : q.ia, second = (l & 1) ? q.ia : q.ib;
  const uint32_t cd = (uint32_t)(l >> 1);
  return mul(q.ld(q.coord(first, cd)), q.ld(q.coord(second, 2 + cd)));
}
template <class Q>
KG_HD void coop_a2(const Q& q, const typename Q::Field& ta, const typename Q::Field& tb, typename Q::Field& d, typename Q::Field& r) {
  using F = typename Q::Field;
  const int l = q.lane;
  const bool diff = l < 2;                             // lanes 0, 1: d = ta - tb (P, R) and its square (PP, RR); lanes 2, 3: ZZ1 ZZ2, ZZZ1 ZZZ2
  const F a = Q::pick(diff, ta, q.ld(q.coord(q.ia, (uint32_t)(l | 2))));
  const F b = Q::pick(diff, tb, q.ld(q.coord(q.ib, (uint32_t)(l | 2))));
  d = norm(sub<4, 1>(a, b));
  r = mul(Q::pick(diff, d, a), Q::pick(diff, d, b));
}
template <class Q>
KG_HD typename Q::Field coop_a3(const Q& q, const typename Q::Field& d, const typename Q::Field& r2, const typename Q::Field& u1, const typename Q::Field& pp) {
  using F = typename Q::Field;
  const int l = q.lane;
  // lane 0: P PP (PPP)   lane 1: U1 PP (Q)   lane 2: ZZ12 PP -> ZZ3   lane 3: (idle: computes lane 2's product, stores nothing)
  const F a = Q::pick(l == 0, d, Q::pick(l == 1, u1, r2));
  const F r = mul(a, pp);
  if (l == 2) q.st(q.coord(q.io, 2), r);
  return r;
}
template <class Q>
KG_HD void coop_a4(const Q& q, const typename Q::Field& r3, const typename Q::Field& e0, const typename Q::Field& e1, const typename Q::Field& e2,
                   const typename Q::Field& e3) {
  using F = typename Q::Field;
  if (q.lane >= 2) return;
  const bool l0 = q.lane == 0;
  // lane 0: X3 = RR - PPP - 2Q, Y3 = R (Q - X3) - S1 PPP;   lane 1: ZZZ3 = ZZZ12 PPP, the same double product with a zero subtrahend
  const F ppp = Q::pick(l0, r3, e1);
  const F x3 = vred(norm(sub<8, 3>(e2, add(ppp, dbl(e0)))));                      // (lane 1: a value nobody reads)
  const F a = Q::pick(l0, e1, e2);
  const F b = Q::pick(l0, norm(sub<4, 1>(e0, x3)), ppp);
  const F c = Q::pick(l0, e3, F::zero());
  const F r = mul2sub(a, b, c, ppp);
  if (l0) q.st(q.coord(q.io, 0), x3);
  q.st(q.coord(q.io, l0 ? 1u : 3u), r);
}
// doubling: r1, r2 = what a lane returned from steps 1, 2.  before step 2: v = r1 of lanes [0, 0, 0, 1] (V | XX for lane 3);
// before step 3: e0 = r2 of [3, 0, -, -] (MM | W), e1 = r2 of [1, -, -, -] (S), e2 = r1 of [1, -, -, -] (XX)
template <class Q>
KG_HD typename Q::Field coop_d1(const Q& q) {
  using F = typename Q::Field;
  const bool l0 = (q.lane & 1) == 0;                   // lanes 0 (and 2): V = (2Y)^2   lanes 1 (and 3): XX = X^2
  const F a = q.ld(q.coord(q.ia, l0 ? 1u : 0u));
  const F u = Q::pick(l0, norm(dbl(a)), a);
  return mul(u, u);
}
template <class Q>
KG_HD typename Q::Field coop_d2(const Q& q, const typename Q::Field& v) {
  using F = typename Q::Field;
  const int l = q.lane;
  // lane 0: (2Y) V (W)   lane 1: X V (S)   lane 2: V ZZ -> ZZ3   lane 3: (3 XX)^2 (MM), v = XX there
  const F a = Q::pick(l == 3, v, q.ld(q.coord(q.ia, l == 0 ? 1u : (l == 1 ? 0u : 2u))));
  const F two = norm(dbl(a)), three = norm(add(dbl(a), a));
  const F u = Q::pick(l == 0, two, Q::pick(l == 3, three, a));
  const F r = mul(u, Q::pick(l == 3, three, v));
  if (l == 2) q.st(q.coord(q.io, 2), r);
  return r;
}
template <class Q>
KG_HD void coop_d3(const Q& q, const typename Q::Field& r2, const typename Q::Field& e0, const typename Q::Field& e1, const typename Q::Field& e2) {
  using F = typename Q::Field;
  if (q.lane >= 2) return;
  const bool l0 = q.lane == 0;
  // lane 0: X3 = MM - 2S, Y3 = M (S - X3) - W Y, M = 3 XX;   lane 1: ZZZ3 = W ZZZ
  const F m = norm(add(dbl(e2), e2));
  const F x3 = vred(norm(sub<4, 1>(e0, norm(dbl(e1)))));
  const F w = Q::pick(l0, r2, e0);
  const F a = Q::pick(l0, m, w);
  const F b = Q::pick(l0, norm(sub<4, 1>(e1, x3)), q.ld(q.coord(q.ia, 3)));
  const F c = Q::pick(l0, w, F::zero());
  const F r = mul2sub(a, b, c, q.ld(q.coord(q.ia, 1)));
  if (l0) q.st(q.coord(q.io, 0), x3);
  q.st(q.coord(q.io, l0 ? 1u : 3u), r);
}

#if defined(__HIPCC__)
// One level of additions over a workgroup: quad t of the workgroup (lanes 4t .. 4t+3) adds items ia + ib -> io when `active`.  Every
// thread of the workgroup must call it (one workgroup barrier, at its end: between the steps only the quad's own lanes exchange values, by
// DPP).  tmp, flg: unused since the values between the steps live in registers (kept in the signature: the LDS layout of the callers).
// KG_COOP_LDS builds keep the older form (temporaries in LDS, four barriers) for A/B runs.
template <class F>
__device__ __forceinline__ void coop_add_level(uint32_t* img, uint32_t cap, uint32_t* tmp, uint32_t* flg, bool active, uint32_t ia, uint32_t ib, uint32_t io) {
  const CoopQuad<F> q{img, cap, tmp, blockDim.x >> 2, threadIdx.x >> 2, flg, ia, ib, io, (int)(threadIdx.x & 3u)};
  const int mode = active ? coop_add_mode(q) : COOP_IDLE;
#if defined(KG_COOP_LDS) || !defined(__HIP_DEVICE_COMPILE__)
  if (mode == COOP_ADD) coop_add_s1(q);
  __syncthreads();
  if (mode == COOP_ADD) coop_add_s2(q);
  __syncthreads();
  coop_add_s3(q, mode);
  __syncthreads();
  coop_add_s4(q, mode);
  __syncthreads();
#else
  if (mode == COOP_ADD) {                                   // (uniform over the quad: all four lanes are here together)
    const F r1 = coop_a1(q);
    F d, r2;
    coop_a2(q, coop_xchg<coop_perm(1, 3, 2, 3)>(r1), coop_xchg<coop_perm(0, 2, 2, 3)>(r1), d, r2);
    const F pp = coop_xchg<coop_perm(0, 0, 0, 0)>(r2);
    const F rr = coop_xchg<coop_perm(1, 1, 1, 1)>(d);     // R, for the test of the exceptional case and step 4
    if (is_zero_2p(pp)) {                                   // equal x (weierstrass.rs:114-120): the one-lane formulas; rare
      if (q.lane == 0) {
        if (is_zero(rr)) coop_set_point(q, q.io, double_xyzz(coop_point(q, q.ia)));
        else coop_set_point(q, q.io, XYZZ<F>::identity());
      }
    } else {
      const F r3 = coop_a3(q, d, r2, coop_xchg<coop_perm(0, 0, 0, 0)>(r1), pp);
      const F e1 = CoopQuad<F>::pick(q.lane == 0, rr, coop_xchg<coop_perm(0, 0, 0, 0)>(r3));
      coop_a4(q, r3, coop_xchg<coop_perm(1, 1, 1, 1)>(r3), e1, coop_xchg<coop_perm(1, 3, 2, 3)>(r2), coop_xchg<coop_perm(2, 2, 2, 2)>(r1));
    }
  } else if (mode == COOP_TAKE_A || mode == COOP_TAKE_B) {
    const uint32_t src = mode == COOP_TAKE_A ? q.ia : q.ib;
    if (src != q.io) q.st(q.coord(q.io, (uint32_t)q.lane), q.ld(q.coord(src, (uint32_t)q.lane)));
  }
  __syncthreads();
#endif
}
// `times` doublings in place of item `it` by quad t (times may differ between quads; uniform trip count `max_times` for the barriers)
template <class F>
__device__ __forceinline__ void coop_dbl_level(uint32_t* img, uint32_t cap, uint32_t* tmp, uint32_t* flg, uint32_t it, uint32_t times, uint32_t max_times) {
  const CoopQuad<F> q{img, cap, tmp, blockDim.x >> 2, threadIdx.x >> 2, flg, it, it, it, (int)(threadIdx.x & 3u)};
  for (uint32_t k = 0; k < max_times; ++k) {
    const bool on = k < times && coop_dbl_active(q);
#if defined(KG_COOP_LDS) || !defined(__HIP_DEVICE_COMPILE__)
    if (on) coop_dbl_s1(q);
    __syncthreads();
    if (on) coop_dbl_s2(q);
    __syncthreads();
    if (on) coop_dbl_s3(q);
    __syncthreads();
#else
    if (on) {
      const F r1 = coop_d1(q);
      const F r2 = coop_d2(q, coop_xchg<coop_perm(0, 0, 0, 1)>(r1));
      coop_d3(q, r2, coop_xchg<coop_perm(3, 0, 2, 3)>(r2), coop_xchg<coop_perm(1, 1, 1, 1)>(r2), coop_xchg<coop_perm(1, 1, 1, 1)>(r1));
    }
    __syncthreads();                                        // the next doubling reads the item this one wrote (its lanes 0 .. 2 wrote, all four read)
#endif
  }
}
#endif

}  // namespace kg

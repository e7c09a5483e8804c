// fp2s.h -- Fq2 = Fq[u]/(u^2 + 1) spread over a LANE PAIR (device only): the even lane of a pair holds c0, the odd lane
// c1, and one task (a bucket's partial sum, a reduction tree node) runs on the pair.
//
// Why: with both coordinates in one lane a G2 point addition needs ~256 VGPRs (XYZZ accumulator 72 + affine point 36 +
// products), i.e. 1-2 waves per SIMD, where this arithmetic issues at 69-82 % of its four-wave rate; the G2 halving
// kernel even spills 137 registers to AGPRs.  Split over two lanes every coordinate is a base-field value again (the
// register footprint of the G1 kernels, four waves per SIMD), the multiply-accumulate count is unchanged (each lane
// computes ONE of the two coordinates of a product: a*b + sigma*c*d through mul2pm, the sign being data), and the price
// is the partner's 9 limbs per operand through a quad-permute DPP move.
//
// Same interface as Fp2<F> as far as curve.h uses it, so XYZZ<Fp2S<F>> and the point formulas are the same templates.
#pragma once
#include "fp29.h"
#include "curve.h"

namespace kg {

template <class F>
struct Fp2S {
  F v;                                             // this lane's coordinate
  static __device__ __forceinline__ int half() { return (int)(threadIdx.x & 1u); }
  static __device__ __forceinline__ Fp2S zero() { return {F::zero()}; }
  static __device__ __forceinline__ Fp2S one() {
    const F o = F::one();
    Fp2S r;
#pragma unroll
    for (int k = 0; k < 9; ++k) r.v.l[k] = half() ? 0u : o.l[k];
    return r;
  }
};

template <class F>
__device__ __forceinline__ F partner(const F& a) {                 // the other lane's coordinate
  F r;
#pragma unroll
  for (int k = 0; k < 9; ++k) r.l[k] = (uint32_t)__shfl_xor((int)a.l[k], 1);
  return r;
}
template <class F>
__device__ __forceinline__ F pick(bool first, const F& a, const F& b) {
  F r;
#pragma unroll
  for (int k = 0; k < 9; ++k) r.l[k] = first ? a.l[k] : b.l[k];
  return r;
}

template <class F> __device__ __forceinline__ Fp2S<F> add(const Fp2S<F>& a, const Fp2S<F>& b) { return {add(a.v, b.v)}; }
template <class F> __device__ __forceinline__ Fp2S<F> dbl(const Fp2S<F>& a) { return {dbl(a.v)}; }
template <int C, int T, class F> __device__ __forceinline__ Fp2S<F> sub(const Fp2S<F>& a, const Fp2S<F>& b) { return {sub<C, T>(a.v, b.v)}; }
template <class F> __device__ __forceinline__ Fp2S<F> norm(const Fp2S<F>& a) { return {norm(a.v)}; }
template <class F> __device__ __forceinline__ Fp2S<F> vred(const Fp2S<F>& a) { return {vred(a.v)}; }

// (a0 + a1 u)(b0 + b1 u):  even lane  a0*b0 - a1*b1 = mine_a*mine_b - other_a*other_b
//                          odd lane   a1*b0 + a0*b1 = mine_a*other_b + other_a*mine_b
template <class F>
__device__ __forceinline__ Fp2S<F> mul(const Fp2S<F>& a, const Fp2S<F>& b) {
  const bool even = Fp2S<F>::half() == 0;
  const F oa = partner(a.v), ob = partner(b.v);
  return {mul2pm(a.v, pick(even, b.v, ob), oa, pick(even, ob, b.v), even)};
}
// (a0 + a1 u)^2:  even lane (a0 + a1)(a0 - a1),  odd lane (2 a1) * a0     (inputs: normalised limbs, K <= 6)
template <class F>
__device__ __forceinline__ Fp2S<F> sqr(const Fp2S<F>& a) {
  const bool even = Fp2S<F>::half() == 0;
  const F o = partner(a.v);
  return {mul(pick(even, add(a.v, o), dbl(a.v)), pick(even, norm(sub<8, 1>(a.v, o)), o))};
}
// a*b - c*d (the point formulas' Y3)
template <class F>
__device__ __forceinline__ Fp2S<F> mul2sub(const Fp2S<F>& a, const Fp2S<F>& b, const Fp2S<F>& c, const Fp2S<F>& d) {
  return {vred(norm(sub<4, 1>(mul(a, b).v, mul(c, d).v)))};
}
template <class F>
__device__ __forceinline__ bool is_zero_2p(const Fp2S<F>& a) {
  const int z = is_zero_2p(a.v) ? 1 : 0;
  const int o = __shfl_xor(z, 1);                  // unconditionally: both lanes of the pair must execute the exchange
  return (z & o) != 0;
}
template <class F>
__device__ __forceinline__ bool is_zero(const Fp2S<F>& a) {
  const int z = is_zero(a.v) ? 1 : 0;
  const int o = __shfl_xor(z, 1);
  return (z & o) != 0;
}

// P + (+-a) for the bucket kernel: like Fq2 in one lane, the point is negated first (curve.h: the folded form's R reaches
// K = 12, which the Fq2 square's inner subtraction does not admit)
template <class G>
__device__ __forceinline__ XYZZ<Fp2S<G>> add_mixed_signed(const XYZZ<Fp2S<G>>& p, const Affine<Fp2S<G>>& a, bool negate) {
  return add_mixed(p, negate ? neg_affine(a) : a);
}

// lanes that cooperate on one task
template <class F> struct Lanes { static constexpr int N = 1; };
template <class F> struct Lanes<Fp2S<F>> { static constexpr int N = 2; };

}  // namespace kg

// fp29.h -- BN254 Fq / Fr arithmetic for gfx950: 9 limbs of 29 bits, Montgomery radix R' = 2^261.
//
// Why this shape (measured on MI355X, tools/ubench/valu_rates.hip, profiles/r01_valu_rates.txt):
//   v_mad_u64_u32 (32x32 multiply + 64-bit accumulate) issues at the same ~33 Top/s as a single
//   v_addc_co_u32, so carry chains cost as much as the multiplies.  With 29-bit limbs every column of the
//   product (<= 9 a*b terms + 9 m*p terms, each < 2^58..2^60) fits one 64-bit accumulator, so a whole
//   Montgomery multiplication is 162 mads + ~60 shifts/masks and NO carry instructions, and field
//   add/sub are 9 independent v_add_u32 (lazy, unsaturated limbs).
//
// What it replaces in the reference: zkstd/src/arithmetic/limbs/bits_256/normal.rs:4-253 (add, sub,
// double, neg, mul, square, mont on 4 x u64, R = 2^256).  Results are identical as field elements; values
// cross the ABI in the reference's form (4 x u64 little endian, x*2^256 mod p, fully reduced) via
// from_ref()/to_ref().
//
// Value discipline.  An element is "loose": limbs 0..7 may exceed 29 bits and the value may be any
// representative below K*p.  mul() accepts loose inputs as long as
//     9*max_limb(a)*max_limb(b) + 9*2^58 + 2^36 < 2^64      and      K_a*K_b < 2^261/p (~169),
// and returns a normalised element (limbs < 2^29) below (K_a*K_b/169 + 1)*p.  The worst-case bounds of
// every formula in curve.h / the kernels are machine-checked on the host by instantiating the same
// templates with FpChecked (tests/host/, `-DKG_HOST_TEST`).
#pragma once
#include <type_traits>
#include <cstdint>
#include "fp_consts.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define KG_HD __host__ __device__ __forceinline__
#else
#define KG_HD inline __attribute__((always_inline))
#endif

namespace kg {

constexpr uint32_t M29 = 0x1fffffffu;

template <class P>
struct Fp {
  using Params = P;
  uint32_t l[9];

  static KG_HD Fp zero() {
    Fp r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = 0;
    return r;
  }
  template <class A>
  static KG_HD Fp from_const(const A& c) {
    Fp r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = c[i];
    return r;
  }
  static KG_HD Fp one() { return from_const(P::ONE); }
};

// f(integral_constant<int, I>) for I in [LO, HI): the column index of a product is a compile-time constant
template <int I, int HI, class Fn>
KG_HD void static_for(Fn&& f) {
  if constexpr (I < HI) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, HI>(f);
  }
}

// ---------------------------------------------------------------------------------------------
// acc += sum x[j] * y[j] as ONE chain of v_mad_u64_u32 whose addend is the running column sum.  Written as instructions on the
// device: left to itself the compiler's reassociation starts every column with a fresh zero-seeded chain and joins the
// carried-in sum with a separate 64-bit add (v_lshl_add_u64, the issue cost of a multiply-accumulate) -- 20-22 of them per
// product, a tenth of its instructions.  One asm statement per chain: the compiler separates dependent asm statements by a
// wait state (it must assume an inline instruction writes a partial register).  macs_k: y are compile-time constants (limbs
// of p), kept in scalar registers; macs_i: signed 32-bit factors, signed sum.  The carry-out mask of the 64-bit sum is never
// set (column bounds) and never read.
// ---------------------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__) && !defined(KG_NO_ASM_MAC)
#define KG_ASM_MAC 1
// the chains assume wave64: the carry-out operand of v_mad_u64_u32 is an SGPR PAIR ("=&s" of a 64-bit variable); and the gfx9
// constant-bus rule (one SGPR operand per VALU instruction: the "s" constants of macs_k)
#if defined(__AMDGCN_WAVEFRONT_SIZE)
static_assert(__AMDGCN_WAVEFRONT_SIZE == 64, "fp29.h multiply-accumulate chains are written for wave64 (gfx950)");
#endif
#endif
// operands of a chain statement: %0 the 64-bit sum, %1 the carry-out pair, then (x[j], y[j]) as %(2 + 2j), %(3 + 2j)
#define KG_MAD(op, a, b) op " %0, %1, %" #a ", %" #b ", %0\n\t"
#define KG_CHAIN_1(op) KG_MAD(op, 2, 3)
#define KG_CHAIN_2(op) KG_CHAIN_1(op) KG_MAD(op, 4, 5)
#define KG_CHAIN_3(op) KG_CHAIN_2(op) KG_MAD(op, 6, 7)
#define KG_CHAIN_4(op) KG_CHAIN_3(op) KG_MAD(op, 8, 9)
#define KG_CHAIN_5(op) KG_CHAIN_4(op) KG_MAD(op, 10, 11)
#define KG_CHAIN_6(op) KG_CHAIN_5(op) KG_MAD(op, 12, 13)
#define KG_CHAIN_7(op) KG_CHAIN_6(op) KG_MAD(op, 14, 15)
#define KG_CHAIN_8(op) KG_CHAIN_7(op) KG_MAD(op, 16, 17)
#define KG_CHAIN_9(op) KG_CHAIN_8(op) KG_MAD(op, 18, 19)
#define KG_IN_1(yc) "v"(x[0]), yc(y[0])
#define KG_IN_2(yc) KG_IN_1(yc), "v"(x[1]), yc(y[1])
#define KG_IN_3(yc) KG_IN_2(yc), "v"(x[2]), yc(y[2])
#define KG_IN_4(yc) KG_IN_3(yc), "v"(x[3]), yc(y[3])
#define KG_IN_5(yc) KG_IN_4(yc), "v"(x[4]), yc(y[4])
#define KG_IN_6(yc) KG_IN_5(yc), "v"(x[5]), yc(y[5])
#define KG_IN_7(yc) KG_IN_6(yc), "v"(x[6]), yc(y[6])
#define KG_IN_8(yc) KG_IN_7(yc), "v"(x[7]), yc(y[7])
#define KG_IN_9(yc) KG_IN_8(yc), "v"(x[8]), yc(y[8])
#define KG_Y_VGPR(e) "v"(e)
#define KG_Y_SGPR(e) "s"(e)
#define KG_CHAIN_CASE(n, op, yc) if constexpr (N == n) asm(KG_CHAIN_##n(op) : "+v"(acc), "=&s"(cy) : KG_IN_##n(yc));
#define KG_CHAIN_CASES(op, yc)                                                                                          \
  KG_CHAIN_CASE(1, op, yc) KG_CHAIN_CASE(2, op, yc) KG_CHAIN_CASE(3, op, yc) KG_CHAIN_CASE(4, op, yc) KG_CHAIN_CASE(5, op, yc) \
  KG_CHAIN_CASE(6, op, yc) KG_CHAIN_CASE(7, op, yc) KG_CHAIN_CASE(8, op, yc) KG_CHAIN_CASE(9, op, yc)
template <int N>
KG_HD void macs(uint64_t& acc, const uint32_t (&x)[N], const uint32_t (&y)[N]) {
  static_assert(N >= 1 && N <= 9, "one column of a 9-limb product");
#ifdef KG_ASM_MAC
  uint64_t cy;
  KG_CHAIN_CASES("v_mad_u64_u32", KG_Y_VGPR)
#else
  for (int j = 0; j < N; ++j) acc += (uint64_t)x[j] * y[j];
#endif
}
template <int N>
KG_HD void macs_k(uint64_t& acc, const uint32_t (&x)[N], const uint32_t (&y)[N]) {
  static_assert(N >= 1 && N <= 9, "one column of a 9-limb product");
#ifdef KG_ASM_MAC
  uint64_t cy;
  KG_CHAIN_CASES("v_mad_u64_u32", KG_Y_SGPR)
#else
  for (int j = 0; j < N; ++j) acc += (uint64_t)x[j] * y[j];
#endif
}
template <int N>
KG_HD void macs_i(int64_t& acc, const int32_t (&x)[N], const int32_t (&y)[N]) {
  static_assert(N >= 1 && N <= 9, "one column of a 9-limb product");
#ifdef KG_ASM_MAC
  uint64_t cy;
  KG_CHAIN_CASES("v_mad_i64_i32", KG_Y_VGPR)
#else
  for (int j = 0; j < N; ++j) acc += (int64_t)x[j] * (int64_t)y[j];
#endif
}
// signed columns (mul2sub, mul2pm): the unsigned chains on the two's-complement bit pattern
template <int N>
KG_HD void macs(int64_t& acc, const uint32_t (&x)[N], const uint32_t (&y)[N]) { uint64_t u = (uint64_t)acc; macs(u, x, y); acc = (int64_t)u; }
template <int N>
KG_HD void macs_k(int64_t& acc, const uint32_t (&x)[N], const uint32_t (&y)[N]) { uint64_t u = (uint64_t)acc; macs_k(u, x, y); acc = (int64_t)u; }
// column K of x * y: sum over i in [LO, HI] of x[i] * y[K - i]
template <int K, int LO, int HI, class Acc, class X, class Y>
KG_HD void mac_col(Acc& acc, const X& x, const Y& y) {
  if constexpr (HI >= LO) {
    uint32_t xs[HI - LO + 1], ys[HI - LO + 1];
#pragma unroll
    for (int i = LO; i <= HI; ++i) { xs[i - LO] = x[i]; ys[i - LO] = y[K - i]; }
    macs(acc, xs, ys);
  }
}
// the same against a constant table: Tab::at(j), j = K - i, a compile-time constant (an immediate, not a load)
template <class P> struct ModulusLimbs { static KG_HD constexpr uint32_t at(int j) { return P::P[j]; } };
template <class P> struct ModulusBarLimbs { static KG_HD constexpr uint32_t at(int j) { return P::PBAR[j]; } };
template <class Tab, int K, int LO, int HI, class Acc, class X>
KG_HD void mac_col_k(Acc& acc, const X& x) {
  if constexpr (HI >= LO) {
    uint32_t xs[HI - LO + 1], ys[HI - LO + 1];
#pragma unroll
    for (int i = LO; i <= HI; ++i) { xs[i - LO] = x[i]; ys[i - LO] = Tab::at(K - i); }
    macs_k(acc, xs, ys);
  }
}
template <int K, int LO, int HI, class X, class Y>
KG_HD void mac_col_i(int64_t& acc, const X& x, const Y& y) {
  if constexpr (HI >= LO) {
    int32_t xs[HI - LO + 1], ys[HI - LO + 1];
#pragma unroll
    for (int i = LO; i <= HI; ++i) { xs[i - LO] = (int32_t)x[i]; ys[i - LO] = (int32_t)y[K - i]; }
    macs_i(acc, xs, ys);
  }
}

// ---------------------------------------------------------------------------------------------
// Montgomery product a*b/2^261 (finely integrated product scanning; one 64-bit accumulator)
// ---------------------------------------------------------------------------------------------
template <class P>
KG_HD Fp<P> mul(const Fp<P>& a, const Fp<P>& b) {
  uint32_t m[9];
  Fp<P> r;
  uint64_t acc = 0;
  static_for<0, 9>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    mac_col<k, 0, k>(acc, a.l, b.l);
    mac_col_k<ModulusLimbs<P>, k, 0, k - 1>(acc, m);
    m[k] = ((uint32_t)acc * P::INV) & M29;
    mac_col_k<ModulusLimbs<P>, k, k, k>(acc, m);
    acc >>= 29;
  });
  static_for<9, 17>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    mac_col<k, k - 8, 8>(acc, a.l, b.l);
    mac_col_k<ModulusLimbs<P>, k, k - 8, 8>(acc, m);
    r.l[k - 9] = (uint32_t)acc & M29;
    acc >>= 29;
  });
  r.l[8] = (uint32_t)acc;
  return r;
}

// Montgomery square: the 36 cross products are taken once against the doubled operand.
template <class P>
KG_HD Fp<P> sqr(const Fp<P>& a) {
  uint32_t m[9], d[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) d[i] = a.l[i] << 1;
  Fp<P> r;
  uint64_t acc = 0;
  static_for<0, 9>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    mac_col<k, 0, (k + 1) / 2 - 1>(acc, d, a.l);                 // the cross products: 2 i < k
    if constexpr ((k & 1) == 0) mac_col<k, k / 2, k / 2>(acc, a.l, a.l);
    mac_col_k<ModulusLimbs<P>, k, 0, k - 1>(acc, m);
    m[k] = ((uint32_t)acc * P::INV) & M29;
    mac_col_k<ModulusLimbs<P>, k, k, k>(acc, m);
    acc >>= 29;
  });
  static_for<9, 17>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    mac_col<k, k - 8, (k + 1) / 2 - 1>(acc, d, a.l);
    if constexpr ((k & 1) == 0) mac_col<k, k / 2, k / 2>(acc, a.l, a.l);
    mac_col_k<ModulusLimbs<P>, k, k - 8, 8>(acc, m);
    r.l[k - 9] = (uint32_t)acc & M29;
    acc >>= 29;
  });
  r.l[8] = (uint32_t)acc;
  return r;
}

// Montgomery (a*b + c*d)/2^261 with one reduction (lazy reduction for the Fq2 products): 243 mads.
// Column bound: 9*(max_a*max_b + max_c*max_d) + 9*2^58 + 2^36 < 2^64; value: (Ka*Kb + Kc*Kd) < 169.
template <class P>
KG_HD Fp<P> mul2add(const Fp<P>& a, const Fp<P>& b, const Fp<P>& c, const Fp<P>& d) {
  uint32_t m[9];
  Fp<P> r;
  uint64_t acc = 0;
  static_for<0, 9>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    mac_col<k, 0, k>(acc, a.l, b.l);
    mac_col<k, 0, k>(acc, c.l, d.l);
    mac_col_k<ModulusLimbs<P>, k, 0, k - 1>(acc, m);
    m[k] = ((uint32_t)acc * P::INV) & M29;
    mac_col_k<ModulusLimbs<P>, k, k, k>(acc, m);
    acc >>= 29;
  });
  static_for<9, 17>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    mac_col<k, k - 8, 8>(acc, a.l, b.l);
    mac_col<k, k - 8, 8>(acc, c.l, d.l);
    mac_col_k<ModulusLimbs<P>, k, k - 8, 8>(acc, m);
    r.l[k - 9] = (uint32_t)acc & M29;
    acc >>= 29;
  });
  r.l[8] = (uint32_t)acc;
  return r;
}

// Montgomery (a*b - c*d)/2^261 with one reduction over SIGNED 64-bit columns; result in [0, 2p).
// Needs limbs < 2^31, 9*max_a*max_b + 9*2^58 + 2^36 < 2^63, 9*max_c*max_d < 2^63, Ka*Kb < 169, Kc*Kd < 169.
template <class P>
KG_HD Fp<P> mul2sub(const Fp<P>& a, const Fp<P>& b, const Fp<P>& c, const Fp<P>& d) {
  uint32_t m[9];
  int32_t nc[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) nc[i] = -(int32_t)c.l[i];
  Fp<P> r;
  int64_t acc = 0;
  static_for<0, 9>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    mac_col<k, 0, k>(acc, a.l, b.l);
    mac_col_i<k, 0, k>(acc, nc, d.l);
    mac_col_k<ModulusLimbs<P>, k, 0, k - 1>(acc, m);
    m[k] = ((uint32_t)acc * P::INV) & M29;
    mac_col_k<ModulusLimbs<P>, k, k, k>(acc, m);
    acc >>= 29;
  });
  static_for<9, 17>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    mac_col<k, k - 8, 8>(acc, a.l, b.l);
    mac_col_i<k, k - 8, 8>(acc, nc, d.l);
    mac_col_k<ModulusLimbs<P>, k, k - 8, 8>(acc, m);
    r.l[k - 9] = (uint32_t)acc & M29;
    acc >>= 29;
  });
  // value in (-p, 2p): add p when negative, then propagate carries
  const uint32_t neg_mask = (uint32_t)((int32_t)(acc >> 32) >> 31);
  uint32_t cy = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint32_t v = r.l[i] + (P::P[i] & neg_mask) + cy;
    r.l[i] = v & M29;
    cy = v >> 29;
  }
  r.l[8] = (uint32_t)acc + (P::P[8] & neg_mask) + cy;
  return r;
}

// ---------------------------------------------------------------------------------------------
// Product with a precomputed constant (Shoup / Barrett form): c = {w, q = floor(w * 2^261 / p)}, w canonical.
// Returns a * w mod p -- NO Montgomery factor: the operand keeps its domain -- as a normalised value below
// (2 + Ka / 169) p.  143 multiply-accumulates and no v_mul_lo against the 162 + 9 of mul(): the quotient estimate takes
// columns 7..16 of a * q (the dropped columns are worth less than one unit of the quotient), the remainder the nine low
// columns of a * w + qe * (2^261 - p).  Needs limbs of a below ~2^31 (9 * max_a * 2^29 + 9 * 2^58 + carry < 2^64) and
// a < 2^261 (Ka < 169).  The transforms' twiddles are such constants (ntt_tile.h).
// ---------------------------------------------------------------------------------------------
template <class P>
struct FpConst {
  uint32_t w[9], q[9];
};
template <class P>
KG_HD Fp<P> mulc(const Fp<P>& a, const FpConst<P>& c) {
  uint32_t qe[9];
  uint64_t acc = 0;
  mac_col<7, 0, 7>(acc, a.l, c.q);
  acc >>= 29;
  mac_col<8, 0, 8>(acc, a.l, c.q);
  acc >>= 29;
  static_for<9, 17>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    mac_col<k, k - 8, 8>(acc, a.l, c.q);
    qe[k - 9] = (uint32_t)acc & M29;
    acc >>= 29;
  });
  qe[8] = (uint32_t)acc;                       // floor(a * q / 2^261) or one less; < 2^261 / 2 ... fits (a * q < 2^522)
  Fp<P> r;
  acc = 0;
  static_for<0, 9>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    mac_col<k, 0, k>(acc, a.l, c.w);
    mac_col_k<ModulusBarLimbs<P>, k, 0, k>(acc, qe);
    // (only the low 29 bits of the last column are used; as instructions its 18 multiply-accumulates stay multiply-accumulates)
    r.l[k] = (uint32_t)acc & M29;              // the true remainder is below 3p < 2^256: bits 261 and up are the q * 2^261 term
    acc >>= 29;
  });
  return r;
}
// q = floor(w * 2^261 / p) for a canonical w (restoring division, 261 steps; table construction only)
template <class P>
KG_HD FpConst<P> make_const(const Fp<P>& w_canonical) {
  FpConst<P> c;
  uint32_t rem[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) { c.w[i] = w_canonical.l[i]; c.q[i] = 0; rem[i] = w_canonical.l[i]; }
  for (int step = 260; step >= 0; --step) {
    uint32_t cy = 0;                           // rem <<= 1 (rem < p < 2^254: no overflow of the top limb)
#pragma unroll
    for (int i = 0; i < 8; ++i) { const uint32_t v = (rem[i] << 1) | cy; rem[i] = v & M29; cy = v >> 29; }
    rem[8] = (rem[8] << 1) | cy;
    uint32_t d[9], borrow = 0;                 // d = rem - p
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const uint32_t v = rem[i] - P::P[i] - borrow;
      borrow = v >> 31;
      d[i] = i < 8 ? (v & M29) : v;
    }
    const bool ge = borrow == 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) rem[i] = ge ? d[i] : rem[i];
    c.q[step / 29] |= (ge ? 1u : 0u) << (step % 29);
  }
  return c;
}

// lazy limb-wise sum (no carries)
template <class P>
KG_HD Fp<P> add(const Fp<P>& a, const Fp<P>& b) {
  Fp<P> r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + b.l[i];
  return r;
}
template <class P>
KG_HD Fp<P> dbl(const Fp<P>& a) { return add(a, a); }

template <class P, int C, int T>
struct FatZ;
#define KG_FATZ(C_, T_)                                                           \
  template <class P>                                                              \
  struct FatZ<P, C_, T_> {                                                        \
    static KG_HD uint32_t at(int i) { return P::Z##C_##_##T_[i]; }                \
  };
KG_FATZ(2, 1) KG_FATZ(3, 1) KG_FATZ(4, 1) KG_FATZ(8, 1) KG_FATZ(16, 1) KG_FATZ(32, 1) KG_FATZ(4, 3) KG_FATZ(8, 3) KG_FATZ(16, 3) KG_FATZ(32, 3)
#undef KG_FATZ

// lazy difference a + C*p - b, limb-wise without borrows.  Requires limbs 0..7 of b <= T*2^29 - T and
// top limb of b <= top limb of the fat constant (i.e. b < ~(C-1)*p); checked by FpChecked.
template <int C, int T, class P>
KG_HD Fp<P> sub(const Fp<P>& a, const Fp<P>& b) {
  Fp<P> r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + FatZ<P, C, T>::at(i) - b.l[i];
  return r;
}
template <int C, int T, class P>
KG_HD Fp<P> neg(const Fp<P>& b) { return sub<C, T>(Fp<P>::zero(), b); }

// carry propagation: limbs 0..7 back below 2^29, value unchanged
template <class P>
KG_HD Fp<P> norm(const Fp<P>& a) {
  Fp<P> r;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint32_t v = a.l[i] + c;
    r.l[i] = v & M29;
    c = v >> 29;
  }
  r.l[8] = a.l[8] + c;
  return r;
}

// cheap value reduction: normalised limbs, value < 2^27 * 2^232 (K < ~40)  ->  normalised, value < 1.06p.
// q = floor(top * MU / 2^32) never exceeds floor(value / p) and is short of it by at most one.
template <class P>
KG_HD Fp<P> vred(const Fp<P>& a) {
  const uint32_t q = (uint32_t)(((uint64_t)a.l[8] * P::MU) >> 32);
  Fp<P> r;
  int64_t cy = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int64_t t = (int64_t)a.l[i] - (int64_t)((uint64_t)q * P::P[i]) + cy;
    r.l[i] = (uint32_t)t & M29;
    cy = t >> 29;
  }
  r.l[8] = (uint32_t)((int64_t)a.l[8] - (int64_t)((uint64_t)q * P::P[8]) + cy);
  return r;
}

// [0, 2p) normalised -> [0, p)
template <class P>
KG_HD Fp<P> reduce_2p(const Fp<P>& a) {
  Fp<P> d;
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    uint32_t v = a.l[i] - P::P[i] - borrow;
    borrow = v >> 31;                       // limbs < 2^29: a negative difference sets bit 31
    d.l[i] = (i < 8) ? (v & M29) : v;
  }
  Fp<P> r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = borrow ? a.l[i] : d.l[i];
  return r;
}

// any loose value (K < 169, normalised limbs not required beyond mul's rule) -> canonical [0, p)
template <class P>
KG_HD Fp<P> reduce(const Fp<P>& a) { return reduce_2p(mul(a, Fp<P>::one())); }

// zero test for a normalised value known to lie in [0, 2p): it is 0 mod p iff it is 0 or p
template <class P>
KG_HD bool is_zero_2p(const Fp<P>& a) {
  uint32_t z = 0, e = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    z |= a.l[i];
    e |= a.l[i] ^ P::P[i];
  }
  return z == 0 || e == 0;
}
template <class P>
KG_HD bool is_zero(const Fp<P>& a) { return is_zero_2p(mul(a, Fp<P>::one())); }

// bitwise identity of two canonical values
template <class P>
KG_HD bool same_limbs(const Fp<P>& a, const Fp<P>& b) {
  uint32_t e = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) e |= a.l[i] ^ b.l[i];
  return e == 0;
}

// ---------------------------------------------------------------------------------------------
// Boundary conversions.  Reference form: 8 x u32 little endian = 4 x u64 LE, x*2^256 mod p, canonical.
// ---------------------------------------------------------------------------------------------
// raw 256-bit integer -> 29-bit limbs (no domain change)
template <class P>
KG_HD Fp<P> limbs_from_words(const uint32_t w[8]) {
  Fp<P> r;
  r.l[0] = w[0] & M29;
  r.l[1] = ((w[0] >> 29) | (w[1] << 3)) & M29;
  r.l[2] = ((w[1] >> 26) | (w[2] << 6)) & M29;
  r.l[3] = ((w[2] >> 23) | (w[3] << 9)) & M29;
  r.l[4] = ((w[3] >> 20) | (w[4] << 12)) & M29;
  r.l[5] = ((w[4] >> 17) | (w[5] << 15)) & M29;
  r.l[6] = ((w[5] >> 14) | (w[6] << 18)) & M29;
  r.l[7] = ((w[6] >> 11) | (w[7] << 21)) & M29;
  r.l[8] = w[7] >> 8;
  return r;
}
// normalised limbs of a value < 2^256 -> 8 words
template <class P>
KG_HD void words_from_limbs(const Fp<P>& a, uint32_t w[8]) {
  w[0] = a.l[0] | (a.l[1] << 29);
  w[1] = (a.l[1] >> 3) | (a.l[2] << 26);
  w[2] = (a.l[2] >> 6) | (a.l[3] << 23);
  w[3] = (a.l[3] >> 9) | (a.l[4] << 20);
  w[4] = (a.l[4] >> 12) | (a.l[5] << 17);
  w[5] = (a.l[5] >> 15) | (a.l[6] << 14);
  w[6] = (a.l[6] >> 18) | (a.l[7] << 11);
  w[7] = (a.l[7] >> 21) | (a.l[8] << 8);
}
// reference Montgomery form (x*2^256) -> internal (x*2^261), normalised, < 2p
template <class P>
KG_HD Fp<P> from_ref(const uint32_t w[8]) { return mul(limbs_from_words<P>(w), Fp<P>::from_const(P::C_FROM_REF)); }
// internal -> reference form, canonical
template <class P>
KG_HD void to_ref(const Fp<P>& a, uint32_t w[8]) {
  words_from_limbs(reduce_2p(mul(a, Fp<P>::from_const(P::C_TO_REF))), w);
}
// reference Montgomery form -> canonical integer (the reference's montgomery_reduce, bn254/src/fr.rs:122-128)
template <class P>
KG_HD void ref_to_int(const uint32_t w[8], uint32_t k[8]) {
  words_from_limbs(reduce_2p(mul(limbs_from_words<P>(w), Fp<P>::from_const(P::C_REF_TO_INT))), k);
}
// canonical integer -> internal Montgomery form
template <class P>
KG_HD Fp<P> from_int(const uint32_t w[8]) { return mul(limbs_from_words<P>(w), Fp<P>::from_const(P::C_INT_TO_MONT)); }
// canonical integer -> reference form (the reference's to_mont_form, represent.rs:30-32)
template <class P>
KG_HD void int_to_ref(const uint32_t k[8], uint32_t w[8]) {
  words_from_limbs(reduce_2p(mul(limbs_from_words<P>(k), Fp<P>::from_const(P::C_R2_REF))), w);
}

// a^e for a 256-bit exponent given as 8 words (MSB-first square and multiply); a: K small
template <class P>
KG_HD Fp<P> pow_words(const Fp<P>& a, const uint32_t e[8]) {
  Fp<P> r = Fp<P>::one();
  for (int i = 255; i >= 0; --i) {
    r = sqr(r);
    if ((e[i >> 5] >> (i & 31)) & 1) r = mul(r, a);
  }
  return r;
}
// a^(p-2)  (Fermat inverse; zkstd normal.rs:256-270).  Returns 0 for a == 0.
template <class P>
KG_HD Fp<P> inv(const Fp<P>& a) {
  uint32_t e[8];
  Fp<P> pm2 = Fp<P>::from_const(P::P);
  pm2.l[0] -= 2;                       // P[0] >= 2 for both moduli
  words_from_limbs(pm2, e);
  return pow_words(a, e);
}

// ---------------------------------------------------------------------------------------------
// Fq2 = Fq[u]/(u^2+1) over any Fp-like type F (bn254/src/fqn.rs:12-13, 359-369)
// ---------------------------------------------------------------------------------------------
// Montgomery (a*b + sigma*c*d)/2^261 with one reduction over SIGNED 64-bit columns, sigma = -1 if `negate` else +1;
// result in [0, 2p).  The sign is data (a lane's half of an Fq2 product: see Fp2S), the instruction stream is the same
// for both.  Needs limbs < 2^31 and 9*(max_a*max_b + max_c*max_d) + 9*2^58 + 2^36 < 2^63 (both products may add up).
template <class P>
KG_HD Fp<P> mul2pm(const Fp<P>& a, const Fp<P>& b, const Fp<P>& c, const Fp<P>& d, bool negate) {
  uint32_t m[9];
  int32_t sc[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) sc[i] = negate ? -(int32_t)c.l[i] : (int32_t)c.l[i];
  Fp<P> r;
  int64_t acc = 0;
  static_for<0, 9>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    mac_col<k, 0, k>(acc, a.l, b.l);
    mac_col_i<k, 0, k>(acc, sc, d.l);
    mac_col_k<ModulusLimbs<P>, k, 0, k - 1>(acc, m);
    m[k] = ((uint32_t)acc * P::INV) & M29;
    mac_col_k<ModulusLimbs<P>, k, k, k>(acc, m);
    acc >>= 29;
  });
  static_for<9, 17>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    mac_col<k, k - 8, 8>(acc, a.l, b.l);
    mac_col_i<k, k - 8, 8>(acc, sc, d.l);
    mac_col_k<ModulusLimbs<P>, k, k - 8, 8>(acc, m);
    r.l[k - 9] = (uint32_t)acc & M29;
    acc >>= 29;
  });
  // value in (-p, 2p): add p when negative, then propagate carries
  const uint32_t neg_mask = (uint32_t)((int32_t)(acc >> 32) >> 31);
  uint32_t cy = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint32_t v = r.l[i] + (P::P[i] & neg_mask) + cy;
    r.l[i] = v & M29;
    cy = v >> 29;
  }
  r.l[8] = (uint32_t)acc + (P::P[8] & neg_mask) + cy;
  return r;
}

template <class F>
struct Fp2 {
  F c0, c1;
  static KG_HD Fp2 zero() { return {F::zero(), F::zero()}; }
  static KG_HD Fp2 one() { return {F::one(), F::zero()}; }
};
template <class F>
KG_HD Fp2<F> add(const Fp2<F>& a, const Fp2<F>& b) { return {add(a.c0, b.c0), add(a.c1, b.c1)}; }
template <class F>
KG_HD Fp2<F> dbl(const Fp2<F>& a) { return add(a, a); }
template <int C, int T, class F>
KG_HD Fp2<F> sub(const Fp2<F>& a, const Fp2<F>& b) { return {sub<C, T>(a.c0, b.c0), sub<C, T>(a.c1, b.c1)}; }
template <class F>
KG_HD Fp2<F> norm(const Fp2<F>& a) { return {norm(a.c0), norm(a.c1)}; }
// (a0 + a1 u)(b0 + b1 u): each output coordinate is ONE lazily reduced double product (2 x 243 mads: the
// multiply count of Karatsuba without its additions) and lands in [0, 2p) like a base-field product.
template <class F>
KG_HD Fp2<F> mul(const Fp2<F>& a, const Fp2<F>& b) {
#ifdef KG_FP2_MUL_VIA_PM   // host bound checker: drive the lane-pair routine (fp2s.h) through every G2 formula, both signs
  return {mul2pm(a.c0, b.c0, a.c1, b.c1, true), mul2pm(a.c0, b.c1, a.c1, b.c0, false)};
#else
  return {mul2sub(a.c0, b.c0, a.c1, b.c1), mul2add(a.c0, b.c1, a.c1, b.c0)};
#endif
}
// (a0 + a1 u)^2 = (a0 + a1)(a0 - a1) + 2 a0 a1 u     (inputs: normalised limbs, K <= 6)
template <class F>
KG_HD Fp2<F> sqr(const Fp2<F>& a) {
  return {mul(add(a.c0, a.c1), norm(sub<8, 1>(a.c0, a.c1))), mul(dbl(a.c0), a.c1)};
}
template <class F>
KG_HD Fp2<F> vred(const Fp2<F>& a) { return {vred(a.c0), vred(a.c1)}; }
// a*b - c*d in Fq2 (the point formulas' Y3): component-wise from base-field products
template <class F>
KG_HD Fp2<F> mul2sub(const Fp2<F>& a, const Fp2<F>& b, const Fp2<F>& c, const Fp2<F>& d) {
  Fp2<F> x = mul(a, b), y = mul(c, d);
  return {vred(norm(sub<4, 1>(x.c0, y.c0))), vred(norm(sub<4, 1>(x.c1, y.c1)))};
}
template <class F>
KG_HD bool is_zero_2p(const Fp2<F>& a) { return is_zero_2p(a.c0) && is_zero_2p(a.c1); }
template <class F>
KG_HD bool is_zero(const Fp2<F>& a) { return is_zero(a.c0) && is_zero(a.c1); }
template <class F>
KG_HD Fp2<F> reduce(const Fp2<F>& a) { return {reduce(a.c0), reduce(a.c1)}; }
template <class F>
KG_HD Fp2<F> inv(const Fp2<F>& a) {               // bn254/src/fqn.rs:348-357
  F t = inv(norm(add(sqr(a.c0), sqr(a.c1))));
  return {mul(t, a.c0), mul(t, norm(sub<16, 1>(F::zero(), a.c1)))};
}

using Fq = Fp<FqParams>;
using Fr = Fp<FrParams>;
using Fq2 = Fp2<Fq>;

}  // namespace kg
